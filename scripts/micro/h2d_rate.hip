// scripts/micro/h2d_rate.hip -- pinned -> device copy rate: chunk size x number of copy streams (does a second SDMA engine help?)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/h2d_rate.hip -o scripts/micro/h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main()
{
    const size_t total = (size_t)4 << 30;
    char *h = nullptr, *d = nullptr;
    CK(hipHostMalloc((void**)&h, total, hipHostMallocDefault));
    CK(hipMalloc((void**)&d, total));
    memset(h, 1, total);
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t chunk : {(size_t)4 << 20, (size_t)20 << 20, (size_t)32 << 20, (size_t)256 << 20})
        for (int ns : {1, 2, 4}) {
            double best = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                int i = 0;
                for (size_t off = 0; off < total; off += chunk, ++i)
                    CK(hipMemcpyAsync(d + off, h + off, std::min(chunk, total - off), hipMemcpyHostToDevice, st[i % ns]));
                CK(hipDeviceSynchronize());
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                best = std::max(best, total / s / 1e9);
            }
            printf("chunk %4zu MB, %d stream(s): %.1f GB/s\n", chunk >> 20, ns, best);
        }
    return 0;
}
