// scripts/micro/h2d_pipe.hip -- where the host-staged upload's 42 GB/s comes from: pageable source -> (T threads memcpy) -> pinned
// ring -> DMA, each stage alone and together.
//   hipcc --offload-arch=gfx950 -O2 -pthread scripts/micro/h2d_pipe.hip -o scripts/micro/h2d_pipe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* d, const char* s, size_t n, int T)
{
    std::vector<std::thread> th;
    const size_t per = (((n + T - 1) / T) + 4095) & ~(size_t)4095;
    for (int i = 0; i < T; ++i)
        th.emplace_back([=] { const size_t o = (size_t)i * per; if (o < n) memcpy(d + o, s + o, std::min(per, n - o)); });
    for (auto& t : th) t.join();
}
int main()
{
    const size_t total = (size_t)4 << 30, slice = (size_t)20 << 20;
    char* src = (char*)malloc(total);
    memset(src, 1, total);   // first touch by the main thread, like a numpy array filled by the interpreter thread
    char *pin = nullptr, *d = nullptr, *dstpage = (char*)malloc(total);
    memset(dstpage, 2, total);
    CK(hipHostMalloc((void**)&pin, total, hipHostMallocDefault));
    memset(pin, 3, total);
    CK(hipMalloc((void**)&d, total));
    hipStream_t cs;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (int T : {1, 4, 8, 16}) {
        double t = now();
        for (size_t off = 0; off < total; off += slice) par_copy(pin + off, src + off, std::min(slice, total - off), T);
        printf("pageable -> pinned, %2d threads (fresh threads per 20 MB slice): %.1f GB/s\n", T, total / (now() - t) / 1e9);
        t = now();
        for (size_t off = 0; off < total; off += slice) par_copy(dstpage + off, src + off, std::min(slice, total - off), T);
        printf("pageable -> pageable, %2d threads:                               %.1f GB/s\n", T, total / (now() - t) / 1e9);
    }
    {
        double t = now();
        par_copy(pin, src, total, 8);
        printf("pageable -> pinned, 8 threads, ONE 4 GB copy: %.1f GB/s\n", total / (now() - t) / 1e9);
        t = now();
        par_copy(pin, src, total, 32);
        printf("pageable -> pinned, 32 threads, ONE 4 GB copy: %.1f GB/s\n", total / (now() - t) / 1e9);
    }
    {
        CK(hipDeviceSynchronize());
        double t = now();
        for (size_t off = 0; off < total; off += slice) CK(hipMemcpyAsync(d + off, pin + off, std::min(slice, total - off), hipMemcpyHostToDevice, cs));
        CK(hipDeviceSynchronize());
        printf("pinned -> device alone (20 MB copies): %.1f GB/s\n", total / (now() - t) / 1e9);
        t = now();
        for (size_t off = 0; off < total; off += slice) {
            par_copy(pin + off, src + off, std::min(slice, total - off), 8);
            CK(hipMemcpyAsync(d + off, pin + off, std::min(slice, total - off), hipMemcpyHostToDevice, cs));
        }
        CK(hipDeviceSynchronize());
        printf("pipeline (8-thread memcpy of slice i+1 beside the DMA of slice i): %.1f GB/s\n", total / (now() - t) / 1e9);
    }
    return 0;
}
