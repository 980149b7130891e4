#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(const unsigned* src, unsigned* out, unsigned off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    for (unsigned i = lane; i < 163840 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src) + lane * 16);
    __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(smem + off);
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // report where the data landed: scan for word 0 of the source
    for (unsigned i = lane; i < 163840 / 4; i += 64) {
        unsigned v = reinterpret_cast<unsigned*>(smem)[i];
        if (v != 0xdeadbeefu) { unsigned s = atomicAdd(out, 1u); if (s < 15) { out[1 + 2 * s] = i * 4; out[2 + 2 * s] = v; } }
    }
}
int main()
{
    unsigned *src, *out;
    hipMalloc(&src, 1024); hipMalloc(&out, 256);
    std::vector<unsigned> h(256); for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (unsigned off : {0u, 65536u, 131072u - 1024u, 131072u, 140000u - 140000u % 16, 163840u - 1024u}) {
        hipMemset(out, 0, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, off);
        unsigned r[64]; hipMemcpy(r, out, 256, hipMemcpyDeviceToHost);
        unsigned mn = ~0u; for (unsigned s = 0; s < r[0] && s < 15; ++s) if (r[2 + 2 * s] == 1000) mn = r[1 + 2 * s];
        printf("off %u: %u words changed, word 1000 at byte %u\n", off, r[0], mn);
    }
    return 0;
}
