// Microbenchmark: sustained issue cadence of v_mfma_f32_32x32x2_f32 per SIMD with 1, 2, 4 waves per SIMD and
// 4 independent accumulators per wave (the fp32 tICA kernel's register tile).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* clk)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd, const char* name)
{
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD of a CU
    float* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 16 * NACC;  // MFMAs per wave
    const double flops = nm * 4096.0 * 4 * blocks;
    printf("%s waves/SIMD=%d: %.1f cycles per MFMA per wave (%.1f per SIMD), %.1f TFLOP/s, %.3f ms\n", name, waves_per_simd,
           c / nm, c / nm / waves_per_simd, flops / ms / 1e9, ms);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int w : {1, 2, 4}) run<4>(w, "4 acc");
    for (int w : {1, 2}) run<2>(w, "2 acc");
    for (int w : {1, 2}) run<1>(w, "1 acc");
    return 0;
}
