// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the fp64 VALU instructions behind the exact
// libdistance arithmetic -- v_add_f64, v_mul_f64, v_fma_f64, v_cvt_f64_f32, fp32 sub -- with 16 independent chains per
// lane and 4 waves per SIMD (throughput, not latency).  This is the measured denominator of the "fp64-VALU bound" that
// DESIGN 3.5 quotes for assign_nearest (sub, mul, add separately rounded: -ffp-contract=off).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_f64 valu_f64.hip ; run: ./valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* clk, double seed)
{
    double a[16];
    float f[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + threadIdx.x * 1e-3 + i;
        f[i] = (float)a[i];
    }
    const double c = seed * 0.999, d = seed * 1e-3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) a[i] = a[i] + c;                       // v_add_f64
                if (OP == 1) a[i] = a[i] * c;                       // v_mul_f64
                if (OP == 2) a[i] = __builtin_fma(a[i], c, d);      // v_fma_f64
                if (OP == 3) { f[i] = f[i] - 1.0f; a[i] = (double)f[i]; }   // v_sub_f32 + v_cvt_f64_f32
                if (OP == 4) { const double t = a[i] - c; a[i] = a[i] + t * t; }   // the f64 pair-element: sub, mul, add
                if (OP == 5) { const float t = f[i] - (float)c; const double q = (double)t; a[i] = a[i] + q * q; f[i] = t; }  // f32 pair-element
            }
    }
    const long long t1 = clock64();
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP>
void run(const char* name, double n_instr)
{
    const int wps = 4, blocks = 256 * wps;
    double* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&clk, 8);
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk, 1.0000001);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk, 1.0000001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double groups = (double)iters * 8 * 16;   // op groups per wave
    printf("%-34s %.2f shader cycles per group per wave, %.2f per SIMD (4 waves/SIMD) = %.2f per instruction; %.3f ms\n", name,
           c / groups, c / groups / wps, c / groups / wps / n_instr, ms);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("v_add_f64", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_fma_f64", 1);
    run<3>("v_sub_f32 + v_cvt_f64_f32", 2);
    run<4>("f64 pair-element (sub, mul, add)", 3);
    run<5>("f32 pair-element (sub,cvt,mul,add)", 4);
    return 0;
}
