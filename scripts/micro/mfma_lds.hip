// Microbenchmark: the fp32 tICA kernel's phase A -- fragment reads from LDS ([32][128] frame-major panels)
// interleaved with v_mfma_f32_32x32x2_f32 -- with NO global loads, stores or barriers: cycles per MFMA for
// different read placements.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, BK = 32;

#define MFMA4(A0, A1, B0, B1)                                                      \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);  \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);  \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);  \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);

// VARIANT 0: reads for k+1 issued before the 4 MFMAs of k (the product kernel)
// VARIANT 1: reads for k+2 (two k-pairs ahead, three register sets)
// VARIANT 2: reads for k+1 issued BETWEEN the MFMAs of k (after the first)
// VARIANT 3: no LDS reads at all (register operands)
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void k(float* out, int steps, long long* clk)
{
    __shared__ float As[BK * TM], Bs[BK * TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    for (int i = tid; i < BK * TM; i += 256) { As[i] = i * 1e-4f; Bs[i] = i * 2e-4f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* Ab = As + kl * TM + wr * 64 + cl;
    const float* Bb = Bs + kl * TM + wc * 64 + cl;
    const long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        if (VARIANT == 0) {
            float a0 = Ab[0], a1 = Ab[32], b0 = Bb[0], b1 = Bb[32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int kn = (kk + 1 < BK / 2) ? kk + 1 : kk;
                const float na0 = Ab[kn * 2 * TM], na1 = Ab[kn * 2 * TM + 32], nb0 = Bb[kn * 2 * TM], nb1 = Bb[kn * 2 * TM + 32];
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(a0, a1, b0, b1)
                __builtin_amdgcn_sched_barrier(0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        } else if (VARIANT == 1) {
            float a0 = Ab[0], a1 = Ab[32], b0 = Bb[0], b1 = Bb[32];
            float c0 = Ab[2 * TM], c1 = Ab[2 * TM + 32], d0 = Bb[2 * TM], d1 = Bb[2 * TM + 32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int kn = (kk + 2 < BK / 2) ? kk + 2 : BK / 2 - 1;
                const float na0 = Ab[kn * 2 * TM], na1 = Ab[kn * 2 * TM + 32], nb0 = Bb[kn * 2 * TM], nb1 = Bb[kn * 2 * TM + 32];
                __builtin_amdgcn_sched_barrier(0);
                MFMA4(a0, a1, b0, b1)
                __builtin_amdgcn_sched_barrier(0);
                a0 = c0; a1 = c1; b0 = d0; b1 = d1;
                c0 = na0; c1 = na1; d0 = nb0; d1 = nb1;
            }
        } else if (VARIANT == 2) {
            float a0 = Ab[0], a1 = Ab[32], b0 = Bb[0], b1 = Bb[32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int kn = (kk + 1 < BK / 2) ? kk + 1 : kk;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const float na0 = Ab[kn * 2 * TM], na1 = Ab[kn * 2 * TM + 32], nb0 = Bb[kn * 2 * TM], nb1 = Bb[kn * 2 * TM + 32];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        } else {
            float a0 = tid * 1e-3f, a1 = a0 + 1.f, b0 = a0 * 2.f, b1 = a0 + 3.f;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) { MFMA4(a0, a1, b0, b1) }
        }
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int V>
void run(int blocks_per_cu, const char* name)
{
    const int blocks = 256 * blocks_per_cu, steps = 4000;
    float* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, steps, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)steps * 64;
    printf("%-34s waves/SIMD=%d: %.1f cycles per MFMA per wave, %.1f TFLOP/s\n", name, blocks_per_cu, c / nm,
           nm * 4096.0 * 4 * blocks / ms / 1e9);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int w : {1, 2}) {
        run<3>(w, "no LDS reads");
        run<0>(w, "reads k+1 before MFMAs (product)");
        run<1>(w, "reads k+2 (3 register sets)");
        run<2>(w, "reads k+1 after first MFMA");
    }
    return 0;
}
