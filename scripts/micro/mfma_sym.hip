// Microbenchmark for the symmetric fp32 kernel's inner loop at ONE wave per SIMD: 8 MFMAs per k-pair fed from four
// LDS panels (u, d, v, e), optionally with the two ds_write_b128 + adds per k-pair and the barrier per K-step.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o mfma_sym mfma_sym.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, BK = 32, PAN = BK * TM;

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k(float* out, int steps, long long* clk)
{
    extern __shared__ float sm[];
    float *Us = sm, *Ds = sm + 2 * PAN, *Vs = sm + 4 * PAN, *Es = sm + 6 * PAN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    for (int i = tid; i < 8 * PAN; i += 256) sm[i] = i * 1e-5f;
    __syncthreads();
    f32x16 aH[2][2], aD[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) aH[i][j][r] = aD[i][j][r] = 0.f;
    const int fa = kl * TM + wr * 64 + cl, fb = kl * TM + wc * 64 + cl;
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    float4 xa = make_float4(tid, 1, 2, 3), xb = make_float4(1, tid, 2, 3);
    float u0 = Us[fa], u1 = Us[fa + 32], d0 = Ds[fa], d1 = Ds[fa + 32], v0 = Vs[fb], v1 = Vs[fb + 32], e0 = Es[fb], e1 = Es[fb + 32];
    const long long t0 = clock64();
    for (int s2 = 0; s2 < steps; s2 += 2) {
#pragma unroll
      for (int buf = 0; buf < 2; ++buf) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (VARIANT == 2 && kk == BK / 2 - 1) __syncthreads();
            const int ro = (kk == BK / 2 - 1) ? (buf ^ 1) * PAN : buf * PAN + (kk + 1) * 2 * TM;
            const float nu0 = Us[ro + fa], nu1 = Us[ro + fa + 32], nd0 = Ds[ro + fa], nd1 = Ds[ro + fa + 32];
            const float nv0 = Vs[ro + fb], nv1 = Vs[ro + fb + 32], ne0 = Es[ro + fb], ne1 = Es[ro + fb + 32];
            if ((VARIANT == 1 || VARIANT == 2) && kk >= 8) {
                const int o = (buf ^ 1) * PAN + (srow + ((kk - 8) / 2) * 8) * TM + scol;
                float* A = (kk & 1) ? Vs : Us; float* B = (kk & 1) ? Es : Ds;
                *reinterpret_cast<float4*>(A + o) = make_float4(xa.x + xb.x, xa.y + xb.y, xa.z + xb.z, xa.w + xb.w);
                *reinterpret_cast<float4*>(B + o) = make_float4(xa.x - xb.x, xa.y - xb.y, xa.z - xb.z, xa.w - xb.w);
            }
            __builtin_amdgcn_sched_barrier(0);
            aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, aH[0][0], 0, 0, 0);
            aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v1, aH[0][1], 0, 0, 0);
            aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v0, aH[1][0], 0, 0, 0);
            aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v1, aH[1][1], 0, 0, 0);
            if (VARIANT == 3) {
                aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, aD[0][0], 0, 0, 0);
                aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v1, aD[0][1], 0, 0, 0);
                aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v0, aD[1][0], 0, 0, 0);
                aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v1, aD[1][1], 0, 0, 0);
            } else if (VARIANT == 4) {
                aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, e0, aH[0][0], 0, 0, 0);
                aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, e1, aH[0][1], 0, 0, 0);
                aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, e0, aH[1][0], 0, 0, 0);
                aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, e1, aH[1][1], 0, 0, 0);
            } else {
                aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, e0, aD[0][0], 0, 0, 0);
                aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, e1, aD[0][1], 0, 0, 0);
                aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, e0, aD[1][0], 0, 0, 0);
                aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, e1, aD[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            u0 = nu0; u1 = nu1; d0 = nd0; d1 = nd1; v0 = nv0; v1 = nv1; e0 = ne0; e1 = ne1;
        }
      }
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += aH[i][j][r] + aD[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}


typedef float f4v __attribute__((ext_vector_type(4)));
// VARIANT 5: the fragments of a k-pair live in two aligned register quads {u0,u1,v0,v1}, {d0,d1,e0,e1}, so that
// the A and B operand of every MFMA sit in different VGPR banks (bank = register index mod 4)
template <bool SPREAD>
__global__ __launch_bounds__(256, 1) void kq(float* out, int steps, long long* clk)
{
    extern __shared__ float sm[];
    float *Us = sm, *Ds = sm + 2 * PAN, *Vs = sm + 4 * PAN, *Es = sm + 6 * PAN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    for (int i = tid; i < 8 * PAN; i += 256) sm[i] = i * 1e-5f;
    __syncthreads();
    f32x16 aH[2][2], aD[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) aH[i][j][r] = aD[i][j][r] = 0.f;
    const int fa = kl * TM + wr * 64 + cl, fb = kl * TM + wc * 64 + cl;
    f4v h, d;
    h.x = Us[fa]; h.y = Us[fa + 32]; h.z = Vs[fb]; h.w = Vs[fb + 32];
    d.x = Ds[fa]; d.y = Ds[fa + 32]; d.z = Es[fb]; d.w = Es[fb + 32];
    const long long t0 = clock64();
    for (int s2 = 0; s2 < steps; s2 += 2) {
#pragma unroll
      for (int buf = 0; buf < 2; ++buf) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int ro = (kk == BK / 2 - 1) ? (buf ^ 1) * PAN : buf * PAN + (kk + 1) * 2 * TM;
            f4v nh, nd;
            if (!SPREAD) {
                nh.x = Us[ro + fa]; nh.y = Us[ro + fa + 32]; nh.z = Vs[ro + fb]; nh.w = Vs[ro + fb + 32];
                nd.x = Ds[ro + fa]; nd.y = Ds[ro + fa + 32]; nd.z = Es[ro + fb]; nd.w = Es[ro + fb + 32];
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(h), "+v"(d));
                aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.x, h.z, aH[0][0], 0, 0, 0);
                aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.x, h.w, aH[0][1], 0, 0, 0);
                aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.y, h.z, aH[1][0], 0, 0, 0);
                aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.y, h.w, aH[1][1], 0, 0, 0);
                aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, d.z, aD[0][0], 0, 0, 0);
                aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, d.w, aD[0][1], 0, 0, 0);
                aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, d.z, aD[1][0], 0, 0, 0);
                aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, d.w, aD[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // one LDS read per MFMA gap instead of all of them in one gap
                __builtin_amdgcn_sched_barrier(0);
                aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.x, h.z, aH[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                nh.x = Us[ro + fa]; nh.y = Us[ro + fa + 32];
                __builtin_amdgcn_sched_barrier(0);
                aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.x, h.w, aH[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                nh.z = Vs[ro + fb]; nh.w = Vs[ro + fb + 32];
                __builtin_amdgcn_sched_barrier(0);
                aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.y, h.z, aH[1][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                nd.x = Ds[ro + fa]; nd.y = Ds[ro + fa + 32];
                __builtin_amdgcn_sched_barrier(0);
                aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h.y, h.w, aH[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                nd.z = Es[ro + fb]; nd.w = Es[ro + fb + 32];
                __builtin_amdgcn_sched_barrier(0);
                aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, d.z, aD[0][0], 0, 0, 0);
                aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, d.w, aD[0][1], 0, 0, 0);
                aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, d.z, aD[1][0], 0, 0, 0);
                aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, d.w, aD[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            h = nh; d = nd;
        }
      }
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += aH[i][j][r] + aD[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <bool SPREAD>
void runq(const char* name)
{
    const int blocks = 256, steps = 3000;
    float* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kq<SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kq<SPREAD>, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kq<SPREAD>, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)steps * 128;
    printf("%-44s: %.1f cycles per MFMA, %.1f TFLOP/s\n", name, c / nm, nm * 4096.0 * 4 * blocks / ms / 1e9);
}


typedef float f2v __attribute__((ext_vector_type(2)));
// interleaved panels: [k][128][2] = (u, d) pairs for the I columns, (v, e) pairs for the J columns: the 8 fragment
// values of a k-pair arrive with TWO ds_read2_b64 (like the 4-MFMA kernel's two ds_read2_b32)
__global__ __launch_bounds__(256, 1) void ki(float* out, int steps, long long* clk)
{
    extern __shared__ float sm[];
    f2v* UD = reinterpret_cast<f2v*>(sm);            // [2][PAN]
    f2v* VE = UD + 2 * PAN;                          // [2][PAN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    for (int i = tid; i < 8 * PAN; i += 256) sm[i] = i * 1e-5f;
    __syncthreads();
    f32x16 aH[2][2], aD[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) aH[i][j][r] = aD[i][j][r] = 0.f;
    const int fa = kl * TM + wr * 64 + cl, fb = kl * TM + wc * 64 + cl;
    f2v p0 = UD[fa], p1 = UD[fa + 32], q0 = VE[fb], q1 = VE[fb + 32];
    const long long t0 = clock64();
    for (int s2 = 0; s2 < steps; s2 += 2) {
#pragma unroll
      for (int buf = 0; buf < 2; ++buf) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int ro = (kk == BK / 2 - 1) ? (buf ^ 1) * PAN : buf * PAN + (kk + 1) * 2 * TM;
            const f2v np0 = UD[ro + fa], np1 = UD[ro + fa + 32], nq0 = VE[ro + fb], nq1 = VE[ro + fb + 32];
            __builtin_amdgcn_sched_barrier(0);
            aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0.x, q0.x, aH[0][0], 0, 0, 0);
            aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0.x, q1.x, aH[0][1], 0, 0, 0);
            aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1.x, q0.x, aH[1][0], 0, 0, 0);
            aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1.x, q1.x, aH[1][1], 0, 0, 0);
            aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0.y, q0.y, aD[0][0], 0, 0, 0);
            aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0.y, q1.y, aD[0][1], 0, 0, 0);
            aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1.y, q0.y, aD[1][0], 0, 0, 0);
            aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1.y, q1.y, aD[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            p0 = np0; p1 = np1; q0 = nq0; q1 = nq1;
        }
      }
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += aH[i][j][r] + aD[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

void runi()
{
    const int blocks = 256, steps = 3000;
    float* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ki, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(ki, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)steps * 128;
    printf("%-44s: %.1f cycles per MFMA, %.1f TFLOP/s\n", "interleaved (u,d)/(v,e) panels, 2 ds_read2_b64", c / nm, nm * 4096.0 * 4 * blocks / ms / 1e9);
}

template <int V>
void run(const char* name)
{
    const int blocks = 256, steps = 3000;
    float* out; long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 131072, 0, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)steps * 128;
    printf("%-44s: %.1f cycles per MFMA, %.1f TFLOP/s\n", name, c / nm, nm * 4096.0 * 4 * blocks / ms / 1e9);
}

int main()
{
    run<0>("8 MFMAs / k-pair, 4 LDS panels, reads only");
    run<1>("+ 2 ds_write_b128 and 8 adds per k-pair (8-15)");
    run<2>("+ barrier before the last k-pair");
    run<3>("8 accumulators, operands from 2 panels only");
    run<4>("4 accumulators (8 MFMAs), 4 panels read");
    runq<false>("fragments in aligned register quads");
    runq<true>("... and one LDS read per MFMA gap");
    runi();
    return 0;
}
