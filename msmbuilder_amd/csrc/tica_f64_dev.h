// tica_f64_dev.h -- tica_mfma_f64_kernel (fp64 MFMA)
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_cg_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// fp64 kernel: v_mfma_f64_16x16x4_f64 on inputs widened to fp64 while staging.
// fp32 x fp32 products are exact in fp64, so this is the reference's float64
// arithmetic up to summation order.  Structural twin of the fp32 kernel: K-step =
// 16 frames (64 MFMAs of 64 cycles per wave, like 32 frames there), double-
// buffered LDS panels [16][144] fp64 (pitch 144: rows k and k+1 land on disjoint
// bank halves for the 16-lane-per-row ds_read_b64 fragments), two-step-deep
// register pipeline, 32-bit chunk-relative addressing.  Accumulators stay in
// registers for the workgroup's whole life (one slab merge at the end).
// ---------------------------------------------------------------------------
constexpr int P64 = 144;  // LDS row pitch in doubles

template <typename TIn>
struct Stage64 {
    static constexpr int NV = 16 * TM * sizeof(TIn) / 16 / NT;  // 16-byte vectors per thread per panel: 2 (f32) / 4 (f64)
    float4 a[NV], b[NV];
    double sc[NV];
};

template <typename TIn>
__device__ __forceinline__ void stage_load64(Stage64<TIn>& st, const ChunkCtx& cx, int F, int k0, int isG,
                                             int tauB, int I0, int J0, int tid, bool vec)
{
    constexpr int NV = Stage64<TIn>::NV;
    constexpr int E = 16 / sizeof(TIn);       // elements per vector
    constexpr int VPR = TM / E;               // vectors per panel row
    constexpr int RPP = NT / VPR;             // rows covered per pass
    const int ce = (tid % VPR) * E;
    const int rr0 = tid / VPR;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int kr = k0 + rr0 + RPP * j;
        double sc = (kr < cx.hi) ? 1.0 : 0.0;
        if (isG) sc += (kr >= cx.lo && kr < cx.n) ? 1.0 : 0.0;
        const int ra = kr < cx.nmax ? kr : cx.nmax;
        const int rb = kr < cx.nmaxB ? kr : cx.nmaxB;
        const unsigned oa = (unsigned)ra * cx.ldb, ob = (unsigned)rb * cx.ldb;
        if (vec) {
            const int ca = (I0 + ce < F) ? I0 + ce : F - E;
            const int cb = (J0 + ce < F) ? J0 + ce : F - E;
            st.a[j] = load16_global<char>(cx.base + (oa + (unsigned)ca * (unsigned)sizeof(TIn)));
            st.b[j] = load16_global<char>(cx.baseB + (ob + (unsigned)cb * (unsigned)sizeof(TIn)));
        } else {
            TIn* pa = reinterpret_cast<TIn*>(&st.a[j]);
            TIn* pb = reinterpret_cast<TIn*>(&st.b[j]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int ca = (I0 + ce + e < F) ? I0 + ce + e : F - 1;
                const int cb = (J0 + ce + e < F) ? J0 + ce + e : F - 1;
                pa[e] = *(global_ptr<TIn>)(cx.base + (oa + (unsigned)ca * (unsigned)sizeof(TIn)));
                pb[e] = *(global_ptr<TIn>)(cx.baseB + (ob + (unsigned)cb * (unsigned)sizeof(TIn)));
            }
        }
        st.sc[j] = sc;
    }
}

template <typename TIn>
__device__ __forceinline__ void stage_store64(const Stage64<TIn>& st, double* As, double* Bs, int F, int I0,
                                              int J0, int tid)
{
    constexpr int NV = Stage64<TIn>::NV;
    constexpr int E = 16 / sizeof(TIn);
    constexpr int VPR = TM / E;
    constexpr int RPP = NT / VPR;
    const int ce = (tid % VPR) * E;
    const int rr0 = tid / VPR;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int rr = rr0 + RPP * j;
        const TIn* pa = reinterpret_cast<const TIn*>(&st.a[j]);
        const TIn* pb = reinterpret_cast<const TIn*>(&st.b[j]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            As[rr * P64 + ce + e] = (I0 + ce + e < F) ? st.sc[j] * (double)pa[e] : 0.0;
            Bs[rr * P64 + ce + e] = (J0 + ce + e < F) ? (double)pb[e] : 0.0;
        }
    }
}

#ifndef MSM_F64_PRIO
#define MSM_F64_PRIO 1
#endif
#ifndef MSM_F64_PRIO_OFF
#define MSM_F64_PRIO_OFF 1
#endif
template <typename TIn>
__global__ __launch_bounds__(NT, 2) void tica_mfma_f64_kernel(TicaArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* As = reinterpret_cast<double*>(smem);  // [2][BK64][P64]
    double* Bs = As + 2 * BK64 * P64;              // [2][BK64][P64]

    const int tid = threadIdx.x;
    const int p = xcd_linear_id();
    const int cohort = p / P.ntiles, tile = p % P.ntiles;
    int I, J, isG;
    decode_tile(tile, P.T, I, J, isG);
    const int I0 = I * TM, J0 = J * TM;
    const int tauB = isG ? 0 : P.lag;

    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 4, cl = lane & 15;  // A[i=cl][k=kl], B[k=kl][j=cl]
    double* slab = P.slabs + (size_t)p * (TM * TM);
    constexpr int E = 16 / sizeof(TIn);

    f64x4 acc[4][4];
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[bi][bj][r] = 0.0;

    for (long long c = cohort; c < P.nchunks; c += P.S) {
        const TicaChunk ch = get_chunk(P, c);
        const int nsteps = (ch.n + BK64 - 1) / BK64;
        ChunkCtx cx = make_ctx(P, ch);
        cx.base = as_global<char>(ch.base) + (size_t)ch.row0 * (size_t)P.ld * sizeof(TIn);
        cx.ldb = (unsigned)(P.ld * sizeof(TIn));
        cx.baseB = cx.base;
        set_lag(cx, tauB, sizeof(TIn), P.ld);
        const bool vec = (P.F % E == 0) && (P.ld % E == 0) && ((((uintptr_t)ch.base) & 15) == 0);

        Stage64<TIn> st0, st1;
        stage_load64<TIn>(st0, cx, P.F, 0, isG, tauB, I0, J0, tid, vec);
        stage_store64<TIn>(st0, As, Bs, P.F, I0, J0, tid);
        stage_load64<TIn>(st0, cx, P.F, BK64, isG, tauB, I0, J0, tid, vec);
        __syncthreads();
#define MSM_TICA_STEP64(SNEXT, SLOAD, BUF)                                                        \
        {                                                                                         \
            stage_load64<TIn>(SLOAD, cx, P.F, (s + 2) * BK64, isG, tauB, I0, J0, tid, vec);       \
            const double* Ab = As + (BUF) * (BK64 * P64) + kl * P64 + wr * 64 + cl;               \
            const double* Bb = Bs + (BUF) * (BK64 * P64) + kl * P64 + wc * 64 + cl;               \
            _Pragma("unroll") for (int kk = 0; kk < BK64 / 4; ++kk) {                             \
                double a[4], b[4];                                                                \
                if (MSM_F64_PRIO && kk == MSM_F64_PRIO_OFF) __builtin_amdgcn_s_setprio(0);        \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi) a[bi] = Ab[kk * 4 * P64 + bi * 16]; \
                _Pragma("unroll") for (int bj = 0; bj < 4; ++bj) b[bj] = Bb[kk * 4 * P64 + bj * 16]; \
                _Pragma("unroll") for (int bi = 0; bi < 4; ++bi)                                  \
                    _Pragma("unroll") for (int bj = 0; bj < 4; ++bj)                              \
                        acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], b[bj], acc[bi][bj], 0, 0, 0); \
            }                                                                                     \
            if (s + 1 < nsteps)                                                                   \
                stage_store64<TIn>(SNEXT, As + ((BUF) ^ 1) * (BK64 * P64), Bs + ((BUF) ^ 1) * (BK64 * P64), P.F, I0, J0, tid); \
            __syncthreads();                                                                      \
            if (MSM_F64_PRIO) __builtin_amdgcn_s_setprio(MSM_F64_PRIO); /* as in the sum/difference kernel */ \
        }
        for (int s = 0; s < nsteps; s += 2) {
            MSM_TICA_STEP64(st0, st1, 0)
            ++s;
            if (s < nsteps) MSM_TICA_STEP64(st1, st0, 1)
            --s;
        }
#undef MSM_TICA_STEP64
    }
    // C/D layout of the f64 16x16x4 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    unsigned toff = (unsigned)((wr * 64 + kl) * TM + wc * 64 + cl);
    asm volatile("" : "+v"(toff));
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        double old[4][4];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[bj][r] = (slab + (bi * 16 + 4 * r) * TM + bj * 16)[toff];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 4; ++r) (slab + (bi * 16 + 4 * r) * TM + bj * 16)[toff] = old[bj][r] + acc[bi][bj][r];
    }
}

}  // namespace msm
