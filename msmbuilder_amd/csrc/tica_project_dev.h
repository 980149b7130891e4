// tica_project_dev.h -- tica_project_kernel / tica_project_mfma_kernel (transform)
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_common_dev.h"

namespace msm {

// Element types of the projection kernels: float32, float64, and bfloat16-STORED rows (BASELINE configs[4]) as raw 16-bit
// words that the kernels widen themselves -- an exact widening, so bfloat16 rows project like their float32 images.
struct Bf16Raw { unsigned short bits; };
__device__ __forceinline__ double pj_widen(float x) { return (double)x; }
__device__ __forceinline__ double pj_widen(double x) { return x; }
__device__ __forceinline__ double pj_widen(Bf16Raw x) { return (double)__uint_as_float((unsigned)x.bits << 16); }
__device__ __forceinline__ bool pj_finite(float x) { return isfinite(x); }
__device__ __forceinline__ bool pj_finite(double x) { return isfinite(x); }
__device__ __forceinline__ bool pj_finite(Bf16Raw x) { return (x.bits & 0x7f80u) != 0x7f80u; }

// out[n,k] = (X - mean) @ comps^T in fp64 (tica.py:329-333), evaluated as X @ comps^T - (mean @ comps^T)
// with the k constants mean @ comps^T precomputed on the host in fp64.  HBM-bound: reads
// F*sizeof(T) and writes 8k bytes per frame.  A workgroup owns 128 rows; X tiles [128][FC] arrive
// as 16-byte loads (256-byte row segments) and are written TRANSPOSED to LDS ([FC][128+1], raw
// element type) so that lane-per-row reads are consecutive words; wave w accumulates components
// w*NPW .. w*NPW+NPW-1 of the current tile of 4*NPW components for rows lane and lane+64, one
// fp64 FMA chain per output in feature order (deterministic).
template <typename TIn, int NPW>
__global__ __launch_bounds__(NT) void tica_project_kernel(const TIn* __restrict__ X, long long n,
                                                          int F, long long ld,
                                                          const double* __restrict__ muV,
                                                          const double* __restrict__ comps, int k,
                                                          double* __restrict__ out, int* flag, int vec)
{
    constexpr int FC = 64, KT = 4 * NPW, CW = 16 / sizeof(TIn), RW = 128, RP = RW + 1;
    __shared__ TIn Xs[FC * RP];
    __shared__ double Vs[KT][FC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * RW;
    int bad = 0;
    for (int k0 = 0; k0 < k; k0 += KT) {
        const int kt = (k - k0) < KT ? (k - k0) : KT;
        double acc[2][NPW];
#pragma unroll
        for (int a = 0; a < NPW; ++a) acc[0][a] = acc[1][a] = 0.0;
        for (int f0 = 0; f0 < F; f0 += FC) {
            __syncthreads();
            if (vec) {
                constexpr int VPR = FC / CW;  // 16-byte vectors per row segment
                for (int e = tid; e < RW * VPR; e += NT) {
                    const int rr = e / VPR, cc = (e % VPR) * CW;
                    const long long r = row0 + rr;
                    TIn v[CW];
#pragma unroll
                    for (int q = 0; q < CW; ++q) v[q] = TIn{};
                    if (r < n && f0 + cc < F)
                        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(X + r * ld + f0 + cc);
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        bad |= !pj_finite(v[q]);
                        Xs[(cc + q) * RP + rr] = v[q];
                    }
                }
            } else {
                for (int e = tid; e < RW * FC; e += NT) {
                    const int rr = e / FC, ff = e % FC;
                    const long long r = row0 + rr;
                    TIn v = TIn{};
                    if (r < n && f0 + ff < F) v = X[r * ld + f0 + ff];
                    bad |= !pj_finite(v);
                    Xs[ff * RP + rr] = v;
                }
            }
            for (int e = tid; e < KT * FC; e += NT) {
                const int kk = e / FC, ff = e % FC;
                Vs[kk][ff] = (kk < kt && f0 + ff < F) ? comps[(size_t)(k0 + kk) * F + f0 + ff] : 0.0;
            }
            __syncthreads();
            const int fw = (F - f0) < FC ? (F - f0) : FC;
#pragma unroll 4
            for (int ff = 0; ff < fw; ++ff) {
                const double x0 = pj_widen(Xs[ff * RP + lane]);
                const double x1 = pj_widen(Xs[ff * RP + lane + 64]);
#pragma unroll
                for (int a = 0; a < NPW; ++a) {
                    const double v = Vs[wave * NPW + a][ff];
                    acc[0][a] = fma(x0, v, acc[0][a]);
                    acc[1][a] = fma(x1, v, acc[1][a]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long r = row0 + lane + 64 * h;
            if (r < n) {
#pragma unroll
                for (int a = 0; a < NPW; ++a) {
                    const int kk = wave * NPW + a;
                    if (kk < kt) out[r * k + k0 + kk] = acc[h][a] - muV[k0 + kk];
                }
            }
        }
    }
    if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// Projection on the fp64 matrix pipe (16-byte aligned rows): out[N, k] = X . Vp - muV with
// v_mfma_f64_16x16x4_f64 -- 16 rows x 4 features x 16 components per instruction, inputs widened to
// fp64 exactly as the vector kernel does, fp64 accumulation (same products, the sum of a row is
// merely associated in groups of four features).  A wave owns 64 rows (4 row blocks); per
// 128-byte chunk of a row every lane loads its own 32 bytes STRAIGHT from global memory -- lane
// (r = lane & 15, g = lane >> 4) takes the 8 floats / 4 doubles at column 8g (4g) of row r -- and
// MFMA t of the chunk contracts the features {c0 + E2*g + t : g = 0..3}: a fixed permutation that the
// component panel Vp follows (staged in LDS per chunk as [feature][16 comps], pitch 20 doubles, k
// padded with zeros), so X needs no LDS at all.  Per chunk and wave: 8 loads, 8 fragment reads, 32
// MFMAs (2,048 pipe cycles) for 8 KiB of input -- the matrix time per byte is about the HBM time
// per byte, so the kernel streams at HBM rate instead of being bound by LDS broadcast reads like the
// one-lane-per-row kernel above (2.9 TB/s).  Non-finite INPUT makes non-finite OUTPUT (x finite
// always gives a finite sum), so the finite check of validation.py:68-74 is applied to the k outputs.
// ---------------------------------------------------------------------------
constexpr int PVP = 20;  // LDS pitch of a Vp feature row in doubles (16 comps + 4: lanes of different g hit different banks)

// one 256-row tile of a BATCHED projection (msm_tica_project_batch): the tile's first row, the rows of its trajectory from
// there on, and where the tile's first output row goes
struct ProjTile {
    const void* x;
    double* out;
    long long rows;
};

template <typename TIn>
__global__ __launch_bounds__(NT, 2) void tica_project_mfma_kernel(const TIn* __restrict__ X, long long n, int F,
                                                                  long long ld, const double* __restrict__ muV,
                                                                  const double* __restrict__ Vp /* [F][16] */, int k,
                                                                  int kbase, int ktot, double* __restrict__ out,
                                                                  int* flag, const ProjTile* __restrict__ tiles)
{
    if (tiles) {   // a list of trajectories: this workgroup's tile stands for the whole array (uniform branch)
        const ProjTile t = tiles[blockIdx.x];
        X = static_cast<const TIn*>(t.x);
        out = t.out;
        n = t.rows;
    }
    const long long blk = tiles ? 0 : (long long)blockIdx.x;
    constexpr int FCH = 128 / (int)sizeof(TIn);  // features per chunk (64 bf16 / 32 f32 / 16 f64)
    constexpr int NT4 = FCH / 4;                 // MFMAs per chunk and row block (16 / 8 / 4)
    constexpr int NV = (FCH + 31) / 32;          // Vp rows a thread stages per chunk
    constexpr int RB = 4;                        // row blocks of 16 per wave
    __shared__ double Vs[2][FCH * PVP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long long row0 = blk * (4 * RB * 16) + wave * (RB * 16);
    const int nch = (F + FCH - 1) / FCH;
    const unsigned ldb = (unsigned)(ld * sizeof(TIn));

    // per-lane byte offsets of this lane's rows (clamped into [0, n)) relative to the tile's first row
    const long long tile0 = blk * (4 * RB * 16);
    const global_ptr<char> Xg = as_global<char>(X) + (size_t)tile0 * ldb;
    unsigned xo[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        long long i = row0 + b * 16 + r;
        if (i > n - 1) i = n - 1;
        xo[b] = (unsigned)(i - tile0) * ldb;
    }
    // Vp staging: thread -> (feature tid >> 3, component pair (tid & 7) * 2) of the chunk
    const int vf = tid >> 3, vc = (tid & 7) * 2;
    const global_ptr<char> Vg = as_global<char>(Vp);

    f64x4 acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[b][q] = 0.0;

    raw_f32x4 xs0[RB][2], xs1[RB][2], vreg[NV];
    // byte offsets of this lane's two 16-byte halves of chunk c inside a row; a half that lies past the
    // row (partial last chunk) re-reads the row's last 16 bytes instead: finite data against zero Vp rows
#define MSM_PJ_LOAD(XS, C)                                                                        \
    {                                                                                             \
        const int last16 = F * (int)sizeof(TIn) - 16;                                             \
        const int h0 = (C) * 128 + g * 32, h1 = h0 + 16;                                          \
        const unsigned a0 = (unsigned)(h0 < last16 ? h0 : last16), a1 = (unsigned)(h1 < last16 ? h1 : last16); \
        _Pragma("unroll") for (int b = 0; b < RB; ++b) {                                          \
            XS[b][0] = *(global_ptr<raw_f32x4>)(Xg + (xo[b] + a0));                               \
            XS[b][1] = *(global_ptr<raw_f32x4>)(Xg + (xo[b] + a1));                               \
        }                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                          \
            const int lf = vf + 32 * j, feat = (C) * FCH + lf;                                    \
            vreg[j] = *(global_ptr<raw_f32x4>)(Vg + (size_t)(feat < F ? feat : F - 1) * 128 + vc * 8); \
            if (feat >= F || lf >= FCH) vreg[j] = raw_f32x4{0.f, 0.f, 0.f, 0.f};                  \
        }                                                                                         \
    }
#define MSM_PJ_VSTORE(BUF)                                                                        \
    _Pragma("unroll") for (int j = 0; j < NV; ++j)                                                \
        if (vf + 32 * j < FCH) *reinterpret_cast<raw_f32x4*>(&Vs[BUF][(vf + 32 * j) * PVP + vc]) = vreg[j];
    MSM_PJ_LOAD(xs0, 0)
    MSM_PJ_VSTORE(0)
    __syncthreads();
#define MSM_PJ_STEP(XCUR, XNXT, BUF)                                                              \
    {                                                                                             \
        if (c + 1 < nch) MSM_PJ_LOAD(XNXT, c + 1)                                                 \
        const double* vb = &Vs[BUF][(g * (FCH / 4)) * PVP + r];                                   \
        _Pragma("unroll") for (int t = 0; t < NT4; ++t) {                                         \
            const double bv = vb[t * PVP];                                                        \
            _Pragma("unroll") for (int b = 0; b < RB; ++b) {                                      \
                const TIn* xe = reinterpret_cast<const TIn*>(&XCUR[b][0]);                        \
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj_widen(xe[t]), bv, acc[b], 0, 0, 0); \
            }                                                                                     \
        }                                                                                         \
        if (c + 1 < nch) MSM_PJ_VSTORE((BUF) ^ 1)                                                 \
        __syncthreads();                                                                          \
    }
    for (int c = 0; c < nch; c += 2) {
        MSM_PJ_STEP(xs0, xs1, 0)
        ++c;
        if (c < nch) MSM_PJ_STEP(xs1, xs0, 1)
        --c;
    }
#undef MSM_PJ_STEP
#undef MSM_PJ_VSTORE
#undef MSM_PJ_LOAD
    // C/D layout: component = lane & 15, row = (lane >> 4) + 4 * reg
    int bad = 0;
    const int comp = r;
    if (comp < k) {
        const double mv = muV[kbase + comp];
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long i = row0 + b * 16 + g + 4 * q;
                const double v = acc[b][q] - mv;
                if (i < n) {
                    ((double __attribute__((address_space(1)))*)(uintptr_t)out)[i * ktot + kbase + comp] = v;   // (global store also when `out` came from the tile table)
                    bad |= !isfinite(v);
                }
            }
    }
    if (bad) atomicOr(flag, 1);
}

}  // namespace msm
