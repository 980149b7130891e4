// tica_symw_dev.h -- tica_symw_f32_kernel: the sum/difference accumulation for NARROW feature sets (F <= 256), one
// workgroup owning the WHOLE of H and D, and its export kernel.  (round 6; included by tica.hip after tica_sym_dev.h)
//
// Why (profiles/r05_shape_probe.txt, r05_narrow_probe.txt): every accumulation kernel of rounds 1-5 works
// on 128 x 128 output tiles.  Below 128 features the kernel time did not fall with the width (4.2 ms per 8M frames whatever F:
// 0.03 TB/s of rows at F = 4, BASELINE configs[0]'s width), F = 171 (configs[2]: contact features of Fs-peptide,
// featurizer.py:1149-1179) padded to 256 and ran 3 tile pairs on 256 CUs: 4.29 ms per 2M frames where its flops need 1.1.
//
// Same mathematics as tica_sym_f32_kernel (tica.py:401-424 in the sum/difference form: H = sum u u^T, D = sum d d^T over the
// valid pairs, u = y_t + y_{t+tau}, d = y_t - y_{t+tau}, y = x - r), different decomposition:
//  * v_mfma_f32_16x16x4_f32 (32 cycles per 2,048 flop: the same 157.3 TF as the 32 x 32 form, a quarter of its tile), the
//    columns in GROUPS of W = 16 IL (IL = 1, 2, 4): lane (i = l & 15, k = l >> 4) reads IL adjacent floats of frame k at
//    column g W + IL i -- one ds_read_b32 / b64 / b128 -- and holds the A (= B: the matrices are X^T X) operands of the
//    group's IL interleaved 16-column blocks {g W + IL i + a}.  F pads to NG groups: 16, 32, 64, 128, 192, 256.
//  * only the blocks of the upper triangle: a group pair (g < g') has IL^2, a diagonal pair IL (IL + 1) / 2 blocks (the
//    blocks a <= b; a diagonal block computes its full 16 x 16).  F = 171 -> 3 groups -> 78 blocks per matrix, where
//    128-wide tiles execute the equivalent of 192.
//  * a workgroup is a COHORT of its own: it walks the chunks c = p, p + S, ..., reads every frame of them once (x_t and
//    x_{t+tau}: each row twice per launch, the second time out of the L2), keeps all of H and D in its accumulators and
//    merges them into its private fp64 slab [2][FP x FP].  Block-split (NG >= 1 with IL = 4): the 2 NBLK blocks are dealt
//    round-robin to the waves, every wave reads ALL fragments of a k-step (8 NG registers) and issues its share of the MFMAs
//    -- perfect balance, 6 LDS reads per 39 MFMAs at F = 171.  Frame-split (F <= 32: 2 or 6 blocks in all): every wave owns
//    all blocks and takes every NW-th k-step of the K-step; the waves merge into the slab one after the other.
//  * K-step = KS frames with KS FP = 4,096 floats (3,072 at FP = 192 and 96, 2,560 at 160): 16 KiB per plane (u, d), double
//    buffered -- a quarter of that for the frame-split variants, which are latency-bound and want residency (measured, 8M
//    frames: F = 4 0.189 -> 0.145 ms, F = 32 0.342 -> 0.317 ms going from 4,096 to 1,024 floats per plane); the
//    staging is plain -- loads of K-step s + 1 before the MFMAs of K-step s, shift / weights / u, d / LDS writes behind them --
//    because a K-step is >= 1,000 MFMA cycles per wave against a dozen memory instructions.
//  * rows need 4-byte alignment only: global_load_dwordx4 at any dword address (scripts/micro/unaligned_x4.hip), the piece
//    that straddles the row end is loaded at column F - 4 and shifted into place -- F = 171 takes the 16-byte path.
// Peak of the form: 2 NBLK MFMAs of 32 cycles per 4 frames and CU-SIMD: F <= 16: 153G frames/s (HBM-bound from F = 8),
// F <= 32: 51G (HBM-bound from F = 16), F <= 64: 15.4G, F = 128: 4.27G, F = 171..192: 1.97G, F <= 256: 1.13G.
#pragma once
#include <utility>

#include "tica_sym_dev.h"

namespace msm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double raw_f64x2 __attribute__((ext_vector_type(2)));

// the element type of a variant: float (v_mfma_f32_16x16x4_f32) or double (v_mfma_f64_16x16x4_f64: the same 16 x 4 operand and
// 16 x 16 result layout, one double per lane and operand).  Round 6, float64 rows of up to 128 features: every float64 row
// went to the 128-wide fp64 tile kernel, 13 - 17 ms per 8M frames whatever the width (profiles/r05_narrow_probe.txt).
template <typename T>
struct SymwTy;
template <>
struct SymwTy<float> {
    typedef f32x4 acc_t;
    static constexpr int EPP = 4;   // elements per 16-byte piece
};
template <>
struct SymwTy<double> {
    typedef f64x4 acc_t;
    static constexpr int EPP = 2;
};

template <int IL_, int NG_, int KS_, int NW_, bool FSPLIT_>
struct SymwCfg {
    static constexpr int IL = IL_, NG = NG_, KS = KS_, NW = NW_;
    static constexpr bool FSPLIT = FSPLIT_;
    static constexpr int W = 16 * IL;            // columns per group
    static constexpr int FP = W * NG;            // padded width = LDS row pitch (floats)
    static constexpr int NTH = 64 * NW;
    static constexpr int NBLK = NG * (IL * (IL + 1) / 2) + (NG * (NG - 1) / 2) * IL * IL;   // blocks per matrix
    static constexpr int NACC = FSPLIT ? 2 * NBLK : (2 * NBLK + NW - 1) / NW;             // accumulator blocks per wave
    static constexpr int PLANE = KS * FP;        // floats per LDS plane
    static constexpr int NV = (PLANE / 4 + NTH - 1) / NTH;   // 16-byte pieces per thread, plane and K-step
    static constexpr bool NVX = PLANE % (4 * NTH) == 0;      // ... a whole number of them for every thread
    static constexpr size_t LDS = (size_t)(2 * 2 * PLANE + FP) * sizeof(float);   // [2 buffers][u, d] + the shift row
    static_assert(KS % (4 * (FSPLIT ? NW : 1)) == 0, "whole k-steps (per wave)");
    // the same variant on doubles: half the frames per K-step (the same bytes per plane)
    static constexpr int KS64 = KS / 2;
    static constexpr size_t LDS64 = (size_t)(2 * 2 * KS64 * FP + FP) * sizeof(double);
    static_assert(KS64 % (4 * (FSPLIT ? NW : 1)) == 0, "whole k-steps (per wave), doubles");
};
// what depends on the element type
template <typename Cfg, typename T>
struct SymwK {
    static constexpr int EPP = SymwTy<T>::EPP;
    static constexpr int KS = sizeof(T) == 8 ? Cfg::KS64 : Cfg::KS;
    static constexpr int PLANE = KS * Cfg::FP;
    static constexpr int NV = (PLANE / EPP + Cfg::NTH - 1) / Cfg::NTH;
    static constexpr bool NVX = PLANE % (EPP * Cfg::NTH) == 0;
};

// block `id` of the enumeration g, gp, a, b (diagonal pairs: a <= b) -- evaluated at compile time only
struct SymwBlk {
    int g, gp, a, b;
};
template <typename Cfg>
constexpr SymwBlk symw_block(int id)
{
    int n = 0;
    for (int x = 0; x < Cfg::NG; ++x)
        for (int y = x; y < Cfg::NG; ++y)
            for (int p = 0; p < Cfg::IL; ++p)
                for (int q = (x == y ? p : 0); q < Cfg::IL; ++q) {
                    if (n == id) return SymwBlk{x, y, p, q};
                    ++n;
                }
    return SymwBlk{0, 0, 0, 0};
}
// the wave's q-th accumulator holds block gid (0 .. 2 NBLK - 1: H blocks, then D blocks), or nothing (gid >= 2 NBLK)
template <typename Cfg, int WAVE>
constexpr int symw_gid(int q)
{
    return Cfg::FSPLIT ? q : q * Cfg::NW + WAVE;
}

// one k-step of the wave's share: acc[Q] += frag(A)^T frag(B) for its blocks, every register index a compile-time constant
__device__ __forceinline__ f32x4 symw_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 symw_mfma(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <typename Cfg, int WAVE, int Q, typename T>
__device__ __forceinline__ void symw_mfma_one(typename SymwTy<T>::acc_t (&acc)[Cfg::NACC], const T (&fu)[Cfg::NG][Cfg::IL], const T (&fd)[Cfg::NG][Cfg::IL])
{
    constexpr int gid = symw_gid<Cfg, WAVE>(Q);
    if constexpr (gid < 2 * Cfg::NBLK) {
        constexpr SymwBlk k = symw_block<Cfg>(gid % Cfg::NBLK);
        if constexpr (gid < Cfg::NBLK)
            acc[Q] = symw_mfma(fu[k.g][k.a], fu[k.gp][k.b], acc[Q]);
        else
            acc[Q] = symw_mfma(fd[k.g][k.a], fd[k.gp][k.b], acc[Q]);
    }
}
template <typename Cfg, int WAVE, typename T, int... Q>
__device__ __forceinline__ void symw_mfmas(typename SymwTy<T>::acc_t (&acc)[Cfg::NACC], const T (&fu)[Cfg::NG][Cfg::IL], const T (&fd)[Cfg::NG][Cfg::IL],
                                           std::integer_sequence<int, Q...>)
{
    (symw_mfma_one<Cfg, WAVE, Q, T>(acc, fu, fd), ...);
}

// accumulator Q -> the fp64 slab: element (i = 4 kl + r, j = cl) of block (g, a | gp, b) is H[g W + IL i + a][gp W + IL j + b]
template <typename Cfg, int WAVE, int Q, typename ACC>
__device__ __forceinline__ void symw_merge_one(ACC (&acc)[Cfg::NACC], double* slab)
{
    constexpr int gid = symw_gid<Cfg, WAVE>(Q);
    if constexpr (gid < 2 * Cfg::NBLK) {
        constexpr SymwBlk k = symw_block<Cfg>(gid % Cfg::NBLK);
        constexpr int m = gid / Cfg::NBLK, FP = Cfg::FP, W = Cfg::W, IL = Cfg::IL;
        // `slab` arrives with the lane's part of the address folded in (element (4 kl, cl) of block (0, 0 | 0, 0)); the rest is
        // a compile-time constant per block and register
        double* base = slab + ((size_t)m * FP * FP + (size_t)(k.g * W + k.a) * FP + (k.gp * W + k.b));
        // (result layout: float 16 x 16 x 4: row 4 kl + r; double 16 x 16 x 4: row kl + 4 r -- the lane's kl is in `slab`)
        constexpr int RS = sizeof(acc[Q][0]) == 8 ? 4 * IL : IL;
        double old[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = base[(size_t)(RS * r) * FP];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            base[(size_t)(RS * r) * FP] = old[r] + (double)acc[Q][r];
            acc[Q][r] = 0;
        }
        // (one block at a time: left to itself the scheduler batches the loads of ALL the wave's blocks -- hundreds of
        //  registers for a merge that runs once per 8,192 frames -- and the MFMA loop pays for it in spills)
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <typename Cfg, int WAVE, typename ACC, int... Q>
__device__ __forceinline__ void symw_merge_all(ACC (&acc)[Cfg::NACC], double* slab, std::integer_sequence<int, Q...>)
{
    (symw_merge_one<Cfg, WAVE, Q, ACC>(acc, slab), ...);
}

struct SymwArgs {
    TicaArgs T;        // chunks, ld, F, lag, kflush, shift, flag: as for the other kernels (T.S = workgroups = cohorts)
    double* slabs;     // [S][2][FP * FP]
};

// one 16-byte piece of a row, columns c4 .. c4 + 3, from a row of F floats (4-byte aligned): pieces wholly inside the row as
// they are, the piece that straddles the row end from column F - 4, shifted; pieces beyond the row are zero (s = 4)
__device__ __forceinline__ float4 symw_fix(float4 v, int s)
{
    float4 o;
    o.x = s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : s == 3 ? v.w : 0.f;
    o.y = s == 0 ? v.y : s == 1 ? v.z : s == 2 ? v.w : 0.f;
    o.z = s == 0 ? v.z : s == 1 ? v.w : 0.f;
    o.w = s == 0 ? v.w : 0.f;
    return o;
}

// ... and of a row of F doubles (8-byte aligned... 4-byte is enough for the hardware): pieces of two, s = 0, 1 or 2 (beyond the row)
__device__ __forceinline__ raw_f64x2 symw_fix(raw_f64x2 v, int s)
{
    raw_f64x2 o;
    o.x = s == 0 ? v.x : s == 1 ? v.y : 0.0;
    o.y = s == 0 ? v.y : 0.0;
    return o;
}

template <typename Cfg, bool VEC, int WAVE>
__device__ __forceinline__ void symw_body(const SymwArgs& A, float* lds, int wave)
{
    constexpr int IL = Cfg::IL, NG = Cfg::NG, KS = Cfg::KS, NW = Cfg::NW, FP = Cfg::FP, W = Cfg::W, NTH = Cfg::NTH;
    constexpr int NV = Cfg::NV, PLANE = Cfg::PLANE, NACC = Cfg::NACC;
    constexpr bool FSPLIT = Cfg::FSPLIT;
    constexpr int PPR = FP / 4;   // pieces per row
    const TicaArgs& P = A.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kl = lane >> 4, cl = lane & 15;
    const int F = P.F;
    float* rs = lds + 4 * PLANE;   // the shift row r (zeros beyond F or without a shift)
    for (int c = tid; c < FP; c += NTH) rs[c] = (P.shift && c < F) ? P.shift[c] : 0.f;

    f32x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's NV pieces of a plane: frame fr[j] of the K-step, columns c4[j] .. + 3; sh[j] = how the loaded piece is
    // shifted into place (0: as loaded, 1..3: it straddles the row end, 4: beyond the row -> zeros), cs[j] = the column loaded
    int fr[NV], cs[NV], sh[NV], c4[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = tid + NTH * j;
        fr[j] = e / PPR;
        c4[j] = (e % PPR) * 4;
        if (VEC) {
            cs[j] = c4[j] + 4 <= F ? c4[j] : F - 4;
            const int s = c4[j] - cs[j];
            sh[j] = s > 4 ? 4 : s;
        } else {
            cs[j] = c4[j];
            sh[j] = 0;
        }
    }
    double* slab = A.slabs + (size_t)blockIdx.x * (2 * (size_t)FP * FP);
    int rows_acc = 0;

    // accumulators -> the workgroup's fp64 slab (block-split: every block has one owner; frame-split: the waves in turn)
    auto merge = [&]() {
        // the lane's part of every slab address, made opaque HERE: computed ahead of the chunk loop the hundreds of block
        // addresses of the merge are loop invariants, and the compiler keeps them all (in scratch) across the MFMA loop
        unsigned toff = (unsigned)((IL * 4 * kl) * FP + IL * cl);
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int turn = 0; turn < (FSPLIT ? NW : 1); ++turn) {
            if (!FSPLIT || wave == turn) symw_merge_all<Cfg, WAVE, f32x4>(acc, slab + toff, std::make_integer_sequence<int, NACC>{});
            if (FSPLIT) __syncthreads();
        }
    };

    for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
        const TicaChunk ch = get_chunk(P, c);
        const int nsteps = (ch.n + KS - 1) / KS;
        ChunkCtx cx = make_ctx(P, ch);
        set_lag(cx, P.lag, sizeof(float), P.ld);
        float4 xa[NV], xb[NV];
        // raw loads of K-step k0 (rows clamped into the trajectory: every address is valid; validity is the weight below)
#define SYMW_LOAD(K0)                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                                  \
            const int kr = (K0) + (fr[j] < KS ? fr[j] : KS - 1);   /* (a thread's last piece may lie beyond the plane) */ \
            const unsigned ra = (unsigned)(kr < cx.nmax ? kr : cx.nmax) * cx.ldb;                         \
            const unsigned rb = (unsigned)(kr < cx.nmaxB ? kr : cx.nmaxB) * cx.ldb;                       \
            if (VEC) {                                                                                    \
                xa[j] = load16_global<char>(cx.base + (ra + 4u * (unsigned)cs[j]));                       \
                xb[j] = load16_global<char>(cx.baseB + (rb + 4u * (unsigned)cs[j]));                      \
            } else {                                                                                      \
                float* pa_ = reinterpret_cast<float*>(&xa[j]);                                            \
                float* pb_ = reinterpret_cast<float*>(&xb[j]);                                            \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
                    const unsigned cc = 4u * (unsigned)(c4[j] + e < F ? c4[j] + e : F - 1);               \
                    pa_[e] = *(global_ptr<float>)(cx.base + (ra + cc));                                   \
                    pb_[e] = *(global_ptr<float>)(cx.baseB + (rb + cc));                                  \
                }                                                                                         \
            }                                                                                             \
        }
        // ... -> (u, d) of the shifted frames, zero for invalid pairs and padded columns -> LDS buffer BUF
#define SYMW_STORE(K0, BUF)                                                                               \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                                  \
            const int kr = (K0) + fr[j];                                                                  \
            const bool live = kr < cx.hi;                                                                 \
            float4 a_ = xa[j], b_ = xb[j];                                                                \
            if (VEC) {                                                                                    \
                a_ = symw_fix(a_, sh[j]);                                                                 \
                b_ = symw_fix(b_, sh[j]);                                                                 \
            }                                                                                             \
            const float4 r_ = *reinterpret_cast<const float4*>(rs + c4[j]);                               \
            /* (selects, not products with 0: a clamped row or column may hold anything) */              \
            const bool m0 = live && c4[j] + 0 < F, m1 = live && c4[j] + 1 < F, m2 = live && c4[j] + 2 < F, m3 = live && c4[j] + 3 < F; \
            const float ax = m0 ? a_.x - r_.x : 0.f, ay = m1 ? a_.y - r_.y : 0.f, az = m2 ? a_.z - r_.z : 0.f, aw = m3 ? a_.w - r_.w : 0.f; \
            const float bx = m0 ? b_.x - r_.x : 0.f, by = m1 ? b_.y - r_.y : 0.f, bz = m2 ? b_.z - r_.z : 0.f, bw = m3 ? b_.w - r_.w : 0.f; \
            float* pu_ = lds + (BUF) * 2 * PLANE + fr[j] * FP + c4[j];                                    \
            if (Cfg::NVX || fr[j] < KS) {                                                                 \
                *reinterpret_cast<float4*>(pu_) = make_float4(ax + bx, ay + by, az + bz, aw + bw);        \
                *reinterpret_cast<float4*>(pu_ + PLANE) = make_float4(ax - bx, ay - by, az - bz, aw - bw); \
            }                                                                                             \
        }
        SYMW_LOAD(0)
        __syncthreads();   // every wave is done with both buffers (previous chunk) -- and the shift row is in place
        SYMW_STORE(0, 0)
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
            if (s + 1 < nsteps) SYMW_LOAD((s + 1) * KS)
            const float* pl = lds + buf * 2 * PLANE + kl * FP + IL * cl;
#pragma unroll 2
            for (int kk = FSPLIT ? wave : 0; kk < KS / 4; kk += (FSPLIT ? NW : 1)) {
                float fu[NG][IL], fd[NG][IL];   // the k-step's fragments: u and d, IL interleaved blocks of every group
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float* q0 = pl + kk * 4 * FP + g * W;
                    if (IL == 4) {
                        const float4 vu = *reinterpret_cast<const float4*>(q0), vd = *reinterpret_cast<const float4*>(q0 + PLANE);
                        fu[g][0] = vu.x; fu[g][1 % IL] = vu.y; fu[g][2 % IL] = vu.z; fu[g][3 % IL] = vu.w;
                        fd[g][0] = vd.x; fd[g][1 % IL] = vd.y; fd[g][2 % IL] = vd.z; fd[g][3 % IL] = vd.w;
                    } else if (IL == 2) {
                        const float2 vu = *reinterpret_cast<const float2*>(q0), vd = *reinterpret_cast<const float2*>(q0 + PLANE);
                        fu[g][0] = vu.x; fu[g][1 % IL] = vu.y;
                        fd[g][0] = vd.x; fd[g][1 % IL] = vd.y;
                    } else {
                        fu[g][0] = q0[0];
                        fd[g][0] = q0[PLANE];
                    }
                }
                symw_mfmas<Cfg, WAVE, float>(acc, fu, fd, std::make_integer_sequence<int, NACC>{});
            }
            if (s + 1 < nsteps) SYMW_STORE((s + 1) * KS, buf ^ 1)
            __syncthreads();
        }
#undef SYMW_LOAD
#undef SYMW_STORE
        rows_acc += ch.n;
        if (rows_acc + P.kc > P.kflush || c + gridDim.x >= P.nchunks) {
            rows_acc = 0;
            merge();
        }
    }
}

template <typename Cfg, bool VEC>
__global__ __launch_bounds__(Cfg::NTH, 2) void tica_symw_f32_kernel(SymwArgs A)   // (2 waves per SIMD: 256 registers)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = reinterpret_cast<float*>(smem);   // [2 buffers][u, d][KS][FP] | r [FP]
    const int wave = threadIdx.x >> 6;
    if (Cfg::FSPLIT) {
        symw_body<Cfg, VEC, 0>(A, lds, wave);
    } else {
        // block-split: wave w issues the blocks gid % NW == w -- a compile-time set, so each wave gets its own copy of the loop
        // (every copy meets the same barriers the same number of times)
        switch (wave) {
            case 0: symw_body<Cfg, VEC, 0>(A, lds, wave); break;
            case 1: symw_body<Cfg, VEC, 1 % Cfg::NW>(A, lds, wave); break;
            case 2: symw_body<Cfg, VEC, 2 % Cfg::NW>(A, lds, wave); break;
            case 3: symw_body<Cfg, VEC, 3 % Cfg::NW>(A, lds, wave); break;
            case 4: symw_body<Cfg, VEC, 4 % Cfg::NW>(A, lds, wave); break;
            case 5: symw_body<Cfg, VEC, 5 % Cfg::NW>(A, lds, wave); break;
            case 6: symw_body<Cfg, VEC, 6 % Cfg::NW>(A, lds, wave); break;
            default: symw_body<Cfg, VEC, 7 % Cfg::NW>(A, lds, wave); break;
        }
    }
}

// ---- the same kernel on DOUBLES (float64 rows, F <= 128): pieces of two doubles, K-steps of KS / 2 frames, the fp64 matrix pipe.
// Accumulators stay in fp64 registers between merges; the shift row is applied as for the floats (the handle's column sums and
// their un-shifting do not care which kernel formed the products).  Peak of the form: 2 NBLK MFMAs of 64 cycles (16 x 16 x 4
// fp64: 2,048 flop at 78.6 TF) per 4 frames and CU-SIMD -- half the float variant's rate, HBM-bound up to 32 features.
template <typename Cfg, bool VEC, int WAVE>
__device__ __forceinline__ void symw_body64(const SymwArgs& A, double* lds, int wave)
{
    typedef SymwK<Cfg, double> K;
    constexpr int IL = Cfg::IL, NG = Cfg::NG, KS = K::KS, NW = Cfg::NW, FP = Cfg::FP, W = Cfg::W, NTH = Cfg::NTH;
    constexpr int NV = K::NV, PLANE = K::PLANE, NACC = Cfg::NACC;
    constexpr bool FSPLIT = Cfg::FSPLIT;
    constexpr int PPR = FP / 2;   // pieces per row
    const TicaArgs& P = A.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int kl = lane >> 4, cl = lane & 15;
    const int F = P.F;
    double* rs = lds + 4 * PLANE;   // the shift row r (zeros beyond F or without a shift)
    for (int c = tid; c < FP; c += NTH) rs[c] = (P.shift && c < F) ? (double)P.shift[c] : 0.0;

    f64x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};

    int fr[NV], cs[NV], sh[NV], c2[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = tid + NTH * j;
        fr[j] = e / PPR;
        c2[j] = (e % PPR) * 2;
        if (VEC) {
            cs[j] = c2[j] + 2 <= F ? c2[j] : F - 2;
            const int s = c2[j] - cs[j];
            sh[j] = s > 2 ? 2 : s;
        } else {
            cs[j] = c2[j];
            sh[j] = 0;
        }
    }
    double* slab = A.slabs + (size_t)blockIdx.x * (2 * (size_t)FP * FP);
    int rows_acc = 0;
    auto merge = [&]() {
        unsigned toff = (unsigned)((IL * kl) * FP + IL * cl);   // (double results: row kl + 4 r)
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int turn = 0; turn < (FSPLIT ? NW : 1); ++turn) {
            if (!FSPLIT || wave == turn) symw_merge_all<Cfg, WAVE, f64x4>(acc, slab + toff, std::make_integer_sequence<int, NACC>{});
            if (FSPLIT) __syncthreads();
        }
    };

    for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
        const TicaChunk ch = get_chunk(P, c);
        const int nsteps = (ch.n + KS - 1) / KS;
        ChunkCtx cx = make_ctx(P, ch);
        cx.base = as_global<char>(ch.base) + (size_t)ch.row0 * (size_t)P.ld * sizeof(double);
        cx.baseB = cx.base;
        cx.ldb = (unsigned)(P.ld * sizeof(double));
        set_lag(cx, P.lag, sizeof(double), P.ld);
        raw_f64x2 xa[NV], xb[NV];
        auto load = [&](int k0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int kr = k0 + (fr[j] < KS ? fr[j] : KS - 1);   // (a thread's last piece may lie beyond the plane)
                const unsigned ra = (unsigned)(kr < cx.nmax ? kr : cx.nmax) * cx.ldb;
                const unsigned rb = (unsigned)(kr < cx.nmaxB ? kr : cx.nmaxB) * cx.ldb;
                if (VEC) {
                    xa[j] = *(global_ptr<raw_f64x2>)(cx.base + (ra + 8u * (unsigned)cs[j]));
                    xb[j] = *(global_ptr<raw_f64x2>)(cx.baseB + (rb + 8u * (unsigned)cs[j]));
                } else {
                    const unsigned c0 = 8u * (unsigned)(c2[j] < F ? c2[j] : F - 1), c1 = 8u * (unsigned)(c2[j] + 1 < F ? c2[j] + 1 : F - 1);
                    xa[j].x = *(global_ptr<double>)(cx.base + (ra + c0));
                    xa[j].y = *(global_ptr<double>)(cx.base + (ra + c1));
                    xb[j].x = *(global_ptr<double>)(cx.baseB + (rb + c0));
                    xb[j].y = *(global_ptr<double>)(cx.baseB + (rb + c1));
                }
            }
        };
        auto store = [&](int k0, int buf) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int kr = k0 + fr[j];
                const bool live = kr < cx.hi;
                raw_f64x2 a_ = xa[j], b_ = xb[j];
                if (VEC) {
                    a_ = symw_fix(a_, sh[j]);
                    b_ = symw_fix(b_, sh[j]);
                }
                const double r0 = rs[c2[j]], r1 = rs[c2[j] + 1];
                // (selects, not products with 0: a clamped row or column may hold anything)
                const bool m0 = live && c2[j] + 0 < F, m1 = live && c2[j] + 1 < F;
                const double ax = m0 ? a_.x - r0 : 0.0, ay = m1 ? a_.y - r1 : 0.0;
                const double bx = m0 ? b_.x - r0 : 0.0, by = m1 ? b_.y - r1 : 0.0;
                double* pu_ = lds + buf * 2 * PLANE + fr[j] * FP + c2[j];
                if (K::NVX || fr[j] < KS) {
                    *reinterpret_cast<raw_f64x2*>(pu_) = raw_f64x2{ax + bx, ay + by};
                    *reinterpret_cast<raw_f64x2*>(pu_ + PLANE) = raw_f64x2{ax - bx, ay - by};
                }
            }
        };
        load(0);
        __syncthreads();   // every wave is done with both buffers (previous chunk) -- and the shift row is in place
        store(0, 0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
            if (s + 1 < nsteps) load((s + 1) * KS);
            const double* pl = lds + buf * 2 * PLANE + kl * FP + IL * cl;
#pragma unroll 2
            for (int kk = FSPLIT ? wave : 0; kk < KS / 4; kk += (FSPLIT ? NW : 1)) {
                double fu[NG][IL], fd[NG][IL];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double* q0 = pl + kk * 4 * FP + g * W;
#pragma unroll
                    for (int a = 0; a < IL; a += 2) {
                        if (IL >= 2) {
                            const raw_f64x2 vu = *reinterpret_cast<const raw_f64x2*>(q0 + a), vd = *reinterpret_cast<const raw_f64x2*>(q0 + PLANE + a);
                            fu[g][a] = vu.x; fu[g][(a + 1) % IL] = vu.y;
                            fd[g][a] = vd.x; fd[g][(a + 1) % IL] = vd.y;
                        } else {
                            fu[g][0] = q0[0];
                            fd[g][0] = q0[PLANE];
                        }
                    }
                }
                symw_mfmas<Cfg, WAVE, double>(acc, fu, fd, std::make_integer_sequence<int, NACC>{});
            }
            if (s + 1 < nsteps) store((s + 1) * KS, buf ^ 1);
            __syncthreads();
        }
        rows_acc += ch.n;
        if (rows_acc + P.kc > P.kflush || c + gridDim.x >= P.nchunks) {
            rows_acc = 0;
            merge();
        }
    }
}

template <typename Cfg, bool VEC>
__global__ __launch_bounds__(Cfg::NTH, 2) void tica_symw_f64_kernel(SymwArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);   // [2 buffers][u, d][KS / 2][FP] | r [FP]
    const int wave = threadIdx.x >> 6;
    if (Cfg::FSPLIT) {
        symw_body64<Cfg, VEC, 0>(A, lds, wave);
    } else {
        switch (wave) {
            case 0: symw_body64<Cfg, VEC, 0>(A, lds, wave); break;
            case 1: symw_body64<Cfg, VEC, 1 % Cfg::NW>(A, lds, wave); break;
            case 2: symw_body64<Cfg, VEC, 2 % Cfg::NW>(A, lds, wave); break;
            default: symw_body64<Cfg, VEC, 3 % Cfg::NW>(A, lds, wave); break;   // (the double variants have four waves)
        }
    }
}

// packed C and G contributions of the whole-matrix slabs: G += (H + D) / 2, "C" += (H - D) / 4, one thread per (p, q) of
// F x F.  Where the value lives: group pair (gp < gq): [p][q]; (gp > gq): [q][p]; same group: the block (a, b) = (p % IL,
// q % IL) was computed for a <= b only (in full: all 16 x 16 of it), so a < b: [p][q], a > b: [q][p], a == b: either.
__global__ void tica_export_symw_kernel(const double* __restrict__ slabs, double* __restrict__ out, int F, int FP, int W, int IL, int S)
{
    const size_t FF = (size_t)F * F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= FF) return;
    const int p = (int)(idx / F), q = (int)(idx - (size_t)p * F);
    const int gp = p / W, gq = q / W, a = p % IL, b = q % IL;
    const bool pq = gp < gq || (gp == gq && (a < b || (a == b && p <= q)));
    const int r = pq ? p : q, c = pq ? q : p;
    const double* sl = slabs + (size_t)r * FP + c;
    const size_t step = 2 * (size_t)FP * FP, dofs = (size_t)FP * FP;
    double h0 = 0.0, d0 = 0.0, h1 = 0.0, d1 = 0.0;
    int s = 0;
    for (; s + 1 < S; s += 2) {   // two independent chains
        h0 += sl[0];
        d0 += sl[dofs];
        h1 += sl[step];
        d1 += sl[step + dofs];
        sl += 2 * step;
    }
    if (s < S) {
        h0 += sl[0];
        d0 += sl[dofs];
    }
    const double h = h0 + h1, d = d0 + d1;
    out[idx] += 0.25 * (h - d);
    out[FF + idx] += 0.5 * (h + d);
}

// the variants: columns 16 / 32 (frame-split), 64 / 96 / 128 / 160 / 192 (block-split, 4 waves), 256 (block-split, 8 waves)
#ifndef MSM_SYMW_KSA
#define MSM_SYMW_KSA 64
#endif
#ifndef MSM_SYMW_KSB
#define MSM_SYMW_KSB 32
#endif
typedef SymwCfg<1, 1, MSM_SYMW_KSA, 4, true> SymwA;    // F <= 16
typedef SymwCfg<2, 1, MSM_SYMW_KSB, 4, true> SymwB;    // F <= 32
typedef SymwCfg<4, 1, 64, 4, false> SymwC;             // F <= 64
typedef SymwCfg<2, 3, 32, 4, false> SymwG;             // F <= 96   (three 32-column groups: 21 blocks per matrix, not the 36 of 128 columns)
typedef SymwCfg<4, 2, 32, 4, false> SymwD;             // F <= 128
typedef SymwCfg<2, 5, 16, 4, false> SymwH;             // F <= 160  (55 blocks, not 78)
typedef SymwCfg<4, 3, 16, 4, false> SymwE;             // F <= 192
typedef SymwCfg<4, 4, 16, 8, false> SymwF;             // F <= 256

}  // namespace msm
