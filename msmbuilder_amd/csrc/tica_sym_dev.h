// tica_sym_dev.h -- tica_sym_f32_kernel (sum/difference form, the bench kernel) and tica_export_sym_kernel
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_cg_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// Symmetric fp32 kernel (the fp32 default for 128 < F <= 3968): 20 instead of 26 tile products at F = 512.
// Only the SYMMETRISED lagged moment is ever used (offset_correlation = (C + C^T) / 2N' - mu mu^T,
// tica.py:234-241), and with the sum and difference frames of a pair, u = x_t + x_{t+tau},
// d = x_t - x_{t+tau},
//     H = sum_t u u^T = G + (C + C^T),     D = sum_t d d^T = G - (C + C^T)        (over valid pairs)
// so G = (H + D) / 2 and C + C^T = (H - D) / 2: TWO symmetric matrices, T(T+1) upper tile products
// instead of T^2 + T(T+1)/2, and no per-row weights {0,1,2} (a frame counts once per pair it is in).
// A workgroup owns one upper tile (I <= J) and computes BOTH its H and its D block from the same four
// loaded panels (x_t and x_{t+tau}, columns I and J): 128 accumulator registers per lane, two workgroups per
// CU with 64 KiB of LDS each (two buffers of u/d planes, see the kernel).  The sums and differences are formed in
// registers, inside the MFMA stream, before a staged half-step is written to LDS.
// fp32 rounding of u and d is 2^-24 relative and zero-mean: its contribution to the sums is
// ~eps/sqrt(N), far below the fp32 accumulation error, which is bounded by flushing to the fp64 slabs every
// KFLUSH_SYM frames (|H| is up to twice |G|).  The raw, non-symmetrised C is not available in this mode:
// the exported "C" is already (C + C^T) / 2, which is what every consumer of the handle forms anyway.
// Used from T = 2 tiles (F > 128) up to the width whose T(T+1)/2 upper tiles still fit one resident round
// (F <= 3968 on 256 CUs); a single tile has nothing to save (1 H + 1 D against 1 G + 1 C).
// ---------------------------------------------------------------------------
constexpr int KFLUSH_SYM = 4096;

__device__ __forceinline__ float4 f4mul(float4 a, float4 m) { return make_float4(a.x * m.x, a.y * m.y, a.z * m.z, a.w * m.w); }

// a - b on four floats as two v_pk_add_f32 with the negate modifiers on the second source (the compiler splits a
// vector fsub into scalar v_sub_f32: there is no v_pk_sub_f32)
__device__ __forceinline__ float __attribute__((ext_vector_type(4))) pk_sub4(float __attribute__((ext_vector_type(4))) a,
                                                                             float __attribute__((ext_vector_type(4))) b)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(a.xy), "v"(b.xy));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(a.zw), "v"(b.zw));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// Wave priority around the half-step boundary: from the barrier until k-pair MSM_SYM_PRIO_OFF the wave runs at priority
// MSM_SYM_PRIO, so its first fragment reads, its 8 global loads and its first MFMAs are issued ahead of the co-resident
// workgroup's stream (which otherwise lets them through about once per MFMA).  Measured on one box, 10M x 512
// (build variants side by side): off 51.1-51.7 ms; level 1 or 3, dropped at k-pair 2-4: 50.0-50.3 ms; held until k-pair 6
// or raised again for the staging instructions at k-pairs 5-7: no gain (51.2 ms); raised for the exposed staging at a chunk's
// start and for the slab merge: no gain either.  (Starting every other workgroup half a half-step late, so that the two
// workgroups of a CU do not meet their barriers together, COSTS 1 ms: they are better off in lockstep.  Moving the barrier
// in front of the half-step's last quad of MFMAs, with the next half-step's first fragments read behind it: +1.5 ms.)
#ifndef MSM_SYM_PRIO
#define MSM_SYM_PRIO 1
#endif
#ifndef MSM_SYM_PRIO_OFF
#define MSM_SYM_PRIO_OFF 4
#endif
#ifndef MSM_SYM_PRIO_ON2
#define MSM_SYM_PRIO_ON2 99
#endif
// FOLD: the staging lanes also sum the LEFT frames x_t of the valid pairs in fp64, so the separate column-sum pass over X
// goes: eight registers per thread hold the sums of the thread's four x-side columns (two instructions per element: widen,
// add), every tile adds every half-step it stages, and the diagonal tile (I, I) of a cohort writes the cohort's sums of
// column block I to P.colA.  NO branch in the MFMA stream decides anything (a version that shared the sums out over the
// tiles of a block, taking turns, saved the adds and lost 2 ms to the branches): the half-steps whose in-stream loads are
// dummies (their frames are staged by the edge sequence instead) read a row of zeros.  A NaN or an infinity anywhere in the
// left frames ends up in a sum, which is the finite check of the pass this replaces.
// REM (round 4): the grid is ALL resident slots -- P.S whole cohorts of P.ntiles workgroups, which take the chunks
// [0, P.n_main) round-robin as before, plus R = gridDim.x - P.S * P.ntiles workgroups that round 3 left idle (104 of 512 at
// 2,048 features): a REMAINDER cohort that takes the chunks [P.n_main, P.nchunks) in ceil(ntiles / R) rounds of R tiles
// (slab / column-sum row P.S).  The host picks n_main so that every workgroup is busy for the same time.
// ROLE SPLIT (round 5; built, measured, removed -- git history: "role-split fp32 sum/difference kernel"):
// eight waves per workgroup on a 128 x 128 tile of H OR of D, waves 0-3 issuing only fragment reads and MFMAs (64 x 64 each),
// waves 4-7 only staging (loads one half-step ahead in registers, shift, weights, packed adds, LDS writes), one barrier per
// half-step, two workgroups per CU.  Correct (the fp32 test files pass on it, eigenvalues equal to 1e-10) and SLOWER:
// 62.5 ms against 49.8 ms at 10M x 512 (0.67 against 0.84 of the fp32 MFMA peak; profiles/r05_role_split_f32_ab.txt).
// The 256 x 128 tile of H AND D that VERDICT names cannot exist: 4 MFMA waves x (64 x 128) x 2 matrices = 256 accumulators
// per lane, and a kernel's register allocation is uniform over its waves, so the stagers would be charged 256 + too -- one
// workgroup per CU; and 256-wide tiles waste a fifth of their products on F = 512's triangle (10 tiles of 128 do not pair up
// into dominoes without two singles).  With H and D in separate workgroups each reads the raw rows itself (2x the L2 -> CU
// bytes and 2x the staging arithmetic of this kernel, where one staged pair of panels feeds both matrices), the MFMA wave of a
// workgroup is alone on its SIMD with its barrier and LDS latencies, and what the interleaved kernel loses to its in-stream
// staging (matrix pipe busy 0.87) is less than that.  The item is closed.
template <bool PARTIAL, bool FOLD, bool REM = false>
__global__ __launch_bounds__(NT, 2) void tica_sym_f32_kernel(TicaArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HK = BK32 / 2;                   // frames per half-step
    constexpr int PAN = HK * TM;                    // floats per plane
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
#define MSM_F2(O) (*reinterpret_cast<const f2v*>(lds + (O)))
    // (x_t, x_{t+tau}) -> (u, d) of the SHIFTED frames y = x - r (R: this thread's four columns of r; W: row weight x
    // column mask, applied only where USEW): y is exact or rounded at its own (sigma-sized) scale, so the fp32 products
    // never see the column means
#define MSM_SYM_UD(A, B, R, W, USEW)                                                                   \
    {                                                                                                  \
        f4v a_ = pk_sub4(*reinterpret_cast<const f4v*>(&(A)), R), b_ = pk_sub4(*reinterpret_cast<const f4v*>(&(B)), R); \
        if (USEW) {                                                                                    \
            a_ *= *reinterpret_cast<const f4v*>(&(W));                                                 \
            b_ *= *reinterpret_cast<const f4v*>(&(W));                                                 \
        }                                                                                              \
        const f4v u_ = a_ + b_, d_ = pk_sub4(a_, b_);                                                  \
        A = *reinterpret_cast<const float4*>(&u_);                                                     \
        B = *reinterpret_cast<const float4*>(&d_);                                                     \
    }
    // LDS: two buffers (half-steps of 16 frames ping-pong between them) of four planes [16 frames][128 columns]:
    // u = x_t + x_{t+tau} and d = x_t - x_{t+tau} for the I columns, then for the J columns.  A lane's two MFMA row
    // blocks are the ADJACENT columns 2l and 2l+1 (the accumulators hold a permuted tile, undone at the slab merge),
    // so one ds_read2st64_b64 (u plane + d plane, 8 KiB apart) feeds four MFMAs, and the writer forms its
    // sums/differences with packed adds on the loaded float4s -- no lane or register shuffles.
    float* lds = reinterpret_cast<float*>(smem);   // [2 buffers][UI, DI, UJ, DJ][HK][TM]

    const int tid = threadIdx.x;
    const int p = xcd_linear_id();
    const bool rem = REM && p >= P.S * P.ntiles;             // a workgroup of the remainder cohort (uniform)
    const int remR = REM ? (int)gridDim.x - P.S * P.ntiles : 1;
    PROF_DECL;
  for (int round = 0; round < (rem ? (P.ntiles + remR - 1) / remR : 1); ++round) {   // (REM = false: one trip, folded away)
    const int cohort = rem ? P.S : p / P.ntiles;
    const int tile = rem ? p - P.S * P.ntiles + round * remR : p % P.ntiles;  // ntiles = T (T + 1) / 2 upper tiles
    if (rem && tile >= P.ntiles) break;
    int I = 0, u = tile;
    while (u >= P.T - I) {
        u -= P.T - I;
        ++I;
    }
    const int J = I + u;
    const int I0 = I * TM, J0 = J * TM;

    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 5, cl = lane & 31;
    double* slabH = P.slabs + ((size_t)cohort * P.ntiles + tile) * (2 * TM * TM);
    double* slabD = slabH + TM * TM;

    const int c4 = (tid & 31) * 4;
    const float4 ma = make_float4(I0 + c4 + 0 < P.F ? 1.f : 0.f, I0 + c4 + 1 < P.F ? 1.f : 0.f,
                                  I0 + c4 + 2 < P.F ? 1.f : 0.f, I0 + c4 + 3 < P.F ? 1.f : 0.f);
    const float4 mb = make_float4(J0 + c4 + 0 < P.F ? 1.f : 0.f, J0 + c4 + 1 < P.F ? 1.f : 0.f,
                                  J0 + c4 + 2 < P.F ? 1.f : 0.f, J0 + c4 + 3 < P.F ? 1.f : 0.f);

    // mean shift: the reference row r of this tile's I and J columns lives in LDS behind the panels ([2][TM] floats;
    // the kernel has no registers to spare) and is read, 16 bytes per thread, inside the MFMA stream one k-pair
    // before the packed subtractions that use it.  No shift = zeros (x - 0 is exact: bit-identical sums).
    float* rs = lds + 2 * 4 * PAN;
    double cs0 = 0.0, cs1 = 0.0, cs2 = 0.0, cs3 = 0.0;   // FOLD: fp64 sums of this thread's four x-side columns
#define MSM_SYM_COLADD(V)                                                                              \
    {                                                                                                  \
        cs0 += (double)(V).x;                                                                          \
        cs1 += (double)(V).y;                                                                          \
        cs2 += (double)(V).z;                                                                          \
        cs3 += (double)(V).w;                                                                          \
    }
    if (tid < 64) {
        const int col = (tid < 32 ? I0 : J0) + (tid & 31) * 4;
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.shift) rv = *reinterpret_cast<const float4*>(P.shift + (col < P.F ? col : P.F - 4));
        *reinterpret_cast<float4*>(rs + tid * 4) = rv;
    }

    f32x16 aH[2][2], aD[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) aH[bi][bj][r] = aD[bi][bj][r] = 0.f;
    int rows_acc = 0, chunks_done = 0;
    if (P.dbg && blockIdx.x == 0 && tid == 0) {
        P.dbg[0] = clock64();
        P.dbg[2] = wall_clock64();
    }

    const long long c_end = REM ? (rem ? P.nchunks : P.n_main) : P.nchunks, c_step = rem ? 1 : P.S;
    for (long long c = rem ? P.n_main : cohort; c < c_end; c += c_step) {
        PROF_MARK(5)
        const TicaChunk ch = get_chunk(P, c);
        const int nsteps = (ch.n + BK32 - 1) / BK32;
        ChunkCtx cx = make_ctx(P, ch);
        set_lag(cx, P.lag, sizeof(float), P.ld);
        const int srow = tid >> 5, scol = (tid & 31) * 4;
        const unsigned ca = 4u * (unsigned)(I0 + scol < P.F ? I0 + scol : P.F - 4), cb = 4u * (unsigned)(J0 + scol < P.F ? J0 + scol : P.F - 4);
        // TWO workgroups per CU (64 KiB of LDS each).  Section timers of the single-image version showed what a step
        // boundary costs there: each of its instructions (adds, LDS writes) issues only about once per MFMA of the
        // co-resident wave (~90 cycles), while an instruction inside this wave's own MFMA stream costs ~10.  So nothing
        // is left at the boundary: half-steps of 16 frames ping-pong between two LDS buffers, and while the 64 MFMAs of
        // half-step h run, the wave loads half-step h+1 (k-pairs 0-1: 8 global_load_dwordx4, scalar base + one lane
        // offset per panel), turns (x_t, x_{t+tau}) into (u, d) in place (k-pairs 5-6: packed adds) and writes it to
        // the other buffer (k-pairs 6-7: 8 ds_write_b128).  One barrier per half-step.
        // Half-steps that touch a trajectory edge (clamped rows, invalid pairs; a few per chunk) and the first one of
        // a chunk are staged by a plain, exposed sequence instead (MSM_STAGE_EDGE).
        const unsigned offx = (unsigned)srow * cx.ldb + ca, offy = (unsigned)srow * cx.ldb + cb;
        float4 xa[2], xb[2], ya[2], yb[2];  // rows srow, srow + 8 of the half-step: t / t+tau, columns I (x) and J (y)
        const int wofs = srow * TM + scol;  // floats; + buffer, plane, 8 rows
#define MSM_STORE_X(BUF)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                \
            *reinterpret_cast<float4*>(lds + (BUF) * 4 * PAN + 0 * PAN + j * 8 * TM + wofs) = xa[j];   \
            *reinterpret_cast<float4*>(lds + (BUF) * 4 * PAN + 1 * PAN + j * 8 * TM + wofs) = xb[j];   \
        }
#define MSM_STORE_Y(BUF)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                \
            *reinterpret_cast<float4*>(lds + (BUF) * 4 * PAN + 2 * PAN + j * 8 * TM + wofs) = ya[j];   \
            *reinterpret_cast<float4*>(lds + (BUF) * 4 * PAN + 3 * PAN + j * 8 * TM + wofs) = yb[j];   \
        }
#define MSM_STAGE_EDGE(K0, BUF)                                                                        \
        {                                                                                              \
            float sc_[2];                                                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
                const int kr = (K0) + srow + 8 * j;                                                    \
                const unsigned ra = (unsigned)(kr < cx.nmax ? kr : cx.nmax) * cx.ldb;                  \
                const unsigned rb = (unsigned)(kr < cx.nmaxB ? kr : cx.nmaxB) * cx.ldb;                \
                xa[j] = load16_global<char>(cx.base + (ra + ca));                                      \
                xb[j] = load16_global<char>(cx.baseB + (rb + ca));                                     \
                ya[j] = load16_global<char>(cx.base + (ra + cb));                                      \
                yb[j] = load16_global<char>(cx.baseB + (rb + cb));                                     \
                sc_[j] = (kr < cx.hi) ? 1.f : 0.f;                                                     \
            }                                                                                          \
            const f4v rx_ = *reinterpret_cast<const f4v*>(rs + scol), ry_ = *reinterpret_cast<const f4v*>(rs + TM + scol); \
            if (FOLD) {                                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
                    if (sc_[j] != 0.f) MSM_SYM_COLADD(xa[j])                                           \
            }                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
                const float4 wa = PARTIAL ? make_float4(sc_[j] * ma.x, sc_[j] * ma.y, sc_[j] * ma.z, sc_[j] * ma.w) \
                                          : make_float4(sc_[j], sc_[j], sc_[j], sc_[j]);               \
                MSM_SYM_UD(xa[j], xb[j], rx_, wa, true)                                                \
                MSM_SYM_UD(ya[j], yb[j], ry_, mb, PARTIAL)                                             \
            }                                                                                          \
            MSM_STORE_X(BUF)                                                                           \
            MSM_STORE_Y(BUF)                                                                           \
        }
#define MSM_SYM_FRAGS(BUF, KK)                                                                         \
                    const f2v npu = MSM_F2((BUF) * 4 * PAN + 0 * PAN + (KK) * 2 * TM + fa),            \
                              npd = MSM_F2((BUF) * 4 * PAN + 1 * PAN + (KK) * 2 * TM + fa),            \
                              nqu = MSM_F2((BUF) * 4 * PAN + 2 * PAN + (KK) * 2 * TM + fb),            \
                              nqd = MSM_F2((BUF) * 4 * PAN + 3 * PAN + (KK) * 2 * TM + fb);
#define MSM_SYM_MFMAS                                                                                  \
                    __builtin_amdgcn_sched_barrier(0);                                                 \
                    aH[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pu.x, qu.x, aH[0][0], 0, 0, 0);    \
                    aH[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pu.x, qu.y, aH[0][1], 0, 0, 0);    \
                    aH[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pu.y, qu.x, aH[1][0], 0, 0, 0);    \
                    aH[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pu.y, qu.y, aH[1][1], 0, 0, 0);    \
                    aD[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd.x, qd.x, aD[0][0], 0, 0, 0);    \
                    aD[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd.x, qd.y, aD[0][1], 0, 0, 0);    \
                    aD[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd.y, qd.x, aD[1][0], 0, 0, 0);    \
                    aD[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pd.y, qd.y, aD[1][1], 0, 0, 0);    \
                    __builtin_amdgcn_sched_barrier(0);                                                 \
                    pu = npu; pd = npd; qu = nqu; qd = nqd;
        __syncthreads();  // every wave is done with both buffers (previous chunk)
        MSM_STAGE_EDGE(0, 0)
        if (P.cosync && !rem && chunks_done > 0 && tid == 0) {  // cohort pacing (opt-in, see the C/G kernel): bounded wait
            const unsigned target = (unsigned)P.ntiles * (unsigned)chunks_done;
            const long long t0 = clock64();
            while (__hip_atomic_load(P.cosync + cohort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (clock64() - t0 > 200000) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        PROF_MARK(0)
        const int fa = kl * TM + wr * 64 + 2 * cl, fb = kl * TM + wc * 64 + 2 * cl;  // floats
        for (int s = 0; s < nsteps; ++s) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {  // half-step h = 2 s + b reads buffer b and fills buffer b ^ 1 with h + 1
                const int k1 = s * BK32 + (b + 1) * HK;
                const bool more = b == 0 || s + 1 < nsteps;
                const int lastrow = k1 + HK - 1;
                const bool fast = more && lastrow <= cx.nmax && lastrow <= cx.nmaxB && lastrow < cx.hi;  // uniform
                // ONE code path through the MFMAs (two variants of the loop make the compiler keep two copies of the 128
                // accumulators): a half-step that must not take the fast staging still runs it, on row 0 of the chunk
                // (always readable), and the edge sequence after the loop overwrites what it wrote
                const size_t kb = fast ? (size_t)k1 * cx.ldb : 0, r8 = fast ? (size_t)8 * cx.ldb : 0;  // scalar
                const global_ptr<char> zb = as_global<char>(P.zrow);
                const global_ptr<char> pa = FOLD && !fast ? zb : cx.base + kb, pb = FOLD && !fast ? zb : cx.baseB + kb;
                const unsigned ox = fast ? offx : ca, oy = fast ? offy : cb;
                f2v pu = MSM_F2(b * 4 * PAN + 0 * PAN + fa), pd = MSM_F2(b * 4 * PAN + 1 * PAN + fa);
                f2v qu = MSM_F2(b * 4 * PAN + 2 * PAN + fb), qd = MSM_F2(b * 4 * PAN + 3 * PAN + fb);
                f4v rsh;
                PROF_MARK(1)
#pragma unroll
                for (int kk = 0; kk < HK / 2; ++kk) {
                    MSM_SYM_FRAGS(b, (kk + 1 < HK / 2 ? kk + 1 : kk))
                    if (kk == 0) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            xa[j] = load16_global<char>(pa + j * r8 + ox);
                            xb[j] = load16_global<char>(pb + j * r8 + ox);
                        }
                    } else if (kk == 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            ya[j] = load16_global<char>(pa + j * r8 + oy);
                            yb[j] = load16_global<char>(pb + j * r8 + oy);
                        }
                    } else if (kk == 4) {
                        rsh = *reinterpret_cast<const f4v*>(rs + scol);  // r, I columns (waited on with the fragments)
                    } else if (kk == 5) {
                        if (FOLD) {   // (all sixteen here: split over k-pairs 4 and 5 the kernel is 0.7 ms slower)
#pragma unroll
                            for (int j = 0; j < 2; ++j) MSM_SYM_COLADD(xa[j])
                        }
#pragma unroll
                        for (int j = 0; j < 2; ++j) MSM_SYM_UD(xa[j], xb[j], rsh, ma, PARTIAL)
                        rsh = *reinterpret_cast<const f4v*>(rs + TM + scol);  // r, J columns
                    } else if (kk == 6) {
                        MSM_STORE_X(b ^ 1)
#pragma unroll
                        for (int j = 0; j < 2; ++j) MSM_SYM_UD(ya[j], yb[j], rsh, mb, PARTIAL)
                    } else if (kk == 7) {
                        MSM_STORE_Y(b ^ 1)
                    }
                    if (MSM_SYM_PRIO && kk == MSM_SYM_PRIO_OFF) __builtin_amdgcn_s_setprio(0);
                    if (MSM_SYM_PRIO && kk == MSM_SYM_PRIO_ON2) __builtin_amdgcn_s_setprio(MSM_SYM_PRIO);
                    MSM_SYM_MFMAS
                }
                if (more && !fast) MSM_STAGE_EDGE(k1, b ^ 1)
                PROF_MARK(2)
                __syncthreads();  // buffer b ^ 1 is complete, buffer b is free
                if (MSM_SYM_PRIO) __builtin_amdgcn_s_setprio(MSM_SYM_PRIO);  // first fragment reads + MFMAs of the new half-step first
                PROF_MARK(3)
            }
        }
#undef MSM_STAGE_EDGE
#undef MSM_SYM_MFMAS
#undef MSM_SYM_FRAGS
#undef MSM_STORE_X
#undef MSM_STORE_Y
        if (P.cosync && !rem) {
            ++chunks_done;
            if (tid == 0) __hip_atomic_fetch_add(P.cosync + cohort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rows_acc += ch.n;
        if (rows_acc + P.kc > P.kflush || c + c_step >= c_end) {
            rows_acc = 0;
            // accumulator register r of block (bi, bj), lane (kl, cl) = tile row wr*64 + 2*rho + bi with
            // rho = (r & 3) + 8 (r >> 2) + 4 kl, tile column wc*64 + 2*cl + bj: the two bj of a lane are adjacent doubles
            unsigned toff = (unsigned)((wr * 64 + 8 * kl) * TM + wc * 64 + 2 * cl);
            asm volatile("" : "+v"(toff));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double* slab = h ? slabD : slabH;
#pragma unroll
                for (int bi = 0; bi < 2; ++bi) {
                    double2 old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[r] = *reinterpret_cast<const double2*>(slab + (2 * ((r & 3) + 8 * (r >> 2)) + bi) * TM + toff);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        double2* q = reinterpret_cast<double2*>(slab + (2 * ((r & 3) + 8 * (r >> 2)) + bi) * TM + toff);
                        if (h) {
                            *q = make_double2(old[r].x + (double)aD[bi][0][r], old[r].y + (double)aD[bi][1][r]);
                            aD[bi][0][r] = aD[bi][1][r] = 0.f;
                        } else {
                            *q = make_double2(old[r].x + (double)aH[bi][0][r], old[r].y + (double)aH[bi][1][r]);
                            aH[bi][0][r] = aH[bi][1][r] = 0.f;
                        }
                    }
                }
            }
        }
    }
    PROF_MARK(4)
    if (FOLD) {
        __syncthreads();  // the panels are free
        double* cs = reinterpret_cast<double*>(smem);   // [4 elements][NT threads]
        cs[0 * NT + tid] = cs0;
        cs[1 * NT + tid] = cs1;
        cs[2 * NT + tid] = cs2;
        cs[3 * NT + tid] = cs3;
        __syncthreads();
        if (I == J && tid < TM) {  // column tid of the block = element tid & 3 of the threads (srow, tid >> 2), srow = 0..7
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) a += cs[(tid & 3) * NT + r * 32 + (tid >> 2)];
            P.colA[(size_t)cohort * P.F + I0 + tid] = a;
        }
        __syncthreads();  // (the next round's first staging writes the panels this sum was read from)
    }
  }   // round
#ifdef MSM_TICA_PROFILE
    if (P.dbg && tid == 0 && blockIdx.x < 5) {
        for (int i = 0; i < 6; ++i) P.dbg[8 + 8 * blockIdx.x + i] = pf_acc[i];
    }
#endif
    if (P.dbg && blockIdx.x == 0 && tid == 0) {
        P.dbg[1] = clock64();
        P.dbg[3] = wall_clock64();
    }
}
#undef MSM_F2
#undef MSM_SYM_UD
#undef MSM_SYM_COLADD

// packed C and G contributions of the symmetric kernel's slabs: G += (H + D) / 2 and "C" += (H - D) / 4
// (a symmetric matrix whose symmetrisation (C + C^T) / 2 is the lagged moment's).  One thread per element of an UPPER tile
// (diagonal tiles: r <= c): every slab word is read once -- coalesced along the tile row -- and the four outputs it feeds
// (C and G, (i, j) and its mirror image) are written from the same thread.
__global__ void tica_export_sym_kernel(const double* __restrict__ slabs, double* __restrict__ out, int F, int T,
                                       int ntiles, int S)
{
    const size_t FF = (size_t)F * F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)ntiles * TM * TM) return;
    const int tile = (int)(idx / (TM * TM));
    const int off = (int)(idx - (size_t)tile * (TM * TM));
    const int r = off / TM, c = off - r * TM;
    // tile -> (ti, tj), ti <= tj, in the row-major order of the upper triangle
    int ti = 0, first = 0;
    while (tile >= first + (T - ti)) {
        first += T - ti;
        ++ti;
    }
    const int tj = ti + (tile - first);
    const int i = ti * TM + r, j = tj * TM + c;
    if (i >= F || j >= F || (ti == tj && r > c)) return;
    double h0 = 0.0, d0 = 0.0, h1 = 0.0, d1 = 0.0;
    const double* sl = slabs + (size_t)tile * (2 * TM * TM) + off;
    const size_t step = (size_t)ntiles * (2 * TM * TM);
    int s = 0;
    for (; s + 1 < S; s += 2) {   // two independent chains: the loads of consecutive slabs overlap
        h0 += sl[0];
        d0 += sl[TM * TM];
        h1 += sl[step];
        d1 += sl[step + TM * TM];
        sl += 2 * step;
    }
    if (s < S) {
        h0 += sl[0];
        d0 += sl[TM * TM];
    }
    const double h = h0 + h1, d = d0 + d1;
    const double cv = 0.25 * (h - d), gv = 0.5 * (h + d);
    out[(size_t)i * F + j] += cv;
    out[FF + (size_t)i * F + j] += gv;
    if (i != j) {
        out[(size_t)j * F + i] += cv;
        out[FF + (size_t)j * F + i] += gv;
    }
}

}  // namespace msm
