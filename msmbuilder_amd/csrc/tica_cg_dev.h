// tica_cg_dev.h -- staging helpers (ChunkCtx, Stage32) and tica_mfma_f32_kernel, the fp32 C/G kernel
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_common_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// fp32 kernel: v_mfma_f32_32x32x2_f32.  LDS panels are frame-major [BK][128]
// exactly like X in HBM (coalesced 512-B row segments in, conflict-free
// ds_read_b32 fragment reads out: lanes 0-31 read 32 consecutive floats of
// frame k, lanes 32-63 of frame k+1).
// ---------------------------------------------------------------------------
template <bool VEC4>
struct Stage32 {
    float4 a[4], b[4];
    float sc[4];  // per-row weight applied when the stage is written to LDS (NOT at load time:
                  // touching a loaded value early would park the wave on vmcnt before the MFMA loop)
};

// Per-chunk, wave-uniform addressing context.  Everything per-lane is 32-bit and chunk
// relative: rows are clamped into the trajectory and columns into [0, F) so every address is
// valid; validity is carried by the A-side weight (0 kills the whole rank-1 term, B only has to
// be finite) and, for partial tiles, by column masks applied at LDS-store time.  Loads become
// `global_load_dwordx4 v, v_off32, s[base]`: no 64-bit VALU address math in the K loop.
struct ChunkCtx {
    global_ptr<char> base;  // &X[row0][0]
    global_ptr<char> baseB; // &X[row0 + tau][0] (lagged panel; == base for Gram tiles or when no pair is valid)
    int nmaxB;              // kr <= nmaxB keeps the lagged row inside the trajectory
    int n;                  // rows in the chunk
    int lo;                 // kr >= lo  <=>  row >= lag           (second Gram term)
    int hi;                 // kr <  hi  <=>  row <  len - lag, and kr < n
    int nmax;               // kr <= nmax keeps the row inside the trajectory
    unsigned ldb;           // row pitch in bytes
};

__device__ __forceinline__ int sat_i32(long long v)
{
    return v > 0x3fffffff ? 0x3fffffff : (v < -0x3fffffff ? -0x3fffffff : (int)v);
}

__device__ __forceinline__ ChunkCtx make_ctx(const TicaArgs& P, const TicaChunk& ch)
{
    ChunkCtx c;
    c.base = as_global<char>(ch.base) + (size_t)ch.row0 * (size_t)P.ld * sizeof(float);
    c.n = ch.n;
    c.lo = sat_i32((long long)P.lag - ch.row0);
    const int hi = sat_i32(ch.len - P.lag - ch.row0);
    c.hi = hi < ch.n ? hi : ch.n;
    c.nmax = sat_i32((ch.last < ch.len - 1 ? ch.last : ch.len - 1) - ch.row0);
    c.ldb = (unsigned)(P.ld * sizeof(float));
    c.baseB = c.base;
    c.nmaxB = c.nmax;
    return c;
}

// The lag goes into a 64-bit base pointer, never into the 32-bit per-lane offsets (lag * pitch can
// exceed 4 GiB); if the lagged row of the chunk's first frame is already past the trajectory end
// no pair of this chunk is valid (all weights are 0) and the B panel may read the A rows instead.
__device__ __forceinline__ void set_lag(ChunkCtx& c, long long tauB, size_t elem_bytes, long long ld)
{
    if (tauB > 0 && c.nmax >= tauB) {
        c.baseB = c.base + (size_t)tauB * (size_t)ld * elem_bytes;
        c.nmaxB = sat_i32((long long)c.nmax - tauB);
    }
}

template <bool VEC4>
__device__ __forceinline__ float4 load_row4(global_ptr<char> base, unsigned rowoff, int col, int F)
{
    if (VEC4) {
        const int c = col < F ? col : F - 4;
        return load16_global<char>(base + (rowoff + (unsigned)c * 4u));
    } else {
        float4 v;
        v.x = *(global_ptr<float>)(base + (rowoff + 4u * (unsigned)(col + 0 < F ? col + 0 : F - 1)));
        v.y = *(global_ptr<float>)(base + (rowoff + 4u * (unsigned)(col + 1 < F ? col + 1 : F - 1)));
        v.z = *(global_ptr<float>)(base + (rowoff + 4u * (unsigned)(col + 2 < F ? col + 2 : F - 1)));
        v.w = *(global_ptr<float>)(base + (rowoff + 4u * (unsigned)(col + 3 < F ? col + 3 : F - 1)));
        return v;
    }
}

template <bool VEC4>
__device__ __forceinline__ void stage_load32(Stage32<VEC4>& st, const ChunkCtx& cx, int F, int k0,
                                             int isG, int tauB, int I0, int J0, int tid)
{
    const int c4 = (tid & 31) * 4;
    const int rr0 = tid >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kr = k0 + rr0 + 8 * j;
        float sc = (kr < cx.hi) ? 1.f : 0.f;
        if (isG) sc += (kr >= cx.lo && kr < cx.n) ? 1.f : 0.f;
        const int ra = kr < cx.nmax ? kr : cx.nmax;
        const int rb = kr < cx.nmaxB ? kr : cx.nmaxB;
        st.a[j] = load_row4<VEC4>(cx.base, (unsigned)ra * cx.ldb, I0 + c4, F);
        st.b[j] = load_row4<VEC4>(cx.baseB, (unsigned)rb * cx.ldb, J0 + c4, F);
        st.sc[j] = sc;
    }
}

template <bool VEC4, bool PARTIAL>
__device__ __forceinline__ void stage_store32(const Stage32<VEC4>& st, float* As, float* Bs, int tid,
                                              float4 ma, float4 mb)
{
    const int c4 = (tid & 31) * 4;
    const int rr0 = tid >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = rr0 + 8 * j;
        const float sc = st.sc[j];
        if (PARTIAL) {
            *reinterpret_cast<float4*>(As + rr * TM + c4) =
                make_float4(st.a[j].x * (sc * ma.x), st.a[j].y * (sc * ma.y), st.a[j].z * (sc * ma.z),
                            st.a[j].w * (sc * ma.w));
            *reinterpret_cast<float4*>(Bs + rr * TM + c4) =
                make_float4(st.b[j].x * mb.x, st.b[j].y * mb.y, st.b[j].z * mb.z, st.b[j].w * mb.w);
        } else {
            *reinterpret_cast<float4*>(As + rr * TM + c4) =
                make_float4(st.a[j].x * sc, st.a[j].y * sc, st.a[j].z * sc, st.a[j].w * sc);
            *reinterpret_cast<float4*>(Bs + rr * TM + c4) = st.b[j];
        }
    }
}

// Compile-time section timers (make EXTRA_tica=-DMSM_TICA_PROFILE): every wave reads s_memtime at the
// section boundaries of the fp32 kernel and wave 0 of a few workgroups reports the sums through
// P.dbg[8 + 8*slot ..].  Perturbs the kernel (each read drains lgkmcnt); never built into the product.
#ifdef MSM_TICA_PROFILE
#define PROF_DECL long long pf_t = clock64(), pf_acc[6] = {0, 0, 0, 0, 0, 0}
#define PROF_MARK(i) { const long long pf_n = clock64(); pf_acc[i] += pf_n - pf_t; pf_t = pf_n; }
#else
#define PROF_DECL
#define PROF_MARK(i)
#endif

// ---- staging with an INTERIOR fast path ------------------------------------------------------
// Section timers showed that a wave's non-MFMA instructions run ~10x slower than their count
// suggests while the co-resident wave streams MFMAs (the matrix instruction monopolises the SIMD's
// issue port / register ports: ~100 VALU instructions of clamps, weights and address products cost
// 2,000+ cycles per K-step).  So the K-step is put on a diet.  A step is INTERIOR when none of its 32
// frames needs a clamp and all of them carry the same weight (97 % of the steps of a 10,000-frame
// trajectory): its 8 loads then use per-lane offsets that are CONSTANT for the whole chunk on top
// of a scalar base that advances by 32 rows (SALU), and its LDS store writes the loaded registers
// unchanged.  To make the Gram weight of an interior frame 1 instead of 2, Gram tiles accumulate
// HALF weights {0, 1/2, 1} (exact scalings) and the slab merge multiplies by 2 (exact): bit-identical
// results.  Loads stay unconditional; only VALU work sits inside the branch.
struct LaneOffs {
    unsigned a[4], b[4];  // (rr0 + 8 j) * ldb + column bytes, relative to the step's first row
};

template <bool VEC4>
__device__ __forceinline__ LaneOffs make_lane_offs(const ChunkCtx& cx, int F, int I0, int J0, int tid)
{
    LaneOffs o;
    const int c4 = (tid & 31) * 4;
    const int rr0 = tid >> 5;
    const int ca = I0 + c4 < F ? I0 + c4 : F - 4, cb = J0 + c4 < F ? J0 + c4 : F - 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o.a[j] = (unsigned)(rr0 + 8 * j) * cx.ldb + 4u * (unsigned)ca;
        o.b[j] = (unsigned)(rr0 + 8 * j) * cx.ldb + 4u * (unsigned)cb;
    }
    return o;
}

// uniform: may step k0 (32 frames) take the fast path?  wsel: 0 = lagged tile (weight [t < len - lag]),
// 1 = Gram tile (half weights: 1 needs lag <= t < len - lag)
__device__ __forceinline__ bool step_interior(const ChunkCtx& cx, int k0, int isG)
{
    const int last = k0 + BK32 - 1;
    bool ok = last <= cx.nmax && last <= cx.nmaxB && last < cx.hi;
    if (isG) ok = ok && k0 >= cx.lo && last < cx.n;
    return ok;
}

template <bool VEC4>
__device__ __forceinline__ void stage_load32x(Stage32<VEC4>& st, int& uniform, const ChunkCtx& cx, const LaneOffs& lo,
                                              int F, int k0, int isG, int tauB, int I0, int J0, int tid)
{
    if (!VEC4) {  // element-wise loads: no fast path
        stage_load32<VEC4>(st, cx, F, k0, isG, tauB, I0, J0, tid);
        if (isG) {
#pragma unroll
            for (int j = 0; j < 4; ++j) st.sc[j] *= 0.5f;
        }
        uniform = 0;
        return;
    }
    // the branch holds VALU/SALU work only; the 8 loads are issued after the join so that the
    // compiler keeps counting vmcnt across it
    global_ptr<char> pa = cx.base, pb = cx.baseB;
    unsigned oa[4], ob[4];
    if (step_interior(cx, k0, isG)) {
        pa += (size_t)k0 * cx.ldb;  // scalar
        pb += (size_t)k0 * cx.ldb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            oa[j] = lo.a[j];
            ob[j] = lo.b[j];
        }
        uniform = 1;
    } else {
        const int c4 = (tid & 31) * 4;
        const int rr0 = tid >> 5;
        const float wfull = isG ? 0.5f : 1.f;
        const unsigned ca = 4u * (unsigned)(I0 + c4 < F ? I0 + c4 : F - 4), cb = 4u * (unsigned)(J0 + c4 < F ? J0 + c4 : F - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = k0 + rr0 + 8 * j;
            float sc = (kr < cx.hi) ? wfull : 0.f;
            if (isG) sc += (kr >= cx.lo && kr < cx.n) ? 0.5f : 0.f;
            const int ra = kr < cx.nmax ? kr : cx.nmax;
            const int rb = kr < cx.nmaxB ? kr : cx.nmaxB;
            oa[j] = (unsigned)ra * cx.ldb + ca;
            ob[j] = (unsigned)rb * cx.ldb + cb;
            st.sc[j] = sc;
        }
        uniform = 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st.a[j] = load16_global<char>(pa + oa[j]);
        st.b[j] = load16_global<char>(pb + ob[j]);
    }
}

// addresses (and, on edge steps, weights) of the 8 loads of step k0 -- no load is issued here.
// VEC4: a scalar base pair + one 32-bit byte offset per load; !VEC4: the offset addresses the ROW,
// the four elements are fetched one by one with clamped columns (stage_ld).
struct StageAddr {
    global_ptr<char> pa, pb;
    unsigned oa[4], ob[4];
};

template <bool VEC4>
__device__ __forceinline__ void stage_addr32(StageAddr& sa, Stage32<VEC4>& st, int& uniform, const ChunkCtx& cx,
                                             const LaneOffs& lo, int F, int k0, int isG, int I0, int J0, int tid)
{
    sa.pa = cx.base;
    sa.pb = cx.baseB;
    if (VEC4 && step_interior(cx, k0, isG)) {
        sa.pa += (size_t)k0 * cx.ldb;  // scalar
        sa.pb += (size_t)k0 * cx.ldb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sa.oa[j] = lo.a[j];
            sa.ob[j] = lo.b[j];
        }
        uniform = 1;
    } else {
        const int c4 = (tid & 31) * 4;
        const int rr0 = tid >> 5;
        const float wfull = isG ? 0.5f : 1.f;
        const unsigned ca = VEC4 ? 4u * (unsigned)(I0 + c4 < F ? I0 + c4 : F - 4) : 0u;
        const unsigned cb = VEC4 ? 4u * (unsigned)(J0 + c4 < F ? J0 + c4 : F - 4) : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = k0 + rr0 + 8 * j;
            float sc = (kr < cx.hi) ? wfull : 0.f;
            if (isG) sc += (kr >= cx.lo && kr < cx.n) ? 0.5f : 0.f;
            const int ra = kr < cx.nmax ? kr : cx.nmax;
            const int rb = kr < cx.nmaxB ? kr : cx.nmaxB;
            sa.oa[j] = (unsigned)ra * cx.ldb + ca;
            sa.ob[j] = (unsigned)rb * cx.ldb + cb;
            st.sc[j] = sc;
        }
        uniform = 0;
    }
}

template <bool VEC4>
__device__ __forceinline__ float4 stage_ld(global_ptr<char> base, unsigned off, int F, int col)
{
    if (VEC4) return load16_global<char>(base + off);
    return load_row4<false>(base, off, col, F);
}

// shift (x - r), then apply the per-row weight (edge steps) and the column masks (partial tiles) to a loaded stage
// in place.  Interior steps of full tiles do not come here: their shift is applied inside the MFMA stream.
template <bool VEC4, bool PARTIAL>
__device__ __forceinline__ void stage_scale32(Stage32<VEC4>& st, int uniform, float4 ma, float4 mb, float4 ra, float4 rb)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sc = uniform ? 1.f : st.sc[j];
        const float4 a = make_float4(st.a[j].x - ra.x, st.a[j].y - ra.y, st.a[j].z - ra.z, st.a[j].w - ra.w);
        const float4 b = make_float4(st.b[j].x - rb.x, st.b[j].y - rb.y, st.b[j].z - rb.z, st.b[j].w - rb.w);
        if (PARTIAL) {
            st.a[j] = make_float4(a.x * (sc * ma.x), a.y * (sc * ma.y), a.z * (sc * ma.z), a.w * (sc * ma.w));
            st.b[j] = make_float4(b.x * mb.x, b.y * mb.y, b.z * mb.z, b.w * mb.w);
        } else {
            st.a[j] = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
            st.b[j] = b;
        }
    }
}

// x - f * r with f in {0, 1} (wave-uniform): exact product, so this is x - r or x bit for bit.  Two v_pk_fma_f32.
__device__ __forceinline__ float4 shift_fma4(float4 x, float4 r, float nf)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v n2 = {nf, nf};
    f2v lo, hi;  // (the builtin elementwise fma is split into scalar v_fma_f32)
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(n2), "v"(f2v{r.x, r.y}), "v"(f2v{x.x, x.y}));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(n2), "v"(f2v{r.z, r.w}), "v"(f2v{x.z, x.w}));
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// this thread's four columns of the reference row (clamped like the data loads; zeros without a shift)
__device__ __forceinline__ float4 load_shift4(const float* shift, int col, int F)
{
    if (!shift) return make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(shift[col + 0 < F ? col + 0 : F - 1], shift[col + 1 < F ? col + 1 : F - 1],
                       shift[col + 2 < F ? col + 2 : F - 1], shift[col + 3 < F ? col + 3 : F - 1]);
}

template <bool VEC4, bool PARTIAL>
__device__ __forceinline__ void stage_store32x(const Stage32<VEC4>& st, int uniform, float* As, float* Bs, int tid,
                                               float4 ma, float4 mb)
{
    if (!PARTIAL && uniform) {
        const int c4 = (tid & 31) * 4;
        const int rr0 = tid >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(As + (rr0 + 8 * j) * TM + c4) = st.a[j];
            *reinterpret_cast<float4*>(Bs + (rr0 + 8 * j) * TM + c4) = st.b[j];
        }
        return;
    }
    Stage32<VEC4> t = st;
    if (uniform) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t.sc[j] = 1.f;
    }
    stage_store32<VEC4, PARTIAL>(t, As, Bs, tid, ma, mb);
}

#ifndef MSM_CG_PRIO
#define MSM_CG_PRIO 1
#endif
#ifndef MSM_CG_PRIO_OFF
#define MSM_CG_PRIO_OFF 8
#endif
template <bool VEC4, bool PARTIAL>
__global__ __launch_bounds__(NT, 2) void tica_mfma_f32_kernel(TicaArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);  // [2][BK32][TM]
    float* Bs = As + 2 * BK32 * TM;              // [2][BK32][TM]

    const int tid = threadIdx.x;
    const int p = xcd_linear_id();
    const int cohort = p / P.ntiles, tile = p % P.ntiles;
    int I, J, isG;
    decode_tile(tile, P.T, I, J, isG);
    const int I0 = I * TM, J0 = J * TM;
    const int tauB = isG ? 0 : P.lag;

    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 5, cl = lane & 31;
    double* slab = P.slabs + (size_t)p * (TM * TM);

    // column masks of this thread's staging float4 (only partial tiles of F % 128 != 0 have zeros)
    const int c4 = (tid & 31) * 4;
    const float4 ma = make_float4(I0 + c4 + 0 < P.F ? 1.f : 0.f, I0 + c4 + 1 < P.F ? 1.f : 0.f,
                                  I0 + c4 + 2 < P.F ? 1.f : 0.f, I0 + c4 + 3 < P.F ? 1.f : 0.f);
    const float4 mb = make_float4(J0 + c4 + 0 < P.F ? 1.f : 0.f, J0 + c4 + 1 < P.F ? 1.f : 0.f,
                                  J0 + c4 + 2 < P.F ? 1.f : 0.f, J0 + c4 + 3 < P.F ? 1.f : 0.f);

    // mean shift: this thread's staging columns of the reference row r; both panels hold (x - r)
    const float4 ra = load_shift4(P.shift, I0 + c4, P.F), rb = load_shift4(P.shift, J0 + c4, P.F);

    f32x16 acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
    const double gscale = isG ? 2.0 : 1.0;  // Gram tiles accumulate half weights (see stage_load32x)
    int rows_acc = 0;
    int chunks_done = 0;
    if (P.dbg && blockIdx.x == 0 && tid == 0) {
        P.dbg[0] = clock64();
        P.dbg[2] = wall_clock64();
    }

    PROF_DECL;
    for (long long c = cohort; c < P.nchunks; c += P.S) {
        PROF_MARK(5)
        const TicaChunk ch = get_chunk(P, c);
        const int nsteps = (ch.n + BK32 - 1) / BK32;
        ChunkCtx cx = make_ctx(P, ch);
        set_lag(cx, tauB, sizeof(float), P.ld);
        // Register-staged software pipeline, TWO K-steps deep: while step s runs on the MFMA pipe
        // the panel of step s+1 sits in one register set (written to the other LDS buffer during
        // step s) and the loads of step s+2 go into the other.  (One step of lookahead is not
        // enough: the lagged panel misses L2 on first touch and an HBM round trip under load is as
        // long as a step.)
        // The 8 global loads and the 8 LDS writes of a step are interleaved INTO the unrolled MFMA
        // stream (k-pairs 0-3 and 8-15), where they issue in the shadow of this wave's own MFMAs;
        // issued in a block before / after the loop they wait on the CO-RESIDENT wave's MFMAs instead
        // (section timers: 17 % of the kernel).  Everything data-dependent -- edge clamps, weights,
        // column masks -- is resolved in two uniform branches at the top of the step that hold VALU
        // work only and are skipped on interior steps, so the stream itself is branch-free.
        Stage32<VEC4> st0, st1;
        int un0 = 0, un1 = 0;
        const LaneOffs lofs = make_lane_offs<VEC4>(cx, P.F, I0, J0, tid);
        stage_load32x<VEC4>(st0, un0, cx, lofs, P.F, 0, isG, tauB, I0, J0, tid);
        stage_scale32<VEC4, PARTIAL>(st0, un0, ma, mb, ra, rb);
        stage_store32x<VEC4, false>(st0, 1, As, Bs, tid, ma, mb);  // already shifted, weighted and masked
        stage_load32x<VEC4>(st0, un0, cx, lofs, P.F, BK32, isG, tauB, I0, J0, tid);
        if (P.cosync && chunks_done > 0) {
            if (tid == 0) {
                const unsigned target = (unsigned)P.ntiles * (unsigned)chunks_done;
                const long long t0 = clock64();
                while (__hip_atomic_load(P.cosync + cohort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (clock64() - t0 > 200000) break;  // ~90 us: give up, never hang
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        __syncthreads();
        PROF_MARK(0)  /* chunk prologue: descriptor, first two stage loads, first LDS store */
        const int srow = tid >> 5, scol = (tid & 31) * 4;  // this thread's staging row / column
#define MSM_TICA_STEP(SNEXT, UNEXT, SLOAD, ULOAD, BUF)                                            \
        {                                                                                         \
            const float* Ab = As + (BUF) * (BK32 * TM) + kl * TM + wr * 64 + cl;                  \
            const float* Bb = Bs + (BUF) * (BK32 * TM) + kl * TM + wc * 64 + cl;                  \
            float* Aw = As + ((BUF) ^ 1) * (BK32 * TM) + srow * TM + scol;                        \
            float* Bw = Bs + ((BUF) ^ 1) * (BK32 * TM) + srow * TM + scol;                        \
            /* addresses of step s+2 and (edge steps only) weights; no loads issued here */       \
            StageAddr sa;                                                                         \
            stage_addr32<VEC4>(sa, SLOAD, ULOAD, cx, lofs, P.F, (s + 2) * BK32, isG, I0, J0, tid); \
            /* step s+1's panel becomes what LDS must hold: weights / masks applied in registers */ \
            /* (shifted there too; interior steps of full tiles are shifted inside the stream)   */ \
            if (PARTIAL || !UNEXT) stage_scale32<VEC4, PARTIAL>(SNEXT, UNEXT, ma, mb, ra, rb);    \
            const float nfs = (PARTIAL || !UNEXT) ? 0.f : -1.f;                                   \
            PROF_MARK(1) /* step head */                                                          \
            /* fully unrolled: an inner loop makes the compiler's vmcnt bookkeeping give up and     */ \
            /* wait vmcnt(0) at the top of every step, which cuts the register pipeline to 1 step */ \
            _Pragma("unroll") for (int kk = 0; kk < BK32 / 2; ++kk) {                             \
                /* fragment reads run one k-pair ahead of the MFMAs that consume them -- across  */ \
                /* the step boundary too: before its last four MFMAs a step passes the barrier     */ \
                /* (every wave has written step s+1's panel by then) and fetches the first          */ \
                /* fragments of step s+1, so the next step starts without an LDS round trip        */ \
                if (kk == BK32 / 2 - 1) {                                                         \
                    __syncthreads();                                                              \
                    if (MSM_CG_PRIO) __builtin_amdgcn_s_setprio(MSM_CG_PRIO); /* as in the sum/difference kernel */ \
                }                                                                                 \
                if (MSM_CG_PRIO && kk == MSM_CG_PRIO_OFF) __builtin_amdgcn_s_setprio(0);          \
                const float* An = (kk == BK32 / 2 - 1) ? Ab + (((BUF) ^ 1) - (BUF)) * (BK32 * TM) : Ab + (kk + 1) * 2 * TM; \
                const float* Bn = (kk == BK32 / 2 - 1) ? Bb + (((BUF) ^ 1) - (BUF)) * (BK32 * TM) : Bb + (kk + 1) * 2 * TM; \
                const float na0 = An[0], na1 = An[32];                                            \
                const float nb0 = Bn[0], nb1 = Bn[32];                                            \
                if (kk < 4) { /* step s+2 -> registers */                                         \
                    SLOAD.a[kk] = stage_ld<VEC4>(sa.pa, sa.oa[kk], P.F, I0 + scol);               \
                    SLOAD.b[kk] = stage_ld<VEC4>(sa.pb, sa.ob[kk], P.F, J0 + scol);               \
                }                                                                                 \
                if (kk >= 7 && kk < 15 && (kk & 1) == 1) { /* shift one k-pair ahead of its store */ \
                    SNEXT.a[(kk - 7) / 2] = shift_fma4(SNEXT.a[(kk - 7) / 2], ra, nfs);           \
                    SNEXT.b[(kk - 7) / 2] = shift_fma4(SNEXT.b[(kk - 7) / 2], rb, nfs);           \
                }                                                                                 \
                if (kk >= 8 && (kk & 1) == 0) { /* step s+1 -> the other LDS buffer */            \
                    *reinterpret_cast<float4*>(Aw + ((kk - 8) / 2) * 8 * TM) = SNEXT.a[(kk - 8) / 2]; \
                    *reinterpret_cast<float4*>(Bw + ((kk - 8) / 2) * 8 * TM) = SNEXT.b[(kk - 8) / 2]; \
                }                                                                                 \
                __builtin_amdgcn_sched_barrier(0); /* keep the reads ABOVE the MFMAs they do not feed */ \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);     \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);     \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);     \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);     \
                __builtin_amdgcn_sched_barrier(0);                                                \
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;                                           \
            }                                                                                     \
            PROF_MARK(2) /* MFMA loop (with the barrier before its last k-pair) */                \
        }
        /* first fragments of step 0 (lane: frame kl, columns wr*64+cl / +32 of the tile) */
        float a0 = As[kl * TM + wr * 64 + cl], a1 = As[kl * TM + wr * 64 + cl + 32];
        float b0 = Bs[kl * TM + wc * 64 + cl], b1 = Bs[kl * TM + wc * 64 + cl + 32];
        for (int s = 0; s < nsteps; s += 2) {
            MSM_TICA_STEP(st0, un0, st1, un1, 0)
            ++s;
            if (s < nsteps) MSM_TICA_STEP(st1, un1, st0, un0, 1)
            --s;
        }
#undef MSM_TICA_STEP
        // Cohort pacing.  The cohort's workgroups read the SAME frames; left alone they drift apart
        // by more than the 4 MB L2 holds and every panel is re-fetched from the Infinity Cache
        // (measured 10x the algorithmic bytes).  A relaxed arrival counter per cohort, waited on
        // at chunk boundaries, keeps them within one chunk of each other.  No data is exchanged
        // (no fences needed) and the wait is BOUNDED: if a member is not resident the others
        // simply run on, so this can cost performance but never correctness or liveness.
        if (P.cosync) {
            ++chunks_done;  // arrive now, wait later (after the slab merge and the next chunk's prologue)
            if (tid == 0)
                __hip_atomic_fetch_add(P.cosync + cohort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // fp64 merge of the fp32 partial into the workgroup's private slab, once <= KFLUSH frames
        // are in the registers.  Per 64x32 half all 32 loads are issued before the first add/store
        // (a plain `*q += x` loop compiles to 64 dependent round trips); addresses are a
        // wave-uniform base plus ONE 32-bit per-lane offset so they cost no VGPR pairs.
        rows_acc += ch.n;
        if (rows_acc + P.kc > KFLUSH || c + P.S >= P.nchunks) {
            rows_acc = 0;
            unsigned toff = (unsigned)((wr * 64 + 4 * kl) * TM + wc * 64 + cl);
            // opaque to the optimiser: otherwise the 64 slab addresses are hoisted out of the chunk
            // loop as loop invariants (128 VGPRs -> scratch spills in the MFMA loop)
            asm volatile("" : "+v"(toff));
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                double old[2][16];
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const double* q = slab + (bi * 32 + (r & 3) + 8 * (r >> 2)) * TM + bj * 32;
                        old[bj][r] = q[toff];
                    }
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        double* q = slab + (bi * 32 + (r & 3) + 8 * (r >> 2)) * TM + bj * 32;
                        q[toff] = old[bj][r] + gscale * (double)acc[bi][bj][r];
                        acc[bi][bj][r] = 0.f;
                    }
            }
        }
    }
    PROF_MARK(4) /* since the last step: slab merges (and the idle tail of the last chunk) */
#ifdef MSM_TICA_PROFILE
    if (P.dbg && tid == 0 && (blockIdx.x < 3 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) {
        const int slot = blockIdx.x < 3 ? blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 3 : 4);
        for (int i = 0; i < 6; ++i) P.dbg[8 + 8 * slot + i] = pf_acc[i];
    }
#endif
    if (P.dbg && blockIdx.x == 0 && tid == 0) {
        P.dbg[1] = clock64();
        P.dbg[3] = wall_clock64();
    }
}

}  // namespace msm
