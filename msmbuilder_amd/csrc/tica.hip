// tica.hip -- time-lagged second-moment accumulation for tICA on gfx950.
//
// Replaces the body of tICA._fit (/root/reference/msmbuilder/decomposition/tica.py:401-424):
//   C  += X[:-tau].T @ X[tau:]                                  (:417)
//   G  += X[:-tau].T @ X[:-tau] + X[tau:].T @ X[tau:]           (:421-422; only the sum is read, :245)
//   s0 += X[:-tau].sum(0),  stau += X[tau:].sum(0)              (:418-419)
// The reference does three float64 dgemm per trajectory; here C and G are ONE
// MFMA kernel over frame-major X (X is read as both operands: A = X^T needs no
// transpose because the MFMA A and B fragments are both k-major):
//   * C tile (I,J):  sum_t  a(t) * X[t, I]^T X[t+tau, J],  a(t) = [t < len-tau]
//   * G tile (I<=J): sum_t  w(t) * X[t, I]^T X[t, J],      w(t) = [t < len-tau] + [t >= tau]
//     (w in {0,1,2} is exact in fp32, so S0+Stau needs no head/tail correction pass;
//      the lower triangle is mirrored at export).
// Decomposition: a persistent grid of S cohorts x ntiles workgroups (<= resident
// slots, one round, no tail).  Every workgroup owns ONE 128x128 output tile for
// its whole life and walks the frame chunks c = cohort, cohort+S, ...; a chunk is
// <= 4096 frames of one trajectory, accumulated in fp32 MFMA registers and then
// merged in fp64 into the workgroup's private slab (no atomics, deterministic).
// The cohort's workgroups read the same frames at the same time and are placed on
// one XCD where possible, so the 13x panel re-read (F=512) is L2 traffic, not HBM.
#include "common.h"
#include "tica_img_dev.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace msm {

constexpr int TM = 128;     // output tile is TM x TM features
constexpr int NT = 256;     // threads per workgroup: 4 waves as 2x2, 64x64 outputs per wave
constexpr int BK32 = 32;    // frames per K-step, fp32 kernel
constexpr int BK64 = 16;    // frames per K-step, fp64 kernel
constexpr int KCMAX = 4096; // max frames per chunk (load-balance granule)
constexpr int KFLUSH = 8192; // max frames accumulated in fp32 registers before an fp64 merge
constexpr int NCB = 1024;   // column-sum partial slots (4 blocks per CU)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct TicaChunk {
    const void* base;  // row 0 of the trajectory
    long long row0;    // first row of this chunk inside the trajectory
    long long len;     // trajectory length
    int n;             // rows in this chunk
    int pad;
    long long last;    // last ADDRESSABLE row of the trajectory's storage (len - 1, or the end of the
                       // slice a rank holds when one long trajectory is split over ranks)
    long long g0;      // bf16 image path: first 8-pair group of this chunk in the packed image
};

struct TicaArgs {
    const TicaChunk* chunks;  // device table, or nullptr -> `single` split arithmetically by kc
    TicaChunk single;
    long long nchunks;
    long long ld;
    int kc;
    int F, lag, T, ntiles, S;
    double* slabs;    // [S*ntiles][TM*TM] fp64, owned per workgroup
    double* colpart;  // [NCB][2][F] fp64 partial column sums (temporary buffer)
    int* flag;        // sticky non-finite flag
    unsigned* cosync; // [S] per-cohort arrival counters (zeroed per launch): keeps a cohort's workgroups within one chunk of each other
    long long* dbg;   // profiling only: [shader clock start, end, 100 MHz wall start, end] of workgroup 0
    const float* shift; // [F] per-column reference row r (or nullptr): the fp32 / bf16 kernels accumulate (x - r), see "mean shift"
    int kflush;         // sum/difference kernel: frames accumulated in fp32 registers before the fp64 slab merge
    const float* zrow;  // [F] zeros: where the dummy loads of a non-staging half-step read when the column sums are folded
    double* colA;       // sum/difference kernel with folded column sums: [S (+ 1)][F] fp64 sums of the LEFT frames, one row per cohort
    long long n_main;   // sum/difference kernel, REM: chunks [0, n_main) belong to the whole cohorts, the rest to the remainder cohort
};

__device__ __forceinline__ TicaChunk get_chunk(const TicaArgs& P, long long c)
{
    if (P.chunks) return P.chunks[c];
    TicaChunk ch = P.single;
    ch.row0 = c * (long long)P.kc;
    long long rem = ch.len - ch.row0;
    ch.n = (int)(rem < P.kc ? rem : P.kc);
    return ch;
}

// persistent block id -> (cohort, tile); blocks land on XCD (blockIdx % 8), so remap to
// make consecutive p (= one cohort's tiles) share an XCD's L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_linear_id()
{
    const int G = gridDim.x, b = blockIdx.x;
    const int q = G / 8, r = G % 8, xcd = b % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

__device__ __forceinline__ void decode_tile(int tile, int T, int& I, int& J, int& isG)
{
    if (tile < T * T) {
        isG = 0;
        I = tile / T;
        J = tile % T;
    } else {
        isG = 1;
        int u = tile - T * T;
        I = 0;
        while (u >= T - I) {
            u -= T - I;
            ++I;
        }
        J = I + u;
    }
}

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

// ---------------------------------------------------------------------------
// bf16 image path (modes `bf16` / `bf16x2`, BASELINE configs[4]): the sum/difference form of 3.1b on the bf16 matrix
// pipe, in two kernels.
//
//  1. tica_img_kernel: ONE streaming pass turns the frame-major trajectories into a packed bf16 IMAGE of the
//     pair frames u = (x_t - r) + (x_{t+tau} - r) and d = x_t - x_{t+tau} (formed in fp32; r = mean shift row), laid out as
//     the bf16 MFMA wants its operands: 16-byte packets [8 consecutive pairs] of one feature, [pair group][feature][8].
//     The lag, the trajectory edges (a trajectory's pairs are padded with zero packets to a whole K-step), the shift, the
//     fp32 -> bf16 rounding (RNE; bf16x2: hi + mid = 16 significant bits, mid image alongside) and partial feature tiles
//     (the image is zero-padded to a multiple of 256 features) are all handled HERE, once per element.  bf16-STORED
//     trajectories (dtype_bytes = 2) enter through the same kernel.  Traffic: F sizeof(T) read (+ the lagged row, an
//     L2 hit) and 4 B (bf16x2: 8 B) written per pair and feature.
//  2. tica_img_mfma_kernel: H = sum u u^T and D = sum d d^T on the upper tiles, v_mfma_f32_32x32x16_bf16, straight from
//     the image: both operands of a product come from the SAME image at the SAME pair index, so there is no lag, no
//     edge, no mask and no conversion left in the hot loop -- 16-byte packets go global -> (registers) -> LDS unchanged and
//     come back as conflict-free ds_read_b128 fragments.  A workgroup is 8 waves and owns a 256 x 256 tile of H or of D
//     (wave: 64 x 128 outputs, 128 fp32 accumulators): per 32-pair K-step it stages 32 KiB for 128 MFMAs, HALF the
//     L2 -> LDS bytes per flop of the 128 x 128 tiles of round 1 (whose bf16 kernel sat at 0.11 of the bf16 peak, bound by
//     exactly that traffic plus the in-register transpose).  bf16x2 forms hi.hi + hi.mid + mid.hi + mid.mid per 16 pairs.
//     fp32 partials go to the fp64 slabs of the sum/difference layout every <= 8192 pairs; export and un-shift are 3.1b's.
// ---------------------------------------------------------------------------
struct ImgArgs {
    const TicaChunk* chunks;
    long long nchunks;
    long long ld;
    int F, Fp, lag, dtype_bytes;
    const float* shift;
    long long g_off; // first 8-pair group of the super-chunk being packed: the ring slot holds groups [g_off, g_off + G)
    bf16x8* u_hi;   // [G][Fp] packets
    bf16x8* d_hi;
    bf16x8* u_mid;  // bf16x2 only
    bf16x8* d_mid;
    double* colA;   // folded column sums: [nchunks][F] fp64 sums of the chunk's LEFT frames (nullptr: a separate pass made them)
};

// thread -> 4 consecutive features (one 16-byte load per row for float32, 8 bytes for bfloat16) x the 8 pairs of one group:
// 16 row loads in flight, an 8 x 4 transpose in registers, four 16-byte packets per image written back to back (a wave
// writes 4 KiB contiguous).  A workgroup handles one K-step (4 groups) of a 256-feature block per iteration.
// (nontemporal stores of the image were tried here: no change, 11.6 -> 11.6 ms per 1M x 2048 fit)
#define IMG_PACK_STORE(PTR, V) (*(PTR) = (V))
template <bool X2>
__global__ __launch_bounds__(256) void tica_img_kernel(ImgArgs P)
{
    const TicaChunk ch = P.chunks[blockIdx.x];
    const int fq = threadIdx.x & 63, gq = threadIdx.x >> 6;   // feature quad, group within the K-step
    const int f0 = blockIdx.y * 256 + fq * 4;                 // < Fp
    const bool vec = (P.F % 4 == 0) && (P.ld % 4 == 0) && ((((uintptr_t)ch.base) & 15) == 0);
    const bool vec2 = (P.F % 4 == 0) && (P.ld % 4 == 0) && ((((uintptr_t)ch.base) & 7) == 0);   // bfloat16 rows: 8-byte loads
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (P.shift && f0 + q < P.F) ? P.shift[f0 + q] : 0.f;
    long long nv = ch.len - P.lag - ch.row0;   // valid pairs of this chunk: left frames row0 .. with t < len - lag
    if (nv > ch.n) nv = ch.n;
    if (nv < 0) nv = 0;
    const long long nsteps = (nv + 31) / 32;   // whole K-steps of 32 pairs, zero padded
    const size_t esz = (size_t)P.dtype_bytes;
    const global_ptr<char> base = as_global<char>(ch.base);
    const int fc = f0 < P.F ? f0 : (P.F >= 4 ? P.F - 4 : 0);  // clamped column of the vector loads
    double cs[4] = {0.0, 0.0, 0.0, 0.0};   // P.colA: fp64 sums of the left frames this thread loads (its four features)
    __shared__ bf16x8 img_stage[4][X2 ? 4 : 2][256];   // per wave: the packets of its group, one tile per image
    for (long long st = 0; st < nsteps; ++st) {
        const long long gi = st * 4 + gq;
        float a[8][4], b[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long pidx = gi * 8 + e;
            const long long t = ch.row0 + (pidx < nv ? pidx : 0);
            const global_ptr<char> rowa = base + (size_t)t * (size_t)P.ld * esz;
            const global_ptr<char> rowb = rowa + (size_t)P.lag * (size_t)P.ld * esz;
            if (P.dtype_bytes == 4 && vec) {
                const raw_f32x4 va = *(global_ptr<raw_f32x4>)(rowa + (size_t)fc * 4), vb = *(global_ptr<raw_f32x4>)(rowb + (size_t)fc * 4);
                a[e][0] = va.x; a[e][1] = va.y; a[e][2] = va.z; a[e][3] = va.w;
                b[e][0] = vb.x; b[e][1] = vb.y; b[e][2] = vb.z; b[e][3] = vb.w;
            } else if (P.dtype_bytes == 2 && vec2) {
                // four bfloat16 = one 8-byte load (element-wise 2-byte loads made this pre-pass 2.1 TB/s on bfloat16-stored
                // input against 3.8 TB/s on float32)
                typedef unsigned raw_u32x2 __attribute__((ext_vector_type(2)));
                const raw_u32x2 va = *(global_ptr<raw_u32x2>)(rowa + (size_t)fc * 2), vb = *(global_ptr<raw_u32x2>)(rowb + (size_t)fc * 2);
                a[e][0] = __uint_as_float(va.x << 16); a[e][1] = __uint_as_float(va.x & 0xffff0000u);
                a[e][2] = __uint_as_float(va.y << 16); a[e][3] = __uint_as_float(va.y & 0xffff0000u);
                b[e][0] = __uint_as_float(vb.x << 16); b[e][1] = __uint_as_float(vb.x & 0xffff0000u);
                b[e][2] = __uint_as_float(vb.y << 16); b[e][3] = __uint_as_float(vb.y & 0xffff0000u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t col = (size_t)(f0 + q < P.F ? f0 + q : P.F - 1);   // clamped: masked below
                    if (P.dtype_bytes == 4) {
                        a[e][q] = *(global_ptr<float>)(rowa + col * 4);
                        b[e][q] = *(global_ptr<float>)(rowb + col * 4);
                    } else {
                        a[e][q] = (float)*(global_ptr<__bf16>)(rowa + col * 2);
                        b[e][q] = (float)*(global_ptr<__bf16>)(rowb + col * 2);
                    }
                }
            }
        }
        if (P.colA) {   // this pre-pass is bandwidth-bound: the 64 widenings and adds per thread and step are free
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool okp = (gi * 8 + e) < nv;
#pragma unroll
                for (int q = 0; q < 4; ++q) cs[q] += okp ? (double)a[e][q] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 uh, dh, um, dm;
            const bool inF = f0 + q < P.F;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = inF && (gi * 8 + e) < nv;
                float u, d;
                if (P.dtype_bytes == 2) {
                    // bfloat16 rows: a + b and a - b are exact in fp32 (8-bit significands), so ONE rounding each -- and the
                    // arithmetic of the fused kernel (tica_img_dev.h), which this path must match bit for bit
                    u = ok ? (a[e][q] + b[e][q]) - 2.f * r[q] : 0.f;
                    d = ok ? a[e][q] - b[e][q] : 0.f;
                } else {
                    // float32 rows: x - r first (exact by Sterbenz when |mean| >> std, the case the shift exists for)
                    const float ya = ok ? a[e][q] - r[q] : 0.f, yb = ok ? b[e][q] - r[q] : 0.f;
                    u = ya + yb;
                    d = ya - yb;
                }
                const __bf16 u1 = (__bf16)u, d1 = (__bf16)d;
                uh[e] = u1;
                dh[e] = d1;
                if (X2) {
                    um[e] = (__bf16)(u - (float)u1);
                    dm[e] = (__bf16)(d - (float)d1);
                }
            }
            // Round 5: the lane's four packets (64 contiguous bytes per image) go through an LDS staging tile and leave as
            // 1 KiB-contiguous wave stores.  Stored straight from the lane, a store instruction wrote 16 bytes of every 64
            // (lane stride 64 B): four partial passes over each cache line.  XOR swizzle: conflict-free both ways.
            const int sl = 4 * fq + q;
            img_stage[gq][0][sl ^ ((sl >> 3) & 7)] = uh;
            img_stage[gq][1][sl ^ ((sl >> 3) & 7)] = dh;
            if (X2) {
                img_stage[gq][2][sl ^ ((sl >> 3) & 7)] = um;
                img_stage[gq][3][sl ^ ((sl >> 3) & 7)] = dm;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave's LDS operations complete in order: its own packets are all there)
        __builtin_amdgcn_wave_barrier();
        {
            const size_t o0 = (size_t)(ch.g0 - P.g_off + gi) * (size_t)P.Fp + (size_t)blockIdx.y * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sr = 64 * i + fq;
                const int sp = sr ^ ((sr >> 3) & 7);
                IMG_PACK_STORE(P.u_hi + o0 + sr, img_stage[gq][0][sp]);
                IMG_PACK_STORE(P.d_hi + o0 + sr, img_stage[gq][1][sp]);
                if (X2) {
                    IMG_PACK_STORE(P.u_mid + o0 + sr, img_stage[gq][2][sp]);
                    IMG_PACK_STORE(P.d_mid + o0 + sr, img_stage[gq][3][sp]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read out before the next step's packets overwrite it
        __builtin_amdgcn_wave_barrier();
    }
    if (P.colA) {   // the four groups of a feature quad -> one sum per (chunk, feature): plain stores, one writer each
        __shared__ double red[4][64][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) red[gq][fq][q] = cs[q];
        __syncthreads();
        if (gq == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (f0 + q < P.F)
                    P.colA[(size_t)blockIdx.x * P.F + f0 + q] = (red[0][fq][q] + red[1][fq][q]) + (red[2][fq][q] + red[3][fq][q]);
        }
    }
}

// (ImgMfmaArgs and the MFMA kernels of the image path: tica_img_dev.h)

// Round 5, fused kernel (tica_img_dev.h): the K-step records {row of the step's first pair, valid pairs} from the chunk table.
// A chunk's pairs are padded to whole 32-pair steps exactly as tica_img_kernel padded the image (g0 = the chunk's first
// 8-pair group); bf16x2 splits a 32-pair step into two 16-pair steps (the second may hold no pair: nvalid 0).
__global__ void tica_img_steps_kernel(const TicaChunk* __restrict__ chunks, long long ld, int lag, int x2, ImgStep* __restrict__ steps)
{
    const TicaChunk ch = chunks[blockIdx.x];
    long long nv = ch.len - lag - ch.row0;
    if (nv > ch.n) nv = ch.n;
    if (nv < 0) nv = 0;
    const long long n32 = (nv + 31) / 32, first = ch.g0 / 4;
    for (long long j = threadIdx.x; j < n32; j += blockDim.x) {
        const char* base = (const char*)ch.base + (size_t)(ch.row0 + j * 32) * (size_t)ld * 2;
        const int n = (int)(nv - j * 32 < 32 ? nv - j * 32 : 32);
        if (!x2) {
            steps[first + j] = ImgStep{base, n, 0};
        } else {
            steps[2 * (first + j)] = ImgStep{base, n < 16 ? n : 16, 0};
            steps[2 * (first + j) + 1] = n > 16 ? ImgStep{base + (size_t)16 * (size_t)ld * 2, n - 16, 0} : ImgStep{base, 0, 0};
        }
    }
}

// ---------------------------------------------------------------------------
// Column sums s0 / stau (tica.py:418-419) + the finite check of
// utils/validation.py:68-74, one streaming pass, fp64 accumulation.
// Block b owns partial slot b and walks chunks b, b+grid, ...
// ---------------------------------------------------------------------------
__device__ __forceinline__ double to_f64(float v) { return (double)v; }
__device__ __forceinline__ double to_f64(double v) { return v; }
__device__ __forceinline__ double to_f64(__bf16 v) { return (double)(float)v; }

template <typename TIn>
__global__ __launch_bounds__(NT) void tica_colsum_kernel(TicaArgs P)
{
    // thread -> a group of CW consecutive columns (one 16-byte load per row when aligned) and a
    // row lane; RU rows are kept in flight per thread so the pass is HBM-bound, not latency-bound
    constexpr int CW = 16 / sizeof(TIn);
    constexpr int RU = 8;
    __shared__ double red[2][NT][CW];
    const int tid = threadIdx.x;
    const int ngroups = (P.F + CW - 1) / CW;
    int cpb = 1;
    while (cpb < ngroups && cpb < NT) cpb <<= 1;  // column groups per pass (power of two <= 256)
    const int rl = NT / cpb;                      // row lanes
    const int tc = tid % cpb, tr = tid / cpb;
    const bool vec = (P.F % CW == 0) && (P.ld % CW == 0);
    double* part = P.colpart + (size_t)blockIdx.x * 2 * P.F;
    int bad = 0;
    for (int g0 = 0; g0 < ngroups; g0 += cpb) {
        const int col = (g0 + tc) * CW;
        double s0[CW], st[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) s0[e] = st[e] = 0.0;
        if (col < P.F) {
            for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
                const TicaChunk ch = get_chunk(P, c);
                const global_ptr<TIn> X = as_global<TIn>(ch.base);
                const bool al = vec && ((((uintptr_t)ch.base) & 15) == 0);
                for (int k0 = tr; k0 < ch.n; k0 += rl * RU) {
                    TIn v[RU][CW];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int kr = k0 + u * rl;
                        const long long r = ch.row0 + kr;
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[u][e] = (TIn)0.f;
                        if (kr < ch.n) {
                            const global_ptr<TIn> p = X + r * P.ld + col;
                            if (al) {
                                *reinterpret_cast<float4*>(&v[u][0]) = load16_global<TIn>(p);
                            } else {
#pragma unroll
                                for (int e = 0; e < CW; ++e)
                                    if (col + e < P.F) v[u][e] = p[e];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int kr = k0 + u * rl;
                        const long long r = ch.row0 + kr;
                        const bool in0 = (kr < ch.n) && (r < ch.len - P.lag);
                        const bool in1 = (kr < ch.n) && (r >= P.lag);
#pragma unroll
                        for (int e = 0; e < CW; ++e) {
                            const double x = to_f64(v[u][e]);
                            bad |= !isfinite(x);
                            if (in0) s0[e] += x;
                            if (in1) st[e] += x;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            red[0][tid][e] = s0[e];
            red[1][tid][e] = st[e];
        }
        __syncthreads();
        if (tr == 0 && col < P.F) {
            for (int k = 1; k < rl; ++k)
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    s0[e] += red[0][k * cpb + tc][e];
                    st[e] += red[1][k * cpb + tc][e];
                }
#pragma unroll
            for (int e = 0; e < CW; ++e)
                if (col + e < P.F) {
                    part[col + e] += s0[e];
                    part[P.F + col + e] += st[e];
                }
        }
        __syncthreads();
    }
    if (bad) atomicOr(P.flag, 1);
}

// colpart (persistent) += coltmp, then coltmp = 0
__global__ void tica_colmerge_kernel(double* __restrict__ dst, double* __restrict__ tmp, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dst[i] += tmp[i];
        tmp[i] = 0.0;
    }
}

// ---- folded column sums (sum/difference kernel, FOLD) ---------------------------------------------------------------
// r for a handle's first launch when no column-sum pass runs ahead of the MFMA kernel: the mean of up to FOLD_NS frames
// spread evenly over the launch's chunks (any r within a fraction of sigma of the mean serves: the shifted moments are
// restored exactly whatever r is).  One block per 64 columns, four row lanes, fp64.
constexpr int FOLD_NS = 4096, FOLD_NB = 32;   // samples, and the blocks (per 64 columns) that share them
template <typename TIn>
__global__ __launch_bounds__(256) void tica_fold_sample_kernel(TicaArgs P, double* __restrict__ part)
{
    __shared__ double red[256];
    constexpr int PER = FOLD_NS / FOLD_NB / 4;   // samples per row lane
    const int tid = threadIdx.x, col = blockIdx.x * 64 + (tid & 63), rl = tid >> 6;
    double a = 0.0;
    if (col < P.F)
        for (int i0 = 0; i0 < PER; i0 += 8) {
            TIn v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = (blockIdx.y * 4 + rl) * PER + i0 + u;
                const TicaChunk ch = get_chunk(P, ((long long)s * P.nchunks) / FOLD_NS);
                const int row = (int)((((unsigned)s * 2654435761u) >> 8) % (unsigned)ch.n);
                v[u] = as_global<TIn>(ch.base)[(ch.row0 + row) * P.ld + col];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += to_f64(v[u]);
        }
    red[tid] = a;
    __syncthreads();
    if (rl == 0 && col < P.F) part[(size_t)blockIdx.y * P.F + col] = a + red[tid + 64] + red[tid + 128] + red[tid + 192];
}

__global__ void tica_fold_setr_kernel(const double* __restrict__ part, float* __restrict__ r, int F)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= F) return;
    double v[FOLD_NB], a = 0.0;
#pragma unroll
    for (int k = 0; k < FOLD_NB; ++k) v[k] = part[(size_t)k * F + col];
#pragma unroll
    for (int k = 0; k < FOLD_NB; ++k) a += v[k];
    r[col] = (float)(a / (double)FOLD_NS);
}

// After the FOLD kernel: colA[c][:] = sums of cohort c's left frames; tmp (the [NCB][2][F] temporary partials) holds what a
// column-sum pass over the trajectories' first and last tau rows left there: [k][0] = a_k (rows [0, tau)), [k][1] = b_k
// (rows [len - tau, len)).  s0 = sum A, stau = sum of the right frames = A - a + b, so slot k becomes
// [A_k | A_k - a_k + b_k] (A_k = 0 beyond the S cohorts) -- the layout an ordinary column-sum pass leaves.  A non-finite
// A_k raises the flag (the boundary pass checked its own rows element by element).
// the bf16 image path's variant: colA[c][:] per CHUNK (tica_img_kernel); slot k takes chunks k, k + NCB, ... in order
__global__ void tica_fold_fix_img_kernel(double* __restrict__ tmp, const double* __restrict__ colA, int F, long long nchunks, int* flag)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)NCB * F) return;
    const int k = (int)(idx / F), col = (int)(idx - (size_t)k * F);
    double A = 0.0;
    for (long long c = k; c < nchunks; c += NCB) A += colA[(size_t)c * F + col];
    double* t = tmp + (size_t)k * 2 * F;
    const double a = t[col], b = t[F + col];
    t[col] = A;
    t[F + col] = (A - a) + b;
    if (!isfinite(A)) atomicOr(flag, 1);
}

__global__ void tica_fold_fix_kernel(double* __restrict__ tmp, const double* __restrict__ colA, int F, int S, int* flag)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)NCB * F) return;
    const int k = (int)(idx / F), col = (int)(idx - (size_t)k * F);
    const double A = k < S ? colA[(size_t)k * F + col] : 0.0;
    double* t = tmp + (size_t)k * 2 * F;
    const double a = t[col], b = t[F + col];
    t[col] = A;
    t[F + col] = (A - a) + b;
    if (!isfinite(A)) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// Mean shift.  The covariance is G / 2N - mu mu^T (tica.py:228-259): an error of eps * |G| in an fp32-accumulated
// G is a RELATIVE covariance error of eps * (mu / sigma)^2, i.e. 1e-3 for features whose mean is 100 standard
// deviations (contact and atom-pair distances), where the reference -- float64 throughout, tica.py:402 -- loses nothing.
// So the fp32 and bf16 kernels accumulate the moments of y = x - r for a per-handle reference row r (fp32, the column
// mean of the first launch: the column-sum pass runs before the MFMA pass anyway), whose entries are sigma-sized, and
// the raw moments are restored in fp64 at export time from the exact fp64 column sums:
//     C = C' + A' r^T + r B'^T + n r r^T          A' = A - n r,  B' = B - n r      (A, B: sums of the left / right frames
//     G = G' + W' r^T + r W'^T + nW r r^T         W' = W - nW r                     of the n shifted pairs; W, nW: weighted
// frame sum and total weight of the Gram term -- A + B and 2n except when a trajectory is split over ranks, where the
// C/G kernel's Gram tiles own FRAMES, not pairs).  r never changes while a handle accumulates, so launches add up.
// ---------------------------------------------------------------------------
// part: the [NCB][2][F] column-sum partials (a = "s0" half, b = "stau" half) of ONE column-sum launch; `what` says where
// they go: SH_A_a: A += a, SH_B_b: B += b, SH_W_ab: W += a + b, SH_B_a: B += a, SH_W_a: W += a.
//   whole trajectories                       A|B_b|W_ab   (left sums, right sums, both)
//   segments, owned rows, C/G or bf16 kernel A|W_ab       (the Gram tiles weight the OWNED frames)
//   segments, owned rows, H/D kernel         A|W_a        (its Gram is over owned PAIRS: W = A + B)
//   segments, the pairs' right rows          B_a (|W_a for the H/D kernel)
enum { SH_A_a = 1, SH_B_b = 2, SH_W_ab = 4, SH_B_a = 8, SH_W_a = 16 };
__global__ __launch_bounds__(512) void tica_shift_kernel(const double* __restrict__ part, double* __restrict__ shsum,
                                                         float* __restrict__ r, int F, double inv_n, int set_r, int what)
{
    // 64 columns per workgroup of 512; thread = (column, half a / b, one of four row lanes) with ONE accumulator and sixteen
    // partials requested per trip -- the shape tica_export_cols_kernel has.  (Round 3's loop -- four row lanes, `a += ...; b += ...`
    // under `if (col < F)` -- compiled to pairs of loads each waited for on the spot: 256 dependent round trips per
    // thread, 84 us per launch; two batched accumulators per thread were paired up again by the scheduler.)
    __shared__ double red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, half = (tid >> 6) & 1, rl = tid >> 7;
    const int col = blockIdx.x * 64 + lane;
    const int cc = col < F ? col : F - 1;
    static_assert(NCB % 64 == 0, "whole groups of 16 per row lane");
    double acc = 0.0;
    for (int k0 = rl; k0 < NCB; k0 += 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(k0 + 4 * u) * 2 * F + (size_t)half * F + cc];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    red[half][rl][lane] = acc;
    __syncthreads();
    if (tid < 64 && col < F) {
        const double a = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        const double b = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        if (set_r) r[col] = (float)((a + b) * inv_n);
        if (what & SH_A_a) shsum[col] += a;
        if (what & SH_B_b) shsum[F + col] += b;
        if (what & SH_B_a) shsum[F + col] += a;
        if (what & SH_W_ab) shsum[2 * F + col] += a + b;
        if (what & SH_W_a) shsum[2 * F + col] += a;
    }
}

__global__ void tica_unshift_kernel(double* __restrict__ packed, const double* __restrict__ shsum,
                                    const float* __restrict__ r, double n, double nW, int F, int sym)
{
    const size_t FF = (size_t)F * F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * FF) return;
    const int type = idx >= FF;
    const size_t e = idx - (type ? FF : 0);
    int i = (int)(e / F), j = (int)(e % F);
    if ((type || sym) && i > j) {  // symmetric corrections: evaluate the mirrored element with the SAME operand order, so
        const int t = i;           // the result is symmetric bit for bit whatever the compiler contracts into FMAs
        i = j;
        j = t;
    }
    const double ri = (double)r[i], rj = (double)r[j];
    double v;
    if (type) {
        const double wi = shsum[2 * F + i] - nW * ri, wj = shsum[2 * F + j] - nW * rj;
        v = (ri * wj + wi * rj) + nW * ri * rj;
    } else {
        const double ai = shsum[i] - n * ri, aj = shsum[j] - n * rj;
        const double bi = shsum[F + i] - n * ri, bj = shsum[F + j] - n * rj;
        if (sym)
            v = 0.5 * ((ai * rj + aj * ri) + (ri * bj + rj * bi)) + n * ri * rj;
        else
            v = (ai * rj + ri * bj) + n * ri * rj;
    }
    packed[idx] += v;
}

// packed[C | G | s0 | stau | n_obs | n_seq] = base + sum over slabs / column partials
__global__ void tica_export_kernel(const double* __restrict__ slabs, const double* __restrict__ colpart,
                                   const double* __restrict__ base, double* __restrict__ out, int F,
                                   int T, int ntiles, int S)
{
    const size_t FF = (size_t)F * F;
    const size_t total = 2 * FF + 2 * (size_t)F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    double v = base[idx];
    if (idx < 2 * FF) {
        const int type = idx >= FF;
        const size_t e = idx - (type ? FF : 0);
        int i = (int)(e / F), j = (int)(e % F);
        int ti = i / TM, tj = j / TM;
        int tile;
        if (type == 0) {
            tile = ti * T + tj;
        } else {
            if (i > j) {  // lower triangle (also inside a diagonal tile): mirror of the upper element, so the
                          // result is exactly symmetric whatever the kernel's product order was
                int t = i; i = j; j = t;
                t = ti; ti = tj; tj = t;
            }
            // upper-triangle tiles are enumerated row by row: (0,0..T-1), (1,1..T-1), ...
            tile = T * T + ti * T - ti * (ti - 1) / 2 + (tj - ti);
        }
        const size_t off = (size_t)(i % TM) * TM + (j % TM);
        int s = 0;
        for (; s + 4 <= S; s += 4) {   // four slabs' loads in flight per trip
            double q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = slabs[((size_t)(s + u) * ntiles + tile) * (TM * TM) + off];
#pragma unroll
            for (int u = 0; u < 4; ++u) v += q[u];
        }
        for (; s < S; ++s) v += slabs[((size_t)s * ntiles + tile) * (TM * TM) + off];
    } else {
        return;   // [s0 | stau]: tica_export_cols_kernel (NCB partials per column: a reduction, not a per-thread loop)
    }
    out[idx] = v;
}

// out[2 F^2 + e] = base[2 F^2 + e] + sum over the NCB column partials, e in [s0 | stau].  64 columns per workgroup, four
// waves take a quarter of the partials each with 16 loads in flight per trip (the per-thread loop over all 1024 partials
// was 1024 dependent L2 round trips: 70 us of a 2 ms solve), summed in partial order (deterministic).
__global__ __launch_bounds__(256) void tica_export_cols_kernel(const double* __restrict__ colpart, const double* __restrict__ base,
                                                               double* __restrict__ out, int F)
{
    __shared__ double red[4][64];
    const size_t FF = (size_t)F * F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool in = e < 2 * F;
    double acc = 0.0;
    constexpr int PER = NCB / 4;
    for (int b0 = wave * PER; b0 < (wave + 1) * PER; b0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in ? colpart[(size_t)(b0 + u) * 2 * F + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && in) out[2 * FF + e] = base[2 * FF + e] + ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

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

// ---------------------------------------------------------------------------
// Device-resident finalisation (tica.py:228-259, 492-524): from the packed raw moments [C | G | s0 | stau] to
//     mu = (s0 + stau) / 2N',   OC = (C + C^T) / 2N' - mu mu^T,   S = G / 2N' - mu mu^T      (each scaled by 1 / (sc_i sc_j)
// when an input scaling is folded in), per-block partials of tr S and sum S^2 for the Rao-Blackwell Ledoit-Wolf
// intensity, then  B = (1 - rho) S + rho (tr S / p) I.  Operation for operation what decomposition/_moments.py does on
// the host in numpy (division by 2N', outer product subtracted, scaling, shrink), so the two paths agree to rounding.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tica_finalise_kernel(const double* __restrict__ packed, const double* __restrict__ scale,
                                                            double two_n, int F, double* __restrict__ A, double* __restrict__ B,
                                                            double* __restrict__ mu, double* __restrict__ part, int* __restrict__ flags)
{
    __shared__ double red[2][256];
    const size_t FF = (size_t)F * F;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    double tr = 0.0, sq = 0.0;
    if (idx < FF) {
        const int i = (int)(idx / F), j = (int)(idx % F);
        const double* s0 = packed + 2 * FF;
        const double* st = s0 + F;
        const double mi = (s0[i] + st[i]) / two_n, mj = (s0[j] + st[j]) / two_n;
        double oc = (packed[idx] + packed[(size_t)j * F + i]) / two_n - mi * mj;
        double sv = packed[FF + idx] / two_n - mi * mj;
        if (scale) {
            const double d = scale[i] * scale[j];
            oc /= d;
            sv /= d;
        }
        A[idx] = oc;
        B[idx] = sv;
        if (!isfinite(oc)) atomicOr(flags, 1);
        if (!isfinite(sv)) atomicOr(flags + 1, 1);
        if (i == 0) mu[j] = mj;
        if (i == j) tr = sv;
        sq = sv * sv;
    }
    red[0][threadIdx.x] = tr;
    red[1][threadIdx.x] = sq;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = red[0][0];
        part[2 * blockIdx.x + 1] = red[1][0];
    }
}

// scal[0] = rho, scal[1] = tr S, scal[2] = rho tr S / p  (one workgroup; fixed summation order)
__global__ __launch_bounds__(256) void tica_rblw_kernel(const double* __restrict__ part, int nblocks, double shrinkage, double n, int p,
                                                        double* __restrict__ scal)
{
    __shared__ double red[2][256];
    double tr = 0.0, sq = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        tr += part[2 * b];
        sq += part[2 * b + 1];
    }
    red[0][threadIdx.x] = tr;
    red[1][threadIdx.x] = sq;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tr = red[0][0];
        sq = red[1][0];
        double rho = shrinkage;
        if (!(shrinkage >= 0.0)) {  // tica.py:492-524 (Chen, Wiesel, Hero), n = n_observations
            const double alpha = (n - 2.0) / (n * (n + 2.0));
            const double beta = ((p + 1.0) * n - 2.0) / (n * (n + 2.0));
            const double U = (double)p * sq / (tr * tr) - 1.0;
            rho = alpha + beta / U;
            if (!(rho < 1.0)) rho = 1.0;
        }
        scal[0] = rho;
        scal[1] = tr;
        scal[2] = rho * tr / (double)p;
    }
}

__global__ void tica_shrink_kernel(double* __restrict__ B, const double* __restrict__ scal, int F)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)F * F) return;
    const int i = (int)(idx / F), j = (int)(idx % F);
    double v = (1.0 - scal[0]) * B[idx];
    if (i == j) v += scal[2];
    B[idx] = v;
}

// vecs[j][:] = column (n - 1 - j) of the column-major Z (eigenvector of the j-th LARGEST eigenvalue), vals[j] = D[n - 1 - j]
__global__ void tica_top_pairs_kernel(const double* __restrict__ Z, const double* __restrict__ D, int n, int k,
                                      double* __restrict__ vecs, double* __restrict__ vals)
{
    const int j = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        vecs[(size_t)j * n + i] = Z[(size_t)(n - 1 - j) * n + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) vals[j] = D[n - 1 - j];
}

}  // namespace msm

using namespace msm;

struct msm_tica {
    int F = 0, lag = 0, mode = 0, T = 0, ntiles = 0;
    int S32 = 0, S64 = 0, S = 0, G = 0;  // cohorts per kernel flavour; S = max (slab count), G = S * ntiles
    int sym = 0, ntiles_sym = 0, S_sym = 0;                // symmetric fp32 kernel: upper tiles, slab / column-sum ROWS (cohorts, + 1 with a remainder cohort)
    int sym_cohorts = 0, sym_grid = 0;                     // ... whole cohorts, workgroups of a launch (sym_grid > sym_cohorts * ntiles_sym: remainder cohort)
    double* slabs_sym = nullptr;                           // [S_sym * ntiles_sym][2][TM*TM]: H and D blocks
    double* slabs = nullptr;    // [G][TM*TM]
    double* base = nullptr;     // packed [2FF+2F] imported state
    double* colpart = nullptr;  // [NCB][2][F]
    double* coltmp = nullptr;   // [NCB][2][F]
    double* packed = nullptr;   // [2FF+2F+2] export scratch
    int* flag = nullptr;        // [2]: [0] sticky, [1] per-call
    long long* dbg = nullptr;   // [4] profiling clocks
    unsigned* cosync = nullptr; // [S] cohort pacing counters
    float* shift = nullptr;     // [F] reference row r of the mean shift (fp32 / bf16 kernels); valid once have_shift
    double* shsum = nullptr;    // [3F] raw column sums [A | B | W] of everything accumulated under the shift
    bool shift_on = true, have_shift = false;
    double* fold = nullptr;     // folded column sums: [S_sym][F] per-cohort sums of the left frames | [FOLD_NB][F] sample partials | [F] zeros
    bool last_folded = false;   // the most recent launch took the folded path
    bool cg_dirty = false;      // the C/G slabs (`slabs`) hold something since the last reset: only then does the export sum them
    bool slabs_dirty = false;   // slabs_sym hold something since the last reset (a rejected folded launch must be able to undo itself)
    DevBuf snap;                // ... from this copy
    DevBuf foldimg;             // bf16 image path: [nchunks][F] per-chunk sums of the left frames
    DevBuf imgsteps;            // fused bf16 kernel: the launch's K-step records (tica_img_steps_kernel)
    int last_fused = 0;         // the last accumulate ran the fused kernel (msm_tica_last_img_fused)
    long long n_sh = 0, nw_sh = 0;  // shifted pairs, and the total weight of their Gram terms (2 n_sh for whole trajectories)
    long long n_obs = 0, n_seq = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // bracket the most recent MFMA launch (bf16 image path: the whole pack + multiply pipeline)
    bool timed = false;
    DevBuf table, table2, staging;
    // Round 5: the chunk tables of the last launch stay on the device with a key of what they were built from.  Fitting the
    // SAME trajectories again (the bench's steps, partial_fit loops, cross-validated re-fits through a pooled handle) then
    // skips the table build, its upload and the two stream synchronisations that cover the pageable host vectors: ~0.3 ms
    // of idle GPU at 1,000 trajectories, and the host gets to the MFMA launch that much earlier.
    unsigned long long table_key = 0, table2_key = 0;
    long long table_n = 0, table2_n = 0, table_img_groups = 0, table_total = -1, table2_total = -1;   // (frames of the keyed launch: a second guard)
    std::vector<TicaChunk> table_host;   // bf16 image path: the super-chunk loop walks the table on the host
    int img_on = 0, T2 = 0, ntile2 = 0, S_img = 0, img_grid = 0;  // 256-wide tiles per side, H and D tiles of the upper triangle, whole cohorts, workgroups
    double* solve_pin = nullptr; // pinned host staging of the solve's results (one device-to-host copy per solve)
    size_t solve_pin_n = 0;
    DevBuf solve;                // device-resident solve: [A (F*F) | B (F*F) | mu F | D F | E F | scal 4 | part 2*nblk | scale F | Y k*F | vals F | ints]
    bool reduced = false;        // solve.A / solve.B hold the reduced matrix and the Cholesky factor of the current state
    size_t packed_len() const { return 2 * (size_t)F * F + 2 * (size_t)F + 2; }
};

namespace {

template <typename K>
int query_slots(K kernel, size_t lds, int* slots)
{
    int occ = 0;
    MSM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, NT, lds));
    if (occ < 1) occ = 1;
    *slots = occ * num_cus();
    return MSM_OK;
}

constexpr size_t LDS32 = 2 * 2 * BK32 * TM * sizeof(float);  // 64 KiB
constexpr size_t LDSSYM = 4 * BK32 * TM * sizeof(float) + 2 * TM * sizeof(float);  // 64 KiB: the (u, d) images for columns I and J, + 1 KiB: the shift row
constexpr int IMG_LAG = 1;                                    // tica_img_pp_kernel: load batches left in flight (tica_img_dev.h)
constexpr size_t IMG_PP_LDS = (size_t)(3 + IMG_LAG) * IMG_SLOT;  // 128 KiB: a ring of four K-steps
constexpr size_t LDS64 = 2 * 2 * BK64 * P64 * sizeof(double);  // 72 KiB (double-buffered, pitch 144)

int tica_zero(msm_tica* h)
{
    const size_t FF2 = 2 * (size_t)h->F * h->F + 2 * (size_t)h->F;
    MSM_HIP_CHECK(hipMemsetAsync(h->slabs, 0, (size_t)h->G * TM * TM * sizeof(double), stream()));
    if (h->slabs_sym)
        MSM_HIP_CHECK(hipMemsetAsync(h->slabs_sym, 0, (size_t)h->S_sym * h->ntiles_sym * 2 * TM * TM * sizeof(double), stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->base, 0, FF2 * sizeof(double), stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->colpart, 0, (size_t)NCB * 2 * h->F * sizeof(double), stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->coltmp, 0, (size_t)NCB * 2 * h->F * sizeof(double), stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->flag, 0, 2 * sizeof(int), stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->shsum, 0, 3 * (size_t)h->F * sizeof(double), stream()));
    h->have_shift = false;
    h->slabs_dirty = false;
    h->cg_dirty = false;
    h->reduced = false;
    h->n_sh = h->nw_sh = 0;
    {
        const char* sh_env = getenv("MSM_TICA_SHIFT");  // 0: accumulate raw moments (A/B switch for the tests); read per reset
        h->shift_on = !(sh_env && atoi(sh_env) == 0);
    }
    h->n_obs = 0;
    h->n_seq = 0;
    return MSM_OK;
}

// A slice of one trajectory: `ptr` is trajectory row `off`, the slice holds n_rows rows, and this
// call owns the LEFT indices t in [ob, oe) of the lagged pairs (t, t + lag) -- i.e. it adds
// w_t x_t x_t^T, [t < len - lag] x_t x_{t+lag}^T and the matching column sums for those t only.
// The slice must reach row min(oe + lag, len) - 1 (the right halo).  Whole trajectory: {n, 0, 0, n}.
struct SegInfo {
    long long len, off, ob, oe;
};

// device-resident trajectories only
int tica_accumulate_device(msm_tica* h, const void* const* ptrs, const msm_idx_t* n_rows,
                           msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld, int check_finite,
                           msm_idx_t* n_skipped, const SegInfo* segs = nullptr)
{
    long long total = 0, nvalid = 0, skipped = 0;
    bool aligned = (h->F % 4 == 0) && (ld % 4 == 0);
    auto seg_of = [&](msm_idx_t s) {
        SegInfo g;
        if (segs) g = segs[s];
        else { g.len = n_rows[s]; g.off = 0; g.ob = 0; g.oe = n_rows[s]; }
        return g;
    };
    for (msm_idx_t s = 0; s < n_seq; ++s) {
        const SegInfo g = seg_of(s);
        if (g.len > h->lag && g.oe > g.ob) {
            total += g.oe - g.ob;
            ++nvalid;
            if (((uintptr_t)ptrs[s]) & 15) aligned = false;
        } else if (g.len <= h->lag) {
            ++skipped;
        }
    }
    if (n_skipped) *n_skipped = skipped;
    if (nvalid == 0) return MSM_OK;
    h->reduced = false;

    // chunk size: every cohort gets work, fp32 partials stay <= KCMAX frames
    const bool bfmode = (h->mode == MSM_TICA_BF16 || h->mode == MSM_TICA_BF16X2);
    const bool useimg = bfmode && h->img_on && (dtype_bytes == 4 || dtype_bytes == 2);  // packed bf16 image + 256 x 256 tiles
    // Round 5, MSM_TICA_IMG_FUSED=1 (read per launch): bfloat16-STORED rows of whole 256-feature panels skip the image -- the MFMA
    // kernel's load role stages the raw rows in LDS and forms the packets itself (tica_img_fused_kernel; its slabs equal the
    // packed-image pipeline's bit for bit).  Half the fabric traffic and no ring, but NOT faster (1M x 2048: 13.9 ms + a
    // column-sum pass against 12.3 ms; DESIGN 3.2c has the measurements), so the packed image stays the default.
    bool usefused = false;
    if (useimg && dtype_bytes == 2 && h->F % 256 == 0 && ld % 8 == 0) {
        const char* fe = getenv("MSM_TICA_IMG_FUSED");
        usefused = fe && atoi(fe) == 1;
        for (msm_idx_t s = 0; s < n_seq && usefused; ++s)
            if (((uintptr_t)ptrs[s]) & 15) usefused = false;   // 16-byte LDS-direct row pieces
    }
    // (a bf16 mode whose 256-wide tiles do not fit one resident round -- beyond 3,840 features -- runs the fp32 C/G kernel:
    //  the mode is an accuracy floor, not a promise of the bf16 pipe; bfloat16-stored rows there take the fp64 kernel)
    const bool use32 = dtype_bytes == 4 && (h->mode == MSM_TICA_F32 || (bfmode && !useimg));
    const int bk = (use32 || useimg) ? BK32 : BK64;
    const bool usesym = (use32 && h->mode == MSM_TICA_F32 && h->sym && aligned) || useimg;  // pair semantics: sum/difference slabs (H/D kernel: 16-byte aligned rows only)
    const int S = useimg ? h->S_img : usesym ? h->sym_cohorts : use32 ? h->S32 : h->S64;  // one resident round
    const bool symrem = usesym && !useimg && h->sym_grid > S * h->ntiles_sym;              // ... + a remainder cohort
    const int G = symrem ? h->sym_grid : S * (usesym ? h->ntiles_sym : h->ntiles);
    long long kc = ceil_div(total, S);
    kc = ceil_div(kc, bk) * bk;
    if (kc > KCMAX) kc = KCMAX;
    // a chunk = one workgroup column of the packing pre-pass: finer chunks, more of them in flight (round 5, measured at 1M x 2048
    // bfloat16-stored, pack + multiply: 2048 -> 12.7 ms, 1024 -> 12.3, 512 -> 11.9, 256 -> 11.6; 1024 keeps the per-chunk
    // column sums of a 6.25M-frame fit at 100 MB)
    if (useimg && kc > 1024) kc = 1024;
    if (kc < bk) kc = bk;
    if (kc == KCMAX && total < 16LL * KCMAX * S && !useimg) {
        // Few chunks per cohort (one rank's share of a strong-scaled fit: 1.25M frames = 7.35 chunks of 4096 per cohort, the
        // busiest cohort does 8): cohorts take chunks round-robin, so the launch lasts as long as the fullest one.  Try smaller
        // chunks and keep the size whose fullest cohort -- plus ~16 frames' worth of prologue per chunk -- is lightest.
        long long best = -1, best_kc = kc;
        std::vector<long long> load((size_t)S);
        for (long long cand : {4096LL, 3072LL, 2560LL, 2048LL, 1536LL, 1024LL}) {
            std::fill(load.begin(), load.end(), 0LL);
            long long c = 0;
            for (msm_idx_t s = 0; s < n_seq; ++s) {
                const SegInfo g = seg_of(s);
                if (g.len <= h->lag || g.oe <= g.ob) continue;
                const long long own = g.oe - g.ob, nch = ceil_div(own, cand);
                const long long piece = ceil_div(ceil_div(own, nch), bk) * bk;
                for (long long r0 = 0; r0 < own; r0 += piece, ++c) load[(size_t)(c % S)] += std::min(piece, own - r0) + 16;
            }
            const long long worst = *std::max_element(load.begin(), load.end());
            if (best < 0 || worst < best) {
                best = worst;
                best_kc = cand;
            }
        }
        kc = best_kc;
    }

    TicaArgs P;
    memset(&P, 0, sizeof(P));
    P.ld = ld;
    P.kc = (int)kc;
    P.F = h->F;
    P.lag = h->lag;
    P.T = h->T;
    P.ntiles = usesym ? h->ntiles_sym : h->ntiles;
    P.S = S;
    P.slabs = usesym ? h->slabs_sym : h->slabs;
    P.colpart = h->coltmp;
    P.flag = h->flag;
    P.dbg = h->dbg;
    {
        // fp32 partial sums of the SHIFTED frames are sigma^2-sized, so two chunks (8192 frames) can share a merge; raw
        // moments (no shift) keep the 4096-frame partials of round 1
        P.kflush = h->shift_on ? 2 * KFLUSH_SYM : KFLUSH_SYM;
    }
    {
        // Cohort pacing (the workgroups of a cohort wait for each other at chunk boundaries, bounded).  C/G kernel, measured
        // at 10M x 512: the L2 fabric-side fetch drops from 207 GB to 79-82 GB per launch but the kernel is 4 % slower
        // (78.4 -> 81.7 ms): opt-in there.  Sum/difference kernel WITH the wave-priority window (round 2): 110 GB -> 32 GB
        // fetched per launch (1.8x the algorithmic bytes instead of 5.5x) AND 1 % faster (50.45 -> 49.8 ms; without the
        // priority window pacing cost 2 %): on there.
        const bool pace = usesym && !useimg;
        P.cosync = pace ? h->cosync : nullptr;
    }

    // what the chunk tables are a function of (FNV-1a over the pointer / length tables and the launch's parameters; 0 = no key)
    unsigned long long table_key = 0;
    if (!segs) {
        unsigned long long hsh = 1469598103934665603ULL;
        auto mix = [&](unsigned long long v) {
            for (int b = 0; b < 8; ++b) {
                hsh ^= (v >> (8 * b)) & 0xffULL;
                hsh *= 1099511628211ULL;
            }
        };
        mix((unsigned long long)n_seq);
        mix((unsigned long long)dtype_bytes);
        mix((unsigned long long)ld);
        mix((unsigned long long)kc);
        mix((unsigned long long)h->lag);
        mix((unsigned long long)bk);
        mix(useimg ? 1ULL : 0ULL);
        for (msm_idx_t s = 0; s < n_seq; ++s) {
            mix((unsigned long long)(uintptr_t)ptrs[s]);
            mix((unsigned long long)n_rows[s]);
        }
        table_key = hsh ? hsh : 1ULL;
    }
    long long img_groups = 0;  // bf16 image path: 8-pair groups of the packed image (whole K-steps per chunk)
    std::vector<TicaChunk> img_tab;   // ... and its chunk table (the super-chunk loop below cuts it into ring slots)
    if (nvalid == 1 && n_seq == 1 && !segs && !useimg) {
        P.chunks = nullptr;
        P.single.base = ptrs[0];
        P.single.row0 = 0;
        P.single.len = n_rows[0];
        P.single.last = n_rows[0] - 1;
        P.single.n = 0;
        P.nchunks = ceil_div(n_rows[0], kc);
    } else if (!segs && table_key != 0 && h->table_key == table_key && h->table_total == total) {
        P.chunks = h->table.as<TicaChunk>();     // the same trajectories as the last launch: its table is still on the device
        P.nchunks = h->table_n;
        img_groups = h->table_img_groups;
        if (useimg) img_tab = h->table_host;
    } else {
        std::vector<TicaChunk> tab;
        tab.reserve((size_t)(total / kc + nvalid + 1));
        for (msm_idx_t s = 0; s < n_seq; ++s) {
            const SegInfo g = seg_of(s);
            if (g.len <= h->lag || g.oe <= g.ob) continue;
            const long long own = g.oe - g.ob;
            const long long nch = ceil_div(own, kc);
            long long piece = ceil_div(ceil_div(own, nch), bk) * bk;
            for (long long r0 = g.ob; r0 < g.oe; r0 += piece) {
                TicaChunk ch;
                // virtual row 0 of the trajectory (never dereferenced outside the slice: rows are
                // clamped to [row0, last])
                ch.base = (const char*)ptrs[s] - (ptrdiff_t)g.off * (ptrdiff_t)ld * dtype_bytes;
                ch.row0 = r0;
                ch.len = g.len;
                ch.n = (int)((g.oe - r0) < piece ? (g.oe - r0) : piece);
                ch.pad = 0;
                ch.last = g.off + n_rows[s] - 1;
                ch.g0 = img_groups;
                if (useimg) {
                    long long nv = std::min<long long>(ch.n, g.len - h->lag - r0);
                    if (nv < 0) nv = 0;
                    img_groups += ceil_div(nv, 32) * 4;
                }
                tab.push_back(ch);
            }
        }
        int rc = h->table.reserve(tab.size() * sizeof(TicaChunk));
        if (rc) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(h->table.p, tab.data(), tab.size() * sizeof(TicaChunk),
                                     hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // `tab` is pageable host memory
        P.chunks = h->table.as<TicaChunk>();
        P.nchunks = (long long)tab.size();
        h->table_key = segs ? 0 : table_key;
        h->table_total = total;
        h->table_n = P.nchunks;
        h->table_img_groups = img_groups;
        if (useimg) {
            if (!segs) h->table_host = tab;
            img_tab.swap(tab);
        }
    }

    P.n_main = P.nchunks;
    if (symrem) {
        // chunks of the remainder cohort: its R workgroups walk them once per round, the whole cohorts share the others
        // S ways -- equal time when n_rem x rounds = n_main / S
        const int R = G - S * h->ntiles_sym, rounds = (int)ceil_div(h->ntiles_sym, R);
        const long long n_rem = (P.nchunks + ((long long)S * rounds + 1) / 2) / ((long long)S * rounds + 1);
        P.n_main = P.nchunks - n_rem;
    }

    // Folded column sums (sum/difference kernel, whole trajectories of >= 2 lag frames, full tiles, launches big enough
    // for a pass over X to matter): no column-sum pass over X ahead of the MFMA kernel -- the kernel's staging lanes sum the
    // left frames, a column-sum pass over the first and last `lag` rows of every trajectory supplies what separates the
    // right frames' sums from the left frames' (and checks those rows), and the finite check is made on the sums afterwards.
    bool fold = false;
    if (usesym && !segs && h->fold && !usefused && (useimg || h->F % TM == 0)) {   // (bf16 image path: the pre-pass sums while it packs; the fused kernel has no pre-pass: column-sum pass)
        // MSM_TICA_FOLD, read per launch (A/B switch of the tests): 0 = never, 2 = whatever the size; default: launches of
        // at least 2^26 elements (frames x features)
        const char* fe = getenv("MSM_TICA_FOLD");
        const int fmode = fe ? atoi(fe) : 1;
        fold = fmode != 0 && (fmode == 2 || (double)total * h->F >= 67108864.0);
        for (msm_idx_t s = 0; s < n_seq && fold; ++s)
            if (n_rows[s] > h->lag && n_rows[s] < 2 * (long long)h->lag) fold = false;
        if (2 * (long long)h->lag * nvalid > total / 4) fold = false;   // the boundary rows would be a pass of their own
    }
    const bool shifted = h->shift_on && (use32 || useimg);
    // 1b) mean shift bookkeeping (fp32 / bf16 kernels): the raw column sums of what this launch accumulates under the
    //     shift, and -- first shifted launch of the handle, `set_r` -- the reference row r = this launch's column means
    auto shift_and_merge = [&](int set_r) -> int {
        if (shifted) {
            long long n_call = 0, nw_call = 0, nmean = 0;
            for (msm_idx_t s = 0; s < n_seq; ++s) {
                const SegInfo g = seg_of(s);
                if (g.len <= h->lag || g.oe <= g.ob) continue;
                const long long n0 = std::max<long long>(0, std::min<long long>(g.oe, g.len - h->lag) - g.ob);
                const long long nt = std::max<long long>(0, g.oe - std::max<long long>(g.ob, h->lag));
                n_call += n0;
                nmean += n0 + nt;
                nw_call += (segs && !usesym) ? n0 + nt : 2 * n0;
            }
            const int what = !segs ? (SH_A_a | SH_B_b | SH_W_ab) : usesym ? (SH_A_a | SH_W_a) : (SH_A_a | SH_W_ab);
            hipLaunchKernelGGL(tica_shift_kernel, dim3((unsigned)ceil_div(h->F, 64)), dim3(512), 0, stream(), h->coltmp,
                               h->shsum, h->shift, h->F, 1.0 / (double)std::max<long long>(1, nmean), set_r, what);
            MSM_HIP_CHECK(hipGetLastError());
            h->have_shift = true;
            h->n_sh += n_call;
            h->nw_sh += nw_call;
        }
        const size_t n = (size_t)NCB * 2 * h->F;
        hipLaunchKernelGGL(tica_colmerge_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(),
                           h->colpart, h->coltmp, n);
        MSM_HIP_CHECK(hipGetLastError());
        return MSM_OK;
    };
    bool snapshot = false;
    const size_t slab_bytes = usesym && h->slabs_sym ? (size_t)h->S_sym * h->ntiles_sym * 2 * TM * TM * sizeof(double) : 0;
    if (!fold) {
        // 1) column sums + finite check into the temporary partials
        if (dtype_bytes == 4)
            hipLaunchKernelGGL(tica_colsum_kernel<float>, dim3(NCB), dim3(NT), 0, stream(), P);
        else if (dtype_bytes == 2)
            hipLaunchKernelGGL(tica_colsum_kernel<__bf16>, dim3(NCB), dim3(NT), 0, stream(), P);
        else
            hipLaunchKernelGGL(tica_colsum_kernel<double>, dim3(NCB), dim3(NT), 0, stream(), P);
        MSM_HIP_CHECK(hipGetLastError());
        if (check_finite) {
            int f[2] = {0, 0};
            MSM_HIP_CHECK(hipMemcpyAsync(f, h->flag, sizeof(f), hipMemcpyDeviceToHost, stream()));
            MSM_HIP_CHECK(hipStreamSynchronize(stream()));
            if (f[0]) {
                MSM_HIP_CHECK(hipMemsetAsync(h->coltmp, 0, (size_t)NCB * 2 * h->F * sizeof(double), stream()));
                MSM_HIP_CHECK(hipMemsetAsync(h->flag, 0, sizeof(int), stream()));
                return fail(MSM_ERR_NONFINITE, "Input contains NaN, infinity or a value too large");
            }
        }
        if (shifted) P.shift = h->shift;
        int rc = shift_and_merge(h->have_shift ? 0 : 1);
        if (rc) return rc;
    } else {
        // 1') the boundary rows: [0, lag) count as left frames only (chunk length "infinite"), [len - lag, len) as right
        //     frames only, so the temporary partials receive a = sum of the first rows | b = sum of the last rows
        std::vector<TicaChunk> tab;
        const bool have2 = table_key != 0 && h->table2_key == table_key && h->table2_total == total;   // (the boundary table of the same trajectories: still there)
        for (msm_idx_t s = 0; s < n_seq && !have2; ++s) {
            const long long len = n_rows[s];
            if (len <= h->lag) continue;
            for (int side = 0; side < 2; ++side) {
                const long long b0 = side ? len - h->lag : 0, b1 = b0 + h->lag;
                for (long long r0 = b0; r0 < b1; r0 += kc) {
                    TicaChunk ch;
                    ch.base = ptrs[s];
                    ch.row0 = r0;
                    ch.len = side ? len : (long long)1 << 60;
                    ch.n = (int)std::min<long long>(kc, b1 - r0);
                    ch.pad = 0;
                    ch.last = len - 1;
                    ch.g0 = 0;
                    tab.push_back(ch);
                }
            }
        }
        int rc = MSM_OK;
        if (!have2) {
            if ((rc = h->table2.reserve(tab.size() * sizeof(TicaChunk)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(h->table2.p, tab.data(), tab.size() * sizeof(TicaChunk), hipMemcpyHostToDevice, stream()));
            MSM_HIP_CHECK(hipStreamSynchronize(stream()));
            h->table2_key = table_key;
            h->table2_total = total;
            h->table2_n = (long long)tab.size();
        }
        TicaArgs Q = P;
        Q.chunks = h->table2.as<TicaChunk>();
        Q.nchunks = h->table2_n;
        if (dtype_bytes == 4)
            hipLaunchKernelGGL(tica_colsum_kernel<float>, dim3(NCB), dim3(NT), 0, stream(), Q);
        else
            hipLaunchKernelGGL(tica_colsum_kernel<__bf16>, dim3(NCB), dim3(NT), 0, stream(), Q);
        MSM_HIP_CHECK(hipGetLastError());
        if (shifted) {
            if (!h->have_shift) {
                double* sp = h->fold + (size_t)h->S_sym * h->F;
                if (dtype_bytes == 4)
                    hipLaunchKernelGGL(tica_fold_sample_kernel<float>, dim3((unsigned)ceil_div(h->F, 64), FOLD_NB), dim3(256), 0, stream(), P, sp);
                else
                    hipLaunchKernelGGL(tica_fold_sample_kernel<__bf16>, dim3((unsigned)ceil_div(h->F, 64), FOLD_NB), dim3(256), 0, stream(), P, sp);
                hipLaunchKernelGGL(tica_fold_setr_kernel, dim3((unsigned)ceil_div(h->F, 256)), dim3(256), 0, stream(), sp, h->shift, h->F);
                MSM_HIP_CHECK(hipGetLastError());
            }
            P.shift = h->shift;
        }
        if (useimg) {
            if ((rc = h->foldimg.reserve((size_t)P.nchunks * h->F * sizeof(double)))) return rc;   // every word is written by the pre-pass
        } else {
            P.colA = h->fold;
            P.zrow = reinterpret_cast<const float*>(h->fold + (size_t)(h->S_sym + FOLD_NB) * h->F);   // zeroed at creation, never written
            MSM_HIP_CHECK(hipMemsetAsync(h->fold, 0, (size_t)h->S_sym * h->F * sizeof(double), stream()));
        }
        if (check_finite && h->slabs_dirty) {   // a rejected launch leaves the state untouched (utils/validation.py:68-74 raises
            if ((rc = h->snap.reserve(slab_bytes))) return rc;   // before tica.py:401 accumulates anything)
            MSM_HIP_CHECK(hipMemcpyAsync(h->snap.p, h->slabs_sym, slab_bytes, hipMemcpyDeviceToDevice, stream()));
            snapshot = true;
        }
    }
    if (shifted && segs) {
        // a trajectory split over ranks: the RIGHT frames of the owned pairs, rows [own_begin + lag, min(own_end, len - lag)
        // + lag), are not the owned rows the pass above summed -- one more column-sum pass over exactly those rows (into
        // the temporary partials, which the merge above has just zeroed; they are zeroed again afterwards)
        std::vector<TicaChunk> tab;
        for (msm_idx_t s = 0; s < n_seq; ++s) {
            const SegInfo g = seg_of(s);
            if (g.len <= h->lag || g.oe <= g.ob) continue;
            const long long b0 = g.ob + h->lag, b1 = std::min<long long>(g.oe, g.len - h->lag) + h->lag;
            for (long long r0 = b0; r0 < b1; r0 += kc) {
                TicaChunk ch;
                ch.base = (const char*)ptrs[s] - (ptrdiff_t)g.off * (ptrdiff_t)ld * dtype_bytes;
                ch.row0 = r0;
                ch.len = (long long)1 << 60;  // every row counts (as "s0")
                ch.n = (int)std::min<long long>(kc, b1 - r0);
                ch.pad = 0;
                ch.last = g.off + n_rows[s] - 1;
                tab.push_back(ch);
            }
        }
        if (!tab.empty()) {
            h->table2_key = 0;
            int rc = h->table2.reserve(tab.size() * sizeof(TicaChunk));
            if (rc) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(h->table2.p, tab.data(), tab.size() * sizeof(TicaChunk), hipMemcpyHostToDevice, stream()));
            MSM_HIP_CHECK(hipStreamSynchronize(stream()));
            TicaArgs Q = P;
            Q.chunks = h->table2.as<TicaChunk>();
            Q.nchunks = (long long)tab.size();
            Q.flag = h->flag + 1;  // these rows were (or will be) checked by the launch that owns them
            if (dtype_bytes == 4)
                hipLaunchKernelGGL(tica_colsum_kernel<float>, dim3(NCB), dim3(NT), 0, stream(), Q);
            else if (dtype_bytes == 2)
                hipLaunchKernelGGL(tica_colsum_kernel<__bf16>, dim3(NCB), dim3(NT), 0, stream(), Q);
            else
                hipLaunchKernelGGL(tica_colsum_kernel<double>, dim3(NCB), dim3(NT), 0, stream(), Q);
            hipLaunchKernelGGL(tica_shift_kernel, dim3((unsigned)ceil_div(h->F, 64)), dim3(512), 0, stream(), h->coltmp,
                               h->shsum, h->shift, h->F, 0.0, 0, usesym ? (SH_B_a | SH_W_a) : SH_B_a);
            MSM_HIP_CHECK(hipGetLastError());
            MSM_HIP_CHECK(hipMemsetAsync(h->coltmp, 0, (size_t)NCB * 2 * h->F * sizeof(double), stream()));
        }
    }
    // 2) the MFMA pass
    MSM_HIP_CHECK(hipMemsetAsync(h->cosync, 0, (size_t)(std::max(h->S, h->S_sym) + 1) * sizeof(unsigned), stream()));
    if (!useimg && h->ev0) MSM_HIP_CHECK(hipEventRecord(h->ev0, stream()));   // (image path: recorded below, around the whole pipeline)
    h->last_fused = 0;
    if (usefused && img_groups > 0 && img_groups / 2 < 0x7fffffffLL) {
        // ONE launch over every K-step of the call: step records from the chunk table, then the fused kernel
        const bool x2 = h->mode == MSM_TICA_BF16X2;
        const long long nsteps = x2 ? img_groups / 2 : img_groups / 4;
        int rc = h->imgsteps.reserve((size_t)nsteps * sizeof(ImgStep));
        if (rc) return rc;
        if (h->ev0) MSM_HIP_CHECK(hipEventRecord(h->ev0, stream()));
        hipLaunchKernelGGL(tica_img_steps_kernel, dim3((unsigned)P.nchunks), dim3(64), 0, stream(), P.chunks, (long long)ld, h->lag, x2 ? 1 : 0,
                           h->imgsteps.as<ImgStep>());
        MSM_HIP_CHECK(hipGetLastError());
        ImgFusedArgs FA;
        memset(&FA, 0, sizeof(FA));
        FA.steps = h->imgsteps.as<ImgStep>();
        FA.shift = P.shift;
        FA.row_bytes = (long long)ld * 2;
        FA.lag_bytes = (long long)h->lag * (long long)ld * 2;
        FA.nsteps = (int)nsteps;
        FA.T = h->T;
        FA.T2 = h->T2;
        FA.ntiles_sym = h->ntiles_sym;
        FA.ntile2 = h->ntile2;
        FA.S = h->S_img;
        FA.main_steps = img_main_steps(nsteps, h->img_grid, h->ntile2);
        FA.kflush_steps = std::max(1, x2 ? P.kflush / 16 : 8 * P.kflush / 32);   // (the image path's merge interval)
        FA.slabs = h->slabs_sym;
        if (x2)
            hipLaunchKernelGGL((tica_img_fused_kernel<true>), dim3((unsigned)h->img_grid), dim3(IMG_NT), IMG_FUSED_LDS, stream(), FA);
        else
            hipLaunchKernelGGL((tica_img_fused_kernel<false>), dim3((unsigned)h->img_grid), dim3(IMG_NT), IMG_FUSED_LDS, stream(), FA);
        MSM_HIP_CHECK(hipGetLastError());
        h->last_fused = 1;
    } else if (useimg) {
        // The image is produced and consumed in SUPER-CHUNKS through a ring that the library owns (runtime.hip, img_ring):
        // tica_img_kernel packs as many chunks as the ring holds, tica_img_pp_kernel multiplies them, and so on, all on
        // stream().  Round 3 packed the WHOLE input into a per-handle image first (2x - 4x the input bytes of scratch,
        // reserved inside the timed fit).  (Packing super-chunk k + 1 WHILE super-chunk k is multiplied was built and
        // measured this round -- CU-masked streams, the MFMA kernel on the other CUs -- and is slower than taking turns:
        // the packing pass needs the whole chip's memory pipelines, 40 - 64 CUs deliver 1.0 - 1.6 TB/s)
        const bool x2 = h->mode == MSM_TICA_BF16X2;
        const int Fp = h->T2 * 256;
        ImgRing* ring = img_ring();
        if (!ring) return MSM_ERR_HIP;
        const int nimg = x2 ? 4 : 2;
        const size_t gbytes = (size_t)Fp * 16;                                  // one 8-pair group of ONE image
        long long slot_groups = (long long)(ring->bytes / (gbytes * nimg));
        slot_groups -= slot_groups % 4;
        const long long max_chunk_groups = ceil_div(kc, 32) * 4;
        if (slot_groups < max_chunk_groups)
            return fail(MSM_ERR_INVALID, "bf16 image ring too small for %d features (MSM_TICA_IMG_RING_MB)", h->F);
        const size_t one = (size_t)slot_groups * gbytes;                        // bytes of one image inside the ring
        if (h->ev0) MSM_HIP_CHECK(hipEventRecord(h->ev0, stream()));
        const std::vector<TicaChunk>& tab = img_tab;
        for (size_t c0 = 0; c0 < tab.size() && img_groups > 0;) {
            // chunks [c0, c1): as many as fit the ring
            const long long g_start = tab[c0].g0;
            size_t c1 = c0;
            long long g_end = g_start;
            while (c1 < tab.size()) {
                const long long ge = c1 + 1 < tab.size() ? tab[c1 + 1].g0 : img_groups;
                if (ge - g_start > slot_groups) break;
                g_end = ge;
                ++c1;
            }
            ImgArgs IA;
            memset(&IA, 0, sizeof(IA));
            IA.chunks = P.chunks + c0;
            IA.nchunks = (long long)(c1 - c0);
            IA.ld = ld;
            IA.F = h->F;
            IA.Fp = Fp;
            IA.lag = h->lag;
            IA.dtype_bytes = dtype_bytes;
            IA.shift = P.shift;
            IA.colA = fold ? h->foldimg.as<double>() + c0 * (size_t)h->F : nullptr;
            IA.g_off = g_start;
            IA.u_hi = reinterpret_cast<bf16x8*>(ring->p);
            IA.d_hi = reinterpret_cast<bf16x8*>(ring->p + one);
            IA.u_mid = x2 ? reinterpret_cast<bf16x8*>(ring->p + 2 * one) : nullptr;
            IA.d_mid = x2 ? reinterpret_cast<bf16x8*>(ring->p + 3 * one) : nullptr;
            const dim3 g1((unsigned)(c1 - c0), (unsigned)h->T2);
            if (x2)
                hipLaunchKernelGGL(tica_img_kernel<true>, g1, dim3(256), 0, stream(), IA);
            else
                hipLaunchKernelGGL(tica_img_kernel<false>, g1, dim3(256), 0, stream(), IA);
            MSM_HIP_CHECK(hipGetLastError());
            ImgMfmaArgs MA;
            memset(&MA, 0, sizeof(MA));
            MA.u_hi = IA.u_hi;
            MA.d_hi = IA.d_hi;
            MA.u_mid = IA.u_mid;
            MA.d_mid = IA.d_mid;
            MA.nsteps = x2 ? (g_end - g_start) / 2 : (g_end - g_start) / 4;
            MA.Fp = Fp;
            MA.T = h->T;
            MA.T2 = h->T2;
            MA.ntiles_sym = h->ntiles_sym;
            MA.ntile2 = h->ntile2;
            MA.S = h->S_img;
            MA.main_steps = img_main_steps(MA.nsteps, h->img_grid, h->ntile2);
            // fp32 partials: bf16 inputs carry 8 significant bits, their products' partial sums can run 8x longer than the
            // fp32 kernels' before the merge costs accuracy that matters (stated tolerance of the mode: 1e-3)
            MA.kflush_steps = std::max(1, x2 ? P.kflush / 16 : 8 * P.kflush / 32);
            MA.slabs = h->slabs_sym;
            if (x2)
                hipLaunchKernelGGL((tica_img_pp_kernel<true, IMG_LAG>), dim3((unsigned)h->img_grid), dim3(IMG_NT), IMG_PP_LDS, stream(), MA);
            else
                hipLaunchKernelGGL((tica_img_pp_kernel<false, IMG_LAG>), dim3((unsigned)h->img_grid), dim3(IMG_NT), IMG_PP_LDS, stream(), MA);
            MSM_HIP_CHECK(hipGetLastError());
            c0 = c1;
        }
    } else if (usesym) {
        if (symrem) {
            if (fold)
                hipLaunchKernelGGL((tica_sym_f32_kernel<false, true, true>), dim3(G), dim3(NT), LDSSYM, stream(), P);
            else if (h->F % TM == 0)
                hipLaunchKernelGGL((tica_sym_f32_kernel<false, false, true>), dim3(G), dim3(NT), LDSSYM, stream(), P);
            else
                hipLaunchKernelGGL((tica_sym_f32_kernel<true, false, true>), dim3(G), dim3(NT), LDSSYM, stream(), P);
        } else if (fold)
            hipLaunchKernelGGL((tica_sym_f32_kernel<false, true>), dim3(G), dim3(NT), LDSSYM, stream(), P);
        else if (h->F % TM == 0)
            hipLaunchKernelGGL((tica_sym_f32_kernel<false, false>), dim3(G), dim3(NT), LDSSYM, stream(), P);
        else
            hipLaunchKernelGGL((tica_sym_f32_kernel<true, false>), dim3(G), dim3(NT), LDSSYM, stream(), P);
    } else if (use32) {
        if (aligned && h->F % TM == 0)
            hipLaunchKernelGGL((tica_mfma_f32_kernel<true, false>), dim3(G), dim3(NT), LDS32, stream(), P);
        else if (aligned)
            hipLaunchKernelGGL((tica_mfma_f32_kernel<true, true>), dim3(G), dim3(NT), LDS32, stream(), P);
        else
            hipLaunchKernelGGL((tica_mfma_f32_kernel<false, true>), dim3(G), dim3(NT), LDS32, stream(), P);
    } else if (dtype_bytes == 4) {
        hipLaunchKernelGGL(tica_mfma_f64_kernel<float>, dim3(G), dim3(NT), LDS64, stream(), P);
    } else {
        hipLaunchKernelGGL(tica_mfma_f64_kernel<double>, dim3(G), dim3(NT), LDS64, stream(), P);
    }
    MSM_HIP_CHECK(hipGetLastError());
    if (h->ev1) {
        MSM_HIP_CHECK(hipEventRecord(h->ev1, stream()));
        h->timed = true;
    }
    if (fold) {
        // 3') temporary partials -> [left sums | right sums] per slot, finite check of the folded sums
        if (useimg)
            hipLaunchKernelGGL(tica_fold_fix_img_kernel, dim3((unsigned)ceil_div((size_t)NCB * h->F, 256)), dim3(256), 0, stream(),
                               h->coltmp, h->foldimg.as<double>(), h->F, P.nchunks, h->flag);
        else
            hipLaunchKernelGGL(tica_fold_fix_kernel, dim3((unsigned)ceil_div((size_t)NCB * h->F, 256)), dim3(256), 0, stream(),
                               h->coltmp, h->fold, h->F, h->S_sym, h->flag);
        MSM_HIP_CHECK(hipGetLastError());
        if (check_finite) {
            int f[2] = {0, 0};
            MSM_HIP_CHECK(hipMemcpyAsync(f, h->flag, sizeof(f), hipMemcpyDeviceToHost, stream()));
            MSM_HIP_CHECK(hipStreamSynchronize(stream()));
            if (f[0]) {   // undo the launch: the slabs as they were, nothing of it in the column sums or the shift
                if (snapshot)
                    MSM_HIP_CHECK(hipMemcpyAsync(h->slabs_sym, h->snap.p, slab_bytes, hipMemcpyDeviceToDevice, stream()));
                else
                    MSM_HIP_CHECK(hipMemsetAsync(h->slabs_sym, 0, slab_bytes, stream()));
                MSM_HIP_CHECK(hipMemsetAsync(h->coltmp, 0, (size_t)NCB * 2 * h->F * sizeof(double), stream()));
                MSM_HIP_CHECK(hipMemsetAsync(h->flag, 0, sizeof(int), stream()));
                MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                return fail(MSM_ERR_NONFINITE, "Input contains NaN, infinity or a value too large");
            }
        }
        int rc = shift_and_merge(0);
        if (rc) return rc;
    }
    if (usesym) h->slabs_dirty = true;
    else h->cg_dirty = true;
    h->last_folded = fold;
    for (msm_idx_t s = 0; s < n_seq; ++s) {
        const SegInfo g = seg_of(s);
        if (g.len > h->lag && g.oe > g.ob) {
            h->n_obs += g.oe - g.ob;       // summed over the ranks sharing a trajectory this is its length
            h->n_seq += (g.ob == 0) ? 1 : 0;  // ... and the rank owning row 0 counts the sequence
        }
    }
    return MSM_OK;
}

int tica_export_device(msm_tica* h)
{
    const size_t total = 2 * (size_t)h->F * h->F + 2 * (size_t)h->F;
    hipLaunchKernelGGL(tica_export_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream(),
                       h->slabs, h->colpart, h->base, h->packed, h->F, h->T, h->ntiles, h->cg_dirty ? h->S : 0);
    hipLaunchKernelGGL(tica_export_cols_kernel, dim3((unsigned)ceil_div(2 * (int64_t)h->F, 64)), dim3(256), 0, stream(), h->colpart, h->base,
                       h->packed, h->F);
    MSM_HIP_CHECK(hipGetLastError());
    if (h->sym) {
        hipLaunchKernelGGL(tica_export_sym_kernel, dim3((unsigned)ceil_div((size_t)h->ntiles_sym * TM * TM, 256)), dim3(256), 0, stream(),
                           h->slabs_sym, h->packed, h->F, h->T, h->ntiles_sym, h->S_sym);
        MSM_HIP_CHECK(hipGetLastError());
    }
    if (h->have_shift) {  // restore the raw moments from the shifted ones (fp64)
        const size_t ff2 = 2 * (size_t)h->F * h->F;
        hipLaunchKernelGGL(tica_unshift_kernel, dim3((unsigned)ceil_div(ff2, 256)), dim3(256), 0, stream(), h->packed,
                           h->shsum, h->shift, (double)h->n_sh, (double)h->nw_sh, h->F, h->sym);
        MSM_HIP_CHECK(hipGetLastError());
    }
    const double cnt[2] = {(double)h->n_obs, (double)h->n_seq};
    MSM_HIP_CHECK(hipMemcpyAsync(h->packed + total, cnt, sizeof(cnt), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // namespace

extern "C" {

int msm_tica_create(msm_tica_t** out, msm_idx_t n_features, msm_idx_t lag_time, int mode)
{
    if (!out) return fail(MSM_ERR_INVALID, "msm_tica_create: null handle pointer");
    if (n_features < 1 || n_features > (1 << 15)) return fail(MSM_ERR_INVALID, "n_features=%lld out of range", (long long)n_features);
    if (lag_time < 1) return fail(MSM_ERR_INVALID, "lag_time must be >= 1");
    if (mode < MSM_TICA_F32 || mode > MSM_TICA_BF16X2) return fail(MSM_ERR_INVALID, "unknown tica mode %d", mode);
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    msm_tica* h = new msm_tica();
    h->F = (int)n_features;
    h->lag = (int)lag_time;
    h->mode = mode;
    h->T = (int)ceil_div(n_features, TM);
    h->ntiles = h->T * h->T + h->T * (h->T + 1) / 2;
    int slots32 = 0, slots64 = 0, rc;
    MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_mfma_f32_kernel<true, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32));
    MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_mfma_f32_kernel<true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32));
    MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_mfma_f32_kernel<false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32));
    {
        // all three flavours must be resident in one round: size the cohorts by the tightest
        int sa = 0, sb = 0, sc = 0;
        if ((rc = query_slots(tica_mfma_f32_kernel<true, false>, LDS32, &sa))) { delete h; return rc; }
        if ((rc = query_slots(tica_mfma_f32_kernel<true, true>, LDS32, &sb))) { delete h; return rc; }
        if ((rc = query_slots(tica_mfma_f32_kernel<false, true>, LDS32, &sc))) { delete h; return rc; }
        slots32 = sa < sb ? sa : sb;
        slots32 = slots32 < sc ? slots32 : sc;
    }
    MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_mfma_f64_kernel<float>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS64));
    MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_mfma_f64_kernel<double>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS64));
    {
        int sa = 0, sb = 0;
        if ((rc = query_slots(tica_mfma_f64_kernel<float>, LDS64, &sa))) { delete h; return rc; }
        if ((rc = query_slots(tica_mfma_f64_kernel<double>, LDS64, &sb))) { delete h; return rc; }
        slots64 = sa < sb ? sa : sb;
    }
    // one resident round per launch: S cohorts of ntiles workgroups, per kernel flavour
    h->S32 = slots32 / h->ntiles;
    h->S64 = slots64 / h->ntiles;
    if (h->S32 < 1) h->S32 = 1;
    if (h->S64 < 1) h->S64 = 1;
    {
        // symmetric fp32 kernel (H/D blocks of the upper tiles): two 64-KiB workgroups per CU
        const char* sym_env = getenv("MSM_TICA_SYM");
        const bool sym_off = sym_env && atoi(sym_env) == 0;
        h->ntiles_sym = h->T * (h->T + 1) / 2;
        constexpr int tmax = 64;  // and one resident cohort must fit (checked below)
        if (mode == MSM_TICA_F32 && !sym_off && h->T >= 2 && h->T <= tmax && n_features % 4 == 0) {
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<false, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<true, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            int sa = 0, sb = 0, sc = 0;
            if ((rc = query_slots(tica_sym_f32_kernel<false, false>, LDSSYM, &sa))) { delete h; return rc; }
            if ((rc = query_slots(tica_sym_f32_kernel<true, false>, LDSSYM, &sb))) { delete h; return rc; }
            if ((rc = query_slots(tica_sym_f32_kernel<false, true>, LDSSYM, &sc))) { delete h; return rc; }
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<false, false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<true, false, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_sym_f32_kernel<false, true, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSSYM));
            int sd = 0, se = 0, sf = 0;
            if ((rc = query_slots(tica_sym_f32_kernel<false, false, true>, LDSSYM, &sd))) { delete h; return rc; }
            if ((rc = query_slots(tica_sym_f32_kernel<true, false, true>, LDSSYM, &se))) { delete h; return rc; }
            if ((rc = query_slots(tica_sym_f32_kernel<false, true, true>, LDSSYM, &sf))) { delete h; return rc; }
            const int slots = std::min(std::min(std::min(sa, sb), sc), std::min(std::min(sd, se), sf));
            h->sym_cohorts = slots / h->ntiles_sym;
            h->sym = h->sym_cohorts >= 1;  // at least one whole cohort resident (F <= 3968 on 256 CUs), else the C/G kernel
            // remainder cohort: the slots beyond the whole cohorts, when they are worth a launch flavour of their own --
            // at least a sixteenth of the chip and at most three rounds over the tiles (2,048 features: 104 of 512 slots,
            // two rounds; 512 features: 2 slots, not worth it)
            const int R = slots - h->sym_cohorts * h->ntiles_sym;
            const bool rem = h->sym && R * 16 >= slots && 3 * R >= h->ntiles_sym;
            h->sym_grid = rem ? slots : h->sym_cohorts * h->ntiles_sym;
            h->S_sym = h->sym_cohorts + (rem ? 1 : 0);
        }
    }
    {
        // bf16 image path (256 x 256 tiles of H and D on the upper triangle, one 8-wave workgroup per CU)
        h->T2 = (int)ceil_div(n_features, 256);
        h->ntile2 = h->T2 * (h->T2 + 1);
        if ((mode == MSM_TICA_BF16 || mode == MSM_TICA_BF16X2) && h->ntile2 <= num_cus()) {
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_pp_kernel<false, IMG_LAG>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_PP_LDS));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_pp_kernel<true, IMG_LAG>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_PP_LDS));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_fused_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_FUSED_LDS));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_fused_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_FUSED_LDS));
            if (!img_ring()) { delete h; return MSM_ERR_HIP; }   // the process's image ring exists before any fit is timed
            h->img_on = 1;
            h->img_grid = std::max(num_cus(), h->ntile2);   // one workgroup per CU: whole cohorts + a remainder cohort (tica_img_dev.h)
            h->S_img = h->img_grid / h->ntile2;
            h->sym = 1;                 // the exported lagged moment is the symmetrised one
            h->S_sym = h->S_img + 1;    // slab rows: the cohorts' and the remainder cohort's
            h->sym_cohorts = h->S_img;
        }
    }
    h->S = std::max(h->S32, h->S64);  // slabs exist for the largest; unused ones stay zero
    h->G = h->S * h->ntiles;
    const size_t FF2 = 2 * (size_t)h->F * h->F + 2 * (size_t)h->F;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc((void**)&h->slabs, (size_t)h->G * TM * TM * sizeof(double));
    if (e == hipSuccess && h->sym) e = hipMalloc((void**)&h->slabs_sym, (size_t)h->S_sym * h->ntiles_sym * 2 * TM * TM * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->base, FF2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->colpart, (size_t)NCB * 2 * h->F * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->coltmp, (size_t)NCB * 2 * h->F * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->packed, (FF2 + 2) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&h->flag, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&h->dbg, 64 * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void**)&h->cosync, (size_t)(std::max(h->S, h->S_sym) + 1) * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void**)&h->shift, (size_t)h->F * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->shsum, 3 * (size_t)h->F * sizeof(double));
    if (e == hipSuccess && h->sym) e = hipMalloc((void**)&h->fold, (size_t)(h->S_sym + FOLD_NB + 1) * h->F * sizeof(double));
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        msm_tica_destroy(h);
        return fail(MSM_ERR_HIP, "msm_tica_create: hipMalloc failed: %s", hipGetErrorString(e));
    }
    if (h->fold) (void)hipMemsetAsync(h->fold, 0, (size_t)(h->S_sym + FOLD_NB + 1) * h->F * sizeof(double), stream());
    rc = tica_zero(h);
    if (rc) {
        msm_tica_destroy(h);
        return rc;
    }
    *out = h;
    return MSM_OK;
}

int msm_tica_destroy(msm_tica_t* h)
{
    if (!h) return MSM_OK;
    (void)hipStreamSynchronize(stream());
    if (h->slabs) (void)hipFree(h->slabs);
    if (h->slabs_sym) (void)hipFree(h->slabs_sym);
    if (h->base) (void)hipFree(h->base);
    if (h->colpart) (void)hipFree(h->colpart);
    if (h->coltmp) (void)hipFree(h->coltmp);
    if (h->packed) (void)hipFree(h->packed);
    if (h->flag) (void)hipFree(h->flag);
    if (h->dbg) (void)hipFree(h->dbg);
    if (h->cosync) (void)hipFree(h->cosync);
    if (h->shift) (void)hipFree(h->shift);
    if (h->shsum) (void)hipFree(h->shsum);
    if (h->fold) (void)hipFree(h->fold);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->solve_pin) (void)hipHostFree(h->solve_pin);
    delete h;
    return MSM_OK;
}

int msm_tica_reset(msm_tica_t* h)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    return tica_zero(h);
}

static int tica_accumulate_any(msm_tica_t* h, const void* const* X_ptrs, const msm_idx_t* n_rows,
                               msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld, int on_device,
                               int check_finite, msm_idx_t* n_skipped, const SegInfo* segs)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    if (n_seq < 0 || (n_seq > 0 && (!X_ptrs || !n_rows))) return fail(MSM_ERR_INVALID, "bad sequence table");
    if (dtype_bytes != 4 && dtype_bytes != 8 && dtype_bytes != 2) return fail(MSM_ERR_INVALID, "dtype_bytes must be 2 (bfloat16), 4 or 8");
    if (dtype_bytes == 2 && !h->img_on)
        return fail(MSM_ERR_INVALID, "bfloat16 trajectories need MSM_TICA_BF16 / MSM_TICA_BF16X2 mode (got mode %d)", h->mode);
    if (ld < h->F) return fail(MSM_ERR_INVALID, "ld=%lld < n_features=%d", (long long)ld, h->F);
    for (msm_idx_t s = 0; s < n_seq; ++s)
        if (n_rows[s] < 0 || (n_rows[s] > 0 && !X_ptrs[s])) return fail(MSM_ERR_INVALID, "bad sequence %lld", (long long)s);
    if (n_skipped) *n_skipped = 0;
    if (n_seq == 0) return MSM_OK;
    if (on_device) return tica_accumulate_device(h, X_ptrs, n_rows, n_seq, dtype_bytes, ld, check_finite, n_skipped, segs);

    // host trajectories: stage groups of them (compacted to ld = F) through a device buffer
    const size_t row_bytes = (size_t)h->F * dtype_bytes;
    const size_t budget = (size_t)1 << 30;
    msm_idx_t skipped_total = 0;
    msm_idx_t s = 0;
    while (s < n_seq) {
        size_t bytes = 0;
        msm_idx_t e = s;
        while (e < n_seq && (e == s || bytes + (size_t)n_rows[e] * row_bytes <= budget)) {
            bytes += ((size_t)n_rows[e] * row_bytes + 255) & ~(size_t)255;
            ++e;
        }
        int rc = h->staging.reserve(bytes ? bytes : 256);
        if (rc) return rc;
        std::vector<const void*> dptrs((size_t)(e - s));
        size_t off = 0;
        for (msm_idx_t i = s; i < e; ++i) {
            char* d = h->staging.as<char>() + off;
            dptrs[(size_t)(i - s)] = d;
            if (n_rows[i] > 0) {
                if (ld == h->F) {
                    int rcb = h2d_bulk(d, X_ptrs[i], (size_t)n_rows[i] * row_bytes);
                    if (rcb) return rcb;
                } else {
                    MSM_HIP_CHECK(hipMemcpy2DAsync(d, row_bytes, X_ptrs[i], (size_t)ld * dtype_bytes, row_bytes,
                                                   (size_t)n_rows[i], hipMemcpyHostToDevice, stream()));
                }
            }
            off += ((size_t)n_rows[i] * row_bytes + 255) & ~(size_t)255;
        }
        msm_idx_t sk = 0;
        rc = tica_accumulate_device(h, dptrs.data(), n_rows + s, e - s, dtype_bytes, h->F, check_finite, &sk, segs ? segs + s : nullptr);
        if (rc) return rc;
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // staging buffer is reused by the next group
        skipped_total += sk;
        s = e;
    }
    if (n_skipped) *n_skipped = skipped_total;
    return MSM_OK;
}

int msm_tica_accumulate_batch(msm_tica_t* h, const void* const* X_ptrs, const msm_idx_t* n_rows,
                              msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld, int on_device,
                              int check_finite, msm_idx_t* n_skipped)
{
    return tica_accumulate_any(h, X_ptrs, n_rows, n_seq, dtype_bytes, ld, on_device, check_finite, n_skipped, nullptr);
}

int msm_tica_accumulate_segments(msm_tica_t* h, const void* const* X_ptrs, const msm_idx_t* n_rows,
                                 const msm_idx_t* seg4, msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld,
                                 int on_device, int check_finite, msm_idx_t* n_skipped)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    if (n_seq < 0 || (n_seq > 0 && (!X_ptrs || !n_rows || !seg4))) return fail(MSM_ERR_INVALID, "bad segment table");
    std::vector<SegInfo> segs((size_t)n_seq);
    for (msm_idx_t s = 0; s < n_seq; ++s) {
        SegInfo g;
        g.len = seg4[4 * s + 0];
        g.off = seg4[4 * s + 1];
        g.ob = seg4[4 * s + 2];
        g.oe = seg4[4 * s + 3];
        if (g.len < 0 || g.off < 0 || n_rows[s] < 0 || g.off + n_rows[s] > g.len)
            return fail(MSM_ERR_INVALID, "segment %lld: slice [%lld, %lld) is not inside a trajectory of %lld rows",
                        (long long)s, (long long)g.off, (long long)(g.off + n_rows[s]), (long long)g.len);
        if (g.oe > g.ob) {
            const long long need = (g.oe + h->lag < g.len ? g.oe + h->lag : g.len);  // right halo
            if (g.ob < g.off || need > g.off + n_rows[s])
                return fail(MSM_ERR_INVALID, "segment %lld: owned rows [%lld, %lld) + lag %d need rows [%lld, %lld) but the slice holds [%lld, %lld)",
                            (long long)s, (long long)g.ob, (long long)g.oe, h->lag, (long long)g.ob, need,
                            (long long)g.off, (long long)(g.off + n_rows[s]));
        }
        segs[(size_t)s] = g;
    }
    return tica_accumulate_any(h, X_ptrs, n_rows, n_seq, dtype_bytes, ld, on_device, check_finite, n_skipped, segs.data());
}

int msm_tica_accumulate(msm_tica_t* h, const void* X, int dtype_bytes, msm_idx_t n_rows,
                        msm_idx_t ld, int on_device, int check_finite, int* skipped)
{
    msm_idx_t sk = 0;
    const void* ptrs[1] = {X};
    int rc = msm_tica_accumulate_batch(h, ptrs, &n_rows, 1, dtype_bytes, ld, on_device, check_finite, &sk);
    if (skipped) *skipped = (int)sk;
    return rc;
}

int msm_tica_nonfinite(msm_tica_t* h, int* flag)
{
    if (!h || !flag) return fail(MSM_ERR_STATE, "null argument");
    int f[2];
    MSM_HIP_CHECK(hipMemcpyAsync(f, h->flag, sizeof(f), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    *flag = f[0];
    return MSM_OK;
}

int msm_tica_lagged_symmetrised(msm_tica_t* h, int* flag)
{
    if (!h || !flag) return fail(MSM_ERR_STATE, "null argument");
    *flag = h->sym ? 1 : 0;
    return MSM_OK;
}

int msm_tica_last_kernel_ms(msm_tica_t* h, float* ms)
{
    if (!h || !ms) return fail(MSM_ERR_STATE, "null argument");
    if (!h->timed) return fail(MSM_ERR_STATE, "no accumulation launch recorded yet");
    MSM_HIP_CHECK(hipEventSynchronize(h->ev1));
    MSM_HIP_CHECK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MSM_OK;
}

int msm_tica_last_folded(msm_tica_t* h, int* flag)
{
    if (!h || !flag) return fail(MSM_ERR_STATE, "null argument");
    *flag = h->last_folded ? 1 : 0;
    return MSM_OK;
}

int msm_tica_last_img_fused(msm_tica_t* h, int* flag)
{
    if (!h || !flag) return fail(MSM_ERR_INVALID, "msm_tica_last_img_fused: null argument");
    *flag = h->last_fused;
    return MSM_OK;
}

int msm_tica_debug_clocks(msm_tica_t* h, long long* out4)
{
    if (!h || !out4) return fail(MSM_ERR_STATE, "null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(out4, h->dbg, 4 * sizeof(long long), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_tica_debug_profile(msm_tica_t* h, long long* out64)
{
    if (!h || !out64) return fail(MSM_ERR_STATE, "null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(out64, h->dbg, 64 * sizeof(long long), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

msm_idx_t msm_tica_packed_size(msm_tica_t* h) { return h ? (msm_idx_t)h->packed_len() : 0; }

int msm_tica_export_packed(msm_tica_t* h, double* buf, int on_device)
{
    if (!h || !buf) return fail(MSM_ERR_STATE, "null argument");
    int rc = tica_export_device(h);
    if (rc) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(buf, h->packed, h->packed_len() * sizeof(double),
                                 on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_tica_import_packed(msm_tica_t* h, const double* buf, int on_device)
{
    if (!h || !buf) return fail(MSM_ERR_STATE, "null argument");
    const size_t FF2 = 2 * (size_t)h->F * h->F + 2 * (size_t)h->F;
    long long keep_flag = 0;
    (void)keep_flag;
    int rc = tica_zero(h);
    if (rc) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(h->base, buf, FF2 * sizeof(double),
                                 on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream()));
    double cnt[2];
    MSM_HIP_CHECK(hipMemcpyAsync(cnt, buf + FF2, sizeof(cnt),
                                 on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    h->n_obs = (long long)(cnt[0] + 0.5);
    h->n_seq = (long long)(cnt[1] + 0.5);
    return MSM_OK;
}

int msm_tica_export(msm_tica_t* h, double* C, double* G, double* s0, double* stau,
                    msm_idx_t* n_observations, msm_idx_t* n_sequences)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    int rc = tica_export_device(h);
    if (rc) return rc;
    const size_t FF = (size_t)h->F * h->F;
    if (C) MSM_HIP_CHECK(hipMemcpyAsync(C, h->packed, FF * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (G) MSM_HIP_CHECK(hipMemcpyAsync(G, h->packed + FF, FF * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (s0) MSM_HIP_CHECK(hipMemcpyAsync(s0, h->packed + 2 * FF, h->F * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (stau) MSM_HIP_CHECK(hipMemcpyAsync(stau, h->packed + 2 * FF + h->F, h->F * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (n_observations) *n_observations = h->n_obs;
    if (n_sequences) *n_sequences = h->n_seq;
    return MSM_OK;
}

int msm_tica_import(msm_tica_t* h, const double* C, const double* G, const double* s0,
                    const double* stau, msm_idx_t n_observations, msm_idx_t n_sequences)
{
    if (!h || !C || !G || !s0 || !stau) return fail(MSM_ERR_STATE, "null argument");
    int rc = tica_zero(h);
    if (rc) return rc;
    const size_t FF = (size_t)h->F * h->F;
    MSM_HIP_CHECK(hipMemcpyAsync(h->base, C, FF * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(h->base + FF, G, FF * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(h->base + 2 * FF, s0, h->F * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(h->base + 2 * FF + h->F, stau, h->F * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    h->n_obs = n_observations;
    h->n_seq = n_sequences;
    return MSM_OK;
}

}  // extern "C"

// ---- device-resident finalise + solve ------------------------------------------------------------------------
namespace {

struct SolveBufs {
    double *A, *B, *mu, *D, *E, *scal, *part, *scale, *Y, *vals, *trdw;
    double *Winv, *T;             // U^-T of the factorisation (n <= 1024: reduction by GEMMs, back-transformation by a thin product), GEMM scratch
    double* sswork;               // subspace iteration workspace (subspace.hip)
    double *lam, *S, *Yk, *res;   // top-k tail (toppairs.hip): eigenvalues [64], back-transformed vectors [64][F], reduced vectors [64][F], checks [320]
    int* ints;  // [0..1] non-finite flags (OC, S), [2] potrf info, [3] syevd info; [8 ..] sytrd barrier flags + status
    int nblk;
};

int solve_bufs(msm_tica* h, SolveBufs* b)
{
    const size_t F = (size_t)h->F, FF = F * F;
    const int nblk = (int)ceil_div((int64_t)FF, 256);
    const size_t nd = 5 * FF + 13 * F + 4 + 2 * (size_t)nblk + 384 + 128 * F + subspace_work_doubles((int)F);
    int rc = h->solve.reserve(nd * sizeof(double) + (8 + F / 16 + 4) * sizeof(int));
    if (rc) return rc;
    double* p = h->solve.as<double>();
    b->A = p;
    b->B = b->A + FF;
    b->Y = b->B + FF;
    b->mu = b->Y + FF;
    b->D = b->mu + F;
    b->E = b->D + F;
    b->scale = b->E + F;
    b->vals = b->scale + F;
    b->scal = b->vals + F;
    b->part = b->scal + 4;
    b->trdw = b->part + 2 * (size_t)nblk;   // 8F doubles of sytrd exchange records
    b->lam = b->trdw + 8 * F;
    b->res = b->lam + 64;
    b->S = b->res + 320;   // res: 2k + k^2 doubles, k <= 16
    b->Yk = b->S + 64 * F;
    b->sswork = b->Yk + 64 * F;
    b->Winv = b->sswork + subspace_work_doubles((int)F);
    b->T = b->Winv + FF;
    b->ints = reinterpret_cast<int*>(b->T + FF);
    b->nblk = nblk;
    return MSM_OK;
}

// export -> finalise -> shrink -> B = L L^T -> A <- L^-1 A L^-T, all queued on the stream (no synchronisation)
int tica_reduce_device(msm_tica* h, double shrinkage, long long n_rblw, const double* scale_host, SolveBufs* b)
{
    int rc = solve_bufs(h, b);
    if (rc) return rc;
    const long long npairs = h->n_obs - (long long)h->lag * h->n_seq;
    if (h->n_obs <= 0 || npairs <= 0) return fail(MSM_ERR_STATE, "the model must be fit() before use");
    // the packed raw moments (slab sums, un-shifted) stay on the device
    const size_t total = 2 * (size_t)h->F * h->F + 2 * (size_t)h->F;
    hipLaunchKernelGGL(tica_export_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream(),
                       h->slabs, h->colpart, h->base, h->packed, h->F, h->T, h->ntiles, h->cg_dirty ? h->S : 0);
    hipLaunchKernelGGL(tica_export_cols_kernel, dim3((unsigned)ceil_div(2 * (int64_t)h->F, 64)), dim3(256), 0, stream(), h->colpart, h->base,
                       h->packed, h->F);
    if (h->sym) {
        hipLaunchKernelGGL(tica_export_sym_kernel, dim3((unsigned)ceil_div((size_t)h->ntiles_sym * TM * TM, 256)), dim3(256), 0, stream(),
                           h->slabs_sym, h->packed, h->F, h->T, h->ntiles_sym, h->S_sym);
    }
    if (h->have_shift) {
        const size_t ff2 = 2 * (size_t)h->F * h->F;
        hipLaunchKernelGGL(tica_unshift_kernel, dim3((unsigned)ceil_div(ff2, 256)), dim3(256), 0, stream(), h->packed,
                           h->shsum, h->shift, (double)h->n_sh, (double)h->nw_sh, h->F, h->sym);
    }
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemsetAsync(b->ints, 0, 8 * sizeof(int), stream()));
    if (scale_host) MSM_HIP_CHECK(hipMemcpyAsync(b->scale, scale_host, h->F * sizeof(double), hipMemcpyHostToDevice, stream()));
    hipLaunchKernelGGL(tica_finalise_kernel, dim3((unsigned)b->nblk), dim3(256), 0, stream(), h->packed,
                       scale_host ? b->scale : (const double*)nullptr, 2.0 * (double)npairs, h->F, b->A, b->B, b->mu, b->part, b->ints);
    hipLaunchKernelGGL(tica_rblw_kernel, dim3(1), dim3(256), 0, stream(), b->part, b->nblk, shrinkage, (double)n_rblw, h->F, b->scal);
    hipLaunchKernelGGL(tica_shrink_kernel, dim3((unsigned)b->nblk), dim3(256), 0, stream(), b->B, b->scal, h->F);
    MSM_HIP_CHECK(hipGetLastError());
    if ((rc = sygv_reduce_device(b->A, b->B, h->F, b->ints + 2, b->Winv, b->T))) return rc;
    h->reduced = true;
    return MSM_OK;
}

// status of the queued reduction, after the stream was synchronised: info = {rho, tr S, flags...}
int tica_reduce_status(const double scal[4], const int ints[8], double* info)
{
    if (info) {
        info[0] = scal[0];
        info[1] = scal[1];
        info[2] = (double)ints[2];
        info[3] = (double)ints[3];
        info[4] = (double)ints[0];
        info[5] = (double)ints[1];
    }
    if (ints[0]) return fail(MSM_ERR_NONFINITE, "offset correlation matrix is not symmetric");
    if (ints[1]) return fail(MSM_ERR_NONFINITE, "correlation matrix is not symmetric");
    if (ints[2] != 0)
        return fail(MSM_ERR_INVALID, "The leading minor of order %d of B is not positive definite. The factorization of B "
                    "could not be completed and no eigenvalues or eigenvectors were computed.", ints[2]);
    return MSM_OK;
}

// the results of a top-k solve gathered into ONE buffer for one device-to-host copy (seven small copies into pageable host
// arrays cost 20-40 us each): out = [vals k | checks 2k + k^2 | scal 4 | ints 8 | mu n | vecs k n]
__global__ void tica_solve_emit_kernel(const double* __restrict__ lam, const double* __restrict__ res, const double* __restrict__ scal,
                                       const int* __restrict__ ints, const double* __restrict__ mu, const double* __restrict__ vecs,
                                       int n, int k, double* __restrict__ out)
{
    const int nres = 2 * k + k * k;
    const size_t o_res = k, o_scal = o_res + nres, o_ints = o_scal + 4, o_mu = o_ints + 8, o_vec = o_mu + n, total = o_vec + (size_t)k * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (e < o_res) v = lam[e];
        else if (e < o_scal) v = res[e - o_res];
        else if (e < o_ints) v = scal[e - o_scal];
        else if (e < o_mu) v = (double)ints[e - o_ints];
        else if (e < o_vec) v = mu[e - o_mu];
        else v = vecs[e - o_vec];
        out[e] = v;
    }
}

}  // namespace

extern "C" {

int msm_tica_reduce(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, double* Cs, double* mu, double* info)
{
    if (!h || !Cs || !mu) return fail(MSM_ERR_STATE, "msm_tica_reduce: null argument");
    SolveBufs b;
    int rc = tica_reduce_device(h, shrinkage, n_rblw, scale, &b);
    if (rc) return rc;
    const size_t FF = (size_t)h->F * h->F;
    double scal[4];
    int ints[8];
    MSM_HIP_CHECK(hipMemcpyAsync(Cs, b.A, FF * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(mu, b.mu, h->F * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(scal, b.scal, sizeof(scal), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(ints, b.ints, sizeof(ints), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return tica_reduce_status(scal, ints, info);
}

int msm_tica_solve_topk(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, msm_idx_t k, double* vals,
                        double* vecs, double* Cs, double* mu, double* info, int* status)
{
    if (!h || !vals || !vecs || !Cs || !mu || !status) return fail(MSM_ERR_STATE, "msm_tica_solve_topk: null argument");
    if (h->F > 1024 || h->F < 128) return fail(MSM_ERR_INVALID, "msm_tica_solve_topk: need 128 <= n_features <= 1024");
    if (k < 1 || k > 16) return fail(MSM_ERR_INVALID, "msm_tica_solve_topk: need 1 <= k <= 16");
    SolveBufs b;
    int rc = tica_reduce_device(h, shrinkage, n_rblw, scale, &b);
    if (rc) return rc;
    const int n = h->F;
    const size_t FF = (size_t)n * n;
    // Chebyshev-filtered subspace iteration (subspace.hip): a few short launch chains when the spectrum has the gap tICA is
    // run for; the reduced tICA matrix has its spectrum in [-1, 1].  When it does not converge (flat spectra: white-noise
    // features) the reduced matrix goes back to the caller, whose verified fallback is LAPACK's dsyevr on it (round 3 had a
    // second device route in between -- cooperative tridiagonalisation + multisection + inverse iteration; removed in round
    // 4: two routes, one fallback).
    int conv = 0, outer = 0;
    // pinned host memory of the handle: [results of the solve | staging of the iteration's Rayleigh-Ritz steps]
    const size_t pin_res = 17 + 320 + 12 + 17 * (size_t)n, pin_total = pin_res + subspace_pin_doubles();
    if (h->solve_pin_n < pin_total) {
        if (h->solve_pin) (void)hipHostFree(h->solve_pin);
        h->solve_pin = nullptr;
        h->solve_pin_n = 0;
        MSM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->solve_pin), pin_total * sizeof(double), hipHostMallocDefault));
        h->solve_pin_n = pin_total;
    }
    // (Round 5 built the iteration QUEUED -- the 32 x 32 Rayleigh-Ritz problems by one-sided Jacobi in one workgroup, every decision
    //  on the device, one synchronisation per solve -- and removed it again: a Jacobi round is a chain of shuffles, square roots
    //  and divisions, ~0.5 us of latency, 31 rounds a sweep, and the kernel took 0.2 ms per Rayleigh-Ritz round against ~0.13 ms
    //  for the two copies, two synchronisations and 25 us of host arithmetic it replaced: solve 1.26 ms against 1.08 ms.  DESIGN 3.9.)
    if ((rc = subspace_topk_device(b.A, n, (int)k, -1.02, 5e-12, 10, 6, b.lam, b.Yk, b.sswork, h->solve_pin + pin_res, &conv, &outer,
                                   0.0, 1.0)))   // prior: the reduced tICA matrix has its spectrum in [-1, 1], the noise bulk near 0
        return rc;
    if (!conv) {
        double scal[4];
        int ints[8];
        MSM_HIP_CHECK(hipMemcpyAsync(Cs, b.A, FF * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(mu, b.mu, n * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(scal, b.scal, sizeof(scal), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(ints, b.ints, sizeof(ints), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        rc = tica_reduce_status(scal, ints, info);
        if (rc) return rc;
        if (info) {
            info[8] = 0.0;
            info[9] = (double)outer;
        }
        *status = 1;
        return MSM_OK;
    }
    if ((rc = pair_residual_device(b.A, n, b.Yk, b.lam, (int)k, b.res))) return rc;
    // the vectors of the original problem: S = U^-1 Yk = W^T Yk (n <= 1024 here: W = U^-T was formed with the factor)
    if ((rc = winv_back_device(b.Winv, n, b.Yk, (int)k, b.S))) return rc;
    // one packed copy of everything the caller gets (b.Y, n^2 doubles, is free at this point)
    const int nres = 2 * (int)k + (int)k * (int)k;
    const size_t o_res = (size_t)k, o_scal = o_res + nres, o_ints = o_scal + 4, o_mu = o_ints + 8, o_vec = o_mu + n, total = o_vec + (size_t)k * n;
    if (total > pin_res) return fail(MSM_ERR_STATE, "msm_tica_solve_topk: result staging too small");
    hipLaunchKernelGGL(tica_solve_emit_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream(), b.lam, b.res, b.scal, b.ints, b.mu,
                       b.S, n, (int)k, b.Y);
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemcpyAsync(h->solve_pin, b.Y, total * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    const double* hp = h->solve_pin;
    const double* res = hp + o_res;
    double scal[4];
    int ints[8];
    for (int i = 0; i < 4; ++i) scal[i] = hp[o_scal + i];
    for (int i = 0; i < 8; ++i) ints[i] = (int)hp[o_ints + i];
    memcpy(vals, hp, (size_t)k * sizeof(double));
    memcpy(mu, hp + o_mu, (size_t)n * sizeof(double));
    memcpy(vecs, hp + o_vec, (size_t)k * n * sizeof(double));
    rc = tica_reduce_status(scal, ints, info);
    if (rc) return rc;
    if (info) {
        info[8] = 1.0;                // the pairs came from the subspace iteration
        info[9] = (double)outer;      // filtered iterations spent (also when they did not converge)
    }
    // self-check on the reduced matrix: every returned pair must satisfy C y = lambda y to rounding, be normalised, and the
    // k vectors must be mutually orthogonal (a rank-deficient block would pass the per-pair checks; the Gram
    // matrix of the vectors comes from the residual kernel).  A failure hands the reduced matrix to the caller's LAPACK
    // route instead of returning a wrong pair.
    double lmax = 1.0, rmax = 0.0, nmax = 0.0, omax = 0.0;
    for (int j = 0; j < (int)k; ++j) {
        lmax = std::max(lmax, std::fabs(vals[j]));
        rmax = std::max(rmax, res[j]);
        nmax = std::max(nmax, res[k + j]);
        for (int i = 0; i < j; ++i) {
            const double dot = res[2 * k + (size_t)j * k + i];
            omax = std::max(omax, (dot == dot) ? std::fabs(dot) : INFINITY);
        }
    }
    if (info) {
        info[6] = rmax;
        info[7] = std::max(nmax, omax);
    }
    *status = (!(rmax <= 1e-11 * lmax) || !(nmax <= 1e-10) || !(omax <= 1e-8)) ? 2 : 0;
    if (*status) {
        MSM_HIP_CHECK(hipMemcpyAsync(Cs, b.A, FF * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    }
    return MSM_OK;
}

int msm_tica_backsolve(msm_tica_t* h, const double* Y, msm_idx_t k, double* V)
{
    if (!h || !Y || !V) return fail(MSM_ERR_STATE, "msm_tica_backsolve: null argument");
    if (!h->reduced) return fail(MSM_ERR_STATE, "msm_tica_backsolve: no reduced problem (call msm_tica_reduce first)");
    if (k < 1 || k > h->F) return fail(MSM_ERR_INVALID, "msm_tica_backsolve: need 1 <= k <= n_features");
    SolveBufs b;
    int rc = solve_bufs(h, &b);
    if (rc) return rc;
    const size_t bytes = (size_t)k * h->F * sizeof(double);
    MSM_HIP_CHECK(hipMemcpyAsync(b.Y, Y, bytes, hipMemcpyHostToDevice, stream()));
    if (h->F <= 1024 && k <= 64) {
        if ((rc = winv_back_device(b.Winv, h->F, b.Y, (int)k, b.S))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(V, b.S, bytes, hipMemcpyDeviceToHost, stream()));
    } else {
        if ((rc = sygv_back_device(b.B, b.Y, h->F, (int)k))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(V, b.Y, bytes, hipMemcpyDeviceToHost, stream()));
    }
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_tica_solve_device(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, msm_idx_t k,
                          double* vals, double* vecs, double* mu, double* info)
{
    if (!h || !vals || !vecs || !mu) return fail(MSM_ERR_STATE, "msm_tica_solve_device: null argument");
    if (k < 1 || k > h->F) return fail(MSM_ERR_INVALID, "msm_tica_solve_device: need 1 <= k <= n_features");
    SolveBufs b;
    int rc = tica_reduce_device(h, shrinkage, n_rblw, scale, &b);
    if (rc) return rc;
    const int n = h->F;
    if ((rc = syevd_device(b.A, n, b.D, b.E, b.ints + 3))) return rc;
    if ((rc = sygv_back_device(b.B, b.A + (size_t)(n - k) * n, n, (int)k))) return rc;
    hipLaunchKernelGGL(tica_top_pairs_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream(), b.A, b.D, n,
                       (int)k, b.Y, b.vals);
    MSM_HIP_CHECK(hipGetLastError());
    double scal[4];
    int ints[8];
    MSM_HIP_CHECK(hipMemcpyAsync(vecs, b.Y, (size_t)k * n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(vals, b.vals, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(mu, b.mu, n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(scal, b.scal, sizeof(scal), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(ints, b.ints, sizeof(ints), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    h->reduced = false;  // A was overwritten by the eigenvectors
    rc = tica_reduce_status(scal, ints, info);
    if (rc) return rc;
    if (ints[3] != 0) return fail(MSM_ERR_INVALID, "eigenvalue iteration did not converge (info = %d)", ints[3]);
    return MSM_OK;
}

int msm_tica_counts(msm_tica_t* h, msm_idx_t* n_observations, msm_idx_t* n_sequences)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    if (n_observations) *n_observations = h->n_obs;
    if (n_sequences) *n_sequences = h->n_seq;
    return MSM_OK;
}

/* one all-reduce of the packed accumulators over the library communicator, device to device */
int msm_tica_allreduce(msm_tica_t* h)
{
    if (!h) return fail(MSM_ERR_STATE, "null tica handle");
    if (!comm_active()) return MSM_OK;
    int rc = tica_export_device(h);  // h->packed = [C | G | s0 | stau | n_obs | n_seq], raw (un-shifted) moments
    if (rc) return rc;
    if ((rc = comm_allreduce_f64(h->packed, h->packed_len()))) return rc;
    return msm_tica_import_packed(h, h->packed, 1);  // resets the local state, keeps the reduced sums as the base
}

/* s0 / stau alone (F doubles each, host): the column sums without the F x F moments */
int msm_tica_export_sums(msm_tica_t* h, double* s0, double* stau)
{
    if (!h || !s0 || !stau) return fail(MSM_ERR_STATE, "null argument");
    int rc = tica_export_device(h);
    if (rc) return rc;
    const size_t FF = (size_t)h->F * h->F;
    MSM_HIP_CHECK(hipMemcpyAsync(s0, h->packed + 2 * FF, h->F * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(stau, h->packed + 2 * FF + h->F, h->F * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"

extern "C" {

int msm_tica_project(const void* X, int dtype_bytes, msm_idx_t n_rows, msm_idx_t n_features,
                     msm_idx_t ld, const double* mean, const double* comps, msm_idx_t k,
                     double* out, int on_device, int check_finite)
{
    if (!X || !mean || !comps || !out) return fail(MSM_ERR_INVALID, "msm_tica_project: null pointer");
    if (dtype_bytes != 2 && dtype_bytes != 4 && dtype_bytes != 8)
        return fail(MSM_ERR_INVALID, "dtype_bytes must be 2 (bfloat16), 4 or 8");
    if (n_rows < 0 || n_features < 1 || k < 1 || ld < n_features) return fail(MSM_ERR_INVALID, "bad shape");
    if (n_rows == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf &dX = pool(PS_X), &dOut = pool(PS_OUT), &dPar = pool(PS_PAR);
    int rc;
    const size_t par_n = (size_t)k + (size_t)k * n_features;
    if ((rc = dPar.reserve(par_n * sizeof(double) + 16))) return rc;
    std::vector<double> muV((size_t)k);
    for (msm_idx_t c = 0; c < k; ++c) {
        double sacc = 0.0;
        for (msm_idx_t f = 0; f < n_features; ++f) sacc += mean[f] * comps[c * n_features + f];
        muV[(size_t)c] = sacc;
    }
    double* dmean = dPar.as<double>();  // holds mean @ comps^T (k values)
    double* dcomps = dmean + k;
    int* dflag = reinterpret_cast<int*>(dcomps + (size_t)k * n_features);
    MSM_HIP_CHECK(hipMemcpyAsync(dmean, muV.data(), k * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(dcomps, comps, (size_t)k * n_features * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
    const void* Xd = X;
    double* outd = out;
    msm_idx_t ldd = ld;
    if (!on_device) {
        if ((rc = dX.reserve((size_t)n_rows * n_features * dtype_bytes))) return rc;
        if ((rc = dOut.reserve((size_t)n_rows * k * sizeof(double)))) return rc;
        if (ld == n_features) {
            if ((rc = h2d_bulk(dX.p, X, (size_t)n_rows * n_features * dtype_bytes))) return rc;
        } else {
            MSM_HIP_CHECK(hipMemcpy2DAsync(dX.p, (size_t)n_features * dtype_bytes, X, (size_t)ld * dtype_bytes,
                                           (size_t)n_features * dtype_bytes, (size_t)n_rows, hipMemcpyHostToDevice, stream()));
        }
        Xd = dX.p;
        outd = dOut.as<double>();
        ldd = n_features;
    }
    const unsigned grid = (unsigned)ceil_div(n_rows, 128);
    const int cw = 16 / dtype_bytes;
    const int vec = (((uintptr_t)Xd) % 16 == 0) && (ldd % cw == 0) && (n_features % cw == 0);
    if (vec && (size_t)256 * ldd * dtype_bytes < ((size_t)1 << 32)) {
        // fp64-MFMA path: components in blocks of 16, panel Vp[F][16] (feature-major, zero padded)
        DevBuf& dVp = pool(PS_W);
        const msm_idx_t nkb = ceil_div(k, 16);
        if ((rc = dVp.reserve((size_t)nkb * n_features * 16 * sizeof(double)))) return rc;
        std::vector<double> vp((size_t)nkb * n_features * 16, 0.0);
        for (msm_idx_t c = 0; c < k; ++c)
            for (msm_idx_t f = 0; f < n_features; ++f)
                vp[((size_t)(c / 16) * n_features + f) * 16 + (c % 16)] = comps[c * n_features + f];
        MSM_HIP_CHECK(hipMemcpyAsync(dVp.p, vp.data(), vp.size() * sizeof(double), hipMemcpyHostToDevice, stream()));
        const unsigned g2 = (unsigned)ceil_div(n_rows, 256);
        for (msm_idx_t kb = 0; kb < nkb; ++kb) {
            const int kk = (int)std::min<msm_idx_t>(16, k - kb * 16);
            const double* vpk = dVp.as<double>() + (size_t)kb * n_features * 16;
            if (dtype_bytes == 2)
                hipLaunchKernelGGL((tica_project_mfma_kernel<Bf16Raw>), dim3(g2), dim3(NT), 0, stream(), (const Bf16Raw*)Xd,
                                   (long long)n_rows, (int)n_features, (long long)ldd, dmean, vpk, kk, (int)(kb * 16), (int)k,
                                   outd, dflag, nullptr);
            else if (dtype_bytes == 4)
                hipLaunchKernelGGL((tica_project_mfma_kernel<float>), dim3(g2), dim3(NT), 0, stream(), (const float*)Xd,
                                   (long long)n_rows, (int)n_features, (long long)ldd, dmean, vpk, kk, (int)(kb * 16), (int)k,
                                   outd, dflag, nullptr);
            else
                hipLaunchKernelGGL((tica_project_mfma_kernel<double>), dim3(g2), dim3(NT), 0, stream(), (const double*)Xd,
                                   (long long)n_rows, (int)n_features, (long long)ldd, dmean, vpk, kk, (int)(kb * 16), (int)k,
                                   outd, dflag, nullptr);
        }
        MSM_HIP_CHECK(hipGetLastError());
        if (!on_device) {
            int rcd = d2h_bulk(out, outd, (size_t)n_rows * k * sizeof(double));
            if (rcd) return rcd;
        }
        int f2 = 0;
        if (check_finite) MSM_HIP_CHECK(hipMemcpyAsync(&f2, dflag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // `vp` and the scratch buffers die with this frame
        if (check_finite && f2) return fail(MSM_ERR_NONFINITE, "Input contains NaN, infinity or a value too large");
        return MSM_OK;
    }
    const int npw = (int)std::min<msm_idx_t>(8, ceil_div(k, 4));  // components per wave
#define MSM_PROJ(TT, NN)                                                                              \
    hipLaunchKernelGGL((tica_project_kernel<TT, NN>), dim3(grid), dim3(NT), 0, stream(), (const TT*)Xd, \
                       (long long)n_rows, (int)n_features, (long long)ldd, dmean, dcomps, (int)k, outd, dflag, vec)
#define MSM_PROJ_T(TT)                                                                                \
    switch (npw) {                                                                                    \
    case 1: MSM_PROJ(TT, 1); break;                                                                   \
    case 2: MSM_PROJ(TT, 2); break;                                                                   \
    case 3: MSM_PROJ(TT, 3); break;                                                                   \
    case 4: MSM_PROJ(TT, 4); break;                                                                   \
    case 5: MSM_PROJ(TT, 5); break;                                                                   \
    case 6: MSM_PROJ(TT, 6); break;                                                                   \
    case 7: MSM_PROJ(TT, 7); break;                                                                   \
    default: MSM_PROJ(TT, 8); break;                                                                  \
    }
    if (dtype_bytes == 2) {
        MSM_PROJ_T(Bf16Raw)
    } else if (dtype_bytes == 4) {
        MSM_PROJ_T(float)
    } else {
        MSM_PROJ_T(double)
    }
#undef MSM_PROJ_T
#undef MSM_PROJ
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device) {
        int rcd = d2h_bulk(out, outd, (size_t)n_rows * k * sizeof(double));
        if (rcd) return rcd;
    }
    int f = 0;
    if (check_finite) MSM_HIP_CHECK(hipMemcpyAsync(&f, dflag, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // scratch buffers die with this frame
    if (check_finite && f) return fail(MSM_ERR_NONFINITE, "Input contains NaN, infinity or a value too large");
    return MSM_OK;
}

/* msm_tica_project for a LIST of device-resident trajectories in one launch per block of 16 components: X_ptrs[s] is
 * [n_rows[s], n_features] (row stride n_features), out_ptrs[s] its [n_rows[s], k] float64 output.  Needs 16-byte aligned
 * rows (n_features a multiple of 16 / dtype_bytes, aligned base pointers); returns MSM_ERR_INVALID otherwise and the
 * caller projects trajectory by trajectory.  (tica.py:329-352 walks the list; a launch per 10,000-frame trajectory keeps a
 * sixth of the GPU busy.) */
int msm_tica_project_batch(const void* const* X_ptrs, double* const* out_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                           msm_idx_t n_features, const double* mean, const double* comps, msm_idx_t k, int check_finite)
{
    if (!X_ptrs || !out_ptrs || !n_rows || !mean || !comps) return fail(MSM_ERR_INVALID, "msm_tica_project_batch: null pointer");
    if (dtype_bytes != 2 && dtype_bytes != 4 && dtype_bytes != 8)
        return fail(MSM_ERR_INVALID, "dtype_bytes must be 2 (bfloat16), 4 or 8");
    if (n_seq < 0 || n_features < 1 || k < 1) return fail(MSM_ERR_INVALID, "bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int cw = 16 / dtype_bytes;
    if (n_features % cw != 0 || (size_t)256 * n_features * dtype_bytes >= ((size_t)1 << 32))
        return fail(MSM_ERR_INVALID, "msm_tica_project_batch: rows must be whole 16-byte vectors");
    std::vector<ProjTile> tiles;
    for (msm_idx_t s = 0; s < n_seq; ++s) {
        if (n_rows[s] < 0) return fail(MSM_ERR_INVALID, "bad shape");
        if (n_rows[s] == 0) continue;
        if (!X_ptrs[s] || !out_ptrs[s]) return fail(MSM_ERR_INVALID, "msm_tica_project_batch: null pointer");
        if (((uintptr_t)X_ptrs[s]) % 16 != 0) return fail(MSM_ERR_INVALID, "msm_tica_project_batch: rows must be 16-byte aligned");
        for (msm_idx_t r = 0; r < n_rows[s]; r += 256) {
            ProjTile t;
            t.x = static_cast<const char*>(X_ptrs[s]) + (size_t)r * n_features * dtype_bytes;
            t.out = out_ptrs[s] + (size_t)r * k;
            t.rows = n_rows[s] - r;
            tiles.push_back(t);
        }
    }
    if (tiles.empty()) return MSM_OK;
    int rc;
    DevBuf &dPar = pool(PS_PAR), &dVp = pool(PS_W), &dT = pool(PS_IDX);
    const msm_idx_t nkb = ceil_div(k, 16);
    if ((rc = dPar.reserve((size_t)k * sizeof(double) + 16))) return rc;
    if ((rc = dVp.reserve((size_t)nkb * n_features * 16 * sizeof(double)))) return rc;
    if ((rc = dT.reserve(tiles.size() * sizeof(ProjTile)))) return rc;
    std::vector<double> muV((size_t)k), vp((size_t)nkb * n_features * 16, 0.0);
    for (msm_idx_t c = 0; c < k; ++c) {
        double sacc = 0.0;
        for (msm_idx_t f = 0; f < n_features; ++f) {
            sacc += mean[f] * comps[c * n_features + f];
            vp[((size_t)(c / 16) * n_features + f) * 16 + (c % 16)] = comps[c * n_features + f];
        }
        muV[(size_t)c] = sacc;
    }
    double* dmean = dPar.as<double>();
    int* dflag = reinterpret_cast<int*>(dmean + k);
    MSM_HIP_CHECK(hipMemcpyAsync(dmean, muV.data(), k * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(dVp.p, vp.data(), vp.size() * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(dT.p, tiles.data(), tiles.size() * sizeof(ProjTile), hipMemcpyHostToDevice, stream()));
    const ProjTile* dtiles = dT.as<ProjTile>();
    const unsigned g2 = (unsigned)tiles.size();
    for (msm_idx_t kb = 0; kb < nkb; ++kb) {
        const int kk = (int)std::min<msm_idx_t>(16, k - kb * 16);
        const double* vpk = dVp.as<double>() + (size_t)kb * n_features * 16;
        if (dtype_bytes == 2)
            hipLaunchKernelGGL((tica_project_mfma_kernel<Bf16Raw>), dim3(g2), dim3(NT), 0, stream(), (const Bf16Raw*)nullptr, 0LL,
                               (int)n_features, (long long)n_features, dmean, vpk, kk, (int)(kb * 16), (int)k, (double*)nullptr, dflag, dtiles);
        else if (dtype_bytes == 4)
            hipLaunchKernelGGL((tica_project_mfma_kernel<float>), dim3(g2), dim3(NT), 0, stream(), (const float*)nullptr, 0LL,
                               (int)n_features, (long long)n_features, dmean, vpk, kk, (int)(kb * 16), (int)k, (double*)nullptr, dflag, dtiles);
        else
            hipLaunchKernelGGL((tica_project_mfma_kernel<double>), dim3(g2), dim3(NT), 0, stream(), (const double*)nullptr, 0LL,
                               (int)n_features, (long long)n_features, dmean, vpk, kk, (int)(kb * 16), (int)k, (double*)nullptr, dflag, dtiles);
    }
    MSM_HIP_CHECK(hipGetLastError());
    int f2 = 0;
    if (check_finite) MSM_HIP_CHECK(hipMemcpyAsync(&f2, dflag, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));   // `tiles`, `vp` and `muV` die with this frame
    if (check_finite && f2) return fail(MSM_ERR_NONFINITE, "Input contains NaN, infinity or a value too large");
    return MSM_OK;
}

}  // extern "C"
