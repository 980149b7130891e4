// distance_screen_dev.h -- kcenters_screen_pass_kernel, ksc_convert_kernel: k-centers passes screened on a low-precision copy
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// k-centers pass with a low-precision SCREEN (single GPU, float64 rows in registers, euclidean).
//
// A pass is HBM-bound: per row the float64 coordinates (80 B at m = 10), distances_ (8 B) and, for the pruning test,
// labels_ (8 B).  But from the second pass on almost no row changes -- the new centre takes the rows near it -- and to
// know that a row does NOT change an approximate distance is enough.  At the switch-over (ksc_convert_kernel) the rows
// are copied once, CENTRED on the first centre c0 (distances are translation invariant; the copy's rounding error then
// scales with the data's spread, not with its offset) and rounded to bfloat16 (u = 2^-8; float32, u = 2^-24, is the
// MSM_KC_SCREEN=1 variant), together with `curf` = distances_ rounded UP to float32, G = max ||x - c0|| and R = max ||x||.
// A later pass reads only that copy and curf (24 B per row at m = 10) and evaluates d~ = || x~ - (y - c0) || in float64:
//     | d~ - d | <= || x~ - (x - c0) || + float64 rounding of the two centrings <= u/(1-u) ||x~|| + 2^-48 (R + ||c0||) =: eps
// so  d~ - eps >= curf >= distances_  proves  d >= distances_: the reference's strict `d < distances_` (kcenters.py:93) is
// false and the row is left alone.  Every other row -- the candidates -- is re-evaluated from its float64 coordinates with
// the exact arithmetic of kcenters_pass_kernel and updated by the exact comparison: bit-identical labels_/distances_.
// (eps carries 1.02 x on the first term and an absolute 1e-37 for underflow; the float64 rounding of d~ and the float32
//  are 1e-9 of that margin.  Non-finite data, or data beyond the float32 range, make eps NaN: no row passes the screen and the pass is the
//  exact one.)
// Argmax for the next centre: curf_i > curf_j implies distances_i > distances_j (curf is a monotone rounding and a strictly
// larger float32 value lies above the other's whole rounding interval), so a thread tracks its best row by curf and looks
// at the float64 values only on an exact float32 tie; the block reduction then uses the float64 value of each thread's
// winner -- numpy's argmax (largest, lowest row on ties), as in the plain kernel.
// ---------------------------------------------------------------------------
struct KscArgs {
    const double* X;
    void* xs;                     // screen copy of the rows: [n][2 NP + 1] float32, or [n][NP + 1] words: NP packed bfloat16 pairs and,
                                  // last word, the row's distances_ rounded UP to float32 (`curf` in the text above): one stream
    float* curf;                  // (unused: the rounded-up distance lives in the row)
    unsigned long long* gmax2;    // [0] bits of max ||x - c0||^2, [1] bits of max ||x||^2 (non-negative doubles order like their bits)
    double* c0;                   // [16] the first centre (the copy's origin)
    long long n, m;
    int it, nblk, vecw;
    long long seed;
    const KcPartial* prev;
    KcPartial* next;
    double* dist;
    msm_idx_t* labels;
    msm_idx_t* ids;
    // Row-sharded fit (same protocol as KcArgs): the centre of this pass is reduced from the all-gathered candidate
    // records in the prologue, the shard's record for the next pass is written by the last block to finish
    const double* sel_cands;  // [sel_world][2 + m]; nullptr: single-process fit (centre = argmax of `prev`)
    int sel_world;
    double* sel_centers;      // [K][m]
    msm_idx_t* sel_ids;       // [K]
    double* cand_out;         // [2 + m]
    long long row_offset;
    unsigned* counter;
};

__device__ __forceinline__ float ksc_round_up(double c)
{
    float f = (float)c;
    if ((double)f < c) f = __uint_as_float(__float_as_uint(f) + 1u);  // c > 0 finite here: next float32 up
    return f;
}

__device__ __forceinline__ unsigned ksc_bf16_rne(float f)  // round-to-nearest-even bfloat16 image (upper 16 bits)
{
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7f800000u) == 0x7f800000u) return u >> 16;  // inf / nan as they are
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// Formats of the screen copy (FMT): 0 = float32 coordinates; 1 = bfloat16; 2 = Q8: signed bytes q_j with ONE scale per row,
// x~_j = q_j sf, sf a bfloat16 >= max_j |x_j - c0_j| / 127 (round 3).  A byte has the 8 significant bits a bfloat16 has, and
// with the row's own scale ||x~ - (x - c0)|| <= sqrt(m) sf / 2 is only ~2x the bfloat16 copy's bound (measured on a
// 10-dimensional projection: 2.28 % of the rows of a pass are re-evaluated exactly instead of 2.20 %, 2.12 % really change)
// -- but a row of ten features is 10 + 2 + 4 = 16 bytes instead of 24, and a pass is HBM-bound.  q_j sf is EXACT in float32
// (7 + 8 significant bits), so the pass's float32 arithmetic is the bfloat16 copy's: fl(q_j sf - yc_j) by one fma.
constexpr int ksc_words(int np, int fmt) { return fmt == 0 ? 2 * np : fmt == 1 ? np : (2 * np + 2 + 3) / 4; }
__device__ __forceinline__ unsigned ksc_bf16_up(float f)  // smallest bfloat16 >= f (f > 0, finite), as its 16 bits
{
    return (__float_as_uint(f) + 0xffffu) >> 16;
}

template <int NP, int FMT>  // screen row = NP pairs (m rounded up to even, zero padded)
__global__ __launch_bounds__(DT) void kcenters_screen_pass_kernel(KscArgs P)
{
    constexpr bool BF16 = FMT == 1;
    constexpr int FC = FeatChunk<double>::FC;  // 16
    constexpr int R = 2;                       // rows per thread and tile
    __shared__ double ys[FC];
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const int tid = threadIdx.x;
    const int m = (int)P.m;

    // The tile stream is software-pipelined: a tile's rows (screen copy + rounded-up distance, one stream) are loaded one
    // tile ahead, the first one BEFORE the prologue -- its loads do not depend on the centre, and the prologue's reduction
    // (a few microseconds at the head of every pass) then overlaps the first HBM round trip instead of preceding it.
    constexpr int NW = ksc_words(NP, FMT);  // 32-bit words of coordinates (Q8: + the scale) per row of the copy
    constexpr int RW = NW + 1;              // + the row's rounded-up distance: ONE stream, 16- or 8-byte loads when RW allows
    const long long ntile = (P.n + (long long)R * DT - 1) / ((long long)R * DT);
    unsigned qn[R][RW];
    auto load_tile = [&](long long t) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long p0 = t * (R * DT) + k * DT + tid;
            const long long pc = p0 < P.n ? p0 : P.n - 1;
            const unsigned* xr = static_cast<const unsigned*>(P.xs) + pc * RW;
            if ((RW & 3) == 0) {
#pragma unroll
                for (int j = 0; j < RW / 4; ++j) {
                    const uint4 v = reinterpret_cast<const uint4*>(xr)[j];
                    qn[k][4 * j] = v.x;
                    qn[k][4 * j + 1] = v.y;
                    qn[k][4 * j + 2] = v.z;
                    qn[k][4 * j + 3] = v.w;
                }
            } else if ((RW & 1) == 0) {
#pragma unroll
                for (int j = 0; j < RW / 2; ++j) {
                    const uint2 v = reinterpret_cast<const uint2*>(xr)[j];
                    qn[k][2 * j] = v.x;
                    qn[k][2 * j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < RW; ++j) qn[k][j] = xr[j];
            }
        }
    };
    if ((long long)blockIdx.x < ntile) load_tile(blockIdx.x);

    // ---- prologue: centre of this pass = argmax of the previous pass's per-block candidates (P.it >= 1 here), or -- sharded
    // fit -- of the candidate records all-gathered from the ranks ----
    if (P.sel_cands) {
        __shared__ int sel_win;
        const long long rec = 2 + P.m;
        if (tid == 0) {   // largest distance, ties to the lowest GLOBAL row (numpy's argmax over the concatenated array)
            int w = -1;
            for (int r = 0; r < P.sel_world; ++r) {
                const double v = P.sel_cands[r * rec], g = P.sel_cands[r * rec + 1];
                if (g < 0.0) continue;
                if (w < 0 || v > P.sel_cands[w * rec] || (v == P.sel_cands[w * rec] && g < P.sel_cands[w * rec + 1])) w = r;
            }
            sel_win = w;
            if (blockIdx.x == 0) P.sel_ids[P.it] = w >= 0 ? (msm_idx_t)P.sel_cands[w * rec + 1] : -1;
        }
        __syncthreads();
        if (tid < FC) {
            const double v = (tid < m && sel_win >= 0) ? P.sel_cands[sel_win * rec + 2 + tid] : 0.0;
            ys[tid] = v;
            if (blockIdx.x == 0 && tid < m) P.sel_centers[(long long)P.it * P.m + tid] = v;
        }
        __syncthreads();
    } else {
    double fv = -1.0;
    long long fi = 0x7fffffffffffffffLL;
    {
        // nblk <= KC_MAXBLK = 4 DT: the thread's (up to) four candidates in ONE round trip (unconditional loads at clamped
        // indices, compared afterwards), not four dependent ones -- this sits at the head of every pass
        KcPartial q[KC_MAXBLK / DT];
#pragma unroll
        for (int j = 0; j < KC_MAXBLK / DT; ++j) {
            const int k = tid + j * DT;
            q[j] = P.prev[k < P.nblk ? k : P.nblk - 1];
        }
#pragma unroll
        for (int j = 0; j < KC_MAXBLK / DT; ++j) {
            const int k = tid + j * DT;
            if (k < P.nblk && q[j].i >= 0 && kc_better(q[j].v, q[j].i, fv, fi)) {
                fv = q[j].v;
                fi = q[j].i;
            }
        }
    }
    rv[tid] = fv;
    ri[tid] = fi;
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (tid < k && kc_better(rv[tid + k], ri[tid + k], rv[tid], ri[tid])) {
            rv[tid] = rv[tid + k];
            ri[tid] = ri[tid + k];
        }
        __syncthreads();
    }
    const long long cidx = ri[0];
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) P.ids[P.it] = cidx;
    if (tid < FC) ys[tid] = tid < m ? P.X[cidx * P.m + tid] : 0.0;
    __syncthreads();
    }
    double yr[2 * NP], yc[2 * NP];  // the centre, and the centre relative to the copy's origin
    double c0n2 = 0.0, yn2 = 0.0;
#pragma unroll
    for (int f = 0; f < 2 * NP; ++f) {
        yr[f] = ys[f];
        const double c0f = f < m ? P.c0[f] : 0.0;
        yc[f] = ys[f] - c0f;
        c0n2 = fma(c0f, c0f, c0n2);
        yn2 = fma(ys[f], ys[f], yn2);
    }
    // eps of a row = u' ||x~|| + eps0 + e32 (||x~|| + ||yc||):
    //   u' ||x~||  -- ||x~ - (x - c0)|| <= u/(1-u) ||x~|| per row (tighter than u max||x - c0||: fewer false candidates);
    //   eps0       -- the float64 roundings of the two centrings + an absolute term for underflow;
    //   e32 (...)  -- the screen's own arithmetic is FLOAT32 (round 3: the float64 version was 134 VALU instructions per
    //                 row, 36 us of VALU time in a 60 us pass): with ycf = fl32(yc), t_j = fl32(x~_j - ycf_j), a = sum t_j^2
    //                 by float32 fma and d~ = sqrtf(a),  |d~ - ||x~ - yc||| <= 2^-24 ||yc|| + 11.5 * 2^-24 ||x~ - ycf||
    //                 < 2^-20 * 1.07 (||x~|| + ||yc||)   (2 NP <= 16 terms; subtraction, 17 accumulation steps, sqrt),
    //                 and the float32 subtraction d~ - eps rounds by another 2^-24 d~: e32 = 2^-19 covers both twice over.
    //                 Underflow (products, flushed denormals) only makes d~ SMALLER, i.e. more rows re-evaluated: safe.
    //                 Overflow would make d~ = inf and pass every row: the screen is switched off (eps = NaN) unless
    //                 max ||x - c0|| and ||yc|| are below 1e18 (squares below 1e36, sums of 16 of them below FLT_MAX).
    constexpr float UREL = (float)((BF16 ? 0x1p-8 : 0x1p-24) * 1.02);  // unit roundoff 2^-p: p = 8 significand bits for bfloat16, 24 for float32
    constexpr float E32 = 0x1p-19f;
    // Q8: eps of a row = 0.51 sqrt(2 NP) sf  [|x_j - c0_j - q_j sf| <= sf / 2 per feature: q_j = rint((x_j - c0_j) / sf) in
    // float64, |q_j| <= 127 because 127 sf >= max_j |x_j - c0_j|]  +  e32 (||x~|| + ||yc||) with ||x~|| <= 127 sqrt(2 NP) sf
    // + eps0: one fma per row, no norm of the row to compute.  (The square root of 2 NP <= 16, rounded up by hand.)
    constexpr float QSQ = NP == 1 ? 1.4143f : NP == 2 ? 2.f : NP == 3 ? 2.4495f : NP == 4 ? 2.8285f : NP == 5 ? 3.1623f
                        : NP == 6 ? 3.4642f : NP == 7 ? 3.7417f : 4.f;
    constexpr float QA = 0.51f * 1.02f * QSQ + E32 * 127.f * QSQ * 1.001f;
    float eps0f, ycnf, ycf[2 * NP];
    {
        const double g2 = __longlong_as_double((long long)P.gmax2[0]), r2 = __longlong_as_double((long long)P.gmax2[1]);
        // (the centre's own centring y - c0 rounds too; in a sharded fit y may be another rank's row, outside this shard's R)
        double eps0 = (sqrt(r2) + sqrt(yn2) + 2.0 * sqrt(c0n2)) * 0x1p-48 + 1e-37;
        double ycn2 = 0.0;
#pragma unroll
        for (int f = 0; f < 2 * NP; ++f) {
            ycn2 = fma(yc[f], yc[f], ycn2);
            ycf[f] = (float)yc[f];
        }
        if (!(g2 < 1e36) || !(r2 < 1e76) || !(ycn2 < 1e36)) eps0 = NAN;  // beyond the float32 screen's range (or non-finite): nothing passes
        eps0f = (float)(eps0 * 1.000001);              // rounded to float32 with slack (inf if it does not fit: nothing passes)
        ycnf = (float)(sqrt(ycn2) * 1.000001);
    }

    // this thread's argmax candidate: by curf; the float64 value is fetched on exact float32 ties and at the end
    float bf = -1.f;
    long long bi = -1;
    double bx = 0.0;
    bool bknown = false;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        float cf[R];
        bool cand[R];
        long long pr[R];
#pragma unroll
        for (int k = 0; k < R; ++k) pr[k] = t * (R * DT) + k * DT + tid;
        {
            // this tile's rows were loaded one tile ago; the next tile's loads go out before the arithmetic
            unsigned q[R][RW];
#pragma unroll
            for (int k = 0; k < R; ++k) {
#pragma unroll
                for (int j = 0; j < RW; ++j) q[k][j] = qn[k][j];
                cf[k] = __uint_as_float(q[k][NW]);
            }
            if (t + gridDim.x < ntile) load_tile(t + gridDim.x);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (FMT == 2) {
                    constexpr int SB = 2 * NP;   // byte offset of the scale
                    const float sf = __uint_as_float(((q[k][SB >> 2] >> (8 * (SB & 3))) & 0xffffu) << 16);
                    float a = 0.f;
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) {
                        const int qi = (int)(q[k][f >> 2] << (24 - 8 * (f & 3))) >> 24;   // sign-extended byte f
                        const float d = fmaf((float)qi, sf, -ycf[f]);
                        a = fmaf(d, d, a);
                    }
                    const float eps = fmaf(sf, QA, fmaf(E32, ycnf, eps0f));
                    cand[k] = pr[k] < P.n && !(sqrtf(a) - eps >= cf[k]);
                    continue;
                }
                float a = 0.f;
                float n2 = 0.f;  // ||x~||^2
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    float x0, x1;
                    if (BF16) {
                        x0 = __uint_as_float(q[k][j] << 16);
                        x1 = __uint_as_float(q[k][j] & 0xffff0000u);
                    } else {
                        x0 = __uint_as_float(q[k][2 * j]);
                        x1 = __uint_as_float(q[k][2 * j + 1]);
                    }
                    const float d0 = x0 - ycf[2 * j], d1 = x1 - ycf[2 * j + 1];
                    a = fmaf(d0, d0, a);
                    a = fmaf(d1, d1, a);
                    n2 = fmaf(x0, x0, n2);
                    n2 = fmaf(x1, x1, n2);
                }
                const float nrm = sqrtf(n2);
                const float eps = fmaf(nrm, UREL, fmaf(E32, nrm + ycnf, eps0f));
                cand[k] = pr[k] < P.n && !(sqrtf(a) - eps >= cf[k]);
            }
        }
        bool anyc = false;
#pragma unroll
        for (int k = 0; k < R; ++k) anyc = anyc || cand[k];
        if (anyc) {
            // exact evaluation from the float64 rows (the arithmetic of kcenters_pass_kernel).  A lane with a candidate
            // loads all of its R rows at once (clamped): one round trip, not one per candidate
            double x[R][2 * NP], cur[R];  // (2 NP values, not FC: registers decide the occupancy of this kernel)
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const long long pc = pr[k] < P.n ? pr[k] : P.n - 1;
                const double* xp = P.X + pc * P.m;
                if (P.vecw == 16 && (m & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const raw_f32x4 v = *reinterpret_cast<const raw_f32x4*>(xp + 2 * j);
                        x[k][2 * j] = reinterpret_cast<const double*>(&v)[0];
                        x[k][2 * j + 1] = reinterpret_cast<const double*>(&v)[1];
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) x[k][f] = xp[f < m ? f : m - 1];
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f)
                        if (f >= m) x[k][f] = 0.0;
                }
                cur[k] = P.dist[pc];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (cand[k]) {
                    // zero padding is exact (a 0 - 0 pair adds nothing); features in order, one accumulator: kcenters_pass_kernel's sum
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, x[k][f], yr[f]);
                    const double d = m_final<M_EUCLIDEAN>(a, b, P.m);
                    if (d < cur[k]) {  // strict, kcenters.py:93
                        P.dist[pr[k]] = d;
                        P.labels[pr[k]] = P.it;
                        cf[k] = ksc_round_up(d);
                        static_cast<unsigned*>(P.xs)[pr[k] * RW + NW] = __float_as_uint(cf[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long p = pr[k];
            if (p < P.n) {
                if (cf[k] > bf || bi < 0) {
                    bf = cf[k];
                    bi = p;
                    bknown = false;
                } else if (cf[k] == bf) {  // same float32 image: the float64 values decide (rows come in ascending order)
                    if (!bknown) {
                        bx = P.dist[bi];
                        bknown = true;
                    }
                    const double v = P.dist[p];
                    if (v > bx) {
                        bx = v;
                        bi = p;
                    }
                }
            }
        }
    }
    // block argmax on the float64 values of the threads' winners
    double bvx = -1.0;
    if (bi >= 0) bvx = bknown ? bx : P.dist[bi];
    rv[tid] = bvx;
    ri[tid] = bi;
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (tid < k) {
            const long long oi = ri[tid + k];
            if (oi >= 0 && (ri[tid] < 0 || kc_better(rv[tid + k], oi, rv[tid], ri[tid]))) {
                rv[tid] = rv[tid + k];
                ri[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        KcPartial q;
        q.v = rv[0];
        q.i = ri[0];
        if (P.cand_out) {
            // sharded fit: the last block to arrive reads every block's partial -- published write-through (agent-scope
            // relaxed atomics = sc1 stores, so the release fence finds nothing of this block's dirty in the L2), then an
            // agent-scope RELEASE fence, drained, before the arrival ticket; the last arriver takes an ACQUIRE fence
            __hip_atomic_store(&P.next[blockIdx.x].v, q.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P.next[blockIdx.x].i, q.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // round 4: the ticket below is taken behind a release
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            P.next[blockIdx.x] = q;
        }
    }
    if (P.cand_out) {
        // sharded fit: the last block to arrive reduces all partials to the shard's candidate record (as in kcenters_pass_kernel)
        __shared__ int am_last;
        if (tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(P.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            am_last = prev == gridDim.x - 1;
            if (am_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ... and the last arriver acquires
        }
        __syncthreads();
        if (am_last) {
            double cv = -1.0;
            long long ci = -1;
            for (int k = tid; k < (int)gridDim.x; k += DT) {
                KcPartial q;
                q.v = __hip_atomic_load(&P.next[k].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.i = __hip_atomic_load(&P.next[k].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q.i >= 0 && (ci < 0 || kc_better(q.v, q.i, cv, ci))) {
                    cv = q.v;
                    ci = q.i;
                }
            }
            rv[tid] = cv;
            ri[tid] = ci;
            __syncthreads();
            for (int s = DT / 2; s > 0; s >>= 1) {
                if (tid < s) {
                    const long long oi = ri[tid + s];
                    if (oi >= 0 && (ri[tid] < 0 || kc_better(rv[tid + s], oi, rv[tid], ri[tid]))) {
                        rv[tid] = rv[tid + s];
                        ri[tid] = oi;
                    }
                }
                __syncthreads();
            }
            const long long w = ri[0];
            if (tid == 0) {
                P.cand_out[0] = w >= 0 ? rv[0] : -1.0;
                P.cand_out[1] = w >= 0 ? (double)(P.row_offset + w) : -1.0;
                __hip_atomic_store(P.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (long long f = tid; f < P.m; f += DT) P.cand_out[2 + f] = w >= 0 ? P.X[w * P.m + f] : 0.0;
        }
    }
}

// switch-over from the plain kernel: the centred screen copy, rounded-up distances, max ||x - c0||^2 and max ||x||^2
template <int NP, int FMT>
__global__ __launch_bounds__(DT) void ksc_convert_kernel(KscArgs P)
{
    constexpr bool BF16 = FMT == 1;
    __shared__ double rv[DT];
    __shared__ double rw[DT];
    const int tid = threadIdx.x, m = (int)P.m;
    constexpr int NW = ksc_words(NP, FMT);
    double c0[2 * NP];
#pragma unroll
    for (int f = 0; f < 2 * NP; ++f) c0[f] = f < m ? P.c0[f] : 0.0;
    double gloc = 0.0, rloc = 0.0;
    for (long long p = (long long)blockIdx.x * DT + tid; p < P.n; p += (long long)gridDim.x * DT) {
        const double* x = P.X + p * P.m;
        unsigned* xo = static_cast<unsigned*>(P.xs) + p * (NW + 1);
        double n2 = 0.0, r2 = 0.0;
        float xc[2 * NP];
        double xv[2 * NP], xd[2 * NP];
        if (P.vecw == 16 && (m & 1) == 0) {  // 16-byte loads (the per-feature loads fetched 3x the row's bytes)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const raw_f32x4 q = *reinterpret_cast<const raw_f32x4*>(x + 2 * j);
                xv[2 * j] = reinterpret_cast<const double*>(&q)[0];
                xv[2 * j + 1] = reinterpret_cast<const double*>(&q)[1];
            }
        } else {
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) xv[f] = x[f < m ? f : m - 1];
        }
#pragma unroll
        for (int f = 0; f < 2 * NP; ++f) {
            const double v = xv[f];
            const double c = f < m ? v - c0[f] : 0.0;
            xc[f] = (float)c;
            xd[f] = c;
            if (f < m) {
                n2 = fma(c, c, n2);
                r2 = fma(v, v, r2);
            }
        }
        if (FMT == 2) {
            // one scale per row: the smallest bfloat16 sf with 127 sf >= max |x_j - c0_j| (0 for a row that IS c0: every q_j = 0
            // is then exact).  A NaN anywhere makes sf NaN -- the pass re-evaluates such a row exactly every time -- and so do
            // scales that small that q_j sf could be flushed to zero in the pass's float32 arithmetic.
            double smax = 0.0;
            bool bad = false;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) {
                const double a = fabs(xd[f]);
                bad = bad || !(a == a);
                smax = a > smax ? a : smax;
            }
            unsigned sbits = 0;
            if (bad || !(smax < 1e37) || (smax > 0.0 && smax < 1e-30)) {
                sbits = 0x7fc0u;   // NaN
            } else if (smax > 0.0) {
                sbits = ksc_bf16_up((float)(smax * (1.0000002 / 127.0)));
            }
            const double sfd = (double)__uint_as_float(sbits << 16);
            unsigned w[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) w[j] = 0u;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) {
                int qi = 0;
                if (sbits != 0 && sbits != 0x7fc0u) {
                    const double t = rint(xd[f] / sfd);
                    qi = (int)(t > 127.0 ? 127.0 : t < -127.0 ? -127.0 : t);
                }
                w[f >> 2] |= ((unsigned)qi & 0xffu) << (8 * (f & 3));
            }
            w[(2 * NP) >> 2] |= sbits << (8 * ((2 * NP) & 3));
#pragma unroll
            for (int j = 0; j < NW; ++j) xo[j] = w[j];
        } else
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (BF16) {
                xo[j] = ksc_bf16_rne(xc[2 * j]) | (ksc_bf16_rne(xc[2 * j + 1]) << 16);
            } else {
                xo[2 * j] = __float_as_uint(xc[2 * j]);
                xo[2 * j + 1] = __float_as_uint(xc[2 * j + 1]);
            }
        }
        if (gloc == gloc && (n2 > gloc || n2 != n2)) gloc = n2;  // a NaN sticks
        if (rloc == rloc && (r2 > rloc || r2 != r2)) rloc = r2;
        xo[NW] = __float_as_uint(ksc_round_up(P.dist[p]));   // the row's distance rounded up to float32, in the row
    }
    const unsigned long long gb = (gloc == gloc) ? (unsigned long long)__double_as_longlong(gloc) : 0x7ff8000000000000ull;
    const unsigned long long rb = (rloc == rloc) ? (unsigned long long)__double_as_longlong(rloc) : 0x7ff8000000000000ull;
    rv[tid] = __longlong_as_double((long long)gb);
    rw[tid] = __longlong_as_double((long long)rb);
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (tid < k) {
            if ((unsigned long long)__double_as_longlong(rv[tid + k]) > (unsigned long long)__double_as_longlong(rv[tid])) rv[tid] = rv[tid + k];
            if ((unsigned long long)__double_as_longlong(rw[tid + k]) > (unsigned long long)__double_as_longlong(rw[tid])) rw[tid] = rw[tid + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicMax(P.gmax2, (unsigned long long)__double_as_longlong(rv[0]));
        atomicMax(P.gmax2 + 1, (unsigned long long)__double_as_longlong(rw[0]));
    }
}

}  // namespace msm
