// distance_wide_dev.h -- wide_kernel: the streaming path for wide rows (assign_nearest / cdist / k-centers pass)
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// Wide-row streaming path (m > FC, rows 16-byte aligned, no X_indices): the HBM-bound scans
// (dist, one k-centers pass, assign/cdist against a few centres) and the VALU-bound ones
// (many centres) share one kernel.  A workgroup owns 256 rows per tile, one lane per row, and
// walks the features in 128-byte chunks in the reference's order (one fp64 accumulator per
// (row, centre), sequential features).  Staging is what the scalar path lacked:
//  * every thread issues 8 x 16-byte loads per chunk (rows clamped -> unconditional), the tile
//    goes to LDS as [256][WP = 36 words] with ds_write_b128 and comes back as ds_read_b128 per
//    lane (16 lanes x 4 banks tile all 64 banks: conflict-free);
//  * (tile, centre group, chunk) units form one flat stream with a two-deep register pipeline
//    and double-buffered LDS, one barrier per unit, 2 workgroups per CU: 128 KB of loads in
//    flight per CU, enough to cover HBM latency at full bandwidth.
// Zero padding of a partial last chunk is exact for every metric (a 0/0 pair adds nothing).
// MODE 0 assign_nearest, 1 cdist/dist, 2 one k-centers pass (NC == 1).
// ---------------------------------------------------------------------------
constexpr int WP = 36;   // staged row pitch in 32-bit words (128 B of data + 16 B pad)
constexpr int WRD = 4;    // centre-fragment reads in flight ahead of the arithmetic
constexpr int WSTEP = 2;  // pairs between scheduling barriers
constexpr int WNC = 16;  // centres per register tile in MODE 0/1 (8: every X tile was re-fetched K/8 times -- 5.5 TB/s of L2/MALL traffic at the VALU-bound rate)

struct WideArgs {
    PairArgs pa;   // MODE 0/1
    KcArgs kc;     // MODE 2
};

struct WideStage {
    raw_f32x4 x[8];
    raw_f32x4 y;
    int inb;
};

// NCT = centres per register tile in MODE 0 / 1.  16 is the general choice (see WNC); 8 serves K <= 8 (a 16-centre group
// spends half its arithmetic on padding there: 4M x 512 float32, K = 8: 3.40 -> 2.11 ms, 0.43 -> 0.69 of the fp64-VALU
// bound and at the HBM floor of its 8.2 GB).
template <typename T, int M, int MODE, int NCT = WNC>
__global__ __launch_bounds__(DT, (NCT > WNC ? 1 : 2)) void wide_kernel(WideArgs A)
{
    constexpr int E = 16 / (int)sizeof(T);    // elements per 16-byte vector
    constexpr int FC = 128 / (int)sizeof(T);  // features per chunk
    constexpr int NC = (MODE == 2) ? 1 : NCT;
    constexpr int NCL = (MODE == 2) ? WNC : NCT;   // centre rows the LDS layout provides for
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    float* Xs = reinterpret_cast<float*>(wsm);                   // [2][DT * WP]
    float* Ys = Xs + 2 * DT * WP;                                // [2][NCL * 32]
    double* rv = reinterpret_cast<double*>(Ys + 2 * NCL * 32);   // [DT]
    long long* ri = reinterpret_cast<long long*>(rv + DT);       // [DT]
    const int tid = threadIdx.x;
    const long long n = (MODE == 2) ? A.kc.n : A.pa.n;
    const long long m = (MODE == 2) ? A.kc.m : A.pa.m;
    const long long K = (MODE == 2) ? 1 : A.pa.K;
    const global_ptr<char> Xg = as_global<char>((MODE == 2) ? A.kc.X : A.pa.X);

    // ---- k-centers prologue: centre of this pass = global argmax of the previous partials ----
    long long cidx = 0;
    if (MODE == 2) {
        if (A.kc.ycenter) {
        } else if (A.kc.it == 0) {
            cidx = A.kc.seed;
        } else {
            double bv = -1.0;
            long long bi = 0x7fffffffffffffffLL;
            for (int k = tid; k < A.kc.nblk; k += DT) {
                const KcPartial q = A.kc.prev[k];
                if (q.i >= 0 && kc_better(q.v, q.i, bv, bi)) {
                    bv = q.v;
                    bi = q.i;
                }
            }
            rv[tid] = bv;
            ri[tid] = bi;
            __syncthreads();
            for (int s = DT / 2; s > 0; s >>= 1) {
                if (tid < s && kc_better(rv[tid + s], ri[tid + s], rv[tid], ri[tid])) {
                    rv[tid] = rv[tid + s];
                    ri[tid] = ri[tid + s];
                }
                __syncthreads();
            }
            cidx = ri[0];
            __syncthreads();
        }
        if (!A.kc.ycenter && blockIdx.x == 0 && tid == 0) A.kc.ids[A.kc.it] = cidx;
    }
    const global_ptr<char> Yg =
        (MODE == 2) ? (A.kc.ycenter ? as_global<char>(A.kc.ycenter) : Xg + (size_t)cidx * (size_t)m * sizeof(T))
                    : as_global<char>(A.pa.Y);

    const unsigned rowb = (unsigned)(m * sizeof(T));  // row pitch in bytes (host guarantees 256 * rowb < 2^32)
    const int c8 = tid & 7, r0 = tid >> 3;
    const long long ntile = (n + DT - 1) / DT;
    const int nch = (int)((m + FC - 1) / FC);
    const long long ngrp = (K + NC - 1) / NC;
    const long long mytiles = blockIdx.x < ntile ? (ntile - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const long long total = mytiles * ngrp * nch;

    // load cursor: two units ahead of the compute cursor, parks on the last unit
    long long lt = blockIdx.x, lg = 0;
    int lc = 0;
#define WIDE_LOAD(ST)                                                                             \
    {                                                                                             \
        const long long row0 = lt * DT;                                                           \
        const long long rlim = n - 1 - row0;                                                      \
        const int col = lc * FC + c8 * E;                                                         \
        (ST).inb = col < m;                                                                       \
        const unsigned cb = (unsigned)((col < m ? col : (int)m - E) * (int)sizeof(T));            \
        const global_ptr<char> xb = Xg + (size_t)row0 * rowb;                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                           \
            const int rr = r0 + 32 * j;                                                           \
            (ST).x[j] = *(global_ptr<raw_f32x4>)(xb + ((unsigned)(rr < rlim ? rr : (int)rlim) * rowb + cb)); \
        }                                                                                         \
        {                                                                                         \
            const long long jc = lg * NC + (r0 < NC ? r0 : NC - 1);                               \
            (ST).y = *(global_ptr<raw_f32x4>)(Yg + ((size_t)(jc < K ? jc : K - 1) * rowb + cb));  \
        }                                                                                         \
        if (++lc == nch) {                                                                        \
            lc = 0;                                                                               \
            if (++lg == ngrp) {                                                                   \
                lg = 0;                                                                           \
                if (lt + gridDim.x < ntile) lt += gridDim.x;                                      \
            }                                                                                     \
        }                                                                                         \
    }
#define WIDE_STORE(ST, BUF)                                                                       \
    {                                                                                             \
        const bool in = (ST).inb != 0;                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                           \
            raw_f32x4 v = (ST).x[j];                                                              \
            v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f; \
            *reinterpret_cast<raw_f32x4*>(Xs + (BUF) * (DT * WP) + (r0 + 32 * j) * WP + c8 * 4) = v; \
        }                                                                                         \
        if (r0 < NC) {                                                                            \
            raw_f32x4 v = (ST).y;                                                                 \
            v.x = in ? v.x : 0.f; v.y = in ? v.y : 0.f; v.z = in ? v.z : 0.f; v.w = in ? v.w : 0.f; \
            *reinterpret_cast<raw_f32x4*>(Ys + (BUF) * (NCL * 32) + r0 * 32 + c8 * 4) = v;        \
        }                                                                                         \
    }

    // compute cursor and per-row state
    long long t = blockIdx.x, g = 0;
    int c = 0;
    double a[NC], b[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) a[q] = b[q] = 0.0;
    double min_d = 1.7976931348623157e308;  // assign.hpp:20
    long long lab = 0;
    double min_d2 = 1.7976931348623157e308;  // SPLIT: the lane's second row
    long long lab2 = 0;
    double inertia = 0.0;
    double bv = -1.0;    // k-centers: this block's (max distance, lowest row)
    long long bi = -1;
    // MODE 0/1: a lane owns TWO rows (rl, rl + 128) and HALF of the centre group (hh): every centre fragment read from
    // LDS then serves two rows -- 10 reads per 16 (row fragment, centre fragment) pairs instead of 17.  With one row per
    // lane the broadcast centre reads kept the CU's LDS return path (8 cycles per ds_read_b128) busier than the VALU
    // for float64 rows and ~80% as busy for float32 ones.  Same registers (2 x 8 sums), same tile in LDS.
    constexpr bool SPLIT = MODE != 2;
    constexpr int HQ = SPLIT ? NC / 2 : NC;  // centres per lane
    const int rl = SPLIT ? (tid & (DT / 2 - 1)) : tid;
    const int hh = SPLIT ? (tid / (DT / 2)) : 0;

    WideStage st0, st1;
    if (total > 0) {
        WIDE_LOAD(st0)
        WIDE_STORE(st0, 0)
        WIDE_LOAD(st0)
    }
    __syncthreads();
#define WIDE_STEP(SNEXT, SLOAD, BUF)                                                              \
    {                                                                                             \
        WIDE_LOAD(SLOAD)                                                                          \
        const T* xr = reinterpret_cast<const T*>(Xs + (BUF) * (DT * WP) + rl * WP);               \
        const T* xr2 = reinterpret_cast<const T*>(Xs + (BUF) * (DT * WP) + (rl + DT / 2) * WP);   \
        /* the centre tile's address is uniform; left in SGPRs every fragment read needs its own  */ \
        /* v_mov (and the 128 addresses spill to VGPR lanes): one opaque VGPR base + immediates   */ \
        unsigned yo = (BUF) * (NCL * 32) * 4 + hh * (HQ * 128);                                   \
        asm volatile("" : "+v"(yo));                                                              \
        const T* yr = reinterpret_cast<const T*>(reinterpret_cast<const char*>(Ys) + yo);         \
        /* flat over the 8 x HQ (row fragment, centre fragment) pairs of the chunk, fully unrolled, with the centre   */ \
        /* fragments read WRD pairs ahead and the row fragments one group ahead: a read issued right before its use   */ \
        /* is a stall per 4 pair-elements.  A scheduling barrier every WSTEP pairs keeps that distance (the machine   */ \
        /* scheduler otherwise sinks each read to its use -- or, unpinned, hoists all of them above the arithmetic    */ \
        /* and spills); within a step the pairs' dependent fma chains interleave.                                     */ \
        raw_f32x4 xq = *reinterpret_cast<const raw_f32x4*>(xr), xn = xq;                          \
        raw_f32x4 xq2 = xq, xn2 = xq;                                                             \
        if (SPLIT) xq2 = xn2 = *reinterpret_cast<const raw_f32x4*>(xr2);                          \
        raw_f32x4 yb[WRD];                                                                        \
        _Pragma("unroll") for (int d = 0; d < WRD; ++d)                                           \
            yb[d] = *reinterpret_cast<const raw_f32x4*>(yr + (d % HQ) * FC + (d / HQ) * E);       \
        _Pragma("unroll") for (int idx = 0; idx < 8 * HQ; ++idx) {                                \
            const int v = idx / HQ, q = idx % HQ;                                                 \
            if (q == 0 && v + 1 < 8) {                                                            \
                xn = *reinterpret_cast<const raw_f32x4*>(xr + (v + 1) * E);                       \
                if (SPLIT) xn2 = *reinterpret_cast<const raw_f32x4*>(xr2 + (v + 1) * E);          \
            }                                                                                     \
            const raw_f32x4 yq = yb[idx % WRD];                                                   \
            if (idx + WRD < 8 * HQ)                                                               \
                yb[idx % WRD] = *reinterpret_cast<const raw_f32x4*>(yr + ((idx + WRD) % HQ) * FC + ((idx + WRD) / HQ) * E); \
            m_update_frag<T, M>(a[q], b[q], xq, yq);                                              \
            if (SPLIT) m_update_frag<T, M>(a[HQ + q], b[HQ + q], xq2, yq);                        \
            asm volatile("" : "+v"(a[q]));  /* the sums are formed here, not sunk to the end of the chunk */ \
            if (SPLIT) asm volatile("" : "+v"(a[HQ + q]));                                        \
            if (M == M_BRAYCURTIS || M == M_JACCARD) {                                            \
                asm volatile("" : "+v"(b[q]));                                                    \
                if (SPLIT) asm volatile("" : "+v"(b[HQ + q]));                                    \
            }                                                                                     \
            if (idx % WSTEP == WSTEP - 1) __builtin_amdgcn_sched_barrier(0);                      \
            if (q == HQ - 1) {                                                                    \
                xq = xn;                                                                          \
                xq2 = xn2;                                                                        \
            }                                                                                     \
        }                                                                                         \
        if (u + 1 < total) WIDE_STORE(SNEXT, (BUF) ^ 1)                                           \
        __syncthreads();                                                                          \
        if (++c == nch) {                                                                         \
            c = 0;                                                                                \
            if (SPLIT) {                                                                          \
                wide_group_end_split<T, M, MODE, NC>(A, a, b, t * DT + rl, g * NC, hh, n, m, K, min_d, lab, min_d2, lab2, rv, ri); \
            } else {                                                                              \
                wide_group_end<T, M, MODE, NC>(A, a, b, t * DT + tid, g * NC, n, m, K, min_d, lab); \
            }                                                                                     \
            if (++g == ngrp) {                                                                    \
                g = 0;                                                                            \
                if (SPLIT) {                                                                      \
                    if (hh == 0) {                                                                \
                        wide_tile_end<T, M, MODE>(A, t * DT + rl, n, min_d, lab, inertia, bv, bi); \
                        wide_tile_end<T, M, MODE>(A, t * DT + rl + DT / 2, n, min_d2, lab2, inertia, bv, bi); \
                    }                                                                             \
                } else                                                                            \
                wide_tile_end<T, M, MODE>(A, t * DT + tid, n, min_d, lab, inertia, bv, bi);       \
                t += gridDim.x;                                                                   \
            }                                                                                     \
        }                                                                                         \
    }
    for (long long u = 0; u < total; u += 2) {
        WIDE_STEP(st0, st1, 0)
        ++u;
        if (u < total) WIDE_STEP(st1, st0, 1)
        --u;
    }
#undef WIDE_STEP
#undef WIDE_STORE
#undef WIDE_LOAD
    if (MODE == 0) {
        rv[tid] = inertia;
        __syncthreads();
        for (int s = DT / 2; s > 0; s >>= 1) {
            if (tid < s) rv[tid] += rv[tid + s];
            __syncthreads();
        }
        if (tid == 0) A.pa.partial[blockIdx.x] = rv[0];
    } else if (MODE == 2) {
        rv[tid] = bv;
        ri[tid] = bi;
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
        if (tid == 0) {
            KcPartial q;
            q.v = rv[0];
            q.i = ri[0];
            A.kc.next[blockIdx.x] = q;
        }
    }
}

template <typename T, int M, int MODE, int NC>
__device__ __forceinline__ void wide_group_end(const WideArgs& A, double (&a)[NC], double (&b)[NC], long long i,
                                               long long j0, long long n, long long m, long long K, double& min_d,
                                               long long& lab)
{
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const double d = m_final<M>(a[q], b[q], m);
        if (MODE == 0) {
            if (j0 + q < K && d < min_d) {
                min_d = d;
                lab = j0 + q;
            }
        } else if (MODE == 1) {
            if (j0 + q < K && i < n) A.pa.out[i * K + j0 + q] = d;
        } else {
            min_d = d;  // k-centers: the single distance of this pass
        }
        a[q] = 0.0;
        b[q] = 0.0;
    }
}

// SPLIT layout (MODE 0/1): this lane holds centres j0 + hh HQ + [0, HQ) for rows i and i + DT/2.  cdist writes them out;
// assign_nearest keeps the running (distance, label) of both rows in the hh == 0 lane: that lane's own centres come first
// in index order, the other half's best (its FIRST minimum, through LDS) is taken only when strictly smaller -- the same
// result as the reference's sequential strict `<` scan (assign.hpp:20-31).  All threads of the workgroup call this.
template <typename T, int M, int MODE, int NC>
__device__ __forceinline__ void wide_group_end_split(const WideArgs& A, double (&a)[NC], double (&b)[NC], long long i,
                                                     long long j0, int hh, long long n, long long m, long long K,
                                                     double& min_d, long long& lab, double& min_d2, long long& lab2,
                                                     double* rv, long long* ri)
{
    constexpr int HQ = NC / 2;
    const long long jb = j0 + hh * HQ;
    double d1 = 1.7976931348623157e308, d2 = 1.7976931348623157e308;  // the other half's local scan starts like a fresh one
    long long l1 = -1, l2 = -1;
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const double da = m_final<M>(a[q], b[q], m), db = m_final<M>(a[HQ + q], b[HQ + q], m);
        if (MODE == 0) {
            if (jb + q < K) {
                if (hh == 0) {
                    if (da < min_d) {
                        min_d = da;
                        lab = jb + q;
                    }
                    if (db < min_d2) {
                        min_d2 = db;
                        lab2 = jb + q;
                    }
                } else {
                    if (da < d1) {
                        d1 = da;
                        l1 = jb + q;
                    }
                    if (db < d2) {
                        d2 = db;
                        l2 = jb + q;
                    }
                }
            }
        } else {
            if (jb + q < K) {
                if (i < n) A.pa.out[i * K + jb + q] = da;
                if (i + DT / 2 < n) A.pa.out[(i + DT / 2) * K + jb + q] = db;
            }
        }
        a[q] = b[q] = 0.0;
        a[HQ + q] = b[HQ + q] = 0.0;
    }
    if (MODE == 0) {
        const int rl = threadIdx.x & (DT / 2 - 1);
        if (hh == 1) {
            rv[rl] = d1;
            ri[rl] = l1;
            rv[DT / 2 + rl] = d2;
            ri[DT / 2 + rl] = l2;
        }
        __syncthreads();
        if (hh == 0) {
            const double e1 = rv[rl], e2 = rv[DT / 2 + rl];
            const long long k1 = ri[rl], k2 = ri[DT / 2 + rl];
            if (k1 >= 0 && e1 < min_d) {
                min_d = e1;
                lab = k1;
            }
            if (k2 >= 0 && e2 < min_d2) {
                min_d2 = e2;
                lab2 = k2;
            }
        }
    }
}

template <typename T, int M, int MODE>
__device__ __forceinline__ void wide_tile_end(const WideArgs& A, long long i, long long n, double& min_d,
                                              long long& lab, double& inertia, double& bv, long long& bi)
{
    if (MODE == 0) {
        if (i < n) {
            A.pa.labels[i] = lab;
            if (A.pa.min_dist) A.pa.min_dist[i] = min_d;
            inertia += min_d;
        }
        min_d = 1.7976931348623157e308;
        lab = 0;
    } else if (MODE == 2) {
        if (i < n) {
            const double d = min_d;
            double cur = (A.kc.it == 0) ? INFINITY : A.kc.dist[i];  // distances_.fill(inf), kcenters.py:87-88
            const bool upd = d < cur;                                // strict, kcenters.py:93
            if (upd) cur = d;
            if (A.kc.it == 0 || upd) {
                A.kc.dist[i] = cur;
                A.kc.labels[i] = upd ? A.kc.it : 0;
            }
            if (bi < 0 || kc_better(cur, i, bv, bi)) {
                bv = cur;
                bi = i;
            }
        }
    }
}

}  // namespace msm
