// distance_kcpass_dev.h -- kcenters_pass_kernel: one fused k-centers pass (exact triangle-inequality pruning)
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// One k-centers pass (kcenters.py:91-97), fused: (prologue) global argmax of the
// previous pass's per-block partials -> new centre index c; d = metric(X, X[c]);
// strict `d < distances_` update of distances_/labels_; per-block argmax partial
// (max value, lowest row index) for the next pass.  One launch per centre, no
// host round trip; the kernel boundary is the only inter-block synchronisation.
// ---------------------------------------------------------------------------
struct KcPartial {
    double v;
    long long i;
};

struct KcArgs {
    const void* X;
    long long n, m;
    int it;
    long long seed;
    const KcPartial* prev;  // [nblk] partials of pass it-1
    KcPartial* next;        // [nblk]
    int nblk;
    double* dist;
    msm_idx_t* labels;
    msm_idx_t* ids;         // device [K]
    int vecw;               // > 0: rows in registers (m <= FC), vector width in bytes
    const void* ycenter;    // non-null: explicit centre coordinates (device, m values) instead of X[argmax];
                            // used by the sharded driver, where the centre may live on another rank
    const void* centers;    // sharded driver: coordinates of the centres chosen so far, device [it + 1][m] (else X[ids[j]])
    int prune;              // triangle-inequality pruning of rows that cannot change (register path, norm metrics)
    // Fused sharded pass (register path only): the all-gathered candidate records of the previous pass are reduced to this
    // pass's centre in the PROLOGUE (every block redundantly; block 0 stores it to sel_centers[it] / sel_ids[it]), and the
    // shard's candidate record for the next pass is produced in the EPILOGUE by the last block to finish -- one kernel
    // and one all-gather per centre.
    const double* sel_cands;  // [sel_world][2 + m]
    int sel_world;
    void* sel_centers;        // T [K][m]
    msm_idx_t* sel_ids;       // [K]
    double* cand_out;         // [2 + m]
    long long row_offset;
    unsigned* counter;        // zero before the first pass; the last block resets it
};

// Exact pruning of a k-centers pass.  A row i at distance dist_i from its centre c_l cannot move to the new centre c when
// d(c, c_l) >= 2 dist_i: then d(x_i, c) >= d(c, c_l) - d(x_i, c_l) >= dist_i and the reference's strict `d < dist_i`
// (kcenters.py:93) is false.  Such a row needs neither its coordinates nor the distance evaluation -- only distances_[i]
// and labels_[i] (16 B instead of 16 + m sizeof(T)).  Trajectory frames are time-ordered, so neighbouring rows sit in the
// same cluster and whole wavefronts skip together: the untouched 64-byte sectors never leave HBM.  The comparison carries a
// safety factor far above the rounding of the computed distances (fp64 accumulation: ~m 2^-53; float inputs subtract in
// fp32: 2^-24), so a skipped row is PROVABLY one the reference would not update -- labels_/distances_ stay bit-identical.
// Norm metrics only (euclidean, cityblock, chebyshev): the others are not metrics or are not worth it.
constexpr int KC_PRUNE_MAX = 2048;  // previous centres whose distance to the new one is tabulated per block (16 KiB of LDS)
template <typename T> struct PruneMargin;
template <> struct PruneMargin<double> { static constexpr double F = 2.0 * (1.0 + 1e-9); };
template <> struct PruneMargin<float> { static constexpr double F = 2.0 * (1.0 + 1e-5); };
template <int M> struct IsNormMetric { static constexpr bool V = (M == M_EUCLIDEAN || M == M_CITYBLOCK || M == M_CHEBYSHEV); };

__device__ __forceinline__ bool kc_better(double v, long long i, double bv, long long bi)
{
    // numpy argmax: first occurrence of the maximum
    return (v > bv) || (v == bv && i < bi);
}

// REG: rows live in registers (m <= FC, P.vecw > 0) -- the clustering-in-tICA-space shape.  A separate instantiation
// so that the LDS row tile of the generic path (34 KiB) does not cap the occupancy of the streaming path: with it (and
// the 16 KiB pruning table) only two workgroups fitted a CU, i.e. 8 waves to cover HBM latency.
template <typename T, int M, bool REG>
__global__ __launch_bounds__(DT) void kcenters_pass_kernel(KcArgs P)
{
    constexpr int FC = FeatChunk<T>::FC;
    __shared__ T Xs[REG ? 1 : DT * (FC + 1)];
    __shared__ T ys[FC];
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x;

    // ---- prologue: centre of this pass ----
    long long cidx = 0;
    __shared__ int sel_win;
    if (P.sel_cands) {
        // fused select: largest distance wins, ties to the lowest GLOBAL row (numpy's argmax over the concatenated array)
        const long long rec = 2 + P.m;
        if (tid == 0) {
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
            const T v = (tid < P.m && sel_win >= 0) ? (T)P.sel_cands[sel_win * rec + 2 + tid] : (T)0;
            ys[tid] = v;
            if (blockIdx.x == 0 && tid < P.m) static_cast<T*>(P.sel_centers)[(long long)P.it * P.m + tid] = v;
        }
        __syncthreads();
    } else if (P.ycenter) {
        // centre supplied by the host (multi-rank driver): nothing to reduce
    } else if (P.it == 0) {
        cidx = P.seed;
    } else {
        double bv = -1.0;
        long long bi = 0x7fffffffffffffffLL;
        for (int k = tid; k < P.nblk; k += DT) {
            const KcPartial q = P.prev[k];
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
    if (!P.ycenter && !P.sel_cands && blockIdx.x == 0 && tid == 0) P.ids[P.it] = cidx;
    const T* y = P.ycenter ? static_cast<const T*>(P.ycenter) : X + cidx * P.m;  // (unused by the fused select: ys is set)

    double bv = -1.0;
    long long bi = -1;
    const long long ntile = (P.n + DT - 1) / DT;
    __shared__ double Dc[(IsNormMetric<M>::V && REG) ? KC_PRUNE_MAX : 1];  // d(new centre, centre j) for the pruning test
    const bool prune = REG && IsNormMetric<M>::V && P.prune && P.it > 0;
    const int nprev = P.it < KC_PRUNE_MAX ? P.it : KC_PRUNE_MAX;
    if (REG) {  // centre row once per block, broadcast from LDS
        if (!P.sel_cands) {
            __syncthreads();
            if (tid < FC) ys[tid] = tid < P.m ? y[tid] : (T)0;
            __syncthreads();
        }
        if (prune) {
            for (int j = tid; j < nprev; j += DT) {
                const T* cj = P.centers ? static_cast<const T*>(P.centers) + (long long)j * P.m : X + P.ids[j] * P.m;
                double a = 0.0, b = 0.0;
                for (int f = 0; f < (int)P.m; ++f) m_update<T, M>(a, b, cj[f], ys[f]);
                Dc[j] = m_final<M>(a, b, P.m);
            }
            __syncthreads();
        }
    }
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long row0 = t * DT;
        const long long i = row0 + tid;
        double a = 0.0, b = 0.0;
        if (REG && prune) {
            // register path with pruning: everything per row in one place (the common code below is skipped)
            if (i < P.n) {
                double cur = P.dist[i];
                const long long lab = P.labels[i];
                const bool skip = lab < nprev && Dc[(IsNormMetric<M>::V && REG && lab < nprev) ? lab : 0] >= PruneMargin<T>::F * cur;
                if (!skip) {
                    T x[FC];
                    load_row_regs<T>(x, X + i * P.m, (int)P.m, P.vecw);
#pragma unroll
                    for (int g = 0; g < FC / 4; ++g)
                        if (g * 4 < P.m) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) m_update<T, M>(a, b, x[g * 4 + q], ys[g * 4 + q]);
                        }
                    const double d = m_final<M>(a, b, P.m);
                    if (d < cur) {   // strict, kcenters.py:93
                        cur = d;
                        P.dist[i] = d;
                        P.labels[i] = P.it;
                    }
                }
                if (bi < 0 || kc_better(cur, i, bv, bi)) {
                    bv = cur;
                    bi = i;
                }
            }
            continue;
        }
        if (REG) {
            T x[FC];
            load_row_regs<T>(x, X + (i < P.n ? i : P.n - 1) * P.m, (int)P.m, P.vecw);
#pragma unroll
            for (int g = 0; g < FC / 4; ++g)
                if (g * 4 < P.m) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) m_update<T, M>(a, b, x[g * 4 + q], ys[g * 4 + q]);
                }
        } else
        for (int f0 = 0; f0 < P.m; f0 += FC) {
            const int fw = (int)((P.m - f0) < FC ? (P.m - f0) : FC);
            __syncthreads();
            stage_rows<T>(Xs, X, nullptr, row0, P.n, P.m, f0, fw, tid);
            if (tid < fw) ys[tid] = y[f0 + tid];
            __syncthreads();
            for (int ff = 0; ff < fw; ++ff) m_update<T, M>(a, b, Xs[tid * (FC + 1) + ff], ys[ff]);
        }
        if (i < P.n) {
            const double d = m_final<M>(a, b, P.m);
            double cur = (P.it == 0) ? INFINITY : P.dist[i];  // distances_.fill(inf), kcenters.py:87-88
            const bool upd = d < cur;                          // strict, kcenters.py:93
            if (upd) cur = d;
            if (P.it == 0 || upd) {
                P.dist[i] = cur;
                P.labels[i] = upd ? P.it : 0;
            }
            // NaN never enters distances_ (NaN < x is false), so plain compares are numpy's argmax
            if (bi < 0 || kc_better(cur, i, bv, bi)) {
                bv = cur;
                bi = i;
            }
        }
    }
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
        // fused candidate record: the last block to arrive reduces all partials (published write-through above)
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
            for (long long f = tid; f < P.m; f += DT) P.cand_out[2 + f] = w >= 0 ? (double)X[w * P.m + f] : 0.0;
        }
    }
}

struct WideArgs;
// end of a centre group for one row: finalise the NC distances (assign: running strict minimum in
// centre order, assign.hpp:22-31; cdist: write out[i, j])
template <typename T, int M, int MODE, int NC>
__device__ __forceinline__ void wide_group_end(const WideArgs& A, double (&a)[NC], double (&b)[NC], long long i,
                                               long long j0, long long n, long long m, long long K, double& min_d,
                                               long long& lab);
template <typename T, int M, int MODE>
__device__ __forceinline__ void wide_tile_end(const WideArgs& A, long long i, long long n, double& min_d,
                                              long long& lab, double& inertia, double& bv, long long& bi);
template <typename T, int M, int MODE, int NC>
__device__ __forceinline__ void wide_group_end_split(const WideArgs& A, double (&a)[NC], double (&b)[NC], long long i,
                                                     long long j0, int hh, long long n, long long m, long long K,
                                                     double& min_d, long long& lab, double& min_d2, long long& lab2,
                                                     double* rv, long long* ri);

}  // namespace msm
