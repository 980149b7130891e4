// distance.hip -- exact-arithmetic libdistance kernels (dist / cdist / assign_nearest)
// and the fused k-centers pass for gfx950.
//
// Replaces /root/reference/msmbuilder/libdistance/src/{distance_kernels.h:41-293,
// dist.hpp:4-80, cdist.hpp:4-49, assign.hpp:6-91} and the k-pass loop of
// /root/reference/msmbuilder/cluster/kcenters.py:91-97.
//
// Bit-exactness contract (what makes integer labels identical to the CPU path):
// every (row, centre) pair is owned by ONE lane, which visits the features in
// order i = 0..m-1 with ONE fp64 accumulator; for float inputs u-v / u+v are
// fp32 operations widened afterwards; multiply and add are separately rounded
// (this file is compiled with -ffp-contract=off); euclidean takes the sqrt
// before comparing; comparisons are strict `<` in ascending centre order, so
// the lowest index wins.  Parallelism is over rows (and centre tiles), never
// over the feature axis.  This is HBM/L2- and fp64-VALU-bound work: no MFMA.
//
// Tiling: a workgroup stages a [256 rows x FC features] tile of X through LDS
// (coalesced in, row stride FC+1 so that lane-per-row reads are conflict-free)
// and a [CJ centres x FC] tile of Y (wave-uniform broadcast reads).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "distance_dev.h"

namespace msm {

// MODE 0: assign_nearest (assign.hpp:6-91), MODE 1: cdist (cdist.hpp) / dist (K == 1)
template <typename T, int M, int MODE>
__global__ __launch_bounds__(DT) void pair_kernel(PairArgs P)
{
    constexpr int FC = FeatChunk<T>::FC;
    __shared__ T Xs[DT * (FC + 1)];
    __shared__ T Ys[CJ * FC];
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    const T* Y = static_cast<const T*>(P.Y);
    const int tid = threadIdx.x;
    const bool single = P.m <= FC;
    double inertia = 0.0;
    const long long ntile = (P.n + DT - 1) / DT;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long row0 = t * DT;
        const long long i = row0 + tid;
        double min_d = 1.7976931348623157e308;  // DBL_MAX, assign.hpp:20
        long long lab = 0;                      // np.zeros buffer, libdistance.pyx:383
        if (single) {
            __syncthreads();
            stage_rows<T>(Xs, X, P.X_indices, row0, P.n, P.m, 0, (int)P.m, tid);
        }
        for (long long j0 = 0; j0 < P.K; j0 += CJ) {
            double a[CJ], b[CJ];
#pragma unroll
            for (int c = 0; c < CJ; ++c) {
                a[c] = 0.0;
                b[c] = 0.0;
            }
            for (int f0 = 0; f0 < P.m; f0 += FC) {
                const int fw = (int)((P.m - f0) < FC ? (P.m - f0) : FC);
                __syncthreads();
                if (!single) stage_rows<T>(Xs, X, P.X_indices, row0, P.n, P.m, f0, fw, tid);
                for (int e = tid; e < CJ * fw; e += DT) {
                    const int c = e / fw, ff = e - c * fw;
                    Ys[c * FC + ff] = (j0 + c < P.K) ? Y[(j0 + c) * P.m + f0 + ff] : (T)0;
                }
                __syncthreads();
                for (int ff = 0; ff < fw; ++ff) {
                    const T x = Xs[tid * (FC + 1) + ff];
#pragma unroll
                    for (int c = 0; c < CJ; ++c) m_update<T, M>(a[c], b[c], x, Ys[c * FC + ff]);
                }
            }
#pragma unroll
            for (int c = 0; c < CJ; ++c) {
                if (j0 + c < P.K) {
                    const double d = m_final<M>(a[c], b[c], P.m);
                    if (MODE == 0) {
                        if (d < min_d) {
                            min_d = d;
                            lab = j0 + c;
                        }
                    } else if (i < P.n) {
                        P.out[i * P.K + j0 + c] = d;
                    }
                }
            }
        }
        if (MODE == 0 && i < P.n) {
            P.labels[i] = lab;
            if (P.min_dist) P.min_dist[i] = min_d;
            inertia += min_d;
        }
    }
    if (MODE == 0) {
        red[tid] = inertia;
        __syncthreads();
        for (int s = DT / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) P.partial[blockIdx.x] = red[0];
    }
}

// Same contract as pair_kernel for m <= FC and contiguous rows (no X_indices): rows in registers.
template <typename T, int M, int MODE>
__global__ __launch_bounds__(DT) void pair_small_kernel(PairArgs P)
{
    constexpr int FC = FeatChunk<T>::FC;
    __shared__ T Ys[CJ * FC];
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    const T* Y = static_cast<const T*>(P.Y);
    const int tid = threadIdx.x;
    const int m = (int)P.m;
    double inertia = 0.0;
    const long long ntile = (P.n + DT - 1) / DT;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long i = t * DT + tid;
        T x[FC];
        load_row_regs<T>(x, X + (i < P.n ? i : P.n - 1) * P.m, m, P.vecw);
        double min_d = 1.7976931348623157e308;
        long long lab = 0;
        for (long long j0 = 0; j0 < P.K; j0 += CJ) {
            __syncthreads();
            for (int e = tid; e < CJ * FC; e += DT) {
                const int c = e / FC, ff = e % FC;
                Ys[e] = (j0 + c < P.K && ff < m) ? Y[(j0 + c) * P.m + ff] : (T)0;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CJ; ++c) {
                if (j0 + c < P.K) {
                    double a = 0.0, b = 0.0;
                    // zero padding is exact for every metric (a 0/0 pair adds nothing), so the
                    // feature loop is predicated per group of 4, not per element
#pragma unroll
                    for (int g = 0; g < FC / 4; ++g)
                        if (g * 4 < m) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) m_update<T, M>(a, b, x[g * 4 + q], Ys[c * FC + g * 4 + q]);
                        }
                    const double d = m_final<M>(a, b, P.m);
                    if (MODE == 0) {
                        if (d < min_d) {
                            min_d = d;
                            lab = j0 + c;
                        }
                    } else if (i < P.n) {
                        P.out[i * P.K + j0 + c] = d;
                    }
                }
            }
        }
        if (MODE == 0 && i < P.n) {
            P.labels[i] = lab;
            if (P.min_dist) P.min_dist[i] = min_d;
            inertia += min_d;
        }
    }
    if (MODE == 0) {
        red[tid] = inertia;
        __syncthreads();
        for (int s = DT / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) P.partial[blockIdx.x] = red[0];
    }
}

// assign_nearest for short rows (m <= FC, contiguous, no X_indices), the KCenters.predict shape: VALU-bound exact
// arithmetic, so the kernel is built around the fp64 issue rate.
//  * TWO rows per lane share every centre-element read from LDS (a broadcast ds_read feeds 2 x 3 fp64 operations; with
//    one row per lane the LDS pipe, not the VALU, was the limit: 4 waves x 4 clk per b64 read against 12 VALU cycles);
//  * a whole tile of centres (all of them when K m fits 32 KiB) is staged once per workgroup: no barrier inside the
//    centre loop;
//  * euclidean: the reference compares sqrt(a) (distance_kernels.h:67-77, assign.hpp:22-31), and so does this kernel --
//    but it only EVALUATES a square root when the comparison could depend on its rounding.  sqrt is monotone and
//    correctly rounded, so a candidate with a >= a_best can never win the strict `<`; one with a < a_best (1 - 2^-48)
//    wins for certain (the exact roots differ by more than 2 ulp); only candidates inside that sliver -- exact
//    near-ties -- take both roots and compare them.  One sqrt per row at the end gives min_dist.  Bit-identical
//    labels and distances (tests/test_gpu_libdistance.py, incl. constructed ties), ~1/4 fewer fp64 cycles per pair at m = 10.
template <typename T, int M>
__global__ __launch_bounds__(DT) void assign_small2_kernel(PairArgs P)
{
    constexpr int FC = FeatChunk<T>::FC;
    constexpr int YCAP = 32768 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T Ys[YCAP];
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    const T* Y = static_cast<const T*>(P.Y);
    const int tid = threadIdx.x;
    const int m = (int)P.m;
    constexpr int GS = 16 / (int)sizeof(T);  // features per group = one 16-byte LDS read: 4 floats / 2 doubles (m = 10
                                             // doubles is 5 exact groups; groups of 4 computed 12 elements for 10)
    const int mp = (m + GS - 1) / GS * GS;   // centre pitch: whole groups, zero padded (exact for every metric)
    const int KT = YCAP / mp;                // centres per LDS tile
    double inertia = 0.0;
    const long long ntile = (P.n + 2 * DT - 1) / (2 * DT);
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long i0 = t * (2 * DT) + tid, i1 = i0 + DT;
        T x0[FC], x1[FC];
        load_row_regs<T>(x0, X + (i0 < P.n ? i0 : P.n - 1) * P.m, m, P.vecw);
        load_row_regs<T>(x1, X + (i1 < P.n ? i1 : P.n - 1) * P.m, m, P.vecw);
        // euclidean: (best squared distance, certain-win threshold); other metrics: best distance
        double best0 = INFINITY, best1 = INFINITY, thr0 = INFINITY, thr1 = INFINITY;
        if (M != M_EUCLIDEAN) best0 = best1 = 1.7976931348623157e308;  // DBL_MAX, assign.hpp:20
        int lab0 = -1, lab1 = -1;
        for (long long j0 = 0; j0 < P.K; j0 += KT) {
            const int kt = (int)((P.K - j0) < KT ? (P.K - j0) : KT);
            __syncthreads();
            for (int e = tid; e < kt * mp; e += DT) {
                const int c = e / mp, ff = e - c * mp;
                Ys[e] = ff < m ? Y[(j0 + c) * P.m + ff] : (T)0;
            }
            __syncthreads();
#pragma unroll 2
            for (int c = 0; c < kt; ++c) {
                const T* yc = Ys + c * mp;
                double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
#pragma unroll
                for (int g = 0; g < FC / GS; ++g)
                    if (g * GS < m) {
#pragma unroll
                        for (int q = 0; q < GS; ++q) {
                            const T y = yc[g * GS + q];
                            m_update<T, M>(a0, b0, x0[g * GS + q], y);
                            m_update<T, M>(a1, b1, x1[g * GS + q], y);
                        }
                    }
                const int j = (int)(j0 + c);
                if (M == M_EUCLIDEAN) {
                    // one comparison on the common path (a >= best: cannot win); inside, the certain win or the rare
                    // exact near-tie that needs both roots
                    if (a0 < best0) {
                        if (a0 < thr0 || sqrt(a0) < sqrt(best0)) {
                            best0 = a0;
                            thr0 = a0 * (1.0 - 0x1p-48);
                            lab0 = j;
                        }
                    }
                    if (a1 < best1) {
                        if (a1 < thr1 || sqrt(a1) < sqrt(best1)) {
                            best1 = a1;
                            thr1 = a1 * (1.0 - 0x1p-48);
                            lab1 = j;
                        }
                    }
                } else {
                    const double d0 = m_final<M>(a0, b0, P.m), d1 = m_final<M>(a1, b1, P.m);
                    if (d0 < best0) {
                        best0 = d0;
                        lab0 = j;
                    }
                    if (d1 < best1) {
                        best1 = d1;
                        lab1 = j;
                    }
                }
            }
        }
        // no centre ever compared smaller (NaN rows, K = 0): label 0 and DBL_MAX, as the reference's initial values
        double d0 = 1.7976931348623157e308, d1 = 1.7976931348623157e308;
        if (lab0 >= 0) d0 = (M == M_EUCLIDEAN) ? sqrt(best0) : best0;
        if (lab1 >= 0) d1 = (M == M_EUCLIDEAN) ? sqrt(best1) : best1;
        if (i0 < P.n) {
            P.labels[i0] = lab0 < 0 ? 0 : lab0;
            if (P.min_dist) P.min_dist[i0] = d0;
            inertia += d0;
        }
        if (i1 < P.n) {
            P.labels[i1] = lab1 < 0 ? 0 : lab1;
            if (P.min_dist) P.min_dist[i1] = d1;
            inertia += d1;
        }
    }
    red[tid] = inertia;
    __syncthreads();
    for (int s = DT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) P.partial[blockIdx.x] = red[0];
}


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

// sharded driver: reduce the per-block partials of one pass to (max, lowest row) and fetch that row
template <typename T>
__global__ __launch_bounds__(DT) void kc_finalize_kernel(const KcPartial* __restrict__ part, int nblk,
                                                         const T* __restrict__ X, long long m,
                                                         KcPartial* __restrict__ best, T* __restrict__ row)
{
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const int tid = threadIdx.x;
    double bv = -1.0;
    long long bi = -1;
    for (int k = tid; k < nblk; k += DT) {
        const KcPartial q = part[k];
        if (q.i >= 0 && (bi < 0 || kc_better(q.v, q.i, bv, bi))) {
            bv = q.v;
            bi = q.i;
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
        best->v = rv[0];
        best->i = ri[0];
    }
    if (ri[0] >= 0)
        for (long long f = tid; f < m; f += DT) row[f] = X[ri[0] * m + f];
}

// Device-resident exchange for the multi-GPU driver (no host round trip per centre):
// candidate record of a rank = [max distance | GLOBAL row of the first maximum | its coordinates], all
// float64 (rows < 2^53 and float32 coordinates are exact); -1 / -1 when the shard is empty.
template <typename T>
__global__ __launch_bounds__(DT) void kc_candidate_kernel(const KcPartial* __restrict__ part, int nblk,
                                                          const T* __restrict__ X, long long m, long long row_offset,
                                                          double* __restrict__ cand)
{
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const int tid = threadIdx.x;
    double bv = -1.0;
    long long bi = -1;
    for (int k = tid; k < nblk; k += DT) {
        const KcPartial q = part[k];
        if (q.i >= 0 && (bi < 0 || kc_better(q.v, q.i, bv, bi))) {
            bv = q.v;
            bi = q.i;
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
    const long long w = ri[0];
    if (tid == 0) {
        cand[0] = w >= 0 ? rv[0] : -1.0;
        cand[1] = w >= 0 ? (double)(row_offset + w) : -1.0;
    }
    for (long long f = tid; f < m; f += DT) cand[2 + f] = w >= 0 ? (double)X[w * m + f] : 0.0;
}

// all ranks run this on the all-gathered records [world][2 + m]: the winner is the largest distance,
// ties to the lowest global row (numpy's argmax over the concatenated array); its coordinates become
// the next centre (y, and row `slot` of `centers`), its row id goes to ids[slot].
template <typename T>
__global__ __launch_bounds__(DT) void kc_select_kernel(const double* __restrict__ cands, int world, long long m,
                                                       T* __restrict__ y, T* __restrict__ centers,
                                                       msm_idx_t* __restrict__ ids, long long slot)
{
    __shared__ int win;
    if (threadIdx.x == 0) {
        int w = -1;
        for (int r = 0; r < world; ++r) {
            const double v = cands[(size_t)r * (2 + m)], g = cands[(size_t)r * (2 + m) + 1];
            if (g < 0.0) continue;
            if (w < 0 || v > cands[(size_t)w * (2 + m)] ||
                (v == cands[(size_t)w * (2 + m)] && g < cands[(size_t)w * (2 + m) + 1]))
                w = r;
        }
        win = w;
        ids[slot] = w >= 0 ? (msm_idx_t)cands[(size_t)w * (2 + m) + 1] : -1;
    }
    __syncthreads();
    const int w = win;
    for (long long f = threadIdx.x; f < m; f += DT) {
        const T v = w >= 0 ? (T)cands[(size_t)w * (2 + m) + 2 + f] : (T)0;
        y[f] = v;
        centers[slot * m + f] = v;
    }
}

// ---------------------------------------------------------------------------
// pdist (pdist.hpp:4-88): condensed upper triangle, row i -> out[i*n - i(i+1)/2 + (j-i-1)], and
// sumdist (sumdist.hpp:4-44): sum of metric over a pair list.  Same exact per-pair arithmetic:
// one lane per pair, features in order, one fp64 accumulator.  Row i is staged in LDS (broadcast
// reads); lane j walks its own row.
// ---------------------------------------------------------------------------
struct PdArgs {
    const void* X;
    const msm_idx_t* X_indices;  // nullable
    long long n, m;              // n = number of (indexed) rows
    double* out;
    const msm_idx_t* pairs;      // sumdist: [p][2]
    long long p;
    double* partial;             // sumdist: per-block sums
};

template <typename T, int M>
__global__ __launch_bounds__(DT) void pdist_kernel(PdArgs P)
{
    constexpr int UC = 1024;  // features of row i kept in LDS per sweep
    __shared__ T us[UC];
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x;
    for (long long ii = blockIdx.x; ii < P.n - 1; ii += gridDim.x) {
        const long long i = P.X_indices ? P.X_indices[ii] : ii;
        const long long base = ii * P.n - ii * (ii + 1) / 2 - ii - 1;  // + jj gives the condensed index
        for (long long jj0 = ii + 1; jj0 < P.n; jj0 += DT) {
            const long long jj = jj0 + tid;
            const long long j = (jj < P.n) ? (P.X_indices ? P.X_indices[jj] : jj) : 0;
            double a = 0.0, b = 0.0;
            for (long long f0 = 0; f0 < P.m; f0 += UC) {
                const int fw = (int)((P.m - f0) < UC ? (P.m - f0) : UC);
                __syncthreads();
                for (int f = tid; f < fw; f += DT) us[f] = X[i * P.m + f0 + f];
                __syncthreads();
                if (jj < P.n) {
                    const T* v = X + j * P.m + f0;
                    for (int f = 0; f < fw; ++f) m_update<T, M>(a, b, us[f], v[f]);
                }
            }
            if (jj < P.n) P.out[base + jj] = m_final<M>(a, b, P.m);
        }
    }
}

template <typename T, int M>
__global__ __launch_bounds__(DT) void sumdist_kernel(PdArgs P)
{
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    double s = 0.0;
    for (long long k = (long long)blockIdx.x * DT + threadIdx.x; k < P.p; k += (long long)gridDim.x * DT) {
        const T* u = X + P.pairs[2 * k] * P.m;
        const T* v = X + P.pairs[2 * k + 1] * P.m;
        double a = 0.0, b = 0.0;
        for (long long f = 0; f < P.m; ++f) m_update<T, M>(a, b, u[f], v[f]);
        s += m_final<M>(a, b, P.m);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) P.partial[blockIdx.x] = red[0];
}

// deterministic per-block fp64 sums of a vector (inertia = np.sum(distances_))
__global__ __launch_bounds__(DT) void sum_partial_kernel(const double* __restrict__ v, long long n,
                                                         double* __restrict__ partial)
{
    __shared__ double red[DT];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * DT + threadIdx.x; i < n; i += (long long)gridDim.x * DT) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}


// ---- host-side dispatch ----------------------------------------------------
// widest aligned per-lane vector load for a [*, m] row-major array, 0 if the fast path does not apply
template <typename T>
static int row_vecw(const void* X, long long m, bool has_indices)
{
    if (has_indices || m > FeatChunk<T>::FC) return 0;
    const size_t rb = (size_t)m * sizeof(T);
    const uintptr_t a = (uintptr_t)X;
    if (a % 16 == 0 && rb % 16 == 0) return 16;
    if (a % 8 == 0 && rb % 8 == 0) return 8;
    return (a % sizeof(T) == 0) ? (int)sizeof(T) : 0;
}

constexpr size_t wide_lds(int ncl) { return (size_t)2 * DT * WP * 4 + (size_t)2 * ncl * 32 * 4 + (size_t)DT * 16; }

// wide-row streaming path applies: long rows, 16-byte aligned vectors, no row gather
template <typename T>
static bool wide_ok(const void* X, const void* Y, long long m, bool has_indices)
{
    constexpr int E = 16 / (int)sizeof(T);
    return !has_indices && m > FeatChunk<T>::FC && (m % E) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0 &&
           (size_t)m * sizeof(T) < ((size_t)1 << 24);
}

static int wide_grid(long long n)
{
    return (int)std::min<long long>(ceil_div(n, DT), 2LL * num_cus());  // one resident round (2 workgroups / CU)
}

template <typename T, int MM, int MODE, int NCT>
static void launch_wide_nc(int grid, const WideArgs& A)
{
    constexpr size_t lds = wide_lds(MODE == 2 ? WNC : NCT);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_kernel<T, MM, MODE, NCT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wide_kernel<T, MM, MODE, NCT>), dim3(grid), dim3(DT), lds, stream(), A);
}

template <typename T, int MM, int MODE>
static void launch_wide(int grid, const WideArgs& A)
{
    // assign_nearest picks its centre-group size by shape
    if (MODE == 0) {
        if (A.pa.K <= 8) return launch_wide_nc<T, MM, 0, 8>(grid, A);
        // (32-centre groups for float64 rows were measured too: 2M x 256 x K = 100 8.74 ms against 8.31 ms with 16 -- the
        //  4 KiB of extra LDS cost the second workgroup of a CU and more than the halved re-streaming of the row tile gains)
    }
    launch_wide_nc<T, MM, MODE, WNC>(grid, A);
}

template <typename T, int MODE>
void launch_pair(int metric, int grid, const PairArgs& P)
{
    const bool wide = wide_ok<T>(P.X, P.Y, P.m, P.X_indices != nullptr);
    WideArgs A;
    memset(&A, 0, sizeof(A));
    A.pa = P;
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (P.vecw > 0 && MODE == 0 && (MM == M_EUCLIDEAN || MM == M_SQEUCLIDEAN) &&              \
            (sizeof(T) == 4 ? launch_small3_f32(MM, grid, P) : launch_small3_f64(MM, grid, P))) { \
        } else if (P.vecw > 0 && MODE == 0)                                                       \
            hipLaunchKernelGGL((assign_small2_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P); /* every block writes its partial */ \
        else if (P.vecw > 0)                                                                      \
            hipLaunchKernelGGL((pair_small_kernel<T, MM, MODE>), dim3(grid), dim3(DT), 0, stream(), P); \
        else if (wide)                                                                            \
            launch_wide<T, MM, MODE>(grid, A);                                                    \
        else                                                                                      \
            hipLaunchKernelGGL((pair_kernel<T, MM, MODE>), dim3(grid), dim3(DT), 0, stream(), P); \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

template <typename T>
void launch_kc(int metric, int grid, const KcArgs& P)
{
    const bool wide = P.vecw == 0 && wide_ok<T>(P.X, P.ycenter ? P.ycenter : P.X, P.m, false);
    WideArgs A;
    memset(&A, 0, sizeof(A));
    A.kc = P;
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (wide)                                                                                 \
            launch_wide<T, MM, 2>(grid, A);                                                       \
        else if (P.vecw > 0)                                                                      \
            hipLaunchKernelGGL((kcenters_pass_kernel<T, MM, true>), dim3(grid), dim3(DT), 0, stream(), P); \
        else                                                                                      \
            hipLaunchKernelGGL((kcenters_pass_kernel<T, MM, false>), dim3(grid), dim3(DT), 0, stream(), P); \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

template <typename T, int WHICH>
void launch_pd(int metric, int grid, const PdArgs& P)
{
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (WHICH == 0)                                                                           \
            hipLaunchKernelGGL((pdist_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P);      \
        else                                                                                      \
            hipLaunchKernelGGL((sumdist_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P);    \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

static int sum_partials_host(const double* dpartial, int n, double* out)
{
    std::vector<double> h((size_t)n);
    MSM_HIP_CHECK(hipMemcpyAsync(h.data(), dpartial, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += h[i];
    *out = s;
    return MSM_OK;
}

// Rows whose length is not a multiple of 16 bytes (171 float32 contact features: SURVEY 8(d)'s C3 stress variant) cannot take
// the wide-row streaming kernels and fell to the scalar-staged pair kernel -- 20x slower per pair-element (280,000 x 171 x
// K = 200: assign_nearest 32.7 ms, a k-centers pass 134 us at 1.4 TB/s; profiles/r04_pmc_all_first.txt).  For the NORM
// metrics a zero column is an exact no-op in the reference's arithmetic (a float difference 0 - 0 = +0, then s + 0*0 = s,
// s + |0| = s, max(s, 0) = s for s >= 0), so such rows are copied once into a buffer zero-padded to the next multiple of 16
// bytes -- one streaming pass, 2 x the input bytes -- and everything downstream sees aligned rows.  Bit-identical outputs.
template <typename T>
static bool pad_rows_pays(int mid, long long m, long long n, bool has_indices)
{
    constexpr int E = 16 / (int)sizeof(T);
    return !has_indices && (mid == M_EUCLIDEAN || mid == M_SQEUCLIDEAN || mid == M_CITYBLOCK || mid == M_CHEBYSHEV) &&
           m > FeatChunk<T>::FC && (m % E) != 0 && n * m >= (1LL << 20);
}

template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ X, long long n, long long m, long long mp, T* __restrict__ out)
{
    constexpr int E = 16 / (int)sizeof(T);
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x, per = mp / E;   // 16-byte groups
    if (g >= n * per) return;
    const long long i = g / per, c = (g - i * per) * E;
    T v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = c + e < m ? X[i * m + c + e] : (T)0;
#pragma unroll
    for (int e = 0; e < E; ++e) out[i * mp + c + e] = v[e];
}

template <typename T>
static int pad_rows(const T* X, long long n, long long m, DevBuf& buf, const T** out, long long* mp_out)
{
    constexpr int E = 16 / (int)sizeof(T);
    const long long mp = (m + E - 1) / E * E;
    int rc = buf.reserve((size_t)std::max<long long>(n, 1) * mp * sizeof(T));
    if (rc) return rc;
    if (n > 0) {
        const long long groups = n * (mp / E);
        hipLaunchKernelGGL(pad_rows_kernel<T>, dim3((unsigned)ceil_div(groups, 256)), dim3(256), 0, stream(), X, n, m, mp, buf.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
    }
    *out = buf.as<T>();
    *mp_out = mp;
    return MSM_OK;
}

template <typename T>
int assign_nearest_impl(const T* X, const T* Y, const char* metric, const msm_idx_t* X_indices,
                        msm_idx_t n_X, msm_idx_t n_Y, msm_idx_t m, msm_idx_t n_idx,
                        msm_idx_t* assignments, double* min_dist, double* inertia, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !Y || !assignments) return fail(MSM_ERR_INVALID, "assign_nearest: null pointer");
    if (n_X < 0 || n_Y < 0 || m < 1) return fail(MSM_ERR_INVALID, "assign_nearest: bad shape");
    const long long n = X_indices ? n_idx : n_X;
    if (inertia) *inertia = 0.0;
    if (n == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dY = pool(PS_Y), &dIdx = pool(PS_IDX), &dLab = pool(PS_LAB), &dMin = pool(PS_MIN),
           &dPart = pool(PS_PART);
    int grid = (int)std::min<long long>(ceil_div(n, DT), 2048);
    if ((rc = dY.reserve((size_t)(n_Y ? n_Y : 1) * m * sizeof(T)))) return rc;
    if (n_Y) MSM_HIP_CHECK(hipMemcpyAsync(dY.p, Y, (size_t)n_Y * m * sizeof(T), hipMemcpyHostToDevice, stream()));
    if ((rc = dPart.reserve((size_t)grid * sizeof(double)))) return rc;
    PairArgs P;
    memset(&P, 0, sizeof(P));
    P.Y = dY.p;
    P.n = n;
    P.K = n_Y;
    P.m = m;
    P.partial = dPart.as<double>();
    if (on_device) {
        P.X = X;
        P.X_indices = X_indices;
        P.labels = assignments;
        P.min_dist = min_dist;
    } else {
        if ((rc = dX.reserve((size_t)n_X * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n_X * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)n * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dLab.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
        P.labels = dLab.as<msm_idx_t>();
        if (min_dist) {
            if ((rc = dMin.reserve((size_t)n * sizeof(double)))) return rc;
            P.min_dist = dMin.as<double>();
        }
    }
    if (pad_rows_pays<T>(mid, m, n_X, P.X_indices != nullptr)) {
        const T *xp = nullptr, *yp = nullptr;
        long long mp = m;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.X), n_X, m, pool(PS_PADX), &xp, &mp))) return rc;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.Y), n_Y, m, pool(PS_PADY), &yp, &mp))) return rc;
        P.X = xp;
        P.Y = yp;
        P.m = m = mp;
    }
    P.vecw = row_vecw<T>(P.X, m, P.X_indices != nullptr);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.Y, m, P.X_indices != nullptr)) grid = wide_grid(n);
    launch_pair<T, 0>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device) {
        MSM_HIP_CHECK(hipMemcpyAsync(assignments, P.labels, (size_t)n * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
        if (min_dist)
            MSM_HIP_CHECK(hipMemcpyAsync(min_dist, P.min_dist, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    double s = 0.0;
    if ((rc = sum_partials_host(P.partial, grid, &s))) return rc;  // synchronises
    if (inertia) *inertia = s;
    return MSM_OK;
}

template <typename T>
int cdist_impl(const T* XA, const T* XB, const char* metric, msm_idx_t na, msm_idx_t nb,
               msm_idx_t m, const msm_idx_t* X_indices, msm_idx_t n_idx, double* out, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!XA || !XB || !out) return fail(MSM_ERR_INVALID, "cdist/dist: null pointer");
    if (na < 0 || nb < 0 || m < 1) return fail(MSM_ERR_INVALID, "cdist/dist: bad shape");
    const long long n = X_indices ? n_idx : na;
    if (n == 0 || nb == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dY = pool(PS_Y), &dIdx = pool(PS_IDX), &dOut = pool(PS_OUT);
    int grid = (int)std::min<long long>(ceil_div(n, DT), 2048);
    if ((rc = dY.reserve((size_t)nb * m * sizeof(T)))) return rc;
    if ((rc = h2d_bulk(dY.p, XB, (size_t)nb * m * sizeof(T)))) return rc;
    PairArgs P;
    memset(&P, 0, sizeof(P));
    P.Y = dY.p;
    P.n = n;
    P.K = nb;
    P.m = m;
    if (on_device) {
        P.X = XA;
        P.X_indices = X_indices;
        P.out = out;
    } else {
        if ((rc = dX.reserve((size_t)na * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, XA, (size_t)na * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)n * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dOut.reserve((size_t)n * nb * sizeof(double)))) return rc;
        P.out = dOut.as<double>();
    }
    P.vecw = row_vecw<T>(P.X, m, P.X_indices != nullptr);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.Y, m, P.X_indices != nullptr)) grid = wide_grid(n);
    launch_pair<T, 1>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device)
        MSM_HIP_CHECK(hipMemcpyAsync(out, P.out, (size_t)n * nb * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // scratch (and the staged Y) die with this frame
    return MSM_OK;
}

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

// ---------------------------------------------------------------------------
// Several centres per pass (round 3; byte copy, single process).
//
// k-centers is sequential -- centre k+1 is the argmax of the distances AFTER centre k -- but the argmax can usually be
// read off a short list.  A pass appends every row whose updated, rounded-up distance exceeds a threshold theta to a list
// (`curf > theta` implies distance > theta; every row NOT listed has distance <= theta =: tau, and distances only
// shrink).  kcb_select_kernel, one workgroup, then plays the algorithm on the list alone: the listed row of largest
// distance (lowest row on ties) is the next centre -- it beats every unlisted row strictly; the remaining listed rows get
// d = min(d, dist(row, centre)) in the pass kernel's exact arithmetic; the largest of them is the centre after that IF it
// still exceeds tau, and so on, up to KCB_JMAX centres.  The next pass applies them all, in order, to every row it streams
// (a row's candidate centres by the screen, then the exact `d < distances_` of kcenters.py:93 centre after centre): the
// centres, labels_ and distances_ of the one-centre-per-pass loop, in a fraction of its passes (simulated on a 10-dimensional
// projection: 30 passes instead of 199 with lists of 16).  The per-block argmax partials are still written: the first
// centre of a batch must be the row they name (numpy's argmax under this file's NaN rules), otherwise -- and whenever the
// list is empty or overflowed -- the batch is that one row.  theta follows the data: a pass also counts the rows above five
// lower levels, and the selector takes the lowest level that held at most KCB_TARGET rows (counts at a fixed level can only
// fall from pass to pass, so the next list fits).
// ---------------------------------------------------------------------------
constexpr int KCB_JMAX = 32, KCB_CAP = 2048, KCB_NLEV = 6, KCB_TARGET = 1536;
// ---- wave argmax of (value, row): largest value, lowest row among equal values; rows < 0 do not take part ---------------
// The value goes through DPP row operations and readlanes (a 64-bit __shfl_xor is two ds_bpermute round trips per step:
// the selection kernels make ~35 block reductions between two passes and were 30-45 us, most of it shuffles).
template <int CTRL>
__device__ __forceinline__ double kcb_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kcb_readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void kcb_wave_argmax(double v, long long i, double& ov, long long& oi)
{
    const double w = i >= 0 ? v : -1.0;     // distances are >= 0; a NaN loses every fmax
    double x = w;
    x = fmax(x, kcb_dpp_f64<0xB1>(x));      // quad_perm [1,0,3,2]
    x = fmax(x, kcb_dpp_f64<0x4E>(x));      // quad_perm [2,3,0,1]
    x = fmax(x, kcb_dpp_f64<0x141>(x));     // row_half_mirror
    x = fmax(x, kcb_dpp_f64<0x140>(x));     // row_mirror: every lane holds the maximum of its row of 16
    const double vm = fmax(fmax(kcb_readlane_f64(x, 0), kcb_readlane_f64(x, 16)), fmax(kcb_readlane_f64(x, 32), kcb_readlane_f64(x, 48)));
    unsigned long long mask = __builtin_amdgcn_ballot_w64(i >= 0 && w == vm);
    if (!mask) mask = __builtin_amdgcn_ballot_w64(i >= 0);   // only NaN values took part: the lowest row, like a scan that never sees `>`
    long long best = -1;
    while (mask) {   // one lane, except on exact ties
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const long long c = ((long long)__builtin_amdgcn_readlane((int)(i >> 32), l) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(i & 0xffffffffLL), l);
        if (best < 0 || c < best) best = c;
    }
    ov = vm;
    oi = best;
}


struct KcbState {
    int k_done;               // centres fixed so far: ids[0 .. k_done)
    int J;                    // centres the next pass applies: ids[k_done - J .. k_done)
    int rounds, fallbacks;
    float theta;              // listing threshold of the last pass = bound on every row it did not list
    unsigned count;           // rows it listed (more than KCB_CAP: list unusable)
    unsigned lev[KCB_NLEV];   // rows above theta * kcb_level(l) after it; lev[0] mirrors count
    double cen[KCB_JMAX][16];
    long long list[KCB_CAP];
};
// (levels below 0.91 were tried in round 4 -- twelve levels down to 0.58: the number of rounds did not move, 19 on the bench's
//  projection at 10M and at 1.25M rows: what ends a round is the list's capacity, not the threshold's rate of descent -- and
//  the extra level counters cost 15 % of a fit)
// Rounds the host queues before it looks at the progress counter again.  A synchronisation costs 35-50 us of idle GPU, an
// empty round (all K centres fixed: three early-returning launches) about 14: so the first group aims at the whole fit at
// a typical 12 centres per round, and the later ones at what is left at the rate seen so far, plus one.  The value depends
// on nothing but K and the counter, which every rank of a sharded fit holds identically.
static int kcb_group(int K, int done, int rounds_so_far, int done_at_start)
{
    const int left = K - done;
    if (left <= 0) return 0;
    int per = 12;
    if (rounds_so_far > 0) per = std::max(1, (done - done_at_start) / rounds_so_far);
    const int g = (left + per - 1) / per + (rounds_so_far > 0 ? 1 : 0);
    return std::min(std::max(g, 1), 24);
}

// the state before the first round: `k_done` centres fixed by the plain passes, no list yet.  (A launch instead of a copy
// from the host's stack and the synchronisation that keeps the stack alive: 20-30 us per fit.)
__global__ void kcb_init_kernel(KcbState* S, int k_done)
{
    if (threadIdx.x == 0) {
        S->k_done = k_done;
        S->J = S->rounds = S->fallbacks = 0;
        S->theta = INFINITY;
        S->count = 0;
    }
    if (threadIdx.x < KCB_NLEV) S->lev[threadIdx.x] = 0;
}
__device__ __forceinline__ float kcb_level(int l) { return l == 0 ? 1.f : l == 1 ? 0.985f : l == 2 ? 0.97f : l == 3 ? 0.955f : l == 4 ? 0.94f : 0.91f; }

template <int NP>
__global__ __launch_bounds__(1024) void kcb_select_kernel(KscArgs P, KcbState* S, int K)
{
    __shared__ double rv[1024];
    __shared__ long long ri[1024];
    __shared__ double cs[16];
    const int tid = threadIdx.x, m = (int)P.m;
    const int k0 = S->k_done;
    if (k0 >= K) {
        if (tid == 0) S->J = 0;
        return;
    }
    // block argmax (largest value, lowest row on ties; rows < 0 never win): inside a wave by kcb_wave_argmax, then every wave
    // reduces the 16 wave winners by itself -- two barriers per call (the selection loop makes up to 17 calls between passes)
    auto reduce = [&](double v, long long i, double& ov, long long& oi) {
        double wv;
        long long wi;
        kcb_wave_argmax(v, i, wv, wi);
        __syncthreads();   // the previous call's readers are done with rv / ri
        if ((tid & 63) == 0) {
            rv[tid >> 6] = wv;
            ri[tid >> 6] = wi;
        }
        __syncthreads();
        const int l = tid & 63;
        kcb_wave_argmax(l < 16 ? rv[l] : -1.0, l < 16 ? ri[l] : -1, ov, oi);
    };
    // the row the per-block partials of the last pass name (the one-centre-per-pass loop's choice)
    double vP;
    long long iP;
    {
        double v = -1.0;
        long long i = -1;
        if (tid < P.nblk) {
            const KcPartial q = P.prev[tid];
            if (q.i >= 0) {
                v = q.v;
                i = q.i;
            }
        }
        reduce(v, i, vP, iP);
    }
    const float theta = S->theta;
    const unsigned cnt = S->count;
    const bool usable = cnt > 0 && cnt <= (unsigned)KCB_CAP && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    // this thread's (up to) two listed rows: index, current distance, coordinates
    long long ci[2] = {-1, -1};
    double cv[2] = {-1.0, -1.0}, cx[2][2 * NP];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned c = (unsigned)tid + 1024u * u;
        if (usable && c < cnt) {
            ci[u] = S->list[c];
            cv[u] = P.dist[ci[u]];
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) cx[u][f] = f < m ? P.X[ci[u] * P.m + f] : 0.0;
        }
    }
    int J = 0, fell = 0;
    double vlast = vP;
    for (;;) {
        double v = -1.0, vb;
        long long i = -1, ib;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (ci[u] >= 0 && (i < 0 || kc_better(cv[u], ci[u], v, i))) {
                v = cv[u];
                i = ci[u];
            }
        reduce(v, i, vb, ib);
        long long centre;
        if (J == 0) {
            if (usable && ib == iP) {
                centre = ib;
            } else {
                centre = iP;
                fell = 1;
            }
            vlast = vP;
        } else {
            if (!(ib >= 0 && vb > tau)) break;
            centre = ib;
            vlast = vb;
        }
        if (fell) {
            if (tid < 16) cs[tid] = tid < m ? P.X[centre * P.m + tid] : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ci[u] == centre) {   // the thread that holds the row: no trip to global memory inside the loop
#pragma unroll
                    for (int f = 0; f < 16; ++f) cs[f] = f < 2 * NP ? cx[u][f] : 0.0;
                }
        }
        if (tid == 0) P.ids[k0 + J] = centre;
        __syncthreads();
        if (tid < 16) S->cen[J][tid] = cs[tid];
        ++J;
        if (fell || k0 + J >= K || J >= KCB_JMAX) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ci[u] < 0) continue;
            if (ci[u] == centre) {
                ci[u] = -1;
                continue;
            }
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, cx[u][f], cs[f]);
            const double d = m_final<M_EUCLIDEAN>(a, b, P.m);
            if (d < cv[u]) cv[u] = d;   // the pass's own update (kcenters.py:93)
        }
        __syncthreads();
    }
    if (tid == 0) {
        // threshold of the next list
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (cnt > (unsigned)KCB_TARGET) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (S->lev[q] <= (unsigned)KCB_TARGET) l = q;
            th = theta * kcb_level(l);
        }
        if (th > vl) th = vl;
        S->theta = th;
        S->count = 0;
        for (int q = 0; q < KCB_NLEV; ++q) S->lev[q] = 0;
        S->J = J;
        S->k_done = k0 + J;
        S->rounds += 1;
        S->fallbacks += fell;
    }
}

// ---- row-sharded fit: the same rounds with ONE exchange per round -----------------------------------------------------
// A rank's round record (doubles): [0] rows it listed (more than KCB_CAPR: list unusable), [8] value and [9] GLOBAL row of its
// per-block-partials argmax (-1: none), [10..25] that row's coordinates, [32 + l] the count of level l, then from [KCB_HDR]
// on the listed rows as {distance, global row, coordinates[m]}.  The records are all-gathered and every rank runs the same selection on
// the same numbers: no rank learns anything another does not, so the batches -- and the number of rounds -- agree.
constexpr int KCB_CAPR = 1024, KCB_HDR = 48, KCB_LEV0 = 32;
__host__ __device__ constexpr size_t kcb_rec_doubles(long long m) { return (size_t)KCB_HDR + (size_t)KCB_CAPR * (size_t)(2 + m); }

// the records of the first round, made from the one-centre protocol's gathered candidates {value, global row, coordinates}
__global__ void kcb_boot_records_kernel(const double* __restrict__ cands, int world, long long m, double* __restrict__ recs)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= world) return;
    const double* c = cands + (size_t)r * (2 + m);
    double* o = recs + (size_t)r * kcb_rec_doubles(m);
    if (tid < KCB_HDR) {
        double v = 0.0;
        if (tid == 8) v = c[0];
        else if (tid == 9) v = c[1];
        else if (tid >= 10 && tid < 10 + 16) v = tid - 10 < m ? c[2 + tid - 10] : 0.0;
        o[tid] = v;
    }
}

// the shard's record of a round.  (A launch of its own: folding it into the pass kernel -- the last workgroup to arrive packs
// -- was tried in round 4 and cost 24 us per pass instead of the 7 + 4 us of this launch: the agent-scope release that
// every one of the pass's ~5,000 workgroups must then make before it counts itself in is an L2 write-back each.)
template <int NP>
__global__ __launch_bounds__(DT) void kcb_pack_kernel(KscArgs P, KcbState* S, double* __restrict__ rec)
{
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const int tid = threadIdx.x, m = (int)P.m;
    if (S->J == 0) return;   // an empty round: nobody reads the record
    double bv = -1.0;
    long long bi = -1;
    for (int k = tid; k < P.nblk; k += DT) {
        const KcPartial q = P.next[k];
        if (q.i >= 0 && (bi < 0 || kc_better(q.v, q.i, bv, bi))) {
            bv = q.v;
            bi = q.i;
        }
    }
    rv[tid] = bv;
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
    const long long w = ri[0];
    const unsigned cnt = S->count;
    if (tid < KCB_HDR) {
        double v = 0.0;
        if (tid == 0) v = (double)cnt;
        else if (tid > KCB_LEV0 && tid < KCB_LEV0 + KCB_NLEV) v = (double)S->lev[tid - KCB_LEV0];
        else if (tid == 8) v = w >= 0 ? rv[0] : -1.0;
        else if (tid == 9) v = w >= 0 ? (double)(P.row_offset + w) : -1.0;
        else if (tid >= 10 && tid < 26) v = (w >= 0 && tid - 10 < m) ? P.X[w * P.m + (tid - 10)] : 0.0;
        rec[tid] = v;
    }
    const unsigned ne = cnt <= (unsigned)KCB_CAPR ? cnt : 0u;
    for (unsigned e = tid; e < ne; e += DT) {
        const long long p = S->list[e];
        double* o = rec + KCB_HDR + (size_t)e * (2 + m);
        o[0] = P.dist[p];
        o[1] = (double)(P.row_offset + p);
        for (int f = 0; f < m; ++f) o[2 + f] = P.X[p * P.m + f];
    }
}

template <int NP>
__global__ __launch_bounds__(DT) void kcenters_batch_pass_kernel(KscArgs P, KcbState* S)
{
    constexpr int R = 2;
    constexpr int NW = ksc_words(NP, 2), RW = NW + 1;
    constexpr int SB = 2 * NP;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) f32x2 ycf[KCB_JMAX][NP];
    __shared__ double yd[KCB_JMAX][2 * NP];
    __shared__ float epsb[KCB_JMAX];
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    __shared__ unsigned slev[KCB_NLEV];
    const int tid = threadIdx.x, m = (int)P.m;
    const int J = S->J;
    if (J == 0) return;
    const int kbase = S->k_done - J;
    const float theta = S->theta;
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
    // the batch's centres: exact coordinates, float32 coordinates relative to the copy's origin, and the part of eps that
    // belongs to the centre (see kcenters_screen_pass_kernel for the terms)
    if (tid < KCB_NLEV) slev[tid] = 0;
    for (int e = tid; e < KCB_JMAX * 2 * NP; e += DT) {
        const int j = e / (2 * NP), f = e - j * (2 * NP);
        const double y = j < J ? S->cen[j][f] : 0.0;
        yd[j][f] = y;
        reinterpret_cast<float*>(&ycf[j][0])[f] = (float)(y - (f < m ? P.c0[f] : 0.0));
    }
    if (tid < KCB_JMAX) {
        double c0n2 = 0.0, yn2 = 0.0, ycn2 = 0.0;
        for (int f = 0; f < 2 * NP; ++f) {
            const double y = tid < J ? S->cen[tid][f] : 0.0, c0f = f < m ? P.c0[f] : 0.0;
            c0n2 = fma(c0f, c0f, c0n2);
            yn2 = fma(y, y, yn2);
            ycn2 = fma(y - c0f, y - c0f, ycn2);
        }
        const double g2 = __longlong_as_double((long long)P.gmax2[0]), r2 = __longlong_as_double((long long)P.gmax2[1]);
        double eps0 = (sqrt(r2) + sqrt(yn2) + 2.0 * sqrt(c0n2)) * 0x1p-48 + 1e-37;
        if (!(g2 < 1e36) || !(r2 < 1e76) || !(ycn2 < 1e36)) eps0 = NAN;
        epsb[tid] = fmaf(0x1p-19f, (float)(sqrt(ycn2) * 1.000001), (float)(eps0 * 1.000001));
    }
    __syncthreads();
    constexpr float E32 = 0x1p-19f;
    constexpr float QSQ = NP == 1 ? 1.4143f : NP == 2 ? 2.f : NP == 3 ? 2.4495f : NP == 4 ? 2.8285f : NP == 5 ? 3.1623f
                        : NP == 6 ? 3.4642f : NP == 7 ? 3.7417f : 4.f;
    constexpr float QA = 0.51f * 1.02f * QSQ + E32 * 127.f * QSQ * 1.001f;
    float bf = -1.f;
    long long bi = -1;
    double bx = 0.0;
    bool bknown = false;
    unsigned nlev[KCB_NLEV];
#pragma unroll
    for (int l = 0; l < KCB_NLEV; ++l) nlev[l] = 0;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        float cf[R];
        unsigned cmask[R];
        long long pr[R];
        unsigned q[R][RW];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            pr[k] = t * (R * DT) + k * DT + tid;
#pragma unroll
            for (int j = 0; j < RW; ++j) q[k][j] = qn[k][j];
            cf[k] = __uint_as_float(q[k][NW]);
        }
        if (t + gridDim.x < ntile) load_tile(t + gridDim.x);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const float sf = __uint_as_float(((q[k][SB >> 2] >> (8 * (SB & 3))) & 0xffffu) << 16);
            f32x2 xt[NP];
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                xt[g].x = (float)((int)(q[k][(2 * g) >> 2] << (24 - 8 * ((2 * g) & 3))) >> 24) * sf;       // q sf: exact
                xt[g].y = (float)((int)(q[k][(2 * g + 1) >> 2] << (24 - 8 * ((2 * g + 1) & 3))) >> 24) * sf;
            }
            // a row is left alone by centre j when  sqrt(a_j) - eps_j >= curf.  Compared as squares, without the square root:
            // a_j >= T^2 with T = (curf + eps_j)(1 + 2^-20) evaluated in float32 (three roundings of 2^-24 each, and one more
            // in the product T T, leave T^2 above the real (curf + eps_j)^2): the real-arithmetic inequality with room to
            // spare -- eps_j already allows for float32 roundings of the original form
            const float base = cf[k] + sf * QA;
            unsigned mk = 0;
            for (int j = 0; j < J; ++j) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int g = 0; g < NP; ++g) {
                    const f32x2 d = xt[g] - ycf[j][g];
                    acc = __builtin_elementwise_fma(d, d, acc);
                }
                const float T = (base + epsb[j]) * (1.f + 0x1p-20f);
                if (!(acc.x + acc.y >= T * T)) mk |= 1u << j;
            }
            cmask[k] = pr[k] < P.n ? mk : 0u;
        }
        if (cmask[0] | cmask[1]) {
            double x[R][2 * NP], cur[R];
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
                unsigned mk = cmask[k];
                int lab = -1;
                double c = cur[k];
                while (mk) {   // the batch's centres in order, as the separate passes would meet the row
                    const int j = __builtin_ctz(mk);
                    mk &= mk - 1;
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, x[k][f], yd[j][f]);
                    const double d = m_final<M_EUCLIDEAN>(a, b, P.m);
                    if (d < c) {   // strict, kcenters.py:93
                        c = d;
                        lab = kbase + j;
                    }
                }
                if (lab >= 0) {
                    P.dist[pr[k]] = c;
                    P.labels[pr[k]] = lab;
                    cf[k] = ksc_round_up(c);
                    static_cast<unsigned*>(P.xs)[pr[k] * RW + NW] = __float_as_uint(cf[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long p = pr[k];
            const bool in = p < P.n;
            // the list of the next selection, and the level counts that place its threshold
            const bool lst = in && cf[k] > theta;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(lst);
            if (bal) {
                const int lane = tid & 63, leader = __builtin_ctzll(bal);
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&S->count, (unsigned)__builtin_popcountll(bal));
                base = __shfl(base, leader);
                if (lst) {
                    const unsigned slot = base + (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (slot < (unsigned)KCB_CAP) S->list[slot] = p;
                }
            }
#pragma unroll
            for (int l = 1; l < KCB_NLEV; ++l) nlev[l] += (in && cf[k] > theta * kcb_level(l)) ? 1u : 0u;
            if (in) {
                if (cf[k] > bf || bi < 0) {
                    bf = cf[k];
                    bi = p;
                    bknown = false;
                } else if (cf[k] == bf) {
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
#pragma unroll
    for (int l = 1; l < KCB_NLEV; ++l)
        if (nlev[l]) atomicAdd(&slev[l], nlev[l]);
    double bvx = -1.0;
    if (bi >= 0) bvx = bknown ? bx : P.dist[bi];
    rv[tid] = bvx;
    ri[tid] = bi;
    __syncthreads();
    if (tid >= 1 && tid < KCB_NLEV && slev[tid]) atomicAdd(&S->lev[tid], slev[tid]);
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
        P.next[blockIdx.x] = q;
    }
}

// a rank without rows: nothing listed, no argmax
__global__ void kcb_empty_record_kernel(double* __restrict__ rec)
{
    if (threadIdx.x < KCB_HDR) rec[threadIdx.x] = (threadIdx.x == 8 || threadIdx.x == 9) ? -1.0 : 0.0;
}

template <int NP>
__global__ __launch_bounds__(1024) void kcb_select_sharded_kernel(const double* __restrict__ recs, int world, long long mm, KcbState* S, int K,
                                                                   double* __restrict__ cen_out, msm_idx_t* __restrict__ ids_out)
{
    __shared__ double rv[1024];
    __shared__ long long ri[1024];
    __shared__ double cs[16];
    const int tid = threadIdx.x, m = (int)mm;
    const size_t RD = kcb_rec_doubles(mm);
    const int k0 = S->k_done;
    if (k0 >= K) {
        if (tid == 0) S->J = 0;
        return;
    }
    auto reduce = [&](double v, long long i, double& ov, long long& oi) {
        double wv;
        long long wi;
        kcb_wave_argmax(v, i, wv, wi);
        __syncthreads();   // the previous call's readers are done with rv / ri
        if ((tid & 63) == 0) {
            rv[tid >> 6] = wv;
            ri[tid >> 6] = wi;
        }
        __syncthreads();
        const int l = tid & 63;
        kcb_wave_argmax(l < 16 ? rv[l] : -1.0, l < 16 ? ri[l] : -1, ov, oi);
    };
    // the row the one-centre protocol would take: best of the ranks' own argmax records (value, lowest GLOBAL row on ties)
    double vP;
    long long iP;
    int rP = -1;
    {
        double v = -1.0;
        long long i = -1;
        if (tid < world) {
            const double* h = recs + (size_t)tid * RD;
            if (h[9] >= 0.0) {
                v = h[8];
                i = (long long)h[9];
            }
        }
        reduce(v, i, vP, iP);
        for (int r = 0; r < world; ++r)
            if (iP >= 0 && (long long)recs[(size_t)r * RD + 9] == iP) rP = r;
    }
    // union of the ranks' lists, level counts summed
    unsigned total = 0, truecount = 0;
    bool fits = true;
    unsigned lev[KCB_NLEV];
#pragma unroll
    for (int q = 0; q < KCB_NLEV; ++q) lev[q] = 0;
    for (int r = 0; r < world; ++r) {
        const double* h = recs + (size_t)r * RD;
        const unsigned c = (unsigned)h[0];
        truecount += c;
        if (c > (unsigned)KCB_CAPR) fits = false;
        else total += c;
#pragma unroll
        for (int q = 1; q < KCB_NLEV; ++q) lev[q] += (unsigned)h[KCB_LEV0 + q];
    }
    const float theta = S->theta;
    const bool usable = fits && total > 0 && total <= (unsigned)KCB_CAP && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    long long ci[2] = {-1, -1};
    double cv[2] = {-1.0, -1.0}, cx[2][2 * NP];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        unsigned c = (unsigned)tid + 1024u * u;
        if (usable && c < total) {
            int r = 0;
            for (; r < world; ++r) {
                const unsigned cr = (unsigned)recs[(size_t)r * RD];
                if (c < cr) break;
                c -= cr;
            }
            const double* e = recs + (size_t)r * RD + KCB_HDR + (size_t)c * (2 + m);
            cv[u] = e[0];
            ci[u] = (long long)e[1];
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) cx[u][f] = f < m ? e[2 + f] : 0.0;
        }
    }
    int J = 0, fell = 0;
    double vlast = vP;
    for (;;) {
        double v = -1.0, vb;
        long long i = -1, ib;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (ci[u] >= 0 && (i < 0 || kc_better(cv[u], ci[u], v, i))) {
                v = cv[u];
                i = ci[u];
            }
        reduce(v, i, vb, ib);
        long long centre;
        if (J == 0) {
            if (usable && ib == iP) {
                centre = ib;
            } else {
                centre = iP;
                fell = 1;
            }
            vlast = vP;
        } else {
            if (!(ib >= 0 && vb > tau)) break;
            centre = ib;
            vlast = vb;
        }
        if (fell) {
            if (tid < 16) cs[tid] = (rP >= 0 && tid < m) ? recs[(size_t)rP * RD + 10 + tid] : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ci[u] == centre) {
#pragma unroll
                    for (int f = 0; f < 16; ++f) cs[f] = f < 2 * NP ? cx[u][f] : 0.0;
                }
        }
        if (tid == 0) ids_out[k0 + J] = centre;
        __syncthreads();
        if (tid < 16) S->cen[J][tid] = cs[tid];
        if (tid < m) cen_out[(size_t)(k0 + J) * m + tid] = cs[tid];
        ++J;
        if (fell || k0 + J >= K || J >= KCB_JMAX) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ci[u] < 0) continue;
            if (ci[u] == centre) {
                ci[u] = -1;
                continue;
            }
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, cx[u][f], cs[f]);
            const double d = m_final<M_EUCLIDEAN>(a, b, mm);
            if (d < cv[u]) cv[u] = d;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (truecount > (unsigned)KCB_CAPR) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (lev[q] <= (unsigned)KCB_CAPR) l = q;
            th = theta * kcb_level(l);
        }
        if (th > vl) th = vl;
        S->theta = th;
        S->count = 0;
        for (int q = 0; q < KCB_NLEV; ++q) S->lev[q] = 0;
        S->J = J;
        S->k_done = k0 + J;
        S->rounds += 1;
        S->fallbacks += fell;
    }
}

// c0 = coordinates of the first centre (ids[0]), for the copy's origin
// (sharded fit: `centre0` = the first centre's coordinates as selected from the exchanged records -- it may be another rank's row)
__global__ void ksc_origin_kernel(const double* __restrict__ X, const msm_idx_t* __restrict__ ids, long long m, double* __restrict__ c0,
                                  const double* __restrict__ centre0)
{
    if (threadIdx.x < 16) c0[threadIdx.x] = threadIdx.x < m ? (centre0 ? centre0[threadIdx.x] : X[ids[0] * m + threadIdx.x]) : 0.0;
}

// what the last k-centers fit streamed (for bench.py's bytes-per-pass figure): pass counts and the bytes a pass reads per row
struct KcStats {
    long long rows = 0, plain_passes = 0, screened_passes = 0, plain_row_bytes = 0, screen_row_bytes = 0, batch_fallbacks = 0;
};
static KcStats g_kc_stats;

// The screen copy in use is FMT 2 (bytes + a row scale); the float32 / bfloat16 formats of the kernels' FMT parameter were
// measured in round 3 and are no longer instantiated.
constexpr int KSC_FMT = 2;

struct KscBufs {
    DevBuf xf, curf, misc;
};
static KscBufs& ksc_bufs()
{
    static KscBufs b;
    return b;
}

// out[k][:] = X[ids[k]][:] -- the chosen rows themselves (cluster_centers_), gathered on the device so that they travel
// with the ids in the fit's one final synchronisation (torch indexing with a Python list is three round trips)
template <typename T>
__global__ void kc_gather_centres_kernel(const T* __restrict__ X, long long m, const msm_idx_t* __restrict__ ids, T* __restrict__ out)
{
    const msm_idx_t row = ids[blockIdx.x];
    for (long long f = threadIdx.x; f < m; f += blockDim.x) out[(size_t)blockIdx.x * m + f] = X[(size_t)row * m + f];
}

template <typename T>
int kcenters_impl(const T* X, msm_idx_t n, msm_idx_t m, msm_idx_t K, const char* metric,
                  msm_idx_t seed, msm_idx_t* ids, msm_idx_t* labels, double* distances,
                  double* inertia, int on_device, T* centers_out = nullptr)
{
    const msm_idx_t m_rows = m;   // the caller's row length (the fit may run on a zero-padded copy)
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !ids || !labels || !distances) return fail(MSM_ERR_INVALID, "kcenters_fit: null pointer");
    if (n < 1 || m < 1 || K < 1) return fail(MSM_ERR_INVALID, "kcenters_fit: bad shape");
    if (seed < 0 || seed >= n) return fail(MSM_ERR_INVALID, "kcenters_fit: seed_index out of range");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dLab = pool(PS_LAB), &dDist = pool(PS_MIN), &dPart = pool(PS_PART), &dIds = pool(PS_IDS),
           &dSum = pool(PS_SUM);
    int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)2 * nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dIds.reserve((size_t)K * sizeof(msm_idx_t)))) return rc;
    if ((rc = dSum.reserve((size_t)nblk * sizeof(double)))) return rc;
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.m = m;
    P.seed = seed;
    P.nblk = nblk;
    P.ids = dIds.as<msm_idx_t>();
    P.prune = 1;
    if (on_device) {
        P.X = X;
        P.labels = labels;
        P.dist = distances;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = dLab.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
        if ((rc = dDist.reserve((size_t)n * sizeof(double)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        P.X = dX.p;
        P.labels = dLab.as<msm_idx_t>();
        P.dist = dDist.as<double>();
    }
    if (pad_rows_pays<T>(mid, m, n, false)) {   // odd rows, norm metric: a zero-padded aligned copy (see pad_rows_pays)
        const T* xp = nullptr;
        long long mp = m;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.X), n, m, pool(PS_PADX), &xp, &mp))) return rc;
        P.X = xp;
        P.m = m = mp;
    }
    P.vecw = row_vecw<T>(P.X, m, false);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.X, m, false)) P.nblk = nblk = std::min(nblk, wide_grid(n));
    KcPartial* part = dPart.as<KcPartial>();
    const bool screen = sizeof(T) == 8 && mid == M_EUCLIDEAN && P.vecw > 0 && n >= 65536 && K > 8;
    g_kc_stats = KcStats();
    g_kc_stats.rows = n;
    g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);   // the row, distances_, labels_ (pruning test)
    g_kc_stats.plain_passes = K;
    if (screen) {
        // A few plain passes first (in the first passes most rows change, and a candidate costs the screen's bytes on top of
        // the plain pass's), then the screened ones.  Measured on 10M x 10 float64, K = 200 (scripts/kcperf.py, kcblobs.py),
        // plain / screened from pass 16 / from pass 4 / from pass 1: tICA projection 26.6 / 13.1 / 11.8 / 11.8 ms, white
        // noise 33.7 / 15.1 / 13.3 / 17.6 ms, 40 separated blobs (where per-row pruning is at its best) 17.9 / 17.1 / 17.1 /
        // 17.5 ms -- so no decision is needed.
        // (with several centres per pass the early centres are cheap on the copy too -- one pass applies a batch of them to
        //  every row, whatever fraction changes: 2 plain passes, tICA projection 4.53 -> 4.26 ms; 4 for one centre per pass)
        const bool kcb_on = !(getenv("MSM_KC_BATCH") && atoi(getenv("MSM_KC_BATCH")) == 0) && nblk <= 1024;  // read per fit
        const int KSC_PROBE = kcb_on ? 2 : 4;
        KscBufs& B = ksc_bufs();
        const int np = (int)((m + 1) / 2);
        if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;  // [gmax2[2] | ... | c0[16]]
        MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
        msm_idx_t it = 0;
        for (; it < K && it < KSC_PROBE; ++it) {
            P.it = (int)it;
            P.prev = part + (size_t)((it + 1) & 1) * nblk;
            P.next = part + (size_t)(it & 1) * nblk;
            launch_kc<T>(mid, nblk, P);
        }
        const bool use_screen = it < K;
        if (use_screen) {
            constexpr int fmt = KSC_FMT;
            g_kc_stats.plain_passes = it;
            g_kc_stats.screened_passes = K - it;
            g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, fmt) * 4 + 4);   // the screen copy's row + curf
            if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, fmt) + 1) * sizeof(float)))) return rc;
            KscArgs S;
            memset(&S, 0, sizeof(S));
            S.X = reinterpret_cast<const double*>(P.X);
            S.xs = B.xf.p;
            S.gmax2 = B.misc.as<unsigned long long>();
            S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
            hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, P.ids, (long long)m, S.c0, (const double*)nullptr);
            S.n = n;
            S.m = m;
            S.nblk = nblk;
            S.vecw = P.vecw;
            S.seed = seed;
            S.dist = P.dist;
            S.labels = P.labels;
            S.ids = P.ids;
            const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
            switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, fmt>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
            }
            if (kcb_on) {
                // several centres per pass (kcb_select_kernel / kcenters_batch_pass_kernel): rounds of {selector, pass} are
                // queued four at a time -- a round whose selector finds all K centres fixed is two empty launches -- and the
                // host looks at the progress counter between the groups
                DevBuf& SB = pool(PS_W);
                if ((rc = SB.reserve(sizeof(KcbState)))) return rc;
                KcbState* St = SB.as<KcbState>();
                hipLaunchKernelGGL(kcb_init_kernel, dim3(1), dim3(64), 0, stream(), St, (int)it);
                int rounds = 0, done = (int)it;
                int head4[4] = {(int)it, 0, 0, 0};   // k_done, J, rounds, fallbacks: the head of KcbState
                while (done < (int)K) {
                    const int group = kcb_group((int)K, done, rounds, (int)it);
                    for (int r = 0; r < group; ++r, ++rounds) {
                        S.prev = part + (size_t)((it + 1 + rounds) & 1) * nblk;   // partials of the last pass that ran
                        S.next = part + (size_t)((it + rounds) & 1) * nblk;
                        switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcb_select_kernel<NP_>), dim3(1), dim3(1024), 0, stream(), S, St, (int)K); \
                               hipLaunchKernelGGL((kcenters_batch_pass_kernel<NP_>), dim3(nblk), dim3(DT), 0, stream(), S, St); break;
                            MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                        }
                    }
                    MSM_HIP_CHECK(hipGetLastError());
                    MSM_HIP_CHECK(hipMemcpyAsync(head4, St, sizeof(head4), hipMemcpyDeviceToHost, stream()));
                    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                    done = head4[0];
                    if (rounds > 4 * (int)K) return fail(MSM_ERR_HIP, "k-centers: the batched passes made no progress");
                }
                g_kc_stats.screened_passes = head4[2];   // passes that streamed the copy (the empty rounds of the last group are not counted)
                g_kc_stats.batch_fallbacks = head4[3];
                it = K;
            }
            for (; it < K; ++it) {
                S.it = (int)it;
                S.prev = part + (size_t)((it + 1) & 1) * nblk;
                S.next = part + (size_t)(it & 1) * nblk;
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcenters_screen_pass_kernel<NP_, fmt>), dim3(nblk), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
            }
        }
    } else
    for (msm_idx_t it = 0; it < K; ++it) {
        P.it = (int)it;
        P.prev = part + (size_t)((it + 1) & 1) * nblk;
        P.next = part + (size_t)(it & 1) * nblk;
        launch_kc<T>(mid, nblk, P);
    }
    MSM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nblk), dim3(DT), 0, stream(), P.dist, (long long)n, dSum.as<double>());
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemcpyAsync(ids, P.ids, (size_t)K * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
    if (centers_out) {
        DevBuf& dCen = pool(PS_Y);
        if ((rc = dCen.reserve((size_t)K * m_rows * sizeof(T)))) return rc;
        const T* Xsrc = on_device ? X : static_cast<const T*>(dX.p);
        hipLaunchKernelGGL((kc_gather_centres_kernel<T>), dim3((unsigned)K), dim3(64), 0, stream(), Xsrc, (long long)m_rows, P.ids, dCen.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
        MSM_HIP_CHECK(hipMemcpyAsync(centers_out, dCen.p, (size_t)K * m_rows * sizeof(T), hipMemcpyDeviceToHost, stream()));
    }
    if (!on_device) {
        MSM_HIP_CHECK(hipMemcpyAsync(labels, P.labels, (size_t)n * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(distances, P.dist, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    double s = 0.0;
    if ((rc = sum_partials_host(dSum.as<double>(), nblk, &s))) return rc;  // synchronises
    if (inertia) *inertia = s;
    return MSM_OK;
}

template <typename T>
int pdist_impl(const T* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
               msm_idx_t n_idx, double* out, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !out) return fail(MSM_ERR_INVALID, "pdist: null pointer");
    if (n < 0 || m < 1) return fail(MSM_ERR_INVALID, "pdist: bad shape");
    const long long nn = X_indices ? n_idx : n;
    if (nn < 2) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dIdx = pool(PS_IDX), &dOut = pool(PS_OUT);
    const size_t npairs = (size_t)nn * (size_t)(nn - 1) / 2;
    PdArgs P;
    memset(&P, 0, sizeof(P));
    P.n = nn;
    P.m = m;
    if (on_device) {
        P.X = X;
        P.X_indices = X_indices;
        P.out = out;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)nn * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)nn * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dOut.reserve(npairs * sizeof(double)))) return rc;
        P.out = dOut.as<double>();
    }
    launch_pd<T, 0>(mid, (int)std::min<long long>(nn - 1, 4096), P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device && (rc = d2h_bulk(out, P.out, npairs * sizeof(double)))) return rc;
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

template <typename T>
int sumdist_impl(const T* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                 msm_idx_t p, double* sum, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !sum || (p > 0 && !pairs)) return fail(MSM_ERR_INVALID, "sumdist: null pointer");
    if (n < 0 || m < 1 || p < 0) return fail(MSM_ERR_INVALID, "sumdist: bad shape");
    *sum = 0.0;
    if (p == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dIdx = pool(PS_IDX), &dPart = pool(PS_PART);
    const int grid = (int)std::min<long long>(ceil_div(p, DT), 1024);
    if ((rc = dPart.reserve((size_t)grid * sizeof(double)))) return rc;
    PdArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.m = m;
    P.p = p;
    P.partial = dPart.as<double>();
    if (on_device) {
        P.X = X;
        P.pairs = pairs;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = dIdx.reserve((size_t)p * 2 * sizeof(msm_idx_t)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, pairs, (size_t)p * 2 * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
        P.X = dX.p;
        P.pairs = dIdx.as<msm_idx_t>();
    }
    launch_pd<T, 1>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    return sum_partials_host(P.partial, grid, sum);
}

// One externally driven pass (multi-rank k-centers): centre coordinates come from the host,
// the local (max distance, lowest local row) comes back.
template <typename T>
int kcenters_pass_impl(const T* X, msm_idx_t n, msm_idx_t m, const T* y, msm_idx_t it, const char* metric,
                       msm_idx_t* labels, double* distances, double* max_dist, msm_idx_t* argmax,
                       T* argmax_row, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !y || !labels || !distances || !max_dist || !argmax) return fail(MSM_ERR_INVALID, "kcenters_pass: null pointer");
    if (n < 1 || m < 1 || it < 0) return fail(MSM_ERR_INVALID, "kcenters_pass: bad shape");
    if (!on_device) return fail(MSM_ERR_INVALID, "kcenters_pass: per-row arrays must be device resident");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dPart = pool(PS_PART), &dY = pool(PS_Y), &dIds = pool(PS_IDS);
    int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dY.reserve((size_t)m * sizeof(T)))) return rc;
    if ((rc = dIds.reserve(sizeof(msm_idx_t)))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(dY.p, y, (size_t)m * sizeof(T), hipMemcpyHostToDevice, stream()));
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.X = X;
    P.n = n;
    P.m = m;
    P.it = (int)it;
    P.nblk = nblk;
    P.next = dPart.as<KcPartial>();
    P.prev = nullptr;
    P.dist = distances;
    P.labels = labels;
    P.ids = dIds.as<msm_idx_t>();
    P.ycenter = dY.p;
    P.vecw = row_vecw<T>(P.X, m, false);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.ycenter, m, false)) P.nblk = nblk = std::min(nblk, wide_grid(n));
    launch_kc<T>(mid, nblk, P);
    MSM_HIP_CHECK(hipGetLastError());
    // device-side final reduce + fetch of the winning row: ONE small D2H per pass
    DevBuf& dBest = pool(PS_SUM);
    if ((rc = dBest.reserve(sizeof(KcPartial) + (size_t)m * sizeof(T)))) return rc;
    KcPartial* dbest = dBest.as<KcPartial>();
    T* drow = reinterpret_cast<T*>(dbest + 1);
    hipLaunchKernelGGL((kc_finalize_kernel<T>), dim3(1), dim3(DT), 0, stream(), P.next, nblk, X, (long long)m, dbest, drow);
    MSM_HIP_CHECK(hipGetLastError());
    std::vector<char> hb(sizeof(KcPartial) + (size_t)m * sizeof(T));
    MSM_HIP_CHECK(hipMemcpyAsync(hb.data(), dbest, hb.size(), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    const KcPartial* hp = reinterpret_cast<const KcPartial*>(hb.data());
    *max_dist = hp->v;
    *argmax = hp->i;
    if (argmax_row && hp->i >= 0) memcpy(argmax_row, hb.data() + sizeof(KcPartial), (size_t)m * sizeof(T));
    return MSM_OK;
}

// one pass + candidate record, everything on the stream, no synchronisation
template <typename T>
int kcenters_pass_dev_impl(const T* X, msm_idx_t n, msm_idx_t m, const T* y_dev, msm_idx_t it, const char* metric,
                           msm_idx_t* labels, double* distances, msm_idx_t row_offset, double* cand_dev,
                           const T* centers_dev = nullptr)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!y_dev || !cand_dev || (n > 0 && (!X || !labels || !distances))) return fail(MSM_ERR_INVALID, "kcenters_pass_dev: null pointer");
    if (n < 0 || m < 1 || it < 0) return fail(MSM_ERR_INVALID, "kcenters_pass_dev: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dPart = pool(PS_PART), &dIds = pool(PS_IDS);
    int nblk = (int)std::min<long long>(ceil_div(std::max<long long>(n, 1), DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dIds.reserve(sizeof(msm_idx_t)))) return rc;
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.X = X;
    P.n = n;
    P.m = m;
    P.it = (int)it;
    P.next = dPart.as<KcPartial>();
    P.dist = distances;
    P.labels = labels;
    P.ids = dIds.as<msm_idx_t>();
    P.ycenter = y_dev;
    P.centers = centers_dev;                        // centres 0 .. it of the fit (pruning table); null: no pruning
    P.prune = centers_dev ? 1 : 0;
    if (n > 0) {
        P.vecw = row_vecw<T>(P.X, m, false);
        if (P.vecw == 0 && wide_ok<T>(P.X, P.ycenter, m, false)) nblk = std::min(nblk, wide_grid(n));
        P.nblk = nblk;
        launch_kc<T>(mid, nblk, P);
        MSM_HIP_CHECK(hipGetLastError());
    } else {
        nblk = 0;  // empty shard: the candidate kernel reports "none"
    }
    hipLaunchKernelGGL((kc_candidate_kernel<T>), dim3(1), dim3(DT), 0, stream(), P.next, nblk, X, (long long)m,
                       (long long)row_offset, cand_dev);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

template <typename T>
int kcenters_select_impl(const double* cands_dev, msm_idx_t world, msm_idx_t m, T* y_dev, T* centers_dev,
                         msm_idx_t* ids_dev, msm_idx_t slot)
{
    if (!cands_dev || !y_dev || !centers_dev || !ids_dev || world < 1 || m < 1 || slot < 0)
        return fail(MSM_ERR_INVALID, "kcenters_select: bad argument");
    hipLaunchKernelGGL((kc_select_kernel<T>), dim3(1), dim3(DT), 0, stream(), cands_dev, (int)world, (long long)m, y_dev,
                       centers_dev, ids_dev, (long long)slot);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// centre 0 of a sharded fit: the rank that owns the seed row publishes it through the same record exchange
template <typename T>
__global__ __launch_bounds__(DT) void kc_seed_candidate_kernel(const T* __restrict__ X, long long m, long long local_row,
                                                               long long global_row, double* __restrict__ cand)
{
    if (threadIdx.x == 0) {
        cand[0] = local_row >= 0 ? 1.0 : -1.0;
        cand[1] = local_row >= 0 ? (double)global_row : -1.0;
    }
    for (long long f = threadIdx.x; f < m; f += DT) cand[2 + f] = local_row >= 0 ? (double)X[local_row * m + f] : 0.0;
}

// The whole row-sharded fit: K x (pass, candidate record, all-gather over the library communicator, select), all
// queued on the stream -- the host enqueues and synchronises once at the end.
template <typename T>
int kcenters_fit_sharded_impl(const T* X, msm_idx_t n, msm_idx_t m, msm_idx_t K, const char* metric, msm_idx_t seed,
                              msm_idx_t row_offset, msm_idx_t* labels, double* distances, msm_idx_t* ids, T* centers,
                              double* inertia)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!ids || !centers || (n > 0 && (!X || !labels || !distances))) return fail(MSM_ERR_INVALID, "kcenters_fit_sharded: null pointer");
    if (n < 0 || m < 1 || K < 1 || seed < 0 || row_offset < 0) return fail(MSM_ERR_INVALID, "kcenters_fit_sharded: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int world = comm_world();
    const size_t rec = (size_t)(2 + m);
    // [cand rec | cands world*rec | sums 1024 + 1] doubles, [y m | centers K*m] T, [ids K] int64
    DevBuf& dS = pool(PS_S);
    const size_t nd = rec + (size_t)world * rec + 1032;
    const size_t bytes = nd * sizeof(double) + ((size_t)(K + 1) * m * sizeof(T) + 15) / 16 * 16 + (size_t)K * sizeof(msm_idx_t);
    int rc = dS.reserve(bytes);
    if (rc) return rc;
    double* cand = dS.as<double>();
    // without a communicator (a world of one: bench.py's strong-scaling model, single-process use) the "gathered" records
    // ARE the shard's record: no copy per centre
    double* cands = comm_active() ? cand + rec : cand;
    double* sums = cands + (size_t)world * rec;
    T* y = reinterpret_cast<T*>(sums + 1032);
    T* cen = y + m;
    msm_idx_t* dids = reinterpret_cast<msm_idx_t*>(reinterpret_cast<char*>(y) + ((size_t)(K + 1) * m * sizeof(T) + 15) / 16 * 16);
    const long long local_seed = (seed >= row_offset && seed < row_offset + n) ? (long long)(seed - row_offset) : -1;
    hipLaunchKernelGGL((kc_seed_candidate_kernel<T>), dim3(1), dim3(DT), 0, stream(), X, (long long)m, local_seed, (long long)seed, cand);
    MSM_HIP_CHECK(hipGetLastError());
    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
    const bool fused = n > 0 && row_vecw<T>(X, m, false) > 0;  // register path (m <= FC): one kernel per centre
    // Several centres per exchange (kcb_*: threshold lists, an identical selection on every rank): float64 rows of <= 16
    // features, euclidean.  The decision uses nothing a rank knows alone -- not its shard size, which may be zero --, because
    // it changes the exchange pattern: PROBE all-gathers of one candidate, then one all-gather of a round record per round.
    bool batched = false;
    if constexpr (sizeof(T) == 8) {
        constexpr int probe = 2;
        const char* be = getenv("MSM_KC_BATCH");   // 0: one centre per exchange (A/B switch of the tests; read per fit)
        batched = mid == M_EUCLIDEAN && m <= FeatChunk<T>::FC && K > 8 && K > probe &&
                  !(be && atoi(be) == 0);
        if (batched) {
            const msm_idx_t PROBE = probe;
            const int np = (int)((m + 1) / 2);
            const int nblk = (int)std::min<long long>(std::max<long long>(ceil_div(std::max<long long>(n, 1), DT), 1), KC_MAXBLK);
            DevBuf& dPart = pool(PS_PART);
            if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial) + 16))) return rc;
            unsigned* counter = reinterpret_cast<unsigned*>(dPart.as<KcPartial>() + nblk);
            MSM_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), stream()));
            g_kc_stats = KcStats();
            g_kc_stats.rows = n;
            g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);
            g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, 2) * 4 + 4);
            g_kc_stats.plain_passes = PROBE;
            // ---- the first PROBE centres: one candidate per exchange, plain passes (a rank without rows keeps the pattern) ----
            if (n > 0) {
                KcArgs P;
                memset(&P, 0, sizeof(P));
                P.X = X;
                P.n = n;
                P.m = m;
                P.nblk = nblk;
                P.next = dPart.as<KcPartial>();
                P.dist = distances;
                P.labels = labels;
                P.vecw = row_vecw<T>(X, m, false);
                P.centers = cen;
                P.prune = 1;
                P.sel_cands = cands;
                P.sel_world = world;
                P.sel_centers = cen;
                P.sel_ids = dids;
                P.cand_out = cand;
                P.row_offset = row_offset;
                P.counter = counter;
                for (msm_idx_t it = 0; it < PROBE; ++it) {
                    P.it = (int)it;
                    launch_kc<T>(mid, nblk, P);
                    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                }
            } else {
                if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, 0))) return rc;
                for (msm_idx_t it = 0; it < PROBE; ++it) {
                    if ((rc = kcenters_pass_dev_impl<T>(X, n, m, y, it, metric, labels, distances, row_offset, cand, cen))) return rc;
                    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                    if (it + 1 < PROBE && (rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, it + 1))) return rc;
                }
            }
            MSM_HIP_CHECK(hipGetLastError());
            // ---- rounds ----
            const size_t RD = kcb_rec_doubles(m);
            DevBuf& W = pool(PS_W);
            const size_t stb = (sizeof(KcbState) + 15) / 16 * 16;
            if ((rc = W.reserve(stb + (size_t)(1 + world) * RD * sizeof(double)))) return rc;
            KcbState* St = W.as<KcbState>();
            double* recL = reinterpret_cast<double*>(static_cast<char*>(W.p) + stb);
            double* recsG = comm_active() ? recL + RD : recL;   // a world of one: the gathered records ARE the rank's record
            hipLaunchKernelGGL(kcb_init_kernel, dim3(1), dim3(64), 0, stream(), St, (int)PROBE);
            hipLaunchKernelGGL(kcb_boot_records_kernel, dim3((unsigned)world), dim3(64), 0, stream(), cands, world, (long long)m, recsG);
            KscArgs S;
            memset(&S, 0, sizeof(S));
            if (n > 0) {
                KscBufs& B = ksc_bufs();
                if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;
                if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, 2) + 1) * sizeof(float)))) return rc;
                MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
                S.X = reinterpret_cast<const double*>(X);
                S.xs = B.xf.p;
                S.gmax2 = B.misc.as<unsigned long long>();
                S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
                S.n = n;
                S.m = m;
                S.nblk = nblk;
                S.vecw = row_vecw<T>(X, m, false);
                S.seed = seed;
                S.dist = distances;
                S.labels = labels;
                S.ids = dids;
                S.next = dPart.as<KcPartial>();
                S.row_offset = row_offset;
                hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, dids, (long long)m, S.c0,
                                   reinterpret_cast<const double*>(cen));
                const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, 2>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
            }
            MSM_HIP_CHECK(hipGetLastError());
            int rounds = 0, done = (int)PROBE;
            int head4[4] = {(int)PROBE, 0, 0, 0};
            while (done < (int)K) {
                const int group = kcb_group((int)K, done, rounds, (int)PROBE);
                for (int r = 0; r < group; ++r, ++rounds) {
                    switch (np) {
#define MSM_KSC(NP_) case NP_: \
                        hipLaunchKernelGGL((kcb_select_sharded_kernel<NP_>), dim3(1), dim3(1024), 0, stream(), recsG, world, (long long)m, St, (int)K, \
                                           reinterpret_cast<double*>(cen), dids); \
                        if (n > 0) { \
                            hipLaunchKernelGGL((kcenters_batch_pass_kernel<NP_>), dim3(nblk), dim3(DT), 0, stream(), S, St); \
                            hipLaunchKernelGGL((kcb_pack_kernel<NP_>), dim3(1), dim3(DT), 0, stream(), S, St, recL); \
                        } \
                        break;
                        MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                    }
                    if (n <= 0) hipLaunchKernelGGL(kcb_empty_record_kernel, dim3(1), dim3(64), 0, stream(), recL);
                    MSM_HIP_CHECK(hipGetLastError());
                    if ((rc = comm_allgather(recL, recsG, RD * sizeof(double)))) return rc;
                }
                MSM_HIP_CHECK(hipMemcpyAsync(head4, St, sizeof(head4), hipMemcpyDeviceToHost, stream()));
                MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                done = head4[0];
                if (rounds > 4 * (int)K) return fail(MSM_ERR_HIP, "k-centers: the batched rounds made no progress");
            }
            g_kc_stats.screened_passes = head4[2];
            g_kc_stats.batch_fallbacks = head4[3];
        }
    }
    if (batched) {
    } else if (fused) {
        // every rank must take the same path or the all-gathers would not match: rows are a property of the data type and
        // width only, except for an EMPTY shard -- which therefore runs the generic kernels but keeps the exchange pattern
        DevBuf &dPart = pool(PS_PART);
        const int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
        if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial) + 16))) return rc;
        unsigned* counter = reinterpret_cast<unsigned*>(dPart.as<KcPartial>() + nblk);
        MSM_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), stream()));
        KcArgs P;
        memset(&P, 0, sizeof(P));
        P.X = X;
        P.n = n;
        P.m = m;
        P.nblk = nblk;
        P.next = dPart.as<KcPartial>();
        P.dist = distances;
        P.labels = labels;
        P.vecw = row_vecw<T>(X, m, false);
        P.centers = cen;
        P.prune = 1;
        P.sel_cands = cands;
        P.sel_world = world;
        P.sel_centers = cen;
        P.sel_ids = dids;
        P.cand_out = cand;
        P.row_offset = row_offset;
        P.counter = counter;
        // Screened passes (float64 rows, euclidean; see kcenters_screen_pass_kernel) after a few plain ones, as in the
        // single-process fit.  The decision is LOCAL to a rank: the exchange pattern (one all-gather per centre) is the same
        // for both kernels, so a rank with a small shard may keep the plain kernel while its peers screen.
        constexpr int KSC_PROBE = 4;
        constexpr long long KSC_MIN_ROWS = 65536;
        const bool screen = sizeof(T) == 8 && mid == M_EUCLIDEAN && n >= KSC_MIN_ROWS && K > 8 && K > KSC_PROBE;
        KscArgs S;
        memset(&S, 0, sizeof(S));
        const int np = (int)((m + 1) / 2);
        constexpr int fmt = KSC_FMT;
        if (screen) {
            KscBufs& B = ksc_bufs();
            if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;
            if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, fmt) + 1) * sizeof(float)))) return rc;
            MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
            S.X = reinterpret_cast<const double*>(X);
            S.xs = B.xf.p;
            S.gmax2 = B.misc.as<unsigned long long>();
            S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
            S.n = n;
            S.m = m;
            S.nblk = nblk;
            S.vecw = P.vecw;
            S.seed = seed;
            S.dist = distances;
            S.labels = labels;
            S.ids = dids;
            S.next = dPart.as<KcPartial>();
            S.sel_cands = cands;
            S.sel_world = world;
            S.sel_centers = reinterpret_cast<double*>(cen);
            S.sel_ids = dids;
            S.cand_out = cand;
            S.row_offset = row_offset;
            S.counter = counter;
        }
        g_kc_stats = KcStats();
        g_kc_stats.rows = n;
        g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);
        g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, fmt) * 4 + 4);
        for (msm_idx_t it = 0; it < K; ++it) {
            if (screen && it >= KSC_PROBE) {
                if (it == KSC_PROBE) {
                    // the copy's origin is centre 0 as every rank selected it (possibly another rank's row)
                    hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, dids, (long long)m, S.c0,
                                       reinterpret_cast<const double*>(cen));
                    const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
                    switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, fmt>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                        MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                    }
                }
                S.it = (int)it;
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcenters_screen_pass_kernel<NP_, fmt>), dim3(nblk), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
                ++g_kc_stats.screened_passes;
            } else {
                P.it = (int)it;
                launch_kc<T>(mid, nblk, P);
                ++g_kc_stats.plain_passes;
            }
            if (it + 1 < K && (rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
        }
        MSM_HIP_CHECK(hipGetLastError());
    } else {
        if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, 0))) return rc;
        for (msm_idx_t it = 0; it < K; ++it) {
            if ((rc = kcenters_pass_dev_impl<T>(X, n, m, y, it, metric, labels, distances, row_offset, cand, cen))) return rc;
            if (it + 1 < K) {
                if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, it + 1))) return rc;
            }
        }
    }
    // inertia = sum of ALL ranks' distances_: local fp64 tree sum, then one all-reduce of a single double
    const int nblk = (int)std::min<long long>(std::max<long long>(ceil_div(std::max<long long>(n, 1), DT), 1), 1024);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nblk), dim3(DT), 0, stream(), distances, (long long)n, sums);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(1), dim3(DT), 0, stream(), sums, (long long)nblk, sums + 1024);
    MSM_HIP_CHECK(hipGetLastError());
    if ((rc = comm_allreduce_f64(sums + 1024, 1))) return rc;
    double tot = 0.0;
    MSM_HIP_CHECK(hipMemcpyAsync(&tot, sums + 1024, sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(ids, dids, (size_t)K * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(centers, cen, (size_t)K * m * sizeof(T), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (inertia) *inertia = tot;
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_kcenters_pass_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, float* argmax_row, int on_device)
{
    return kcenters_pass_impl<float>(X, n, m, y, it, metric, labels, distances, max_dist, argmax, argmax_row, on_device);
}

int msm_kcenters_pass_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, double* argmax_row, int on_device)
{
    return kcenters_pass_impl<double>(X, n, m, y, it, metric, labels, distances, max_dist, argmax, argmax_row, on_device);
}

int msm_kcenters_pass_dev_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev)
{
    return kcenters_pass_dev_impl<float>(X, n, m, y_dev, it, metric, labels, distances, row_offset, cand_dev);
}

int msm_kcenters_pass_dev_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev)
{
    return kcenters_pass_dev_impl<double>(X, n, m, y_dev, it, metric, labels, distances, row_offset, cand_dev);
}

int msm_kcenters_select_f32(const double* cands_dev, msm_idx_t world, msm_idx_t m, float* y_dev, float* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot)
{
    return kcenters_select_impl<float>(cands_dev, world, m, y_dev, centers_dev, ids_dev, slot);
}

int msm_kcenters_select_f64(const double* cands_dev, msm_idx_t world, msm_idx_t m, double* y_dev, double* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot)
{
    return kcenters_select_impl<double>(cands_dev, world, m, y_dev, centers_dev, ids_dev, slot);
}

int msm_pdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device)
{
    return pdist_impl<float>(X, metric, n, m, X_indices, n_X_indices, out, on_device);
}

int msm_pdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device)
{
    return pdist_impl<double>(X, metric, n, m, X_indices, n_X_indices, out, on_device);
}

int msm_sumdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device)
{
    return sumdist_impl<float>(X, metric, n, m, pairs, p, sum, on_device);
}

int msm_sumdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device)
{
    return sumdist_impl<double>(X, metric, n, m, pairs, p, sum, on_device);
}

int msm_dist_f32(const float* X, const float* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device)
{
    return cdist_impl<float>(X, y, metric, n, 1, m, X_indices, n_X_indices, out, on_device);
}

int msm_dist_f64(const double* X, const double* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device)
{
    return cdist_impl<double>(X, y, metric, n, 1, m, X_indices, n_X_indices, out, on_device);
}

int msm_cdist_f32(const float* XA, const float* XB, const char* metric, msm_idx_t na, msm_idx_t nb,
                  msm_idx_t m, double* out, int on_device)
{
    return cdist_impl<float>(XA, XB, metric, na, nb, m, nullptr, 0, out, on_device);
}

int msm_cdist_f64(const double* XA, const double* XB, const char* metric, msm_idx_t na,
                  msm_idx_t nb, msm_idx_t m, double* out, int on_device)
{
    return cdist_impl<double>(XA, XB, metric, na, nb, m, nullptr, 0, out, on_device);
}

int msm_assign_nearest_f32(const float* X, const float* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device)
{
    return assign_nearest_impl<float>(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                      assignments, min_dist, inertia, on_device);
}

int msm_assign_nearest_f64(const double* X, const double* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device)
{
    return assign_nearest_impl<double>(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                       assignments, min_dist, inertia, on_device);
}

int msm_kcenters_last_stats(msm_idx_t* out5)
{
    if (!out5) return fail(MSM_ERR_INVALID, "msm_kcenters_last_stats: null pointer");
    out5[0] = g_kc_stats.rows;
    out5[1] = g_kc_stats.plain_passes;
    out5[2] = g_kc_stats.screened_passes;
    out5[3] = g_kc_stats.plain_row_bytes;
    out5[4] = g_kc_stats.screen_row_bytes;
    return MSM_OK;
}

int msm_kcenters_fit_sharded_f32(const float* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, float* centers, double* inertia)
{
    return kcenters_fit_sharded_impl<float>(X, n_local, m, n_clusters, metric, seed_index, row_offset, labels, distances, ids,
                                            centers, inertia);
}

int msm_kcenters_fit_sharded_f64(const double* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, double* centers, double* inertia)
{
    return kcenters_fit_sharded_impl<double>(X, n_local, m, n_clusters, metric, seed_index, row_offset, labels, distances, ids,
                                             centers, inertia);
}

int msm_kcenters_fit_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device)
{
    return kcenters_impl<float>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device);
}

int msm_kcenters_fit_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device)
{
    return kcenters_impl<double>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device);
}

/* the same fit, also returning the chosen rows: centers[n_clusters][m] (HOST) = X[ids] (kcenters.py:98 cluster_centers_) */
int msm_kcenters_fit2_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric, msm_idx_t seed_index,
                          msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia, int on_device, float* centers)
{
    return kcenters_impl<float>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device, centers);
}

int msm_kcenters_fit2_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric, msm_idx_t seed_index,
                          msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia, int on_device, double* centers)
{
    return kcenters_impl<double>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device, centers);
}

}  // extern "C"
