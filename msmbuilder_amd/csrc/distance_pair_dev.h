// distance_pair_dev.h -- pair_kernel / pair_small_kernel / assign_small2_kernel: assign_nearest, cdist, dist in exact arithmetic
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
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

}  // namespace msm
