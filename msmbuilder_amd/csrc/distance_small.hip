// distance_small.hip -- exact assign_nearest for short rows (at most 128 bytes per row), euclidean family, one kernel
// per row length in 16-byte groups.  A separate translation unit only because its 32 instantiations dominate the build time.
#include "distance_dev.h"

namespace msm {

// assign_small2_kernel with the row length as a COMPILE-TIME number of 16-byte groups (NG = ceil(m / GS)).  In the
// generic kernel every group is a uniform branch -- ds_read, s_waitcnt lgkmcnt(0), 12 fp64 operations, branch -- so each 48
// cycles of arithmetic exposed a full LDS round trip (ISA of the round-2 kernel; 0.54 of the fp64-VALU bound measured by
// scripts/micro/valu_f64.hip: 12.3 cycles per float64 pair-element, 14.7 per float32 one).  Here a centre is straight-line
// code: its NG fragments are fetched while the previous centre computes, nothing branches inside a centre.
// Instantiated for the euclidean family only (the default metric of every clusterer; 2 x 8 x 2 kernels).
// one centre against the two rows of a lane: straight-line arithmetic, then the (lazy-sqrt) record update
template <typename T, int M, int NG>
__device__ __forceinline__ void small3_centre(const raw_f32x4 (&y)[NG], const T (&x0)[NG * (16 / (int)sizeof(T))],
                                              const T (&x1)[NG * (16 / (int)sizeof(T))], int j, long long m, double& best0,
                                              double& best1, double& thr0, double& thr1, int& lab0, int& lab1)
{
    constexpr int GS = 16 / (int)sizeof(T);
    __builtin_amdgcn_sched_barrier(0);
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if constexpr (sizeof(T) == 4 && (M == M_EUCLIDEAN || M == M_SQEUCLIDEAN)) {
        // float32 rows: the two differences of a feature pair in one packed subtract (v_pk_add_f32), then cvt + exact fma each
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x2* yv = reinterpret_cast<const f32x2*>(&y[g]);
#pragma unroll
            for (int e = 0; e < GS; e += 2) {
                const f32x2 u0 = {x0[g * GS + e], x0[g * GS + e + 1]}, u1 = {x1[g * GS + e], x1[g * GS + e + 1]};
                const f32x2 d0 = u0 - yv[e / 2], d1 = u1 - yv[e / 2];
                a0 = __builtin_fma((double)d0.x, (double)d0.x, a0);
                a1 = __builtin_fma((double)d1.x, (double)d1.x, a1);
                a0 = __builtin_fma((double)d0.y, (double)d0.y, a0);
                a1 = __builtin_fma((double)d1.y, (double)d1.y, a1);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const T* ye = reinterpret_cast<const T*>(&y[g]);
#pragma unroll
            for (int e = 0; e < GS; ++e) {
                m_update<T, M>(a0, b0, x0[g * GS + e], ye[e]);
                m_update<T, M>(a1, b1, x1[g * GS + e], ye[e]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (M == M_EUCLIDEAN) {
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
        const double d0 = m_final<M>(a0, b0, m), d1 = m_final<M>(a1, b1, m);
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

template <typename T, int M, int NG>
__global__ __launch_bounds__(DT) __attribute__((amdgpu_waves_per_eu(4, 4))) void assign_small3_kernel(PairArgs P)
{
    constexpr int GS = 16 / (int)sizeof(T);
    constexpr int MP = NG * GS;               // padded row length (zero padding is exact)
    constexpr int YCAP = 32768 / (int)sizeof(T);
    constexpr int KT = YCAP / MP;
    __shared__ __attribute__((aligned(16))) T Ys[YCAP];
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    const T* Y = static_cast<const T*>(P.Y);
    const int tid = threadIdx.x;
    const int m = (int)P.m;
    double inertia = 0.0;
    bool staged = false;
    const long long ntile = (P.n + 2 * DT - 1) / (2 * DT);
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long i0 = t * (2 * DT) + tid, i1 = i0 + DT;
        T x0[MP], x1[MP];
        {
            const T* p0 = X + (i0 < P.n ? i0 : P.n - 1) * P.m;
            const T* p1 = X + (i1 < P.n ? i1 : P.n - 1) * P.m;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (P.vecw == 16 && (g + 1) * GS <= m) {
                    const raw_f32x4 q0 = *reinterpret_cast<const raw_f32x4*>(p0 + g * GS), q1 = *reinterpret_cast<const raw_f32x4*>(p1 + g * GS);
#pragma unroll
                    for (int e = 0; e < GS; ++e) {
                        x0[g * GS + e] = reinterpret_cast<const T*>(&q0)[e];
                        x1[g * GS + e] = reinterpret_cast<const T*>(&q1)[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < GS; ++e) {
                        const int f = g * GS + e;
                        x0[f] = f < m ? p0[f] : (T)0;
                        x1[f] = f < m ? p1[f] : (T)0;
                    }
                }
            }
        }
        double best0 = INFINITY, best1 = INFINITY, thr0 = INFINITY, thr1 = INFINITY;
        if (M != M_EUCLIDEAN) best0 = best1 = 1.7976931348623157e308;
        int lab0 = -1, lab1 = -1;
        for (long long j0 = 0; j0 < P.K; j0 += KT) {
            const int kt = (int)((P.K - j0) < KT ? (P.K - j0) : KT);
            if (!staged) {  // all centres fit one tile (the usual case): staged once per workgroup, not once per row tile
                __syncthreads();
                for (int e = tid; e < kt * MP; e += DT) {
                    const int c = e / MP, ff = e - c * MP;
                    Ys[e] = ff < m ? Y[(j0 + c) * P.m + ff] : (T)0;
                }
                __syncthreads();
                staged = P.K <= KT;
            }
            raw_f32x4 ya[NG], yb[NG];  // ping-pong: one centre computes from ya while the next one lands in yb
#pragma unroll
            for (int g = 0; g < NG; ++g) ya[g] = *reinterpret_cast<const raw_f32x4*>(Ys + g * GS);
            int c = 0;
            for (; c + 1 < kt; c += 2) {
                const T* y1 = Ys + (c + 1) * MP;
#pragma unroll
                for (int g = 0; g < NG; ++g) yb[g] = *reinterpret_cast<const raw_f32x4*>(y1 + g * GS);
                small3_centre<T, M, NG>(ya, x0, x1, (int)(j0 + c), P.m, best0, best1, thr0, thr1, lab0, lab1);
                const T* y2 = Ys + (c + 2 < kt ? c + 2 : c + 1) * MP;
#pragma unroll
                for (int g = 0; g < NG; ++g) ya[g] = *reinterpret_cast<const raw_f32x4*>(y2 + g * GS);
                small3_centre<T, M, NG>(yb, x0, x1, (int)(j0 + c + 1), P.m, best0, best1, thr0, thr1, lab0, lab1);
            }
            if (c < kt) small3_centre<T, M, NG>(ya, x0, x1, (int)(j0 + c), P.m, best0, best1, thr0, thr1, lab0, lab1);
        }
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

template <typename T, int M>
static bool launch_small3(int grid, const PairArgs& P)
{
    constexpr int GS = 16 / (int)sizeof(T);
    const int ng = (int)((P.m + GS - 1) / GS);
    switch (ng) {
#define MSM_S3(NG_) case NG_: hipLaunchKernelGGL((assign_small3_kernel<T, M, NG_>), dim3(grid), dim3(DT), 0, stream(), P); return true;
        MSM_S3(1) MSM_S3(2) MSM_S3(3) MSM_S3(4) MSM_S3(5) MSM_S3(6) MSM_S3(7) MSM_S3(8)
#undef MSM_S3
    }
    return false;
}

bool launch_small3_f32(int metric, int grid, const PairArgs& P)
{
    return metric == M_SQEUCLIDEAN ? launch_small3<float, M_SQEUCLIDEAN>(grid, P) : launch_small3<float, M_EUCLIDEAN>(grid, P);
}

bool launch_small3_f64(int metric, int grid, const PairArgs& P)
{
    return metric == M_SQEUCLIDEAN ? launch_small3<double, M_SQEUCLIDEAN>(grid, P) : launch_small3<double, M_EUCLIDEAN>(grid, P);
}

}  // namespace msm
