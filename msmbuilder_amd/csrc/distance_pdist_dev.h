// distance_pdist_dev.h -- pdist_kernel, sumdist_kernel, sum_partial_kernel
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

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

}  // namespace msm
