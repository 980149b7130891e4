// distance_kcsel_dev.h -- kc_finalize / kc_candidate / kc_select kernels: the argmax exchange of the k-centers loops
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

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

}  // namespace msm
