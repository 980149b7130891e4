// tica_solve_dev.h -- finalise / RBLW / shrink / top-pairs kernels of the device-resident solve
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_common_dev.h"

namespace msm {

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
