// subspace.hip -- the k largest eigenpairs of the reduced tICA matrix by Chebyshev-filtered subspace iteration.
//
// tICA._solve needs the top k of the n x n standard problem C y = lambda y that the Cholesky reduction leaves
// (/root/reference/msmbuilder/decomposition/tica.py:188-194 asks LAPACK's dsygvx for exactly those).  The direct route --
// Householder tridiagonalisation (sytrd.hip) -- is n - 2 DEPENDENT exchange steps, 2.6 ms at n = 512 whatever k is.  But
// tICA is run BECAUSE the spectrum has a few slow processes above a bulk, and with such a gap a block of 32 vectors
// converges to the dominant invariant subspace in a few filtered iterations, each a short chain of small launches:
//
//   X (n x 32, orthonormal)  ->  Rayleigh-Ritz: H = X^T C X, H = S diag(theta) S^T on the host (32 x 32 Jacobi), X <- X S
//   ->  residuals ||C x_j - theta_j x_j||  ->  done when the k leading ones are at rounding level, otherwise
//   X <- p(C) X with p the degree-m Chebyshev polynomial that is small on [lower bound, theta_32] and grows fastest above
//   it (scaled three-term recurrence, Zhou & Saad's form), re-orthonormalised by two rounds of Cholesky QR.
//
// The spectrum of the reduced tICA matrix lies in [-1, 1] (|u^T C_sym u| <= u^T Sigma u by Cauchy-Schwarz; shrinkage only
// adds to Sigma), which gives the filter its lower bound.  Nothing here is trusted: the caller verifies the returned pairs
// against the reduced matrix (pair_residual_device) and falls back to the tridiagonalisation when the iteration stalls --
// a flat spectrum, or a matrix that is not a tICA matrix -- so the method can only cost time, never accuracy.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace msm {

namespace {

constexpr int SB = 32;        // block size
constexpr int SS_NT = 256;
constexpr int SS_RT = 8;      // rows of the product per workgroup
constexpr int SS_CK = 64;     // columns staged per chunk

// out = alpha * (C X) + beta * X + gamma * P   (C: n x n row-major; X, P, out: n x SB row-major; out may alias P)
__global__ __launch_bounds__(SS_NT) void ss_product_kernel(const double* __restrict__ Cm, int n, const double* __restrict__ X,
                                                           const double* P, double* out, double alpha, double beta, double gamma)
{
    __shared__ double sc[SS_RT][SS_CK + 1];
    __shared__ double sx[SS_CK][SB];
    const int tid = threadIdx.x, j = tid & (SB - 1), rl = tid >> 5;
    const int r0 = blockIdx.x * SS_RT, r = r0 + rl;
    double acc0 = 0.0, acc1 = 0.0;
    for (int c0 = 0; c0 < n; c0 += SS_CK) {
        __syncthreads();
        for (int q = tid; q < SS_RT * SS_CK; q += SS_NT) {
            const int rr = q / SS_CK, cc = q - rr * SS_CK;
            sc[rr][cc] = (r0 + rr < n && c0 + cc < n) ? Cm[(size_t)(r0 + rr) * n + c0 + cc] : 0.0;
        }
        for (int q = tid; q < SS_CK * SB; q += SS_NT) {
            const int cc = q >> 5, jj = q & (SB - 1);
            sx[cc][jj] = (c0 + cc < n) ? X[(size_t)(c0 + cc) * SB + jj] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < SS_CK; cc += 2) {
            acc0 += sc[rl][cc] * sx[cc][j];
            acc1 += sc[rl][cc + 1] * sx[cc + 1][j];
        }
    }
    if (r < n) {
        const size_t o = (size_t)r * SB + j;
        double v = alpha * (acc0 + acc1);
        if (beta != 0.0) v += beta * X[o];
        if (gamma != 0.0) v += gamma * P[o];
        out[o] = v;
    }
}

// part[blockIdx][i][j] = sum over this block's rows of A[r][i] B[r][j]   (A, B: n x SB)
constexpr int SG_ROWS = 64;
__global__ __launch_bounds__(SS_NT) void ss_gram_kernel(const double* __restrict__ A, const double* __restrict__ B, int n,
                                                        double* __restrict__ part)
{
    __shared__ double sa[SG_ROWS][SB + 1], sb[SG_ROWS][SB + 1];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * SG_ROWS;
    for (int q = tid; q < SG_ROWS * SB; q += SS_NT) {
        const int rr = q >> 5, jj = q & (SB - 1);
        const bool in = r0 + rr < n;
        sa[rr][jj] = in ? A[(size_t)(r0 + rr) * SB + jj] : 0.0;
        sb[rr][jj] = in ? B[(size_t)(r0 + rr) * SB + jj] : 0.0;
    }
    __syncthreads();
    // thread -> entries (i, j0 .. j0 + 3)
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    double acc[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int rr = 0; rr < SG_ROWS; ++rr) {
        const double a = sa[rr][i];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] += a * sb[rr][j0 + b];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) part[(size_t)blockIdx.x * SB * SB + i * SB + j0 + b] = acc[b];
}

// G = sum of the partial Gram matrices (deterministic order); out[0 .. SB*SB) = G (used for H = X^T W)
__global__ __launch_bounds__(SS_NT) void ss_gram_sum_kernel(const double* __restrict__ part, int nparts, double* __restrict__ G)
{
    for (int q = threadIdx.x; q < SB * SB; q += SS_NT) {
        double s = 0.0;
        for (int p = 0; p < nparts; ++p) s += part[(size_t)p * SB * SB + q];
        G[q] = s;
    }
}

// One round of Cholesky QR: G = X^T X (summed from the partials), G = R^T R, X <- X R^-1.  Every workgroup factors the
// 32 x 32 matrix redundantly (no exchange) and handles SS_NT rows, one row per thread in registers.
// *flag is set when G is not numerically positive definite (the block has lost rank): the caller gives up on the method.
__global__ __launch_bounds__(SS_NT) void ss_cholqr_kernel(const double* __restrict__ part, int nparts, double* X, int n,
                                                          int* __restrict__ flag)
{
    __shared__ double sg[SB][SB + 1];
    __shared__ double sinv[SB];
    const int tid = threadIdx.x;
    for (int q = tid; q < SB * SB; q += SS_NT) {
        double s = 0.0;
        for (int p = 0; p < nparts; ++p) s += part[(size_t)p * SB * SB + q];
        sg[q >> 5][q & (SB - 1)] = s;
    }
    __syncthreads();
    // upper Cholesky in place, right-looking: 1024 entries over 256 threads
    for (int p = 0; p < SB; ++p) {
        const double piv = sg[p][p];
        if (!(piv > 0.0) && tid == 0 && blockIdx.x == 0) *flag = 1;
        const double u = sqrt(piv), ui = 1.0 / u;
        __syncthreads();
        if (tid < SB && tid >= p) sg[p][tid] = tid == p ? u : sg[p][tid] * ui;
        if (tid == 0) sinv[p] = ui;
        __syncthreads();
        for (int q = tid; q < SB * SB; q += SS_NT) {
            const int rr = q >> 5, cc = q & (SB - 1);
            if (rr > p && cc >= rr) sg[rr][cc] -= sg[p][rr] * sg[p][cc];
        }
        __syncthreads();
    }
    const int r = blockIdx.x * SS_NT + tid;
    if (r < n) {
        double x[SB];
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) x[jj] = X[(size_t)r * SB + jj];
        // y R = x: y_j = (x_j - sum_{i < j} y_i R[i][j]) / R[j][j]
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) {
            double t = x[jj];
#pragma unroll
            for (int i2 = 0; i2 < jj; ++i2) t -= x[i2] * sg[i2][jj];
            x[jj] = t * sinv[jj];
        }
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) X[(size_t)r * SB + jj] = x[jj];
    }
}

// X <- X S, W <- W S   (S: SB x SB row-major, columns = Ritz vectors in the wanted order)
__global__ __launch_bounds__(SS_NT) void ss_rotate_kernel(double* X, double* W, int n, const double* __restrict__ S)
{
    __shared__ double ss[SB][SB + 1];
    const int tid = threadIdx.x;
    for (int q = tid; q < SB * SB; q += SS_NT) ss[q >> 5][q & (SB - 1)] = S[q];
    __syncthreads();
    const int r = blockIdx.x * (SS_NT / SB) + (tid >> 5), j = tid & (SB - 1);
    const bool in = r < n;
    double ax = 0.0, aw = 0.0;
    double xr[SB], wr[SB];
    // a row belongs to 32 consecutive lanes of ONE wavefront: every lane reads the whole row before any lane writes its entry
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        xr[i] = in ? X[(size_t)r * SB + i] : 0.0;
        wr[i] = in ? W[(size_t)r * SB + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        ax += xr[i] * ss[i][j];
        aw += wr[i] * ss[i][j];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (in) {
        X[(size_t)r * SB + j] = ax;
        W[(size_t)r * SB + j] = aw;
    }
}

// res[j] = || W[:, j] - theta[j] X[:, j] ||_2
__global__ __launch_bounds__(SS_NT) void ss_residual_kernel(const double* __restrict__ X, const double* __restrict__ W, int n,
                                                            const double* __restrict__ theta, double* __restrict__ res)
{
    __shared__ double red[SS_NT / SB][SB];
    const int tid = threadIdx.x, j = tid & (SB - 1), part = tid >> 5;
    const double th = theta[j];
    double s = 0.0;
    for (int r = part; r < n; r += SS_NT / SB) {
        const double dlt = W[(size_t)r * SB + j] - th * X[(size_t)r * SB + j];
        s += dlt * dlt;
    }
    red[part][j] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
        for (int p = 0; p < SS_NT / SB; ++p) t += red[p][j];
        res[j] = sqrt(t);
    }
}

__global__ void ss_init_kernel(double* X, int n)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * SB) return;
    unsigned h = (unsigned)q * 0x9E3779B1u + 0x7F4A7C15u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    X[q] = ((double)h + 0.5) * (2.0 / 4294967296.0) - 1.0;
}

// Yk[j][r] = X[r][j] for j < k
__global__ void ss_emit_kernel(const double* __restrict__ X, int n, int k, double* __restrict__ Yk)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * k) return;
    const int j = q / n, r = q - j * n;
    Yk[q] = X[(size_t)r * SB + j];
}

// cyclic Jacobi for a small symmetric matrix (host): A (m x m, row-major, destroyed) -> eigenvalues w, eigenvectors as
// columns of V, sorted by descending eigenvalue
void jacobi_eigh(std::vector<double>& A, int m, std::vector<double>& w, std::vector<double>& V)
{
    V.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < m; ++i) {
            dia += A[(size_t)i * m + i] * A[(size_t)i * m + i];
            for (int j = i + 1; j < m; ++j) off += A[(size_t)i * m + j] * A[(size_t)i * m + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * dia) break;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                const double apq = A[(size_t)p * m + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (int i = 0; i < m; ++i) {   // columns p, q
                    const double aip = A[(size_t)i * m + p], aiq = A[(size_t)i * m + q];
                    A[(size_t)i * m + p] = c * aip - s * aiq;
                    A[(size_t)i * m + q] = s * aip + c * aiq;
                }
                for (int i = 0; i < m; ++i) {   // rows p, q
                    const double api = A[(size_t)p * m + i], aqi = A[(size_t)q * m + i];
                    A[(size_t)p * m + i] = c * api - s * aqi;
                    A[(size_t)q * m + i] = s * api + c * aqi;
                }
                for (int i = 0; i < m; ++i) {
                    const double vip = V[(size_t)i * m + p], viq = V[(size_t)i * m + q];
                    V[(size_t)i * m + p] = c * vip - s * viq;
                    V[(size_t)i * m + q] = s * vip + c * viq;
                }
            }
    }
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * m + a] > A[(size_t)b * m + b]; });
    w.resize(m);
    std::vector<double> Vs((size_t)m * m);
    for (int j = 0; j < m; ++j) {
        w[j] = A[(size_t)order[j] * m + order[j]];
        for (int i = 0; i < m; ++i) Vs[(size_t)i * m + j] = V[(size_t)i * m + order[j]];
    }
    V.swap(Vs);
}

}  // namespace

// The k largest eigenpairs of the symmetric n x n matrix Cm (device) whose spectrum lies in [lower, +inf):
// lam[k] (device, descending), Yk[k][n] (device, orthonormal rows).  work: 4 n SB + 10 SB SB + 4 SB doubles (device).
// *converged (host) = 1 when the k leading residuals reached tol * max(1, |lambda_1|) within max_outer filtered
// iterations, 0 when the iteration stalled or the block lost rank (outputs are then meaningless).  Synchronises.
int subspace_topk_device(const double* Cm, int n, int k, double lower, double tol, int degree, int max_outer, double* lam,
                         double* Yk, double* work, int* converged, int* outer_used)
{
    *converged = 0;
    if (outer_used) *outer_used = 0;
    if (n < 2 * SB || k < 1 || k > SB / 2) return MSM_OK;   // not this method's case
    const size_t NB = (size_t)n * SB;
    double* X = work;
    double* Y = X + NB;
    double* Z = Y + NB;
    double* W = Z + NB;
    double* part = W + NB;                    // [nparts <= 8 .. n / 64][SB][SB]
    const int nparts = (int)ceil_div(n, SG_ROWS);
    double* G = part + (size_t)nparts * SB * SB;   // [2][SB][SB]: H, then S
    double* dS = G + SB * SB;
    double* dtheta = dS + SB * SB;
    double* dres = dtheta + SB;
    int* dflag = reinterpret_cast<int*>(dres + SB);
    const dim3 gprod((unsigned)ceil_div(n, SS_RT)), gqr((unsigned)ceil_div(n, SS_NT)), grot((unsigned)ceil_div(n, SS_NT / SB));
    auto product = [&](const double* Xin, const double* P, double* out, double a, double b, double c) {
        hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), 0, stream(), Cm, n, Xin, P, out, a, b, c);
    };
    auto cholqr2 = [&](double* Q) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), Q, Q, n, part);
            hipLaunchKernelGGL(ss_cholqr_kernel, gqr, dim3(SS_NT), 0, stream(), part, nparts, Q, n, dflag);
        }
    };
    MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
    hipLaunchKernelGGL(ss_init_kernel, dim3((unsigned)ceil_div(NB, 256)), dim3(256), 0, stream(), X, n);
    cholqr2(X);
    std::vector<double> H(SB * SB), w, V, res(SB);
    double prev_res = INFINITY;
    for (int outer = 0; outer <= max_outer; ++outer) {
        // ---- Rayleigh-Ritz on span(X)
        product(X, nullptr, W, 1.0, 0.0, 0.0);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, W, n, part);
        hipLaunchKernelGGL(ss_gram_sum_kernel, dim3(1), dim3(SS_NT), 0, stream(), part, nparts, G);
        MSM_HIP_CHECK(hipGetLastError());
        int flag = 0;
        MSM_HIP_CHECK(hipMemcpyAsync(H.data(), G, SB * SB * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        if (flag) return MSM_OK;   // rank loss in the Cholesky QR (or non-finite data): let the direct method decide
        for (int i = 0; i < SB; ++i)
            for (int j = i + 1; j < SB; ++j) {
                const double s = 0.5 * (H[i * SB + j] + H[j * SB + i]);
                H[i * SB + j] = H[j * SB + i] = s;
            }
        for (int i = 0; i < SB * SB; ++i)
            if (!(std::fabs(H[i]) < 1e300)) return MSM_OK;
        jacobi_eigh(H, SB, w, V);
        MSM_HIP_CHECK(hipMemcpyAsync(dS, V.data(), SB * SB * sizeof(double), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(dtheta, w.data(), SB * sizeof(double), hipMemcpyHostToDevice, stream()));
        hipLaunchKernelGGL(ss_rotate_kernel, grot, dim3(SS_NT), 0, stream(), X, W, n, dS);
        hipLaunchKernelGGL(ss_residual_kernel, dim3(1), dim3(SS_NT), 0, stream(), X, W, n, dtheta, dres);
        MSM_HIP_CHECK(hipGetLastError());
        MSM_HIP_CHECK(hipMemcpyAsync(res.data(), dres, SB * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        double rmax = 0.0;
        for (int j = 0; j < k; ++j) rmax = std::max(rmax, res[j]);
        if (outer_used) *outer_used = outer;
        if (rmax <= tol * std::max(1.0, std::fabs(w[0]))) {
            MSM_HIP_CHECK(hipMemcpyAsync(lam, dtheta, k * sizeof(double), hipMemcpyDeviceToDevice, stream()));
            hipLaunchKernelGGL(ss_emit_kernel, dim3((unsigned)ceil_div((size_t)n * k, 256)), dim3(256), 0, stream(), X, n, k, Yk);
            MSM_HIP_CHECK(hipGetLastError());
            *converged = 1;
            return MSM_OK;
        }
        // stalled: a filtered iteration that does not gain two orders of magnitude will not get there in time
        if (outer == max_outer || (outer >= 2 && !(rmax < 1e-2 * prev_res))) return MSM_OK;
        prev_res = rmax;
        // ---- filter: damp [lower, cut], cut = the smallest Ritz value of the block; scaled so that theta_1 stays O(1)
        const double cut = w[SB - 1], top = w[0];
        if (!(cut > lower) || !(top > cut)) return MSM_OK;
        const double e = 0.5 * (cut - lower), cen = 0.5 * (cut + lower);
        const double sigma1 = e / (top - cen);
        double sigma = sigma1;
        // Y = (sigma1 / e) (C X - cen X)
        product(X, nullptr, Y, sigma1 / e, -sigma1 * cen / e, 0.0);
        double *xp = X, *xc = Y, *xn = Z;
        for (int i = 2; i <= degree; ++i) {
            const double sn = 1.0 / (2.0 / sigma1 - sigma);
            // xn = (2 sn / e) (C xc - cen xc) - sigma sn xp
            product(xc, xp, xn, 2.0 * sn / e, -2.0 * sn * cen / e, -sigma * sn);
            double* t = xp;
            xp = xc;
            xc = xn;
            xn = t;
            sigma = sn;
        }
        if (xc != X) MSM_HIP_CHECK(hipMemcpyAsync(X, xc, NB * sizeof(double), hipMemcpyDeviceToDevice, stream()));
        cholqr2(X);
    }
    return MSM_OK;
}

size_t subspace_work_doubles(int n)
{
    return 4 * (size_t)n * SB + ((size_t)ceil_div(n, SG_ROWS) + 2) * SB * SB + 4 * SB + 8;
}

}  // namespace msm
