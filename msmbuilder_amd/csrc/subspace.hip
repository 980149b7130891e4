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
constexpr int SS_RT = 4;      // rows of the product per workgroup

// out = alpha * (C X) + beta * X + gamma * P   (C: n x n row-major; X, P, out: n x SB row-major; out may alias P)
// A workgroup owns SS_RT = 4 rows of the product.  Columns are processed in panels of SS_PANEL = 512: the panel of X
// (512 x 32 doubles = 128 KB) and the workgroup's 4 x 512 slice of C are loaded into LDS with every load in flight at
// once, then each wave multiplies a quarter of the panel's columns (lane = a 1 x 2 patch of the 4 x 32 block: one
// broadcast read of C and one 16-byte read of X per two multiply-adds) and the four partial blocks are summed.
constexpr int SS_PANEL = 512;
// Queued (device-driven) iteration: `dcoef` (or null) points at {alpha, beta, gamma} in device memory -- the Chebyshev
// recurrence of a filter whose bounds the DEVICE chose (ss_residual_kernel's decision) -- and `dstop` (or null) at the
// iteration's done flag: a set flag makes the launch a no-op, so that a converged block is not filtered away by the launches
// that were queued behind the decision.
__global__ __launch_bounds__(SS_NT) void ss_product_kernel(const double* __restrict__ Cm, int n, const double* __restrict__ X,
                                                           const double* P, double* out, double alpha, double beta, double gamma,
                                                           const double* __restrict__ dcoef, const int* __restrict__ dstop)
{
    extern __shared__ double ssm[];
    if (dstop && *dstop) return;
    if (dcoef) {
        alpha = dcoef[0];
        beta = dcoef[1];
        gamma = dcoef[2];
    }
    double* sx = ssm;                              // [SS_PANEL][SB]
    double* sc = sx + SS_PANEL * SB;               // [SS_RT][SS_PANEL + 2]
    double* sp = sc + SS_RT * (SS_PANEL + 2);      // [4][SS_RT][SB] partial blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * SS_RT;
    const int pr = lane >> 4, pc = (lane & 15) * 2;   // the lane's patch: row pr; columns pc, pc + 1
    double a00 = 0.0, a01 = 0.0;
    for (int c0 = 0; c0 < n; c0 += SS_PANEL) {
        const int pw = min(SS_PANEL, n - c0);
        __syncthreads();
        {
            // the whole panel in flight before the first LDS write: 32 x 16 bytes per thread, in batches of 16 loads
            // (a plain copy loop waits for every load before it issues the next: one L2 round trip per element)
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2* xg = reinterpret_cast<const d2*>(X + (size_t)c0 * SB);
            d2* xs2 = reinterpret_cast<d2*>(sx);
            const int nv = pw * SB / 2;
#pragma unroll
            for (int b0 = 0; b0 < SS_PANEL * SB / 2 / SS_NT; b0 += 16) {
                d2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int q = tid + SS_NT * (b0 + u);
                    v[u] = q < nv ? xg[q] : d2{0.0, 0.0};
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int q = tid + SS_NT * (b0 + u);
                    if (q < nv) xs2[q] = v[u];
                }
            }
            double cv[SS_RT * SS_PANEL / SS_NT];
#pragma unroll
            for (int u = 0; u < SS_RT * SS_PANEL / SS_NT; ++u) {
                const int q = tid + SS_NT * u, rr = q / SS_PANEL, cc = q - rr * SS_PANEL;
                cv[u] = (r0 + rr < n && cc < pw) ? Cm[(size_t)(r0 + rr) * n + c0 + cc] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < SS_RT * SS_PANEL / SS_NT; ++u) {
                const int q = tid + SS_NT * u, rr = q / SS_PANEL, cc = q - rr * SS_PANEL;
                sc[rr * (SS_PANEL + 2) + cc] = cv[u];
            }
        }
        __syncthreads();
        const int q4 = (pw + 3) / 4, cb = wave * q4, ce = min(pw, cb + q4);
        const double* c_r0 = sc + pr * (SS_PANEL + 2);
#pragma unroll 8
        for (int cc = cb; cc < ce; ++cc) {
            const double c0v = c_r0[cc];
            a00 += c0v * sx[cc * SB + pc];
            a01 += c0v * sx[cc * SB + pc + 1];
        }
    }
    double* mine = sp + wave * (SS_RT * SB);
    mine[pr * SB + pc] = a00;
    mine[pr * SB + pc + 1] = a01;
    __syncthreads();
    const int rl = tid >> 5, j = tid & (SB - 1), r = r0 + rl;
    if (rl < SS_RT && r < n) {
        const int e = rl * SB + j;
        const double acc = (sp[e] + sp[SS_RT * SB + e]) + (sp[2 * SS_RT * SB + e] + sp[3 * SS_RT * SB + e]);
        const size_t o = (size_t)r * SB + j;
        double v = alpha * acc;
        if (beta != 0.0) v += beta * X[o];
        if (gamma != 0.0) v += gamma * P[o];
        out[o] = v;
    }
}
constexpr size_t SS_PRODUCT_LDS = ((size_t)SS_PANEL * SB + (size_t)SS_RT * (SS_PANEL + 2) + 4 * SS_RT * SB) * sizeof(double);

// part[blockIdx][i][j] = sum over this block's rows of A[r][i] B[r][j]   (A, B: n x SB)
constexpr int SG_ROWS = 64;
__global__ __launch_bounds__(SS_NT) void ss_gram_kernel(const double* __restrict__ A, const double* __restrict__ B, int n,
                                                        double* __restrict__ part, const int* __restrict__ dstop)
{
    __shared__ double sa[SG_ROWS][SB + 1], sb[SG_ROWS][SB + 1];
    if (dstop && *dstop) return;
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
__global__ __launch_bounds__(SS_NT) void ss_gram_sum_kernel(const double* __restrict__ part, int nparts, double* __restrict__ G,
                                                            const int* __restrict__ dstop)
{
    if (dstop && *dstop) return;
    const double* mine = part + (size_t)blockIdx.x * nparts * SB * SB;   // block 0: part -> G, block 1: part2 -> H (adjacent)
    const int tid = threadIdx.x;
    double acc[SB * SB / SS_NT] = {0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < nparts; p0 += 4) {   // 16 loads in flight per trip
        double v[4][SB * SB / SS_NT];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int u = 0; u < SB * SB / SS_NT; ++u)
                v[pp][u] = p0 + pp < nparts ? mine[(size_t)(p0 + pp) * SB * SB + tid + SS_NT * u] : 0.0;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int u = 0; u < SB * SB / SS_NT; ++u) acc[u] += v[pp][u];
    }
#pragma unroll
    for (int u = 0; u < SB * SB / SS_NT; ++u) G[(size_t)blockIdx.x * SB * SB + tid + SS_NT * u] = acc[u];
}

// One round of Cholesky QR: G = X^T X (summed from the partials), G = R^T R, X <- X R^-1.  Every workgroup factors the
// 32 x 32 matrix redundantly (no exchange) and handles SS_NT rows, one row per thread in registers.
// *flag is set when G is not numerically positive definite (the block has lost rank): the caller gives up on the method.
__global__ __launch_bounds__(SS_NT) void ss_cholqr_kernel(const double* __restrict__ part, int nparts, double* X, int n,
                                                          int* __restrict__ flag, const int* __restrict__ dstop)
{
    __shared__ double sg[SB][SB + 1];
    __shared__ double sinv[SB];
    const int tid = threadIdx.x;
    if (dstop && *dstop) return;
    {
        double acc[SB * SB / SS_NT] = {0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < nparts; p0 += 4) {   // 16 loads in flight per trip
            double v[4][SB * SB / SS_NT];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int u = 0; u < SB * SB / SS_NT; ++u)
                    v[pp][u] = p0 + pp < nparts ? part[(size_t)(p0 + pp) * SB * SB + tid + SS_NT * u] : 0.0;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int u = 0; u < SB * SB / SS_NT; ++u) acc[u] += v[pp][u];
        }
#pragma unroll
        for (int u = 0; u < SB * SB / SS_NT; ++u) {
            const int q = tid + SS_NT * u;
            sg[q >> 5][q & (SB - 1)] = acc[u];
        }
    }
    __syncthreads();
    // upper Cholesky with unscaled pivot rows (see potrf_blockrow_kernel): step p subtracts g[p][r] g[p][c] / piv_p from the
    // rows r > p, one barrier per step; R[p][c] = g[p][c] / sqrt(piv_p) afterwards.  Thread -> entries (i, 4 j0 .. 4 j0 + 3).
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        for (int p = 0; p < SB - 1; ++p) {
            const double piv = sg[p][p];
            if (i > p) {
                const double f = sg[p][i] / piv;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4)
                    if (j0 + b4 >= i) sg[i][j0 + b4] -= f * sg[p][j0 + b4];
            }
            __syncthreads();
        }
        if (tid < SB) {
            const double piv = sg[tid][tid];
            if (!(piv > 0.0) && blockIdx.x == 0) *flag = 1;
            sinv[tid] = 1.0 / sqrt(piv);
        }
        __syncthreads();
        // R[p][c] = g[p][c] * sinv[p]; 1 / R[p][p] = sinv[p]
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4)
            if (j0 + b4 > i) sg[i][j0 + b4] *= sinv[i];
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
__global__ __launch_bounds__(SS_NT) void ss_rotate_kernel(double* X, double* W, int n, const double* __restrict__ S,
                                                          const int* __restrict__ dstop)
{
    __shared__ double ss[SB][SB + 1];
    const int tid = threadIdx.x;
    if (dstop && *dstop) return;
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

// State of a queued iteration (device memory, written by ss_state_init_kernel / ss_rr_kernel / ss_residual_kernel, read back
// ONCE by the host with the solve's results).
constexpr int SS_MAXDEG = 16;
struct SsState {
    int done;        // set when the iteration has decided (either way): every launch queued behind it is a no-op
    int converged;   // the k leading residuals reached the tolerance
    int outer;       // Rayleigh-Ritz rounds run
    int pad;
    double prev_res; // largest of the k leading residuals at the previous round
    double rmax;     // ... at the last round
    double coef[SS_MAXDEG][4];   // the next filter's recurrence: {alpha, beta, gamma} per product
};

// res[j] = || W[:, j] - theta[j] X[:, j] ||_2; with `st` (queued iteration) thread 0 then DECIDES what the host's loop decided
// between its round trips: converged (k leading residuals <= tol max(1, |theta_0|)), stalled / out of rounds / degenerate
// bounds (done, not converged: the caller's fallback takes over), or the recurrence of the next filter of `degree` products
// damping [lower, theta_31] (Zhou & Saad's scaled three-term form, subspace_topk_device's run_filter).
__global__ __launch_bounds__(SS_NT) void ss_residual_kernel(const double* __restrict__ X, const double* __restrict__ W, int n,
                                                            const double* __restrict__ theta, double* __restrict__ res,
                                                            SsState* st, int k, double tol, double lower, int degree, int last_round)
{
    __shared__ double red[SS_NT / SB][SB];
    const int tid = threadIdx.x, j = tid & (SB - 1), part = tid >> 5;
    if (st && st->done) return;
    const double th = theta[j];
    double s = 0.0;
    for (int r = part; r < n; r += 8 * (SS_NT / SB)) {   // 16 loads in flight per trip
        double wv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = r + u * (SS_NT / SB);
            wv[u] = rr < n ? W[(size_t)rr * SB + j] : 0.0;
            xv[u] = rr < n ? X[(size_t)rr * SB + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double dlt = wv[u] - th * xv[u];
            s += dlt * dlt;
        }
    }
    red[part][j] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
        for (int p = 0; p < SS_NT / SB; ++p) t += red[p][j];
        res[j] = sqrt(t);
        red[0][j] = sqrt(t);
    }
    if (!st) return;
    __syncthreads();
    if (tid == 0) {
        double rmax = 0.0;
        bool finite = true;
        for (int jj = 0; jj < k; ++jj) {
            const double r = red[0][jj];
            if (!(r == r)) finite = false;
            rmax = r > rmax ? r : rmax;
        }
        const double top = theta[0], cut = theta[SB - 1];
        const int outer = st->outer;          // rounds completed BEFORE this one
        st->outer = outer + 1;
        st->rmax = rmax;
        const double scale = fabs(top) > 1.0 ? fabs(top) : 1.0;
        if (!finite) {
            st->done = 1;
        } else if (rmax <= tol * scale) {
            st->converged = 1;
            st->done = 1;
        } else if (last_round || (outer >= 2 && !(rmax < 1e-2 * st->prev_res)) || !(cut > lower) || !(top > cut)) {
            st->done = 1;                     // stalled, out of rounds, or no usable filter bounds: the caller's fallback decides
        } else {
            const double e = 0.5 * (cut - lower), cen = 0.5 * (cut + lower);
            const double sigma1 = e / (top - cen);
            double sigma = sigma1;
            st->coef[0][0] = sigma1 / e;
            st->coef[0][1] = -sigma1 * cen / e;
            st->coef[0][2] = 0.0;
            for (int i = 2; i <= degree; ++i) {
                const double sn = 1.0 / (2.0 / sigma1 - sigma);
                st->coef[i - 1][0] = 2.0 * sn / e;
                st->coef[i - 1][1] = -2.0 * sn * cen / e;
                st->coef[i - 1][2] = -sigma * sn;
                sigma = sn;
            }
        }
        st->prev_res = rmax;
    }
}

__global__ void ss_state_init_kernel(SsState* st)
{
    if (threadIdx.x == 0) {
        st->done = 0;
        st->converged = 0;
        st->outer = 0;
        st->pad = 0;
        st->prev_res = 1.0;   // the residual scale of a random block: what the first measured residual is compared with
        st->rmax = 0.0;
    }
}

// dst <- src unless the iteration has decided (the filter's last product may have landed in a scratch block)
__global__ void ss_copy_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t nelem, const int* __restrict__ dstop)
{
    if (dstop && *dstop) return;
    const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q < nelem) dst[q] = src[q];
}

// ---------------------------------------------------------------------------------------------------------------------
// Rayleigh-Ritz on the DEVICE (round 5): the 32 x 32 generalized problem H s = theta G s that the host solved between two
// copies and two stream synchronisations (small_geigh: ~0.13 ms of GPU idle per round).  One workgroup:
//   G = R^T R (upper Cholesky in LDS), M = R^-T H R^-1 (two sweeps of row substitutions), the symmetric eigenproblem of M by
//   ONE-SIDED Jacobi (Hestenes) on the shifted matrix B = M + shift I > 0 -- column rotations only, 16 disjoint pairs per
//   round each owned by 16 lanes of one wavefront (dot products by DPP-free shuffles, no barrier inside a pair; one barrier
//   per round because the tournament re-pairs the columns) -- theta_j = v_j^T M v_j, descending order, S = R^-1 V.
// The pairs that come out of the iteration are verified against the reduced matrix by the caller (pair_residual_device),
// so a failure here can only cost the fallback's time.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SS_NT) void ss_rr_kernel(const double* __restrict__ GH, double* __restrict__ S, double* __restrict__ theta,
                                                      const int* __restrict__ qrflag, SsState* st)
{
    __shared__ double sg[SB][SB + 1];      // G -> R (upper)
    __shared__ double sm[SB][SB + 1];      // H -> M
    __shared__ double sw[SB][SB + 1];      // columns of W = B V, stored [column][row]
    __shared__ double sv[SB][SB + 1];      // columns of V, stored [column][row]
    __shared__ double sinv[SB], sth[SB];
    __shared__ int sord[SB], srot, sbad;
    const int tid = threadIdx.x;
    if (st->done) return;
    if (tid == 0) sbad = (*qrflag) ? 1 : 0;   // rank loss in a Cholesky QR (or non-finite data)
    for (int q = tid; q < SB * SB; q += SS_NT) {
        const int i = q >> 5, j = q & (SB - 1);
        sg[i][j] = 0.5 * (GH[i * SB + j] + GH[j * SB + i]);
        sm[i][j] = 0.5 * (GH[SB * SB + i * SB + j] + GH[SB * SB + j * SB + i]);
    }
    __syncthreads();
    {
        double bad = 0.0;
        for (int q = tid; q < SB * SB; q += SS_NT) {
            const int i = q >> 5, j = q & (SB - 1);
            if (!(fabs(sg[i][j]) < 1e300) || !(fabs(sm[i][j]) < 1e300)) bad = 1.0;
        }
        if (bad != 0.0) sbad = 1;
    }
    __syncthreads();
    // ---- G = R^T R: ss_cholqr_kernel's loop (unscaled pivot rows, one barrier per step)
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        for (int p = 0; p < SB - 1; ++p) {
            const double piv = sg[p][p];
            if (i > p) {
                const double f = sg[p][i] / piv;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4)
                    if (j0 + b4 >= i) sg[i][j0 + b4] -= f * sg[p][j0 + b4];
            }
            __syncthreads();
        }
        if (tid < SB) {
            const double piv = sg[tid][tid];
            if (!(piv > 0.0)) sbad = 1;
            sinv[tid] = 1.0 / sqrt(piv);
        }
        __syncthreads();
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4)
            if (j0 + b4 > i) sg[i][j0 + b4] *= sinv[i];   // R[p][c] = g[p][c] / sqrt(piv_p); 1 / R[p][p] = sinv[p]
        __syncthreads();
    }
    if (sbad) {
        if (tid == 0) st->done = 1;    // not converged: the caller's fallback takes over
        return;
    }
    // ---- M = R^-T H R^-1: rows of H times R^-1 (thread = a row, ss_cholqr_kernel's substitution), transpose, again
    for (int pass = 0; pass < 2; ++pass) {
        if (tid < SB) {
            double x[SB];
#pragma unroll
            for (int jj = 0; jj < SB; ++jj) x[jj] = sm[tid][jj];
#pragma unroll
            for (int jj = 0; jj < SB; ++jj) {
                double t = x[jj];
#pragma unroll
                for (int i2 = 0; i2 < jj; ++i2) t -= x[i2] * sg[i2][jj];
                x[jj] = t * sinv[jj];
            }
#pragma unroll
            for (int jj = 0; jj < SB; ++jj) sw[jj][tid] = x[jj];   // transposed
        }
        __syncthreads();
        for (int q = tid; q < SB * SB; q += SS_NT) sm[q >> 5][q & (SB - 1)] = sw[q >> 5][q & (SB - 1)];
        __syncthreads();
    }
    // symmetrise M; shift = 1 + the largest absolute row sum (Gershgorin): B = M + shift I has its spectrum in [1, 2 shift]
    for (int q = tid; q < SB * SB; q += SS_NT) {
        const int i = q >> 5, j = q & (SB - 1);
        if (i < j) {
            const double t = 0.5 * (sm[i][j] + sm[j][i]);
            sm[i][j] = sm[j][i] = t;
        }
    }
    __syncthreads();
    if (tid < SB) {
        double rs = 0.0;
        for (int jj = 0; jj < SB; ++jj) rs += fabs(sm[tid][jj]);
        sth[tid] = rs;
    }
    __syncthreads();
    double shift = 0.0;
    for (int jj = 0; jj < SB; ++jj) shift = sth[jj] > shift ? sth[jj] : shift;
    shift += 1.0;
    __syncthreads();
    for (int q = tid; q < SB * SB; q += SS_NT) {
        const int c = q >> 5, r = q & (SB - 1);
        sw[c][r] = sm[r][c] + (r == c ? shift : 0.0);
        sv[c][r] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    // ---- one-sided Jacobi: pair g = tid / 16 of the round, its 16 lanes hold rows 2 l, 2 l + 1 of the pair's two columns
    {
        const int g = tid >> 4, l = tid & 15;
        for (int sweep = 0; sweep < 14; ++sweep) {
            if (tid == 0) srot = 0;
            __syncthreads();
            for (int round = 0; round < SB - 1; ++round) {
                // round-robin tournament on 32 players: player 31 fixed, the others rotate
                int p, qc;
                if (g == 0) {
                    p = SB - 1;
                    qc = round;
                } else {
                    p = (round + g) % (SB - 1);
                    qc = (round + (SB - 1) - g) % (SB - 1);
                }
                double wp0 = sw[p][2 * l], wp1 = sw[p][2 * l + 1], wq0 = sw[qc][2 * l], wq1 = sw[qc][2 * l + 1];
                double al = wp0 * wp0 + wp1 * wp1, be = wq0 * wq0 + wq1 * wq1, ga = wp0 * wq0 + wp1 * wq1;
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    al += __shfl_xor(al, m, 16);
                    be += __shfl_xor(be, m, 16);
                    ga += __shfl_xor(ga, m, 16);
                }
                if (fabs(ga) > 2e-14 * sqrt(al * be)) {   // (uniform over the pair's 16 lanes; the dot product of 32 terms carries ~1e-15 of rounding noise:
                                                          //  a threshold at that level never lets a sweep come out clean -- 30 sweeps, 208 us, measured)
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                    sw[p][2 * l] = c * wp0 - sn * wq0;
                    sw[p][2 * l + 1] = c * wp1 - sn * wq1;
                    sw[qc][2 * l] = sn * wp0 + c * wq0;
                    sw[qc][2 * l + 1] = sn * wp1 + c * wq1;
                    const double vp0 = sv[p][2 * l], vp1 = sv[p][2 * l + 1], vq0 = sv[qc][2 * l], vq1 = sv[qc][2 * l + 1];
                    sv[p][2 * l] = c * vp0 - sn * vq0;
                    sv[p][2 * l + 1] = c * vp1 - sn * vq1;
                    sv[qc][2 * l] = sn * vp0 + c * vq0;
                    sv[qc][2 * l + 1] = sn * vp1 + c * vq1;
                    if (l == 0) srot = 1;
                }
                __syncthreads();
            }
            if (!srot) break;          // (uniform: read after the round's barrier)
            __syncthreads();
        }
    }
    // ---- theta_j = v_j^T M v_j (V is orthogonal to rounding); descending order
    if (tid < SB) {
        double acc = 0.0;
        for (int r = 0; r < SB; ++r) {
            double mv = 0.0;
            for (int c2 = 0; c2 < SB; ++c2) mv += sm[r][c2] * sv[tid][c2];
            acc += sv[tid][r] * mv;
        }
        sth[tid] = acc;
    }
    __syncthreads();
    if (tid < SB) {
        int rank = 0;
        const double mine = sth[tid];
        for (int jj = 0; jj < SB; ++jj) {
            const double o = sth[jj];
            if (o > mine || (o == mine && jj < tid)) ++rank;
        }
        sord[rank] = tid;
    }
    __syncthreads();
    // ---- S = R^-1 V (back substitution per column, thread = wanted column), theta
    if (tid < SB) {
        const int src = sord[tid];
        double x[SB];
#pragma unroll
        for (int r = 0; r < SB; ++r) x[r] = sv[src][r];
#pragma unroll
        for (int r = SB - 1; r >= 0; --r) {
            double t = x[r];
#pragma unroll
            for (int c2 = r + 1; c2 < SB; ++c2) t -= sg[r][c2] * x[c2];
            x[r] = t * sinv[r];
        }
#pragma unroll
        for (int r = 0; r < SB; ++r) S[r * SB + tid] = x[r];
        theta[tid] = sth[src];
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

// Small dense symmetric eigenproblem on the host: Householder tridiagonalisation with accumulated transformations, then
// QL with implicit shifts (the classical tred2 / tql2 pair).  A (m x m, row-major, destroyed) -> eigenvalues w and
// eigenvectors as COLUMNS of V, sorted by descending eigenvalue.  ~m^3 flops: microseconds at m = 32.
// sqrt(a^2 + b^2): the plain form where it cannot over- or underflow (std::hypot's careful scaling costs ~25 ns, and QL
// calls it once per rotation: a third of the function at m = 32), std::hypot otherwise
inline double fast_hypot(double a, double b)
{
    const double s = a * a + b * b;
    return (s > 1e-280 && s < 1e280) ? std::sqrt(s) : std::hypot(a, b);
}

void small_eigh(std::vector<double>& A, int m, std::vector<double>& w, std::vector<double>& V)
{
    std::vector<double> d(m), e(m);
    if (m > 64) return;   // (the accumulation below keeps a 64-entry row on the stack; callers pass the 32 x 32 Ritz problem)
    auto a = [&](int i, int j) -> double& { return A[(size_t)i * m + j]; };
    for (int i = m - 1; i > 0; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k2 = 0; k2 <= l; ++k2) scale += std::fabs(a(i, k2));
            if (scale == 0.0) {
                e[i] = a(i, l);
            } else {
                for (int k2 = 0; k2 <= l; ++k2) {
                    a(i, k2) /= scale;
                    h += a(i, k2) * a(i, k2);
                }
                double f = a(i, l);
                double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                a(i, l) = f - g;
                // The active block [0..l] x [0..l] is kept FULL (both triangles), so that p = A u / h is a contiguous
                // row-times-vector product (four accumulators: a chain of dependent adds is 4 cycles per term) and the
                // rank-2 update A -= u q^T + q u^T runs over whole rows (vectorisable).  The textbook form touches the lower
                // triangle only and walks columns of the row-major matrix for half of its terms.
                const double* __restrict__ ui = &A[(size_t)i * m];
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    const double* __restrict__ aj = &A[(size_t)j * m];
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int k2 = 0;
                    for (; k2 + 4 <= l + 1; k2 += 4) {
                        s0 += aj[k2] * ui[k2];
                        s1 += aj[k2 + 1] * ui[k2 + 1];
                        s2 += aj[k2 + 2] * ui[k2 + 2];
                        s3 += aj[k2 + 3] * ui[k2 + 3];
                    }
                    for (; k2 <= l; ++k2) s0 += aj[k2] * ui[k2];
                    e[j] = ((s0 + s1) + (s2 + s3)) / h;
                    f += e[j] * ui[j];
                }
                for (int j = 0; j <= l; ++j) a(j, i) = ui[j] / h;   // u / h, kept in column i for the accumulation below
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) e[j] -= hh * ui[j];
                for (int j = 0; j <= l; ++j) {
                    const double fj = ui[j], gj = e[j];
                    double* __restrict__ aj = &A[(size_t)j * m];
                    for (int k2 = 0; k2 <= l; ++k2) aj[k2] -= fj * e[k2] + gj * ui[k2];
                }
            }
        } else {
            e[i] = a(i, l);
        }
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < m; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            // Q <- Q - (Q u / h)(u^T Q) on the block [0..l]: g = u^T Q accumulated row by row, then a rank-1 update row
            // by row (the same sums in the same order as the column-wise textbook loops, over contiguous memory)
            double gv[64];
            for (int j = 0; j <= l; ++j) gv[j] = 0.0;
            for (int k2 = 0; k2 <= l; ++k2) {
                const double uk = a(i, k2);
                const double* __restrict__ qk = &A[(size_t)k2 * m];
                for (int j = 0; j <= l; ++j) gv[j] += uk * qk[j];
            }
            for (int k2 = 0; k2 <= l; ++k2) {
                const double c = a(k2, i);
                double* __restrict__ qk = &A[(size_t)k2 * m];
                for (int j = 0; j <= l; ++j) qk[j] -= gv[j] * c;
            }
        }
        d[i] = a(i, i);
        a(i, i) = 1.0;
        for (int j = 0; j <= l; ++j) a(j, i) = a(i, j) = 0.0;
    }
    // QL with implicit shifts on (d, e), rotations accumulated into Zt = A^T (ROWS = eigenvectors: a rotation of two
    // eigenvectors then runs over two contiguous rows -- as columns of the row-major A it was a stride-m walk, and the
    // rotations are most of this function's flops: 72 -> ~25 us at m = 32, inside the solve's two host round trips)
    std::vector<double> Zt((size_t)m * m);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Zt[(size_t)j * m + i] = a(i, j);
    for (int i = 1; i < m; ++i) e[i - 1] = e[i];
    e[m - 1] = 0.0;
    for (int l = 0; l < m; ++l) {
        int iter = 0, mm;
        do {
            for (mm = l; mm < m - 1; ++mm) {
                const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
                if (std::fabs(e[mm]) <= 2.3e-16 * dd) break;
            }
            if (mm != l) {
                if (iter++ == 60) break;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = fast_hypot(g, 1.0);
                g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s2 = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = mm - 1; i >= l; --i) {
                    double f = s2 * e[i];
                    const double bb = c * e[i];
                    e[i + 1] = (r = fast_hypot(f, g));
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[mm] = 0.0;
                        break;
                    }
                    s2 = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s2 + 2.0 * c * bb;
                    d[i + 1] = g + (p = s2 * r);
                    g = c * r - bb;
                    double* __restrict__ z0 = &Zt[(size_t)i * m];
                    double* __restrict__ z1 = &Zt[(size_t)(i + 1) * m];
                    for (int k2 = 0; k2 < m; ++k2) {
                        const double f1 = z1[k2], f0 = z0[k2];
                        z1[k2] = s2 * f0 + c * f1;
                        z0[k2] = c * f0 - s2 * f1;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[mm] = 0.0;
            }
        } while (mm != l);
    }
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
    w.resize(m);
    V.assign((size_t)m * m, 0.0);
    for (int j = 0; j < m; ++j) {
        w[j] = d[order[j]];
        for (int i = 0; i < m; ++i) V[(size_t)i * m + j] = Zt[(size_t)order[j] * m + i];
    }
}

// Generalized small problem H s = theta G s (G symmetric positive definite, close to I): G = L L^T, eig(L^-1 H L^-T),
// S = L^-T S'.  Columns of S are G-orthonormal, so X S is orthonormal when G = X^T X.  Returns false if G is not SPD.
bool small_geigh(std::vector<double>& H, std::vector<double>& G, int m, std::vector<double>& w, std::vector<double>& S)
{
    std::vector<double> L((size_t)m * m, 0.0);
    for (int j = 0; j < m; ++j) {
        double s = G[(size_t)j * m + j];
        for (int k2 = 0; k2 < j; ++k2) s -= L[(size_t)j * m + k2] * L[(size_t)j * m + k2];
        if (!(s > 0.0)) return false;
        const double ljj = std::sqrt(s);
        L[(size_t)j * m + j] = ljj;
        for (int i = j + 1; i < m; ++i) {
            double t = G[(size_t)i * m + j];
            for (int k2 = 0; k2 < j; ++k2) t -= L[(size_t)i * m + k2] * L[(size_t)j * m + k2];
            L[(size_t)i * m + j] = t / ljj;
        }
    }
    // M = L^-1 H L^-T: forward substitutions on the rows, then on the columns
    std::vector<double> M(H);
    // M <- L^-1 M: row i minus the finished rows above it (contiguous inner loops the compiler vectorises; the dot-product
    // forms of these substitutions are chains of dependent adds, 4 cycles each).  Twice, with a transposition in between:
    // (L^-1 H)^T = H L^-T, so the second application gives L^-1 H L^-T.
    auto forward = [&](std::vector<double>& X) {
        for (int i = 0; i < m; ++i) {
            double* __restrict__ xi = &X[(size_t)i * m];
            for (int k2 = 0; k2 < i; ++k2) {
                const double lik = L[(size_t)i * m + k2];
                const double* __restrict__ xk = &X[(size_t)k2 * m];
                for (int c = 0; c < m; ++c) xi[c] -= lik * xk[c];
            }
            const double lii = L[(size_t)i * m + i];
            for (int c = 0; c < m; ++c) xi[c] /= lii;
        }
    };
    forward(M);
    {
        std::vector<double> T((size_t)m * m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) T[(size_t)j * m + i] = M[(size_t)i * m + j];
        forward(T);
        M.swap(T);
    }
    for (int i = 0; i < m; ++i)
        for (int j = i + 1; j < m; ++j) {
            const double t = 0.5 * (M[(size_t)i * m + j] + M[(size_t)j * m + i]);
            M[(size_t)i * m + j] = M[(size_t)j * m + i] = t;
        }
    std::vector<double> Sp;
    small_eigh(M, m, w, Sp);
    S = Sp;
    for (int i = m - 1; i >= 0; --i) {   // S = L^-T S': back substitution, row i minus the finished rows below it
        double* __restrict__ si = &S[(size_t)i * m];
        for (int k2 = i + 1; k2 < m; ++k2) {
            const double lki = L[(size_t)k2 * m + i];
            const double* __restrict__ sk = &S[(size_t)k2 * m];
            for (int c = 0; c < m; ++c) si[c] -= lki * sk[c];
        }
        const double lii = L[(size_t)i * m + i];
        for (int c = 0; c < m; ++c) si[c] /= lii;
    }
    return true;
}

}  // namespace

// The k largest eigenpairs of the symmetric n x n matrix Cm (device) whose spectrum lies in [lower, +inf):
// lam[k] (device, descending), Yk[k][n] (device, orthonormal rows).  work: 4 n SB + 10 SB SB + 4 SB doubles (device).
// *converged (host) = 1 when the k leading residuals reached tol * max(1, |lambda_1|) within max_outer filtered
// iterations, 0 when the iteration stalled or the block lost rank (outputs are then meaningless).  Synchronises.
int subspace_topk_device(const double* Cm, int n, int k, double lower, double tol, int degree, int max_outer, double* lam,
                         double* Yk, double* work, double* pin, int* converged, int* outer_used, double first_cut, double first_top)
{
    *converged = 0;
    if (outer_used) *outer_used = 0;
    if (n < 2 * SB || k < 1 || k > SB / 2) return MSM_OK;   // not this method's case
    const size_t NB = (size_t)n * SB;
    double* X = work;
    double* Y = X + NB;
    double* Z = Y + NB;
    double* W = Z + NB;
    const int nparts = (int)ceil_div(n, SG_ROWS);
    double* part = W + NB;                             // [nparts][SB][SB]
    double* part2 = part + (size_t)nparts * SB * SB;   // [nparts][SB][SB]
    double* GH = part2 + (size_t)nparts * SB * SB;     // [2][SB][SB]: G = X^T X, H = X^T W
    double* dS = GH + 2 * SB * SB;
    double* dtheta = dS + SB * SB;
    double* dres = dtheta + SB;
    int* dflag = reinterpret_cast<int*>(dres + SB);
    // pinned staging for the small host <-> device transfers of the Rayleigh-Ritz steps: `pin`, subspace_pin_doubles()
    // doubles of the CALLER's pinned memory (round 3 kept a function-static buffer: not safe for two handles in two threads)
    double* hGH = pin;                 // 2 SB^2
    double* hS = pin + 2 * SB * SB;    // SB^2
    double* htheta = hS + SB * SB;     // SB
    double* hres = htheta + SB;        // SB
    int* hflag = reinterpret_cast<int*>(hres + SB);
    static bool attr = false;
    if (!attr) {
        MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ss_product_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)SS_PRODUCT_LDS));
        attr = true;
    }
    const dim3 gprod((unsigned)ceil_div(n, SS_RT)), gqr((unsigned)ceil_div(n, SS_NT)), grot((unsigned)ceil_div(n, SS_NT / SB));
    auto product = [&](const double* Xin, const double* P, double* out, double a, double b, double c) {
        hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), SS_PRODUCT_LDS, stream(), Cm, n, Xin, P, out, a, b, c, (const double*)nullptr, (const int*)nullptr);
    };
    // one round of Cholesky QR: orthonormal to ~ cond(X)^2 eps, which the Rayleigh-Ritz step below absorbs by solving
    // the small GENERALIZED problem H s = theta G s with G = X^T X
    auto cholqr = [&](double* Q) {
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), Q, Q, n, part, (const int*)nullptr);
        hipLaunchKernelGGL(ss_cholqr_kernel, gqr, dim3(SS_NT), 0, stream(), part, nparts, Q, n, dflag, (const int*)nullptr);
    };
    MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
    hipLaunchKernelGGL(ss_init_kernel, dim3((unsigned)ceil_div(NB, 256)), dim3(256), 0, stream(), X, n);
    cholqr(X);
    std::vector<double> H(SB * SB), G(SB * SB), w, V;
    double prev_res = INFINITY;
    int last_deg = degree;
    // Chebyshev filter that damps [lower, cut] and is scaled so that `top` stays O(1): `deg` degrees in chunks of at most
    // `degree`, a Cholesky QR after each -- the block's condition number grows with the degree of one polynomial (a single
    // filter of degree 30 lost rank at F = 200), and re-orthonormalising costs 30 us on the device where a Rayleigh-Ritz
    // round trip costs 0.2 ms
    auto run_filter = [&](double cut, double top, int deg) {
        const double e = 0.5 * (cut - lower), cen = 0.5 * (cut + lower);
        const double sigma1 = e / (top - cen);
        const int nchunk = (deg + degree - 1) / degree;
        int left = deg;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int d = (left + (nchunk - ch) - 1) / (nchunk - ch);
            left -= d;
            double sigma = sigma1;
            // Y = (sigma1 / e) (C X - cen X)
            product(X, nullptr, Y, sigma1 / e, -sigma1 * cen / e, 0.0);
            double *xp = X, *xc = Y, *xn = Z;
            for (int i = 2; i <= d; ++i) {
                const double sn = 1.0 / (2.0 / sigma1 - sigma);
                // xn = (2 sn / e) (C xc - cen xc) - sigma sn xp
                product(xc, xp, xn, 2.0 * sn / e, -2.0 * sn * cen / e, -sigma * sn);
                double* t = xp;
                xp = xc;
                xc = xn;
                xn = t;
                sigma = sn;
            }
            if (xc != X) (void)hipMemcpyAsync(X, xc, NB * sizeof(double), hipMemcpyDeviceToDevice, stream());
            cholqr(X);
        }
    };
    // A first filter on the caller's PRIOR for the spectrum (first_cut < first_top; NaN: none) instead of a Rayleigh-Ritz
    // round on the random block, whose only product is a pair of filter bounds no better than a sensible prior (tICA: the
    // reduced matrix has its spectrum in [-1, 1], noise near 0): one round trip (0.25 ms) less per solve.
    int filters = 0;
    if (first_cut > lower && first_top > first_cut) {
        run_filter(first_cut, first_top, degree);
        filters = 1;
        prev_res = 1.0;   // the residual scale of a random block (the spectrum's width): lets the first measured residual size the next filter
    }
    for (int outer = 0; outer <= max_outer; ++outer) {
        // ---- Rayleigh-Ritz on span(X)
        product(X, nullptr, W, 1.0, 0.0, 0.0);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, X, n, part, (const int*)nullptr);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, W, n, part2, (const int*)nullptr);
        hipLaunchKernelGGL(ss_gram_sum_kernel, dim3(2), dim3(SS_NT), 0, stream(), part, nparts, GH, (const int*)nullptr);
        MSM_HIP_CHECK(hipGetLastError());
        MSM_HIP_CHECK(hipMemcpyAsync(hGH, GH, 2 * SB * SB * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        if (*hflag) return MSM_OK;   // rank loss in the Cholesky QR (or non-finite data): let the direct method decide
        for (int i = 0; i < SB; ++i)
            for (int j = 0; j < SB; ++j) {
                G[i * SB + j] = 0.5 * (hGH[i * SB + j] + hGH[j * SB + i]);
                H[i * SB + j] = 0.5 * (hGH[SB * SB + i * SB + j] + hGH[SB * SB + j * SB + i]);
            }
        for (int i = 0; i < SB * SB; ++i)
            if (!(std::fabs(H[i]) < 1e300) || !(std::fabs(G[i]) < 1e300)) return MSM_OK;
        if (!small_geigh(H, G, SB, w, V)) return MSM_OK;
        memcpy(hS, V.data(), SB * SB * sizeof(double));
        memcpy(htheta, w.data(), SB * sizeof(double));
        MSM_HIP_CHECK(hipMemcpyAsync(dS, hS, (SB * SB + SB) * sizeof(double), hipMemcpyHostToDevice, stream()));   // S and theta are adjacent
        hipLaunchKernelGGL(ss_rotate_kernel, grot, dim3(SS_NT), 0, stream(), X, W, n, dS, (const int*)nullptr);
        hipLaunchKernelGGL(ss_residual_kernel, dim3(1), dim3(SS_NT), 0, stream(), X, W, n, dtheta, dres, (SsState*)nullptr, 0, 0.0, 0.0, 0, 0);
        MSM_HIP_CHECK(hipGetLastError());
        MSM_HIP_CHECK(hipMemcpyAsync(hres, dres, SB * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        double rmax = 0.0;
        for (int j = 0; j < k; ++j) rmax = std::max(rmax, hres[j]);
        if (!(rmax == rmax)) return MSM_OK;
        if (outer_used) *outer_used = outer + filters;
        if (rmax <= tol * std::max(1.0, std::fabs(w[0]))) {
            MSM_HIP_CHECK(hipMemcpyAsync(lam, dtheta, k * sizeof(double), hipMemcpyDeviceToDevice, stream()));
            hipLaunchKernelGGL(ss_emit_kernel, dim3((unsigned)ceil_div((size_t)n * k, 256)), dim3(256), 0, stream(), X, n, k, Yk);
            MSM_HIP_CHECK(hipGetLastError());
            *converged = 1;
            return MSM_OK;
        }
        // stalled: a filtered iteration that does not gain two orders of magnitude will not get there in time
        if (outer == max_outer || (outer >= 2 && !(rmax < 1e-2 * prev_res))) return MSM_OK;
        // Degree of the next filter: `degree`, or -- once a filter's gain is known -- what the remaining distance to the
        // tolerance asks for at that rate per degree (up to 3 x degree): one longer filter instead of two filters with a
        // Rayleigh-Ritz round trip (two host synchronisations, ~0.2 ms) between them.
        int deg = degree;
        if ((outer >= 1 || filters) && prev_res < INFINITY && rmax < prev_res) {
            const double tol_abs = tol * std::max(1.0, std::fabs(w[0]));
            const double rate = std::log(rmax / prev_res) / last_deg;        // < 0, per degree
            const double need = std::log(0.3 * tol_abs / rmax) / rate;       // degrees still needed, with a margin
            if (need > degree) deg = (int)std::min(3.0 * degree, std::ceil(need));
        }
        last_deg = deg;
        prev_res = rmax;
        // ---- filter: damp [lower, cut], cut = the smallest Ritz value of the block; scaled so that theta_1 stays O(1)
        const double cut = w[SB - 1], top = w[0];
        if (!(cut > lower) || !(top > cut)) return MSM_OK;
        run_filter(cut, top, deg);
    }
    return MSM_OK;
}

// The same iteration QUEUED (round 5): no host synchronisation, no copy.  The Rayleigh-Ritz problems are solved by
// ss_rr_kernel, the decisions (converged / stalled / next filter's bounds) are taken by ss_residual_kernel's last thread and
// every later launch checks the iteration's `done` flag: a fixed schedule -- the prior filter, then `rounds` x (Rayleigh-Ritz,
// filter of `degree` products), a last Rayleigh-Ritz -- of which the launches behind the decision are no-ops.  lam / Yk are
// emitted unconditionally and are meaningful only when the state read back by the caller (*dstate: SsState, {done, converged,
// outer, -} as four ints at its head) says converged; otherwise the caller runs subspace_topk_device (adaptive degrees, more
// rounds) or its own fallback.  Needs first_cut / first_top (the prior for the first filter).
int subspace_topk_queued(const double* Cm, int n, int k, double lower, double tol, int degree, int rounds, double* lam, double* Yk,
                         double* work, double first_cut, double first_top, const void** dstate)
{
    if (dstate) *dstate = nullptr;
    if (n < 2 * SB || k < 1 || k > SB / 2 || degree < 2 || degree > SS_MAXDEG || !(first_cut > lower) || !(first_top > first_cut))
        return MSM_OK;   // not this route's case: *dstate stays null
    const size_t NB = (size_t)n * SB;
    double* X = work;
    double* Y = X + NB;
    double* Z = Y + NB;
    double* W = Z + NB;
    const int nparts = (int)ceil_div(n, SG_ROWS);
    double* part = W + NB;
    double* part2 = part + (size_t)nparts * SB * SB;
    double* GH = part2 + (size_t)nparts * SB * SB;
    double* dS = GH + 2 * SB * SB;
    double* dtheta = dS + SB * SB;
    double* dres = dtheta + SB;
    int* dflag = reinterpret_cast<int*>(dres + SB);
    SsState* st = reinterpret_cast<SsState*>(dres + SB + 8);
    const int* stop = &st->done;
    static bool attr = false;
    if (!attr) {
        MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ss_product_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)SS_PRODUCT_LDS));
        attr = true;
    }
    const dim3 gprod((unsigned)ceil_div(n, SS_RT)), gqr((unsigned)ceil_div(n, SS_NT)), grot((unsigned)ceil_div(n, SS_NT / SB));
    MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
    hipLaunchKernelGGL(ss_state_init_kernel, dim3(1), dim3(64), 0, stream(), st);
    hipLaunchKernelGGL(ss_init_kernel, dim3((unsigned)ceil_div(NB, 256)), dim3(256), 0, stream(), X, n);
    auto cholqr = [&](const int* guard) {
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, X, n, part, guard);
        hipLaunchKernelGGL(ss_cholqr_kernel, gqr, dim3(SS_NT), 0, stream(), part, nparts, X, n, dflag, guard);
    };
    cholqr(nullptr);
    // a filter of `degree` products: host-known recurrence (the prior) or the device's (st->coef), then one Cholesky QR
    auto run_filter = [&](bool device_coef, double cut, double top) {
        const double e = 0.5 * (cut - lower), cen = 0.5 * (cut + lower);
        const double sigma1 = device_coef ? 0.0 : e / (top - cen);
        double sigma = sigma1;
        const int* guard = device_coef ? stop : nullptr;
        double *xp = X, *xc = Y, *xn = Z;
        for (int i = 1; i <= degree; ++i) {
            double a = 0.0, b = 0.0, c = 0.0;
            if (!device_coef) {
                if (i == 1) {
                    a = sigma1 / e;
                    b = -sigma1 * cen / e;
                } else {
                    const double sn = 1.0 / (2.0 / sigma1 - sigma);
                    a = 2.0 * sn / e;
                    b = -2.0 * sn * cen / e;
                    c = -sigma * sn;
                    sigma = sn;
                }
            }
            const double* coef = device_coef ? &st->coef[i - 1][0] : nullptr;
            if (i == 1) {
                hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), SS_PRODUCT_LDS, stream(), Cm, n, X, (const double*)nullptr, Y, a, b, c, coef, guard);
            } else {
                hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), SS_PRODUCT_LDS, stream(), Cm, n, xc, (const double*)xp, xn, a, b, c, coef, guard);
                double* t = xp;
                xp = xc;
                xc = xn;
                xn = t;
            }
        }
        if (xc != X) hipLaunchKernelGGL(ss_copy_kernel, dim3((unsigned)ceil_div(NB, 256)), dim3(256), 0, stream(), X, (const double*)xc, NB, guard);
        cholqr(guard);
    };
    run_filter(false, first_cut, first_top);
    for (int r = 0; r <= rounds; ++r) {
        hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), SS_PRODUCT_LDS, stream(), Cm, n, X, (const double*)nullptr, W, 1.0, 0.0, 0.0,
                           (const double*)nullptr, stop);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, X, n, part, stop);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, W, n, part2, stop);
        hipLaunchKernelGGL(ss_gram_sum_kernel, dim3(2), dim3(SS_NT), 0, stream(), part, nparts, GH, stop);
        hipLaunchKernelGGL(ss_rr_kernel, dim3(1), dim3(SS_NT), 0, stream(), GH, dS, dtheta, dflag, st);
        hipLaunchKernelGGL(ss_rotate_kernel, grot, dim3(SS_NT), 0, stream(), X, W, n, dS, stop);
        hipLaunchKernelGGL(ss_residual_kernel, dim3(1), dim3(SS_NT), 0, stream(), X, W, n, dtheta, dres, st, k, tol, lower, degree, r == rounds ? 1 : 0);
        if (r < rounds) run_filter(true, 0.0, 0.0);
    }
    MSM_HIP_CHECK(hipMemcpyAsync(lam, dtheta, k * sizeof(double), hipMemcpyDeviceToDevice, stream()));
    hipLaunchKernelGGL(ss_emit_kernel, dim3((unsigned)ceil_div((size_t)n * k, 256)), dim3(256), 0, stream(), X, n, k, Yk);
    MSM_HIP_CHECK(hipGetLastError());
    if (dstate) *dstate = st;
    return MSM_OK;
}

size_t subspace_pin_doubles() { return 4 * (size_t)SB * SB + 4 * SB; }

size_t subspace_work_doubles(int n)
{
    return 4 * (size_t)n * SB + (2 * (size_t)ceil_div(n, SG_ROWS) + 3) * SB * SB + 4 * SB + 8 + (sizeof(SsState) + 7) / 8 + 8;
}

}  // namespace msm
