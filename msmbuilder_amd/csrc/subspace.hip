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
// a matrix that is not a tICA matrix -- so the method can only cost time, never accuracy.  When n_components reaches
// into the flat part of the spectrum the damped interval is narrowed to the noise bulk ("hard" mode, in the loop below).
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
__global__ __launch_bounds__(SS_NT) void ss_product_kernel(const double* __restrict__ Cm, int n, const double* __restrict__ X,
                                                           const double* P, double* out, double alpha, double beta, double gamma)
{
    extern __shared__ double ssm[];
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
                                                          int* __restrict__ flag)
{
    __shared__ double sg[SB][SB + 1];
    __shared__ double sinv[SB];
    const int tid = threadIdx.x;
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
        hipLaunchKernelGGL(ss_product_kernel, gprod, dim3(SS_NT), SS_PRODUCT_LDS, stream(), Cm, n, Xin, P, out, a, b, c);
    };
    // one round of Cholesky QR: orthonormal to ~ cond(X)^2 eps, which the Rayleigh-Ritz step below absorbs by solving
    // the small GENERALIZED problem H s = theta G s with G = X^T X
    auto cholqr = [&](double* Q) {
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), Q, Q, n, part);
        hipLaunchKernelGGL(ss_cholqr_kernel, gqr, dim3(SS_NT), 0, stream(), part, nparts, Q, n, dflag);
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
    double lo = lower;   // lower end of the damped interval: the caller's bound, tightened in "hard" mode (below)
    auto run_filter = [&](double cut, double top, int deg, int chunk) {
        const double e = 0.5 * (cut - lo), cen = 0.5 * (cut + lo);
        const double sigma1 = e / (top - cen);
        const int nchunk = (deg + chunk - 1) / chunk;
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
        run_filter(first_cut, first_top, degree, degree);
        filters = 1;
        prev_res = 1.0;   // the residual scale of a random block (the spectrum's width): lets the first measured residual size the next filter
    }
    bool hard = false, hard_adjusted = false;
    int hard_outer = 0;
    for (int outer = 0; outer <= max_outer + 12; ++outer) {
        // ---- Rayleigh-Ritz on span(X)
        product(X, nullptr, W, 1.0, 0.0, 0.0);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, X, n, part);
        hipLaunchKernelGGL(ss_gram_kernel, dim3(nparts), dim3(SS_NT), 0, stream(), X, W, n, part2);
        hipLaunchKernelGGL(ss_gram_sum_kernel, dim3(2), dim3(SS_NT), 0, stream(), part, nparts, GH);
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
        hipLaunchKernelGGL(ss_rotate_kernel, grot, dim3(SS_NT), 0, stream(), X, W, n, dS);
        hipLaunchKernelGGL(ss_residual_kernel, dim3(1), dim3(SS_NT), 0, stream(), X, W, n, dtheta, dres);
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
        const double tol_abs = tol * std::max(1.0, std::fabs(w[0]));
        // ---- "hard" mode (round 5).  When n_components reaches into the flat part of the spectrum (fewer slow processes than
        // components: the k-th eigenvalue sits at the edge of the noise bulk, percents of the bulk's width above the block's
        // last Ritz value) a filter that damps [-1, cut] gains next to nothing per degree -- the wanted gap is measured
        // against the whole interval -- and the iteration used to give up after three rounds: LAPACK on the host, 7.6 ms at
        // n = 512 and 24 ms at n = 1,024 against 1-2 ms (scripts/solveprobe.py).  The bulk, however, is narrow: with the
        // damped interval shrunk to [just below the bulk, cut] the same gap is a sizeable fraction of the interval and a
        // few dozen degrees suffice.  The lower end is not known, so it is GUESSED (the noise bulk of a reduced tICA matrix
        // is roughly symmetric about 0) and MAINTAINED: a Ritz value always lies above lambda_min, and directions below
        // the guessed bound are amplified by the filter and show up as Ritz values near it -- the bound then moves below
        // them, and the next filter damps them again.  The slow processes above the bulk are amplified (2 x)^degree times
        // more than the wanted edge, x = their distance in half-widths of the narrow interval (~30): the filter is cut into
        // chunks of a few degrees, a Cholesky QR after each, so that no chunk stretches the block by more than ~1e7.
        // Nothing else changes: same convergence test on the ORIGINAL matrix, same verification by the caller, and when
        // this stalls too the reduced matrix still goes to LAPACK.  Measured (scripts/solveprobe.py, k = 10 with 2 or 6
        // slow processes): the solve 7.6 -> 3.1 ms at n = 512, 24 -> 4.9 ms at n = 1,024, two or three hard rounds of
        // 40-64 degrees; the rate per degree is ~0.22 where acosh of the Ritz gap says 0.6 (the 33rd eigenvalue sits right
        // under the block) -- a restart-free recurrence or deflating the converged pairs changes the number of
        // re-orthonormalisations, not that rate (both tried).
        if (!hard) {
            const double cut0 = w[SB - 1], wk = w[k - 1];
            bool slow = false;
            if (cut0 > lo && wk > cut0) {
                const double gap0 = (wk - cut0) / (cut0 - lo);
                slow = std::acosh(1.0 + 2.0 * gap0) * (3.0 * degree) < std::log(1e3);   // a maximal easy filter would not gain 1e3
            }
            const bool stalled = outer >= 2 && !(rmax < 1e-2 * prev_res);
            if ((slow || stalled) && n < 256) return MSM_OK;   // (LAPACK on so small a matrix beats a hundred filter degrees: 0.6 ms at n = 128)
            if (slow || stalled) {
                hard = true;
                const double guess = cut0 > 0.0 ? -1.6 * cut0 : cut0 - 3.0 * (wk - cut0);
                lo = std::max(lower, guess);
                hard_adjusted = true;
            } else if (outer >= max_outer) {
                return MSM_OK;
            }
        }
        if (hard) {
            if (++hard_outer > 12) return MSM_OK;
            const double wk = w[k - 1], wmin = w[SB - 1];
            double mid = 0.5 * (lo + wk);
            if (wmin < mid) {   // directions from below the bound have entered the block: move the bound below them
                lo = std::max(lower, wmin - 0.25 * (wk - wmin));
                mid = 0.5 * (lo + wk);
                hard_adjusted = true;
            }
            if (!hard_adjusted && hard_outer >= 2 && !(rmax < 0.2 * prev_res)) return MSM_OK;   // stalled for good
            hard_adjusted = false;
            int ngood = 0;
            double cut = wk;
            for (int j = 0; j < SB; ++j)
                if (w[j] > mid) {
                    ++ngood;
                    cut = std::min(cut, w[j]);
                }
            const double top = w[0];
            if (ngood < k + 2 || !(cut > lo) || !(wk > cut) || !(top > cut)) return MSM_OK;
            const double e = 0.5 * (cut - lo), cen = 0.5 * (cut + lo);
            const double rate = std::acosh(1.0 + 2.0 * (wk - cut) / (cut - lo));                 // per degree, for the k-th pair
            // degrees still needed: at the rate the last hard filter was measured to deliver, else at 0.6 of the theoretical one
            // (the Ritz values that define cut and the gap are themselves still converging)
            double per_degree = 0.6 * rate;
            if (hard_outer >= 2 && rmax < prev_res) per_degree = std::min(rate, std::log(prev_res / rmax) / last_deg);
            const double need = std::log(std::max(rmax, tol_abs) / (0.3 * tol_abs)) / std::max(per_degree, 1e-3);
            const int deg = (int)std::min(64.0, std::max((double)degree, std::ceil(need)));
            const double xtop = (top - cen) / e;
            const int chunk = (int)std::min((double)degree, std::max(2.0, std::floor(std::log(1e7) / std::log(2.0 * std::max(xtop, 1.0) + 1.0))));
            last_deg = deg;
            prev_res = rmax;
            run_filter(cut, top, deg, chunk);
            continue;
        }
        // Degree of the next filter: `degree`, or -- once a filter's gain is known -- what the remaining distance to the
        // tolerance asks for at that rate per degree (up to 3 x degree): one longer filter instead of two filters with a
        // Rayleigh-Ritz round trip (two host synchronisations, ~0.2 ms) between them.
        int deg = degree;
        if ((outer >= 1 || filters) && prev_res < INFINITY && rmax < prev_res) {
            const double rate = std::log(rmax / prev_res) / last_deg;        // < 0, per degree
            const double need = std::log(0.3 * tol_abs / rmax) / rate;       // degrees still needed, with a margin
            if (need > degree) deg = (int)std::min(3.0 * degree, std::ceil(need));
        }
        last_deg = deg;
        prev_res = rmax;
        // ---- filter: damp [lower, cut], cut = the smallest Ritz value of the block; scaled so that theta_1 stays O(1)
        const double cut = w[SB - 1], top = w[0];
        if (!(cut > lo) || !(top > cut)) return MSM_OK;
        run_filter(cut, top, deg, degree);
    }
    return MSM_OK;
}

size_t subspace_pin_doubles() { return 4 * (size_t)SB * SB + 4 * SB; }

size_t subspace_work_doubles(int n)
{
    return 4 * (size_t)n * SB + (2 * (size_t)ceil_div(n, SG_ROWS) + 3) * SB * SB + 4 * SB + 8;
}

}  // namespace msm
