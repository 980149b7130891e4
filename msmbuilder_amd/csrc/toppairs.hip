// toppairs.hip -- device pieces of tICA._solve around the subspace iteration (reference:
// scipy.linalg.eigh(lhs, b=rhs, eigvals=(F-k, F-1)) at /root/reference/msmbuilder/decomposition/tica.py:188-194).
//
//   potrf_upper_device      B = U^T U, blocked, one launch per 32-row block (rocSOLVER's dpotrf is ~40 launches per block
//                           at this size: 1.3 ms at n = 512, all of it launch latency)
//   pair_residual_device    max_i |(C y_j - lambda_j y_j)_i| on the reduced matrix: the caller falls back to LAPACK when
//                           a returned pair is not an eigenpair
// (Round 3 also kept a direct route here -- multisection + inverse iteration on a device tridiagonalisation, Householder
//  back-transform; removed in round 4: the solve has the subspace iteration, LAPACK on the reduced matrix behind it, and
//  rocSOLVER from 1,536 features.)
//
// All launches go to stream(); nothing synchronises.
#include "common.h"

#include <algorithm>
#include <cfloat>

namespace msm {

// =====================================================================================================================
// Cholesky, upper factor of the row-major buffer: B[j][i] (i >= j) <- U[j][i], B = U^T U.  The row-major UPPER triangle
// is the column-major LOWER one, i.e. exactly what rocsolver_dpotrf(lower) leaves behind and what the dtrsm calls of
// eigsolve.hip read.  Left-looking by block rows of 32: the launch for block row J computes, per 32-column tile,
//     W = B[J, tile] - sum_{K < J} U[K, J]^T U[K, tile],   U[J, J] = chol(W[J, J]),   U[J, tile] = U[J, J]^-T W
// where every workgroup forms and factors the 32 x 32 diagonal block redundantly (identical arithmetic, no exchange).
// *info = 1-based index of the first non-positive pivot (LAPACK's convention), untouched on success.
// =====================================================================================================================
constexpr int CH_NB = 32;

// The diagonal block of a block row is factored by EVERY workgroup of the launch while workgroup 0 writes the factor back
// in place, so the others must not read the block where it is being overwritten: they take it from the row-major LOWER
// triangle (the mirror image, never written) and the diagonal itself from a copy saved before the first launch.
__global__ void potrf_save_diag_kernel(const double* __restrict__ B, int n, double* __restrict__ dsave, int* __restrict__ minidx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dsave[i] = B[(size_t)i * n + i];
    if (i == 0) *minidx = 0x7fffffff;
}

__device__ __forceinline__ double fast_recip(double q)
{
    double r = __builtin_amdgcn_rcp(q);
    r = fma(fma(-q, r, 1.0), r, r);
    r = fma(fma(-q, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double fast_rsqrt(double x)   // x > 0
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}

// Sum over the 64 lanes of a wavefront, result in every lane, on the DPP network (quad swaps, half-row and row mirrors,
// then the four row totals through v_readlane): ~25 instructions with short latencies, where six __shfl_xor levels are
// twelve ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror: every lane holds the sum of its row of 16
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// With `Winv` the launch also forms block row J of W = U^-T (lower triangular, row-major): workgroups nu .. nu + J take the
// tiles (J, I), I <= J, of  W[J, :] = U[J, J]^-T (I[J, :] - sum_{I <= K < J} U[K, J]^T W[K, :])  -- the same product and the same
// elimination with the identity as right-hand side, in the launches the factorisation makes anyway.  The reduction to
// standard form is then two GEMMs (C = W A W^T) and the back-transformation of an eigenvector one thin product (W^T y)
// instead of rocBLAS' dtrsm (two of ~14 launches each for the reduction, one for the vectors: 0.5 ms at n = 512).
__global__ __launch_bounds__(256) void potrf_blockrow_kernel(double* __restrict__ B, int n, int j0, int* __restrict__ minidx,
                                                             const double* __restrict__ dsave, double* __restrict__ Winv, int nu)
{
    // pW: the four waves' partial products ([wave][diagonal block | tile][32][33])
    __shared__ double pW[4][2][CH_NB][CH_NB + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const bool inv = (int)blockIdx.x >= nu;                    // a tile of U^-T
    const int ti = inv ? (int)blockIdx.x - nu : 0;             // ... its block column I
    const int c0 = inv ? CH_NB * ti : j0 + CH_NB * (int)blockIdx.x;
    const bool diag = blockIdx.x == 0;
    const double* Bsrc = inv ? Winv : B;                       // rows K of the product's right operand
    double* Bdst = inv ? Winv : B;
    const int stepB0 = inv ? ti * (CH_NB / 4) : 0;             // W[K, I] = 0 for K < I
    // ---- W = sum_K U[K, J]^T U[K, tile] on the fp64 matrix pipe (v_mfma_f64_16x16x4: 16 x 4 times 4 x 16), operands
    // straight from global memory: lane (r = lane & 15, g = lane >> 4) supplies U[kappa0 + g][j0 + 16 ta + r] as the A operand
    // of row tile ta and U[kappa0 + g][c0 + 16 tb + r] as the B operand of column tile tb -- 128-byte segments, no LDS.  The
    // steps of four rows kappa are dealt out over the four waves, eight steps (32 loads per lane) in flight; the diagonal
    // product re-uses the A operands.  (Round 3 staged 32 x 32 chunks through LDS and multiplied on the VALU: 5 us per
    // chunk and wave, 20 us of the 44 us the last block row took.)
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    f64x4 aD[2][2], aT[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) aD[a][b][q] = aT[a][b][q] = 0.0;
    {
        const int r16 = lane & 15, g4 = lane >> 4;
        const int nstep = j0 / 4;   // j0 is a multiple of CH_NB
        constexpr int UNR = 8;
        const bool inA0 = j0 + r16 < n, inA1 = j0 + 16 + r16 < n;
        const bool inB0 = !diag && c0 + r16 < n, inB1 = !diag && c0 + 16 + r16 < n;
        for (int s0 = wave; s0 < nstep; s0 += 4 * UNR) {
            double va0[UNR], va1[UNR], vb0[UNR], vb1[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int st = s0 + 4 * u;
                const bool ok = st < nstep;
                const size_t ro = (size_t)(4 * (ok ? st : 0) + g4) * n;
                const bool okb = ok && st >= stepB0;
                va0[u] = (ok && inA0) ? B[ro + j0 + r16] : 0.0;
                va1[u] = (ok && inA1) ? B[ro + j0 + 16 + r16] : 0.0;
                vb0[u] = (okb && inB0) ? Bsrc[ro + c0 + r16] : 0.0;
                vb1[u] = (okb && inB1) ? Bsrc[ro + c0 + 16 + r16] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                aD[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va0[u], va0[u], aD[0][0], 0, 0, 0);
                aD[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va0[u], va1[u], aD[0][1], 0, 0, 0);
                aD[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1[u], va0[u], aD[1][0], 0, 0, 0);
                aD[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1[u], va1[u], aD[1][1], 0, 0, 0);
                if (!diag) {
                    aT[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va0[u], vb0[u], aT[0][0], 0, 0, 0);
                    aT[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va0[u], vb1[u], aT[0][1], 0, 0, 0);
                    aT[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1[u], vb0[u], aT[1][0], 0, 0, 0);
                    aT[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va1[u], vb1[u], aT[1][1], 0, 0, 0);
                }
            }
        }
        // partial products of the four waves -> pW[wave][0 / 1] (C/D layout: column = lane & 15, row = (lane >> 4) + 4 q), then
        // summed while the blocks are formed
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pW[wave][0][16 * a + g4 + 4 * q][16 * b + r16] = aD[a][b][q];
                    pW[wave][1][16 * a + g4 + 4 * q][16 * b + r16] = aT[a][b][q];
                }
    }
    __syncthreads();
    // ---- right-looking elimination of the diagonal block and of this workgroup's tile with UNSCALED pivot rows: step p
    // subtracts sD[p][r] sD[p][c] / piv_p (and sD[p][r] sT[p][c] / piv_p) from the rows r > p; the rows are scaled by
    // piv_p^-1/2 afterwards, all at once: U[p][c] = sD[p][c] / sqrt(piv_p).  Every thread keeps its 2 x 2 patch of both
    // blocks in registers; per step only the owners of row p publish it (with 1 / piv_p), one barrier per step.
    __shared__ double rowD[2][CH_NB], rowT[2][CH_NB], rinv[2];
    double eD[2][2], eT[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = 2 * ty + a, c = 2 * tx + b;
            const bool in = j0 + r < n && j0 + c < n;
            const double sumD = (pW[0][0][r][c] + pW[1][0][r][c]) + (pW[2][0][r][c] + pW[3][0][r][c]);
            const double sumT = (pW[0][1][r][c] + pW[1][1][r][c]) + (pW[2][1][r][c] + pW[3][1][r][c]);
            // rows / columns beyond n: identity, so the factorisation below needs no special cases
            const double orig = !in ? 0.0 : (r == c ? dsave[j0 + r] : B[(size_t)(j0 + (r > c ? r : c)) * n + j0 + (r > c ? c : r)]);
            eD[a][b] = in ? orig - sumD : (r == c ? 1.0 : 0.0);
            const bool inT = !diag && j0 + r < n && c0 + c < n;
            const double rhs = !inT ? 0.0 : inv ? ((c0 == j0 && r == c) ? 1.0 : 0.0) : B[(size_t)(j0 + r) * n + c0 + c];
            eT[a][b] = inT ? rhs - sumT : 0.0;
        }
    double pivsave[2] = {1.0, 1.0};   // the pivots of this thread's two rows (valid in every thread of the row)
    // (Fully unrolled, branch-free: the step's eight LDS reads are issued together and waited for once.  Round 3's loop
    //  kept p in a register and branched on r > p / c >= r: five dependent LDS round trips per step, 0.47 us a step --
    //  15 of the 22 us of a launch.)
#pragma unroll
    for (int p = 0; p < CH_NB; ++p) {
        const int buf = p & 1;
        if (ty == (p >> 1)) {   // owners of row p: publish it as it stands (all earlier steps applied)
            rowD[buf][2 * tx] = eD[p & 1][0];
            rowD[buf][2 * tx + 1] = eD[p & 1][1];
            rowT[buf][2 * tx] = eT[p & 1][0];
            rowT[buf][2 * tx + 1] = eT[p & 1][1];
            if (tx == (p >> 1)) rinv[buf] = fast_recip(eD[p & 1][p & 1]);   // the thread holding (p, p)
        }
        __syncthreads();
        const double piv_inv = rinv[buf], dpp = rowD[buf][p];
        const double rD[2] = {rowD[buf][2 * ty], rowD[buf][2 * ty + 1]};
        const double cD[2] = {rowD[buf][2 * tx], rowD[buf][2 * tx + 1]};
        const double cT[2] = {rowT[buf][2 * tx], rowT[buf][2 * tx + 1]};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int r = 2 * ty + a;
            pivsave[a] = (r == p) ? dpp : pivsave[a];
            const bool act = r > p;   // finished rows keep their values whatever the pivot row holds (select, not 0 * x)
            const double f = rD[a] * piv_inv;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // (entries left of the diagonal are updated too: nothing reads them)
                const double nd = fma(-f, cD[b], eD[a][b]), nt = fma(-f, cT[b], eT[a][b]);
                eD[a][b] = act ? nd : eD[a][b];
                eT[a][b] = act ? nt : eT[a][b];
            }
        }
        // (double-buffered rows: the owners of row p + 1 may publish while others still read row p)
    }
    if (diag && tx == ty) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
            if (j0 + 2 * ty + a < n && !(eD[a][a] > 0.0)) atomicMin(minidx, j0 + 2 * ty + a + 1);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = 2 * ty + a;
        if (j0 + r >= n) continue;
        const double ui = fast_rsqrt(pivsave[a]);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = 2 * tx + b;
            if (diag) {
                if (c >= r && j0 + c < n) B[(size_t)(j0 + r) * n + j0 + c] = eD[a][b] * ui;
            } else if (c0 + c < n) {
                Bdst[(size_t)(j0 + r) * n + c0 + c] = eT[a][b] * ui;
            }
        }
    }
}

// *info = LAPACK's info: 0, or the 1-based index of the first non-positive pivot (the running minimum over the launches)
__global__ void potrf_info_kernel(const int* __restrict__ minidx, int* __restrict__ info)
{
    if (threadIdx.x == 0) *info = *minidx == 0x7fffffff ? 0 : *minidx;
}

int potrf_upper_device(double* B, int n, int* dinfo, double* Winv)
{
    const int nb = (int)ceil_div(n, CH_NB);
    DevBuf& sv = pool(PS_PAR);   // n doubles + 1 int of scratch; every entry point that reaches here synchronises before it returns
    int rc = sv.reserve((size_t)n * sizeof(double) + 16);
    if (rc) return rc;
    double* dsave = sv.as<double>();
    int* minidx = reinterpret_cast<int*>(dsave + n);
    hipLaunchKernelGGL(potrf_save_diag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(), B, n, dsave, minidx);
    if (Winv) MSM_HIP_CHECK(hipMemsetAsync(Winv, 0, (size_t)n * n * sizeof(double), stream()));   // the GEMMs read the whole square
    for (int J = 0; J < nb; ++J)
        hipLaunchKernelGGL(potrf_blockrow_kernel, dim3(nb - J + (Winv ? J + 1 : 0)), dim3(256), 0, stream(), B, n, J * CH_NB, minidx, dsave,
                           Winv, nb - J);
    hipLaunchKernelGGL(potrf_info_kernel, dim3(1), dim3(64), 0, stream(), minidx, dinfo);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// V[v][i] = sum_{j >= i} W[j][i] Y[v][j]: the vectors U^-1 y = W^T y of the original problem (k vectors, rows of Y / V).
// One thread per component i (coalesced over i for every j), the vector held in LDS, 16 loads in flight.
__global__ __launch_bounds__(256) void winv_back_kernel(const double* __restrict__ W, int n, const double* __restrict__ Y, double* __restrict__ V)
{
    __shared__ double sy[1024];
    const int v = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    for (int j = tid; j < n; j += 256) sy[j] = Y[(size_t)v * n + j];
    __syncthreads();
    if (i >= n) return;
    const int jb = (blockIdx.x * 256) & ~15;   // uniform start for the workgroup: rows above the diagonal hold zeros
    double acc = 0.0;
    int j = jb;
    for (; j + 16 <= n; j += 16) {
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = W[(size_t)(j + u) * n + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma(w[u], sy[j + u], acc);
    }
    for (; j < n; ++j) acc = fma(W[(size_t)j * n + i], sy[j], acc);
    V[(size_t)v * n + i] = acc;
}

// V (k rows of n) <- W^T Y, W = U^-T from potrf_upper_device (n <= 1024); V may not alias Y
int winv_back_device(const double* W, int n, const double* Y, int k, double* V)
{
    hipLaunchKernelGGL(winv_back_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream(), W, n, Y, V);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

constexpr int TRI_MAXN = 1024;   // rows of the reduced matrix the residual kernel stages

// res[j] = max_r |(C y_j)_r - lambda_j y_j[r]|, res[k + j] = | ||y_j||^2 - 1 |, res[2k + i k + j] = y_i . y_j   (C symmetric
// n x n, Y [k][n]; res[0 .. 2k) zeroed by the caller: the maxima are merged with atomicMax on the bits of non-negative
// doubles).  Grid (ceil(n / 64) + 1, k): workgroup (b, j) forms rows 64 b .. 64 b + 63 of C y_j, the four waves splitting
// the columns (16 loads in flight per lane: round 3's one-workgroup-per-vector kernel made 128 dependent trips, 48 us);
// the last workgroup of a column forms row j of the Gram matrix of the vectors (mutual orthogonality).
__global__ __launch_bounds__(256) void pair_residual_kernel(const double* __restrict__ Cm, int n, const double* __restrict__ Y,
                                                            const double* __restrict__ vals, int k, double* __restrict__ res)
{
    __shared__ double sy[TRI_MAXN];
    __shared__ double part[4][64];
    const int j = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + 63) / 64;
    for (int i = tid; i < n; i += 256) sy[i] = Y[(size_t)j * n + i];
    __syncthreads();
    if ((int)blockIdx.x == nb) {
        // Gram row j: wave w takes the vectors i = w, w + 4, ...
        for (int i = wave; i < k; i += 4) {
            double acc = 0.0;
            for (int c = lane; c < n; c += 64) acc = fma(Y[(size_t)i * n + c], sy[c], acc);
            acc = wave_sum_f64(acc);
            if (lane == 0) res[2 * k + (size_t)j * k + i] = acc;
        }
        return;
    }
    const int r = 64 * blockIdx.x + lane;
    const int rc = r < n ? r : n - 1;
    const int q = (n + 3) / 4, c0 = wave * q, c1 = (c0 + q) < n ? (c0 + q) : n;
    double acc = 0.0;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {   // symmetric: column r read as a row, coalesced
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = Cm[(size_t)(c + u) * n + rc];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma(v[u], sy[c + u], acc);
    }
    for (; c < c1; ++c) acc = fma(Cm[(size_t)c * n + rc], sy[c], acc);
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        const double cy = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        double mx = r < n ? fabs(cy - vals[j] * sy[rc]) : 0.0;
        if (!(mx == mx)) mx = INFINITY;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(res + j), (unsigned long long)__double_as_longlong(mx));
    }
}

// res[k + j] = | y_j . y_j - 1 | from the Gram matrix the kernel above left at res[2k ...]
__global__ void pair_norms_kernel(int k, double* __restrict__ res)
{
    const int j = threadIdx.x;
    if (j < k) {
        const double nn = res[2 * k + (size_t)j * k + j];
        res[k + j] = (nn == nn) ? fabs(nn - 1.0) : INFINITY;
    }
}

// res: 2k + k^2 doubles (residual maxima, norm defects, Gram matrix of the vectors)
int pair_residual_device(const double* Cm, int n, const double* Y, const double* vals, int k, double* res)
{
    MSM_HIP_CHECK(hipMemsetAsync(res, 0, 2 * (size_t)k * sizeof(double), stream()));
    hipLaunchKernelGGL(pair_residual_kernel, dim3((unsigned)ceil_div(n, 64) + 1, (unsigned)k), dim3(256), 0, stream(), Cm, n, Y, vals, k, res);
    hipLaunchKernelGGL(pair_norms_kernel, dim3(1), dim3(64), 0, stream(), k, res);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

/* B = U^T U in place on the row-major UPPER triangle (== LAPACK dpotrf 'L' on the column-major view); *info as LAPACK's.
 * (Building block of the tICA solve; exported for the tests.) */
int msm_potrf(double* B, msm_idx_t n, int* info, int on_device)
{
    if (!B || !info || n < 1 || n > 32768) return fail(MSM_ERR_INVALID, "msm_potrf: bad argument");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf& buf = pool(PS_W);
    const size_t nn = (size_t)n * n;
    int rc = buf.reserve(nn * sizeof(double) + 16);
    if (rc) return rc;
    double* dB = on_device ? B : buf.as<double>();
    int* dinfo = reinterpret_cast<int*>(buf.as<char>() + nn * sizeof(double));
    if (!on_device) MSM_HIP_CHECK(hipMemcpyAsync(dB, B, nn * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(dinfo, 0, sizeof(int), stream()));
    if ((rc = potrf_upper_device(dB, (int)n, dinfo, nullptr))) return rc;
    if (!on_device) MSM_HIP_CHECK(hipMemcpyAsync(B, dB, nn * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(info, dinfo, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"
