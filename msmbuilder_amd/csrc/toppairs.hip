// toppairs.hip -- the device-resident tail of tICA._solve: everything between the finalised moments and the k returned
// eigenpairs without LAPACK and without a host round trip (reference: scipy.linalg.eigh(lhs, b=rhs,
// eigvals=(F-k, F-1)) at /root/reference/msmbuilder/decomposition/tica.py:188-194).
//
//   potrf_upper_device      B = U^T U, blocked, one launch per 32-row block (rocSOLVER's dpotrf is ~40 launches per block
//                           at this size: 1.3 ms at n = 512, all of it launch latency)
//   tri_topk_device         the k largest eigenpairs of the symmetric tridiagonal (d, e) that sytrd.hip produces:
//                           multisection on Sturm counts (one workgroup per eigenvalue, 256 shifts per sweep), then
//                           inverse iteration on the pivoted LU of T - lambda I with modified Gram-Schmidt inside clusters
//                           (LAPACK's dstebz / dstein recipe, restated for one lane per vector)
//   apply_q_device          y_j = Q s_j for the Householder product Q = H_0 H_1 ... H_{n-3} of sytrd.hip, formed ROW BY ROW:
//                           row r of Q is e_r^T H_0 H_1 ..., a chain of n - 2 dot / axpy steps that needs no exchange
//                           between rows, so n independent wavefronts stream the reflectors once
//   pair_residual_device    max_i |(C y_j - lambda_j y_j)_i| on the reduced matrix: the caller falls back to LAPACK when
//                           the cooperative tridiagonalisation, the inverse iteration or the back-transform went wrong
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

__global__ __launch_bounds__(256) void potrf_blockrow_kernel(double* __restrict__ B, int n, int j0, int* __restrict__ minidx,
                                                             const double* __restrict__ dsave)
{
    // pW: wave-private staging of a 32-row chunk of the panel above
    // ([wave][A | B][32][33]) and, afterwards, the four waves' partial products
    __shared__ double pW[4][2][CH_NB][CH_NB + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;   // the lane's 4 x 4 patch of the 32 x 32 product: rows 4 ly.., columns 4 lx..
    const int c0 = j0 + CH_NB * blockIdx.x;
    const bool diag = blockIdx.x == 0;
    // ---- W = sum_K U[K, J]^T U[K, tile]: the K chunks are dealt out over the four waves (no barrier inside the loop: a
    // wave stages and multiplies its own chunk), each lane accumulating a 4 x 4 patch of both products
    double accD[4][4], accT[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) accD[a][b] = accT[a][b] = 0.0;
    const int nchunk = j0 / CH_NB;
    double pa[16], pb[16];   // a 32 x 32 chunk = 1024 values = 16 per lane
    auto fetch = [&](int ch) {
        const int kk = ch * CH_NB;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int q = lane + 64 * s4, r = q >> 5, c = q & 31;
            pa[s4] = (j0 + c < n) ? B[(size_t)(kk + r) * n + j0 + c] : 0.0;
            pb[s4] = (!diag && c0 + c < n) ? B[(size_t)(kk + r) * n + c0 + c] : 0.0;
        }
    };
    if (wave < nchunk) fetch(wave);
    for (int ch = wave; ch < nchunk; ch += 4) {
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const int q = lane + 64 * s4, r = q >> 5, c = q & 31;
            pW[wave][0][r][c] = pa[s4];
            pW[wave][1][r][c] = pb[s4];
        }
        if (ch + 4 < nchunk) fetch(ch + 4);   // the next chunk's loads fly while this one is multiplied
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int q = 0; q < CH_NB; ++q) {
            double av[4], dv[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                av[a] = pW[wave][0][q][4 * ly + a];
                dv[a] = pW[wave][0][q][4 * lx + a];
                bv[a] = pW[wave][1][q][4 * lx + a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    accD[a][b] += av[a] * dv[b];
                    if (!diag) accT[a][b] += av[a] * bv[b];
                }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // partial products of the four waves -> pW[wave][0 / 1], then summed while the blocks are formed
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            pW[wave][0][4 * ly + a][4 * lx + b] = accD[a][b];
            pW[wave][1][4 * ly + a][4 * lx + b] = accT[a][b];
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
            eT[a][b] = (!diag && j0 + r < n && c0 + c < n) ? B[(size_t)(j0 + r) * n + c0 + c] - sumT : 0.0;
        }
    double pivsave[2] = {1.0, 1.0};   // the pivots of this thread's two rows (valid in every thread of the row)
    for (int p = 0; p < CH_NB; ++p) {
        const int buf = p & 1;
        if (ty == (p >> 1)) {   // owners of row p: publish it as it stands (all earlier steps applied)
            const int a = p & 1;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                rowD[buf][2 * tx + b] = eD[a][b];
                rowT[buf][2 * tx + b] = eT[a][b];
            }
            if (tx == (p >> 1)) rinv[buf] = fast_recip(eD[a][p & 1]);   // the thread holding (p, p)
        }
        __syncthreads();
        const double piv_inv = rinv[buf];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int r = 2 * ty + a;
            if (r == p) pivsave[a] = rowD[buf][p];
            if (r > p) {
                const double f = rowD[buf][r] * piv_inv;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int c = 2 * tx + b;
                    if (c >= r) eD[a][b] -= f * rowD[buf][c];
                    eT[a][b] -= f * rowT[buf][c];
                }
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
                B[(size_t)(j0 + r) * n + c0 + c] = eT[a][b] * ui;
            }
        }
    }
}

// *info = LAPACK's info: 0, or the 1-based index of the first non-positive pivot (the running minimum over the launches)
__global__ void potrf_info_kernel(const int* __restrict__ minidx, int* __restrict__ info)
{
    if (threadIdx.x == 0) *info = *minidx == 0x7fffffff ? 0 : *minidx;
}

int potrf_upper_device(double* B, int n, int* dinfo)
{
    const int nb = (int)ceil_div(n, CH_NB);
    DevBuf& sv = pool(PS_PAR);   // n doubles + 1 int of scratch; every entry point that reaches here synchronises before it returns
    int rc = sv.reserve((size_t)n * sizeof(double) + 16);
    if (rc) return rc;
    double* dsave = sv.as<double>();
    int* minidx = reinterpret_cast<int*>(dsave + n);
    hipLaunchKernelGGL(potrf_save_diag_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(), B, n, dsave, minidx);
    for (int J = 0; J < nb; ++J)
        hipLaunchKernelGGL(potrf_blockrow_kernel, dim3(nb - J), dim3(256), 0, stream(), B, n, J * CH_NB, minidx, dsave);
    hipLaunchKernelGGL(potrf_info_kernel, dim3(1), dim3(64), 0, stream(), minidx, dinfo);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// =====================================================================================================================
// k largest eigenvalues of the symmetric tridiagonal T = tridiag(e, d, e), n <= 1024.
// Workgroup j brackets the j-th largest eigenvalue by multisection: every sweep its 256 threads evaluate the Sturm count
// (number of eigenvalues below the shift: the sign count of q_i = d_i - x - e_{i-1}^2 / q_{i-1}, LAPACK dlaebz's form
// with its pivmin guard) at 256 interior points of the bracket and keep the sub-interval where the count passes n - 1 - j;
// seven sweeps take the Gershgorin interval down to rounding.  The division is a reciprocal with one Newton step: relative
// error 2^-48 per step acts like a 2^-48 relative perturbation of e^2 -- far inside the tolerance of the count.
// =====================================================================================================================
constexpr int TRI_MAXN = 1024;
constexpr int TRI_P = 256;   // shifts per sweep: one wavefront per SIMD keeps the dependent chain latency-bound, not issue-bound

__device__ __forceinline__ double fast_recip1(double q)   // 2^-48: enough for a Sturm count
{
    const double r = __builtin_amdgcn_rcp(q);
    return fma(fma(-q, r, 1.0), r, r);
}

__global__ __launch_bounds__(TRI_P) void tri_topk_values_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                                int n, double* __restrict__ vals)
{
    __shared__ double sd[TRI_MAXN], se2[TRI_MAXN];
    __shared__ double rmin[TRI_P / 64], rmax[TRI_P / 64], rtn[TRI_P / 64], re2[TRI_P / 64];
    __shared__ int first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = n - 1 - (int)blockIdx.x;   // ascending index of this workgroup's eigenvalue
    double lo_g = DBL_MAX, hi_g = -DBL_MAX, tn = 0.0, e2m = 0.0;
    for (int i = tid; i < n; i += TRI_P) {
        const double di = d[i];
        const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i + 1 < n ? fabs(e[i]) : 0.0;
        sd[i] = di;
        se2[i] = er * er;   // se2[i] = e_i^2 couples rows i and i + 1
        lo_g = fmin(lo_g, di - el - er);
        hi_g = fmax(hi_g, di + el + er);
        tn = fmax(tn, fmax(fabs(di), er));
        e2m = fmax(e2m, er * er);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        lo_g = fmin(lo_g, __shfl_xor(lo_g, m, 64));
        hi_g = fmax(hi_g, __shfl_xor(hi_g, m, 64));
        tn = fmax(tn, __shfl_xor(tn, m, 64));
        e2m = fmax(e2m, __shfl_xor(e2m, m, 64));
    }
    if (lane == 0) {
        rmin[wave] = lo_g;
        rmax[wave] = hi_g;
        rtn[wave] = tn;
        re2[wave] = e2m;
    }
    __syncthreads();
    for (int w = 0; w < TRI_P / 64; ++w) {
        lo_g = fmin(lo_g, rmin[w]);
        hi_g = fmax(hi_g, rmax[w]);
        tn = fmax(tn, rtn[w]);
        e2m = fmax(e2m, re2[w]);
    }
    const double pivmin = 1e-300 * fmax(1.0, e2m);
    const double pad = 2.0 * tn * DBL_EPSILON * n + 2.0 * pivmin;
    double lo = lo_g - pad, hi = hi_g + pad;   // count(lo) = 0 <= idx < n = count(hi)
    for (int sweep = 0; sweep < 24; ++sweep) {
        if (tid == 0) first = TRI_P;
        __syncthreads();
        const double w = hi - lo;
        const double x = lo + w * ((double)(tid + 1) / (double)(TRI_P + 1));
        int cnt = 0;
        double q = sd[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q < 0.0;
        for (int i = 1; i < n; ++i) {
            q = (sd[i] - x) - se2[i - 1] * fast_recip1(q);
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0;
        }
        if (cnt > idx) atomicMin(&first, tid);
        __syncthreads();
        const int t0 = first;
        const double nhi = t0 < TRI_P ? lo + w * ((double)(t0 + 1) / (double)(TRI_P + 1)) : hi;
        const double nlo = t0 > 0 ? lo + w * ((double)t0 / (double)(TRI_P + 1)) : lo;
        __syncthreads();   // everyone has read `first`
        lo = nlo;
        hi = nhi;
        if (hi - lo <= 2.0 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
    }
    if (tid == 0) vals[blockIdx.x] = 0.5 * (lo + hi);
}

// =====================================================================================================================
// Eigenvectors of T for the k eigenvalues above: inverse iteration, one LANE per vector (LAPACK dstein: dlagtf's LU of
// T - lambda I with partial pivoting, dlagts' solve with tiny pivots replaced by +-tol, a pseudo-random start, modified
// Gram-Schmidt against the earlier vectors of a cluster |lambda_i - lambda_j| < 1e-3 ||T||_1 after every solve).
// The factors of a vector (a^-1, b, c, dd: 4 n doubles, the pivot flags, and the iterate) live in LDS; a chunk of G
// vectors is iterated in lockstep by G lanes, the Gram-Schmidt and normalisation steps use the whole workgroup.
// Without clusters the vectors are independent and every workgroup takes one; with a cluster workgroup 0 takes them all.
// =====================================================================================================================
constexpr int TV_NT = 256;
constexpr int TV_ITERS = 2;   // the first solve already amplifies the eigenvector by ~1 / eps; the pairs are verified afterwards

__device__ __forceinline__ double tv_block_sum(double x, double* red, int tid)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double tv_uniform(unsigned j, unsigned i)   // deterministic start vector in (-1, 1)
{
    unsigned h = j * 0x9E3779B1u + i * 0x85EBCA77u + 0x165667B1u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return ((double)h + 0.5) * (2.0 / 4294967296.0) - 1.0;
}

__global__ __launch_bounds__(TV_NT) void tri_topk_vectors_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                                 int n, int k, int G, const double* __restrict__ vals,
                                                                 double* __restrict__ S /* [k][n] */)
{
    extern __shared__ double lds[];
    __shared__ double red[8];
    __shared__ double slam[64];
    __shared__ int scs[64];
    __shared__ int cross;
    const int tid = threadIdx.x;
    // per-vector LDS block: ainv[n] b[n] c[n] dd[n] x[n] + n pivot bytes (rounded to 8); T itself (d, e) behind the last one
    const size_t vstride = 5 * (size_t)n + (size_t)((n + 7) / 8);
    double* sd = lds + (size_t)G * vstride;
    double* se = sd + n;
    for (int i = tid; i < n; i += TV_NT) {
        sd[i] = d[i];
        se[i] = i + 1 < n ? e[i] : 0.0;
    }
    // ---- ||T||_1, cluster starts, separated eigenvalues (dstein: equal eigenvalues are pushed 10 eps apart)
    double cs = 0.0;
    for (int i = tid; i < n; i += TV_NT)
        cs = fmax(cs, fabs(d[i]) + (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) cs = fmax(cs, __shfl_xor(cs, m, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = cs;
    __syncthreads();
    const double onenrm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double ortol = 1e-3 * onenrm;
    const double ptol = fmax(10.0 * DBL_EPSILON * onenrm, 1e-300);
    if (tid == 0) {
        int cr = 0;
        for (int j = 0; j < k; ++j) {
            double l = vals[j];
            if (j > 0 && slam[j - 1] - l < 10.0 * DBL_EPSILON * fabs(l)) l = slam[j - 1] - 10.0 * DBL_EPSILON * fabs(l);
            slam[j] = l;
            scs[j] = (j > 0 && fabs(l - slam[j - 1]) < ortol) ? scs[j - 1] : j;
            if (scs[j] < j) cr = 1;
        }
        cross = cr;
    }
    __syncthreads();
    // no cluster anywhere (the usual case): the vectors are independent, one workgroup each.  Otherwise workgroup 0 takes
    // them all, in index order, G at a time (Gram-Schmidt needs the earlier members of a cluster).
    const bool serial = cross != 0;
    if (serial && blockIdx.x != 0) return;
    if (!serial) G = 1;
    const int nchunk = (k + G - 1) / G;
    for (int ch = serial ? 0 : (int)blockIdx.x; ch < nchunk; ch += serial ? 1 : (int)gridDim.x) {
        const int j0 = ch * G, g = min(G, k - j0);
        __syncthreads();
        // ---- factorisation: lane t < g, vector j0 + t (dlagtf)
        if (tid < g) {
            double* ainv = lds + (size_t)tid * vstride;
            double* b = ainv + n;
            double* c = b + n;
            double* dd = c + n;
            unsigned char* in = reinterpret_cast<unsigned char*>(dd + 2 * (size_t)n);
            const double lam = slam[j0 + tid];
            double ak = sd[0] - lam;                     // a[k], updated as the elimination proceeds
            double bk = se[0];                           // b[k]
            double scale1 = fabs(ak) + fabs(bk);
            for (int kx = 0; kx + 1 < n; ++kx) {
                const double ck = se[kx];
                double an = sd[kx + 1] - lam;            // a[k+1]
                double bn = se[kx + 1];                  // b[k+1] (0 behind the last coupling)
                const double scale2 = fabs(ck) + fabs(an) + fabs(bn);
                double ddk = 0.0, cmul;
                int piv = 0;
                if (ck == 0.0) {
                    cmul = 0.0;
                    scale1 = scale2;
                } else if (ak != 0.0 && fabs(ck) * scale1 <= fabs(ak) * scale2) {   // piv2 <= piv1: no interchange
                    cmul = ck * fast_recip(ak);
                    an -= cmul * bk;
                    scale1 = scale2;
                } else {                                   // interchange rows k and k + 1
                    piv = 1;
                    const double mult = ak * fast_recip(ck);
                    ak = ck;
                    const double temp = an;
                    an = bk - mult * temp;
                    if (kx + 2 < n) {
                        ddk = bn;
                        bn = -mult * ddk;
                    }
                    bk = temp;
                    cmul = mult;
                }
                ainv[kx] = ak;   // reciprocals are taken by the whole workgroup below
                b[kx] = bk;
                c[kx] = cmul;
                dd[kx] = ddk;
                in[kx] = (unsigned char)piv;
                ak = an;
                bk = bn;
            }
            ainv[n - 1] = ak;
            b[n - 1] = 0.0;
            c[n - 1] = 0.0;
            dd[n - 1] = 0.0;
            in[n - 1] = 0;
        }
        __syncthreads();
        for (int q = tid; q < g * n; q += TV_NT) {
            const int t = q / n, i = q - t * n;
            double* ainv = lds + (size_t)t * vstride;
            double a = ainv[i];
            if (fabs(a) < ptol) a = a < 0.0 ? -ptol : ptol;   // dlagts job = -1: perturb a tiny pivot
            ainv[i] = fast_recip(a);
            ainv[4 * (size_t)n + i] = tv_uniform((unsigned)(j0 + t), (unsigned)i);   // x
        }
        __syncthreads();
        for (int it = 0; it < TV_ITERS; ++it) {
            if (tid < g) {
                double* ainv = lds + (size_t)tid * vstride;
                const double* b = ainv + n;
                const double* c = b + n;
                const double* dd = c + n;
                double* x = ainv + 4 * (size_t)n;
                const unsigned char* in = reinterpret_cast<const unsigned char*>(dd + 2 * (size_t)n);
                // forward: y <- L^-1 P y.  Four steps per trip: the coefficients of the next four rows are loaded before the
                // dependent chain runs (one LDS round trip per four steps instead of one per step)
                double yp = x[0];
                int kx = 1;
                for (; kx + 3 < n; kx += 4) {
                    double cq[4], yq[4];
                    unsigned char iq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cq[u] = c[kx - 1 + u];
                        yq[u] = x[kx + u];
                        iq[u] = in[kx - 1 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        double ynew, out;
                        if (iq[u] == 0) {
                            ynew = yq[u] - cq[u] * yp;
                            out = yp;
                        } else {
                            out = yq[u];
                            ynew = yp - cq[u] * yq[u];
                        }
                        x[kx - 1 + u] = out;
                        yp = ynew;
                    }
                }
                for (; kx < n; ++kx) {
                    const double ck = c[kx - 1], yk = x[kx];
                    double ynew;
                    if (in[kx - 1] == 0) {
                        ynew = yk - ck * yp;
                        x[kx - 1] = yp;
                    } else {
                        x[kx - 1] = yk;
                        ynew = yp - ck * yk;
                    }
                    yp = ynew;
                }
                x[n - 1] = yp;
                // backward: x <- U^-1 y
                double x1 = 0.0, x2 = 0.0;
                kx = n - 1;
                for (; kx >= 3; kx -= 4) {
                    double bq[4], dq[4], aq[4], yq[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bq[u] = b[kx - u];
                        dq[u] = dd[kx - u];
                        aq[u] = ainv[kx - u];
                        yq[u] = x[kx - u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double xk = (yq[u] - bq[u] * x1 - dq[u] * x2) * aq[u];
                        x[kx - u] = xk;
                        x2 = x1;
                        x1 = xk;
                    }
                }
                for (; kx >= 0; --kx) {
                    const double xk = (x[kx] - b[kx] * x1 - dd[kx] * x2) * ainv[kx];
                    x[kx] = xk;
                    x2 = x1;
                    x1 = xk;
                }
            }
            __syncthreads();
            // Gram-Schmidt inside clusters (in index order) and normalisation, whole workgroup
            for (int jj = 0; jj < g; ++jj) {
                const int j = j0 + jj;
                double* xj = lds + (size_t)jj * vstride + 4 * (size_t)n;
                // guard against overflow of the raw iterate: scale by its largest entry first
                double mx = 0.0;
                for (int i = tid; i < n; i += TV_NT) mx = fmax(mx, fabs(xj[i]));
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
                __syncthreads();
                if ((tid & 63) == 0) red[4 + (tid >> 6)] = mx;
                __syncthreads();
                mx = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
                const double sc = mx > 0.0 ? 1.0 / mx : 1.0;
                for (int i = tid; i < n; i += TV_NT) xj[i] *= sc;
                __syncthreads();
                for (int i2 = scs[j]; i2 < j; ++i2) {
                    const double* xi = i2 >= j0 ? lds + (size_t)(i2 - j0) * vstride + 4 * (size_t)n : nullptr;
                    const double* gi = S + (size_t)i2 * n;
                    double part = 0.0;
                    for (int i = tid; i < n; i += TV_NT) part += xj[i] * (xi ? xi[i] : gi[i]);
                    const double dot = tv_block_sum(part, red, tid);
                    for (int i = tid; i < n; i += TV_NT) xj[i] -= dot * (xi ? xi[i] : gi[i]);
                    __syncthreads();
                }
                double part = 0.0;
                for (int i = tid; i < n; i += TV_NT) part += xj[i] * xj[i];
                const double nrm = sqrt(tv_block_sum(part, red, tid));
                const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
                for (int i = tid; i < n; i += TV_NT) xj[i] *= inv;
                __syncthreads();
            }
        }
        for (int q = tid; q < g * n; q += TV_NT) {
            const int t = q / n, i = q - t * n;
            S[(size_t)(j0 + t) * n + i] = lds[(size_t)t * vstride + 4 * (size_t)n + i];
        }
        __threadfence();   // a later chunk of a serial run reads these rows back
        __syncthreads();
    }
}

int tri_topk_device(const double* d, const double* e, int n, int k, double* vals, double* S)
{
    if (n < 1 || n > TRI_MAXN || k < 1 || k > 64 || k > n) return fail(MSM_ERR_INVALID, "tri_topk_device: need n <= %d, k <= 64", TRI_MAXN);
    hipLaunchKernelGGL(tri_topk_values_kernel, dim3(k), dim3(TRI_P), 0, stream(), d, e, n, vals);
    const size_t per = (5 * (size_t)n + (size_t)((n + 7) / 8)) * sizeof(double);
    const size_t extra = 2 * (size_t)n * sizeof(double);   // d, e
    const size_t budget = 150 * 1024;
    int G = (int)std::min<size_t>(std::max<size_t>((budget - extra) / per, 1), (size_t)k);
    static bool attr_set = false;
    if (!attr_set) {
        MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(tri_topk_vectors_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
        attr_set = true;
    }
    hipLaunchKernelGGL(tri_topk_vectors_kernel, dim3(k), dim3(TV_NT), (size_t)G * per + extra, stream(), d, e, n, k, G, vals, S);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// =====================================================================================================================
// Y[j][r] = sum_c Q[r][c] S[j][c] with Q = H_0 H_1 ... H_{n-3}, H_i = I - tau_i v_i v_i^T, the reflectors as sytrd.hip
// stores them (V row i = v_i[1 .. n-1]; v_i[c] = 0 for c <= i, v_i[i+1] = 1).  One wavefront owns AQ_RW rows of Q in
// registers (row r starts as e_r and takes z <- z - tau_i (z . v_i) v_i for i = 0 .. n-3); the workgroup stages blocks of
// AQ_RB reflectors through LDS, double-buffered, so that the chain of n - 2 dependent steps runs at LDS latency.
// =====================================================================================================================
constexpr int AQ_NT = 256;
constexpr int AQ_RW = 2;                      // rows per wavefront
constexpr int AQ_ROWS = AQ_RW * (AQ_NT / 64);   // rows per workgroup
constexpr int AQ_RB = 8;                      // reflectors per staged block

template <int NQ>   // register slots per row: n <= 64 NQ
__global__ __launch_bounds__(AQ_NT) void apply_q_rows_kernel(const double* __restrict__ V, const double* __restrict__ tau, int n,
                                                             int k, const double* __restrict__ S, double* __restrict__ Y)
{
    constexpr int NMAX = 64 * NQ;
    __shared__ double sv[2][AQ_RB][NMAX];
    __shared__ double stau[2][AQ_RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * AQ_ROWS + wave * AQ_RW;
    double z[AQ_RW][NQ];
#pragma unroll
    for (int a = 0; a < AQ_RW; ++a)
#pragma unroll
        for (int q = 0; q < NQ; ++q) z[a][q] = (lane + 64 * q == row0 + a) ? 1.0 : 0.0;
    const int nref = n - 2;
    const int nblk = nref > 0 ? (nref + AQ_RB - 1) / AQ_RB : 0;
    // staging: block `blk` of AQ_RB reflectors -> registers (loads in flight while the previous block is applied) -> LDS;
    // columns >= n are staged as zeros, so the arithmetic below needs no bounds
    constexpr int PFW = NMAX / AQ_NT;   // column slots per thread and reflector
    double pf[AQ_RB][PFW];
    auto fetch = [&](int blk) {
#pragma unroll
        for (int rr = 0; rr < AQ_RB; ++rr) {
            const int i = blk * AQ_RB + rr;
#pragma unroll
            for (int w = 0; w < PFW; ++w) {
                const int c = tid + AQ_NT * w;
                pf[rr][w] = (i < nref && c >= 1 && c < n) ? V[(size_t)i * (n - 1) + (c - 1)] : 0.0;
            }
        }
    };
    auto commit = [&](int blk, int buf) {
#pragma unroll
        for (int rr = 0; rr < AQ_RB; ++rr)
#pragma unroll
            for (int w = 0; w < PFW; ++w) sv[buf][rr][tid + AQ_NT * w] = pf[rr][w];
        if (tid < AQ_RB) {
            const int i = blk * AQ_RB + tid;
            stau[buf][tid] = i < nref ? tau[i] : 0.0;
        }
    };
    if (nblk > 0) {
        fetch(0);
        commit(0, 0);
    }
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const int buf = blk & 1;
        if (blk + 1 < nblk) fetch(blk + 1);
#pragma unroll 2
        for (int rr = 0; rr < AQ_RB; ++rr) {
            const double t = stau[buf][rr];
            double vq[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) vq[q] = sv[buf][rr][lane + 64 * q];
            double dot[AQ_RW];
#pragma unroll
            for (int a = 0; a < AQ_RW; ++a) {
                double s0 = 0.0, s1 = 0.0;   // two chains per row: half the dependent depth
#pragma unroll
                for (int q = 0; q < NQ; q += 2) {
                    s0 += z[a][q] * vq[q];
                    s1 += z[a][q + 1] * vq[q + 1];
                }
                dot[a] = s0 + s1;
            }
#pragma unroll
            for (int a = 0; a < AQ_RW; ++a) dot[a] = wave_sum_f64(dot[a]);
#pragma unroll
            for (int a = 0; a < AQ_RW; ++a) {
                const double f = t * dot[a];
#pragma unroll
                for (int q = 0; q < NQ; ++q) z[a][q] -= f * vq[q];
            }
        }
        if (blk + 1 < nblk) commit(blk + 1, buf ^ 1);
        __syncthreads();
    }
    // Y[j][r] = z_r . S[j]
    for (int j = 0; j < k; ++j) {
        const double* sj = S + (size_t)j * n;
        double sq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) sq[q] = (lane + 64 * q) < n ? sj[lane + 64 * q] : 0.0;
#pragma unroll
        for (int a = 0; a < AQ_RW; ++a) {
            double s2 = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) s2 += z[a][q] * sq[q];
            s2 = wave_sum_f64(s2);
            if (lane == 0 && row0 + a < n) Y[(size_t)j * n + row0 + a] = s2;
        }
    }
}

int apply_q_device(const double* V, const double* tau, int n, int k, const double* S, double* Y)
{
    if (n < 1 || n > TRI_MAXN) return fail(MSM_ERR_INVALID, "apply_q_device: n out of range");
    const dim3 grid((unsigned)ceil_div(n, AQ_ROWS));
    if (n <= 256)
        hipLaunchKernelGGL(apply_q_rows_kernel<4>, grid, dim3(AQ_NT), 0, stream(), V, tau, n, k, S, Y);
    else if (n <= 512)
        hipLaunchKernelGGL(apply_q_rows_kernel<8>, grid, dim3(AQ_NT), 0, stream(), V, tau, n, k, S, Y);
    else
        hipLaunchKernelGGL(apply_q_rows_kernel<16>, grid, dim3(AQ_NT), 0, stream(), V, tau, n, k, S, Y);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// res[j] = max_r |(C y_j)_r - lambda_j y_j[r]|, res[k + j] = | ||y_j||^2 - 1 |   (C symmetric n x n, Y [k][n])
__global__ __launch_bounds__(256) void pair_residual_kernel(const double* __restrict__ Cm, int n, const double* __restrict__ Y,
                                                            const double* __restrict__ vals, int k, double* __restrict__ res)
{
    __shared__ double sy[TRI_MAXN];
    __shared__ double rm[4], rs[4];
    const int j = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) sy[i] = Y[(size_t)j * n + i];
    __syncthreads();
    const double lam = vals[j];
    double mx = 0.0, ss = 0.0;
    for (int r = tid; r < n; r += 256) {
        double acc = 0.0;
        int c = 0;
        for (; c + 8 <= n; c += 8) {   // symmetric: column r read as a row, coalesced; eight loads in flight per trip
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Cm[(size_t)(c + u) * n + r];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u] * sy[c + u];
        }
        for (; c < n; ++c) acc += Cm[(size_t)c * n + r] * sy[c];
        mx = fmax(mx, fabs(acc - lam * sy[r]));
        ss += sy[r] * sy[r];
    }
    if (!(mx == mx)) mx = INFINITY;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, m, 64));
        ss += __shfl_xor(ss, m, 64);
    }
    if ((tid & 63) == 0) {
        rm[tid >> 6] = mx;
        rs[tid >> 6] = ss;
    }
    __syncthreads();
    if (tid == 0) {
        res[j] = fmax(fmax(rm[0], rm[1]), fmax(rm[2], rm[3]));
        const double nn = (rs[0] + rs[1]) + (rs[2] + rs[3]);
        res[k + j] = (nn == nn) ? fabs(nn - 1.0) : INFINITY;
    }
}

int pair_residual_device(const double* Cm, int n, const double* Y, const double* vals, int k, double* res2k)
{
    hipLaunchKernelGGL(pair_residual_kernel, dim3(k), dim3(256), 0, stream(), Cm, n, Y, vals, k, res2k);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

/* k largest eigenpairs of the symmetric tridiagonal (d[n], e[n-1]) on the device: vals[k] descending, vecs[k][n].
 * Host or device pointers per on_device.  (Building block of msm_tica_solve_topk; exported for the tests.) */
int msm_tridiag_topk(const double* d, const double* e, msm_idx_t n, msm_idx_t k, double* vals, double* vecs, int on_device)
{
    if (!d || (n > 1 && !e) || !vals || !vecs) return fail(MSM_ERR_INVALID, "msm_tridiag_topk: null pointer");
    if (n < 1 || n > TRI_MAXN || k < 1 || k > n || k > 64) return fail(MSM_ERR_INVALID, "msm_tridiag_topk: need 1 <= k <= min(n, 64), n <= %d", TRI_MAXN);
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf& buf = pool(PS_W);
    int rc = buf.reserve(((size_t)2 * n + k + (size_t)k * n) * sizeof(double));
    if (rc) return rc;
    double* dd = buf.as<double>();
    double* de = dd + n;
    double* dv = de + n;
    double* dS = dv + k;
    const hipMemcpyKind in = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    MSM_HIP_CHECK(hipMemcpyAsync(dd, d, n * sizeof(double), in, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(de, 0, n * sizeof(double), stream()));
    if (n > 1) MSM_HIP_CHECK(hipMemcpyAsync(de, e, (n - 1) * sizeof(double), in, stream()));
    if ((rc = tri_topk_device(dd, de, (int)n, (int)k, dv, dS))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(vals, dv, k * sizeof(double), out, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(vecs, dS, (size_t)k * n * sizeof(double), out, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

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
    if ((rc = potrf_upper_device(dB, (int)n, dinfo))) return rc;
    if (!on_device) MSM_HIP_CHECK(hipMemcpyAsync(B, dB, nn * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(info, dinfo, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"
