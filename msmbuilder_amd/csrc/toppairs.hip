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

constexpr int TRI_MAXN = 1024;   // rows of the reduced matrix the residual kernel stages

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
