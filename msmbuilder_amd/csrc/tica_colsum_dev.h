// tica_colsum_dev.h -- column sums, folded sums, mean shift and export kernels
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_cg_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// Column sums s0 / stau (tica.py:418-419) + the finite check of
// utils/validation.py:68-74, one streaming pass, fp64 accumulation.
// Block b owns partial slot b and walks chunks b, b+grid, ...
// ---------------------------------------------------------------------------
__device__ __forceinline__ double to_f64(float v) { return (double)v; }
__device__ __forceinline__ double to_f64(double v) { return v; }
__device__ __forceinline__ double to_f64(__bf16 v) { return (double)(float)v; }

template <typename TIn>
__global__ __launch_bounds__(NT) void tica_colsum_kernel(TicaArgs P)
{
    // thread -> a group of CW consecutive columns (one 16-byte load per row when aligned) and a
    // row lane; RU rows are kept in flight per thread so the pass is HBM-bound, not latency-bound
    constexpr int CW = 16 / sizeof(TIn);
    constexpr int RU = 8;
    __shared__ double red[2][NT][CW];
    const int tid = threadIdx.x;
    const int ngroups = (P.F + CW - 1) / CW;
    int cpb = 1;
    while (cpb < ngroups && cpb < NT) cpb <<= 1;  // column groups per pass (power of two <= 256)
    const int rl = NT / cpb;                      // row lanes
    const int tc = tid % cpb, tr = tid / cpb;
    const bool vec = (P.F % CW == 0) && (P.ld % CW == 0);
    double* part = P.colpart + (size_t)blockIdx.x * 2 * P.F;
    int bad = 0;
    for (int g0 = 0; g0 < ngroups; g0 += cpb) {
        const int col = (g0 + tc) * CW;
        double s0[CW], st[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) s0[e] = st[e] = 0.0;
        if (col < P.F) {
            for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
                const TicaChunk ch = get_chunk(P, c);
                const global_ptr<TIn> X = as_global<TIn>(ch.base);
                const bool al = vec && ((((uintptr_t)ch.base) & 15) == 0);
                for (int k0 = tr; k0 < ch.n; k0 += rl * RU) {
                    TIn v[RU][CW];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int kr = k0 + u * rl;
                        const long long r = ch.row0 + kr;
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[u][e] = (TIn)0.f;
                        if (kr < ch.n) {
                            const global_ptr<TIn> p = X + r * P.ld + col;
                            if (al) {
                                *reinterpret_cast<float4*>(&v[u][0]) = load16_global<TIn>(p);
                            } else {
#pragma unroll
                                for (int e = 0; e < CW; ++e)
                                    if (col + e < P.F) v[u][e] = p[e];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int kr = k0 + u * rl;
                        const long long r = ch.row0 + kr;
                        const bool in0 = (kr < ch.n) && (r < ch.len - P.lag);
                        const bool in1 = (kr < ch.n) && (r >= P.lag);
#pragma unroll
                        for (int e = 0; e < CW; ++e) {
                            const double x = to_f64(v[u][e]);
                            bad |= !isfinite(x);
                            if (in0) s0[e] += x;
                            if (in1) st[e] += x;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            red[0][tid][e] = s0[e];
            red[1][tid][e] = st[e];
        }
        __syncthreads();
        if (tr == 0 && col < P.F) {
            for (int k = 1; k < rl; ++k)
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    s0[e] += red[0][k * cpb + tc][e];
                    st[e] += red[1][k * cpb + tc][e];
                }
#pragma unroll
            for (int e = 0; e < CW; ++e)
                if (col + e < P.F) {
                    part[col + e] += s0[e];
                    part[P.F + col + e] += st[e];
                }
        }
        __syncthreads();
    }
    if (bad) atomicOr(P.flag, 1);
}

// colpart (persistent) += coltmp, then coltmp = 0
__global__ void tica_colmerge_kernel(double* __restrict__ dst, double* __restrict__ tmp, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dst[i] += tmp[i];
        tmp[i] = 0.0;
    }
}

// ---- folded column sums (sum/difference kernel, FOLD) ---------------------------------------------------------------
// r for a handle's first launch when no column-sum pass runs ahead of the MFMA kernel: the mean of up to FOLD_NS frames
// spread evenly over the launch's chunks (any r within a fraction of sigma of the mean serves: the shifted moments are
// restored exactly whatever r is).  One block per 64 columns, four row lanes, fp64.
constexpr int FOLD_NS = 4096, FOLD_NB = 32;   // samples, and the blocks (per 64 columns) that share them
template <typename TIn>
__global__ __launch_bounds__(256) void tica_fold_sample_kernel(TicaArgs P, double* __restrict__ part)
{
    __shared__ double red[256];
    constexpr int PER = FOLD_NS / FOLD_NB / 4;   // samples per row lane
    const int tid = threadIdx.x, col = blockIdx.x * 64 + (tid & 63), rl = tid >> 6;
    double a = 0.0;
    if (col < P.F)
        for (int i0 = 0; i0 < PER; i0 += 8) {
            TIn v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = (blockIdx.y * 4 + rl) * PER + i0 + u;
                const TicaChunk ch = get_chunk(P, ((long long)s * P.nchunks) / FOLD_NS);
                const int row = (int)((((unsigned)s * 2654435761u) >> 8) % (unsigned)ch.n);
                v[u] = as_global<TIn>(ch.base)[(ch.row0 + row) * P.ld + col];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += to_f64(v[u]);
        }
    red[tid] = a;
    __syncthreads();
    if (rl == 0 && col < P.F) part[(size_t)blockIdx.y * P.F + col] = a + red[tid + 64] + red[tid + 128] + red[tid + 192];
}

__global__ void tica_fold_setr_kernel(const double* __restrict__ part, float* __restrict__ r, int F)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= F) return;
    double v[FOLD_NB], a = 0.0;
#pragma unroll
    for (int k = 0; k < FOLD_NB; ++k) v[k] = part[(size_t)k * F + col];
#pragma unroll
    for (int k = 0; k < FOLD_NB; ++k) a += v[k];
    r[col] = (float)(a / (double)FOLD_NS);
}

// After the FOLD kernel: colA[c][:] = sums of cohort c's left frames; tmp (the [NCB][2][F] temporary partials) holds what a
// column-sum pass over the trajectories' first and last tau rows left there: [k][0] = a_k (rows [0, tau)), [k][1] = b_k
// (rows [len - tau, len)).  s0 = sum A, stau = sum of the right frames = A - a + b, so slot k becomes
// [A_k | A_k - a_k + b_k] (A_k = 0 beyond the S cohorts) -- the layout an ordinary column-sum pass leaves.  A non-finite
// A_k raises the flag (the boundary pass checked its own rows element by element).
// the bf16 image path's variant: colA[c][:] per CHUNK (tica_img_kernel); slot k takes chunks k, k + NCB, ... in order
__global__ void tica_fold_fix_img_kernel(double* __restrict__ tmp, const double* __restrict__ colA, int F, long long nchunks, int* flag)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)NCB * F) return;
    const int k = (int)(idx / F), col = (int)(idx - (size_t)k * F);
    double A = 0.0;
    for (long long c = k; c < nchunks; c += NCB) A += colA[(size_t)c * F + col];
    double* t = tmp + (size_t)k * 2 * F;
    const double a = t[col], b = t[F + col];
    t[col] = A;
    t[F + col] = (A - a) + b;
    if (!isfinite(A)) atomicOr(flag, 1);
}

__global__ void tica_fold_fix_kernel(double* __restrict__ tmp, const double* __restrict__ colA, int F, int S, int* flag)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)NCB * F) return;
    const int k = (int)(idx / F), col = (int)(idx - (size_t)k * F);
    const double A = k < S ? colA[(size_t)k * F + col] : 0.0;
    double* t = tmp + (size_t)k * 2 * F;
    const double a = t[col], b = t[F + col];
    t[col] = A;
    t[F + col] = (A - a) + b;
    if (!isfinite(A)) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------
// Mean shift.  The covariance is G / 2N - mu mu^T (tica.py:228-259): an error of eps * |G| in an fp32-accumulated
// G is a RELATIVE covariance error of eps * (mu / sigma)^2, i.e. 1e-3 for features whose mean is 100 standard
// deviations (contact and atom-pair distances), where the reference -- float64 throughout, tica.py:402 -- loses nothing.
// So the fp32 and bf16 kernels accumulate the moments of y = x - r for a per-handle reference row r (fp32, the column
// mean of the first launch: the column-sum pass runs before the MFMA pass anyway), whose entries are sigma-sized, and
// the raw moments are restored in fp64 at export time from the exact fp64 column sums:
//     C = C' + A' r^T + r B'^T + n r r^T          A' = A - n r,  B' = B - n r      (A, B: sums of the left / right frames
//     G = G' + W' r^T + r W'^T + nW r r^T         W' = W - nW r                     of the n shifted pairs; W, nW: weighted
// frame sum and total weight of the Gram term -- A + B and 2n except when a trajectory is split over ranks, where the
// C/G kernel's Gram tiles own FRAMES, not pairs).  r never changes while a handle accumulates, so launches add up.
// ---------------------------------------------------------------------------
// part: the [NCB][2][F] column-sum partials (a = "s0" half, b = "stau" half) of ONE column-sum launch; `what` says where
// they go: SH_A_a: A += a, SH_B_b: B += b, SH_W_ab: W += a + b, SH_B_a: B += a, SH_W_a: W += a.
//   whole trajectories                       A|B_b|W_ab   (left sums, right sums, both)
//   segments, owned rows, C/G or bf16 kernel A|W_ab       (the Gram tiles weight the OWNED frames)
//   segments, owned rows, H/D kernel         A|W_a        (its Gram is over owned PAIRS: W = A + B)
//   segments, the pairs' right rows          B_a (|W_a for the H/D kernel)
enum { SH_A_a = 1, SH_B_b = 2, SH_W_ab = 4, SH_B_a = 8, SH_W_a = 16 };
__global__ __launch_bounds__(512) void tica_shift_kernel(const double* __restrict__ part, double* __restrict__ shsum,
                                                         float* __restrict__ r, int F, double inv_n, int set_r, int what)
{
    // 64 columns per workgroup of 512; thread = (column, half a / b, one of four row lanes) with ONE accumulator and sixteen
    // partials requested per trip -- the shape tica_export_cols_kernel has.  (Round 3's loop -- four row lanes, `a += ...; b += ...`
    // under `if (col < F)` -- compiled to pairs of loads each waited for on the spot: 256 dependent round trips per
    // thread, 84 us per launch; two batched accumulators per thread were paired up again by the scheduler.)
    __shared__ double red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, half = (tid >> 6) & 1, rl = tid >> 7;
    const int col = blockIdx.x * 64 + lane;
    const int cc = col < F ? col : F - 1;
    static_assert(NCB % 64 == 0, "whole groups of 16 per row lane");
    double acc = 0.0;
    for (int k0 = rl; k0 < NCB; k0 += 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(k0 + 4 * u) * 2 * F + (size_t)half * F + cc];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    red[half][rl][lane] = acc;
    __syncthreads();
    if (tid < 64 && col < F) {
        const double a = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        const double b = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        if (set_r) r[col] = (float)((a + b) * inv_n);
        if (what & SH_A_a) shsum[col] += a;
        if (what & SH_B_b) shsum[F + col] += b;
        if (what & SH_B_a) shsum[F + col] += a;
        if (what & SH_W_ab) shsum[2 * F + col] += a + b;
        if (what & SH_W_a) shsum[2 * F + col] += a;
    }
}

__global__ void tica_unshift_kernel(double* __restrict__ packed, const double* __restrict__ shsum,
                                    const float* __restrict__ r, double n, double nW, int F, int sym)
{
    const size_t FF = (size_t)F * F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * FF) return;
    const int type = idx >= FF;
    const size_t e = idx - (type ? FF : 0);
    int i = (int)(e / F), j = (int)(e % F);
    if ((type || sym) && i > j) {  // symmetric corrections: evaluate the mirrored element with the SAME operand order, so
        const int t = i;           // the result is symmetric bit for bit whatever the compiler contracts into FMAs
        i = j;
        j = t;
    }
    const double ri = (double)r[i], rj = (double)r[j];
    double v;
    if (type) {
        const double wi = shsum[2 * F + i] - nW * ri, wj = shsum[2 * F + j] - nW * rj;
        v = (ri * wj + wi * rj) + nW * ri * rj;
    } else {
        const double ai = shsum[i] - n * ri, aj = shsum[j] - n * rj;
        const double bi = shsum[F + i] - n * ri, bj = shsum[F + j] - n * rj;
        if (sym)
            v = 0.5 * ((ai * rj + aj * ri) + (ri * bj + rj * bi)) + n * ri * rj;
        else
            v = (ai * rj + ri * bj) + n * ri * rj;
    }
    packed[idx] += v;
}

// packed[C | G | s0 | stau | n_obs | n_seq] = base + sum over slabs / column partials
__global__ void tica_export_kernel(const double* __restrict__ slabs, const double* __restrict__ colpart,
                                   const double* __restrict__ base, double* __restrict__ out, int F,
                                   int T, int ntiles, int S)
{
    const size_t FF = (size_t)F * F;
    const size_t total = 2 * FF + 2 * (size_t)F;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    double v = base[idx];
    if (idx < 2 * FF) {
        const int type = idx >= FF;
        const size_t e = idx - (type ? FF : 0);
        int i = (int)(e / F), j = (int)(e % F);
        int ti = i / TM, tj = j / TM;
        int tile;
        if (type == 0) {
            tile = ti * T + tj;
        } else {
            if (i > j) {  // lower triangle (also inside a diagonal tile): mirror of the upper element, so the
                          // result is exactly symmetric whatever the kernel's product order was
                int t = i; i = j; j = t;
                t = ti; ti = tj; tj = t;
            }
            // upper-triangle tiles are enumerated row by row: (0,0..T-1), (1,1..T-1), ...
            tile = T * T + ti * T - ti * (ti - 1) / 2 + (tj - ti);
        }
        const size_t off = (size_t)(i % TM) * TM + (j % TM);
        int s = 0;
        for (; s + 4 <= S; s += 4) {   // four slabs' loads in flight per trip
            double q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = slabs[((size_t)(s + u) * ntiles + tile) * (TM * TM) + off];
#pragma unroll
            for (int u = 0; u < 4; ++u) v += q[u];
        }
        for (; s < S; ++s) v += slabs[((size_t)s * ntiles + tile) * (TM * TM) + off];
    } else {
        return;   // [s0 | stau]: tica_export_cols_kernel (NCB partials per column: a reduction, not a per-thread loop)
    }
    out[idx] = v;
}

// out[2 F^2 + e] = base[2 F^2 + e] + sum over the NCB column partials, e in [s0 | stau].  64 columns per workgroup, four
// waves take a quarter of the partials each with 16 loads in flight per trip (the per-thread loop over all 1024 partials
// was 1024 dependent L2 round trips: 70 us of a 2 ms solve), summed in partial order (deterministic).
__global__ __launch_bounds__(256) void tica_export_cols_kernel(const double* __restrict__ colpart, const double* __restrict__ base,
                                                               double* __restrict__ out, int F)
{
    __shared__ double red[4][64];
    const size_t FF = (size_t)F * F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool in = e < 2 * F;
    double acc = 0.0;
    constexpr int PER = NCB / 4;
    for (int b0 = wave * PER; b0 < (wave + 1) * PER; b0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in ? colpart[(size_t)(b0 + u) * 2 * F + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && in) out[2 * FF + e] = base[2 * FF + e] + ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

}  // namespace msm
