// preprocess.hip -- the column scan in front of tICA: per-column count / mean / M2 /
// min / max in ONE streaming pass (what msmbuilder.preprocessing.StandardScaler, MinMaxScaler and
// MaxAbsScaler -- thin mixins over scikit-learn's scalers,
// /root/reference/msmbuilder/preprocessing/__init__.py:56-83, base.py:14-199 -- compute in fit /
// partial_fit), and the element-wise (x - shift) / scale of their transform.
//
// Arithmetic restated (scikit-learn, third party, unpinned by the reference -- DESIGN.md):
//   mean/var : sklearn.utils.extmath._incremental_mean_and_var -- float64 sums, NaN = missing value,
//              batches merged with Chan/Golub/LeVeque's update
//                  delta = mean_b - mean;  mean += delta * n_b / (n + n_b)
//                  M2   += M2_b + delta^2 * n * n_b / (n + n_b)
//              here: every thread forms (n_b, mean_b, M2_b) of 8 rows in registers (two-pass, exact
//              enough for any offset), merges into its running triple, and triples are merged
//              lane -> block -> grid in a FIXED order (deterministic, no atomics).
//   transform: numpy in-place `X -= mean_; X /= scale_` on the input dtype: each step is computed in
//              float64 and rounded back to the array's dtype (StandardScaler.transform).
// HBM-bound: F * sizeof(T) bytes per frame read once (scan), read + written once (transform).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace msm {

constexpr int PNT = 256;
constexpr int PNB = 1024;  // scan blocks (= partial slots), at most
constexpr long long SCAN_ROWS = 1024;  // rows per chunk of the column scans

struct ScanChunk {
    const void* base;
    long long n;  // rows (<= SCAN_ROWS; the digit passes use 4096)
};

struct ScanArgs {
    const ScanChunk* chunks;
    long long nchunks;
    long long ld;
    int F;
    double* part;  // [gridDim][5][F]: n, mean, M2, min, max
    int* flag;     // |= 1 if an infinity was seen
};

struct Stat {
    double n, mean, m2, lo, hi;
};

__device__ __forceinline__ void stat_init(Stat& s)
{
    s.n = 0.0;
    s.mean = 0.0;
    s.m2 = 0.0;
    s.lo = INFINITY;
    s.hi = -INFINITY;
}

__device__ __forceinline__ void stat_merge(Stat& a, const Stat& b)
{
    if (b.n == 0.0) return;
    const double tot = a.n + b.n;
    const double delta = b.mean - a.mean;
    const double w = b.n / tot;
    a.mean = a.mean + delta * w;
    a.m2 = a.m2 + b.m2 + delta * delta * a.n * w;
    a.n = tot;
    a.lo = b.lo < a.lo ? b.lo : a.lo;
    a.hi = b.hi > a.hi ? b.hi : a.hi;
}

template <typename T>
__global__ __launch_bounds__(PNT) void colstats_kernel(ScanArgs P)
{
    constexpr int CW = 16 / sizeof(T);
    constexpr int RU = 8;
    __shared__ Stat red[PNT][CW];
    const int tid = threadIdx.x;
    const int ngroups = (P.F + CW - 1) / CW;
    int cpb = 1;
    while (cpb < ngroups && cpb < PNT) cpb <<= 1;
    const int rl = PNT / cpb;
    const int tc = tid % cpb, tr = tid / cpb;
    const bool vec = (P.F % CW == 0) && (P.ld % CW == 0);
    double* part = P.part + (size_t)blockIdx.x * 5 * P.F;
    int inf_seen = 0;
    for (int g0 = 0; g0 < ngroups; g0 += cpb) {
        const int col = (g0 + tc) * CW;
        Stat run[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) stat_init(run[e]);
        if (col < P.F) {
            for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
                const ScanChunk ch = P.chunks[c];
                const global_ptr<T> X = as_global<T>(ch.base);
                const bool al = vec && ((((uintptr_t)ch.base) & 15) == 0);
                // One chunk (<= SCAN_ROWS rows, 1 / rl of them this thread's) per column as SHIFTED sums
                // s1 = sum (x - K), s2 = sum (x - K)^2 in float64, K = the running mean (first chunk: the first value):
                // 4 float64 operations per element and one division per chunk, where the first version (two-pass
                // batches of 8 rows, one Chan merge with two divisions per batch) was VALU-bound at 2.7 TB/s.
                // x - K is rounded once (float64); the cancellation in s2 - s1^2/n costs at most
                // n_chunk * eps * (x - K)^2 <= 1024 * 2^-53 * range^2, against a total M2 >= range^2 / 2.
                double K[CW], s1[CW], s2[CW];
                int cnt[CW];
                T lo[CW], hi[CW];
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    K[e] = run[e].mean;
                    s1[e] = s2[e] = 0.0;
                    cnt[e] = 0;
                    lo[e] = (T)INFINITY;
                    hi[e] = (T)-INFINITY;
                }
                for (long long k0 = tr; k0 < ch.n; k0 += (long long)rl * RU) {
                    T v[RU][CW];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const long long kr = k0 + (long long)u * rl;
                        const long long rr = kr < ch.n ? kr : ch.n - 1;  // clamped: unconditional loads
                        const global_ptr<T> p = X + rr * P.ld + col;
                        if (al) {
                            *reinterpret_cast<float4*>(&v[u][0]) = load16_global<T>(p);
                        } else {
#pragma unroll
                            for (int e = 0; e < CW; ++e) v[u][e] = (col + e < P.F) ? p[e] : (T)0;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < CW; ++e) {
                        if (k0 == tr && run[e].n == 0.0) {  // nothing seen yet: shift by the first finite value
                            const T x0 = v[0][e];
                            K[e] = (x0 == x0 && x0 != (T)INFINITY && x0 != (T)-INFINITY) ? (double)x0 : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < RU; ++u) {
                            const T xt = v[u][e];
                            const bool ok = (k0 + (long long)u * rl < ch.n) && (xt == xt);  // NaN = missing
                            const double d = (double)xt - K[e];
                            if (ok) {
                                ++cnt[e];
                                s1[e] += d;
                                s2[e] = fma(d, d, s2[e]);
                                lo[e] = xt < lo[e] ? xt : lo[e];
                                hi[e] = xt > hi[e] ? xt : hi[e];
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < CW; ++e)
                    if (cnt[e] > 0) {
                        inf_seen |= (lo[e] == (T)-INFINITY || hi[e] == (T)INFINITY) ? 1 : 0;
                        Stat b;
                        b.n = (double)cnt[e];
                        const double m = s1[e] / b.n;
                        b.mean = K[e] + m;
                        const double m2 = s2[e] - s1[e] * m;
                        b.m2 = m2 > 0.0 ? m2 : 0.0;
                        b.lo = (double)lo[e];
                        b.hi = (double)hi[e];
                        stat_merge(run[e], b);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) red[tid][e] = run[e];
        __syncthreads();
        if (tr == 0 && col < P.F) {
            for (int k = 1; k < rl; ++k)
#pragma unroll
                for (int e = 0; e < CW; ++e) stat_merge(run[e], red[k * cpb + tc][e]);
#pragma unroll
            for (int e = 0; e < CW; ++e)
                if (col + e < P.F) {
                    part[0 * P.F + col + e] = run[e].n;
                    part[1 * P.F + col + e] = run[e].mean;
                    part[2 * P.F + col + e] = run[e].m2;
                    part[3 * P.F + col + e] = run[e].lo;
                    part[4 * P.F + col + e] = run[e].hi;
                }
        }
        __syncthreads();
    }
    if (inf_seen) atomicOr(P.flag, 1);
}

// one thread per column: merge the block partials in block order
__global__ void colstats_merge_kernel(const double* __restrict__ part, int nb, int F, double* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= F) return;
    Stat a;
    stat_init(a);
    for (int b = 0; b < nb; ++b) {
        const double* p = part + (size_t)b * 5 * F;
        Stat s;
        s.n = p[c];
        s.mean = p[F + c];
        s.m2 = p[2 * F + c];
        s.lo = p[3 * F + c];
        s.hi = p[4 * F + c];
        stat_merge(a, s);
    }
    out[c] = a.n;
    out[F + c] = a.mean;
    out[2 * F + c] = a.m2;
    out[3 * F + c] = a.lo;
    out[4 * F + c] = a.hi;
}

// mode 0: out = ((T)((double)x - shift)) / scale      (StandardScaler / MaxAbsScaler.transform)
// mode 1: out = ((T)((double)x * scale)) + shift      (MinMaxScaler.transform: X *= scale_; X += min_)
// each step rounded to T (numpy in-place semantics); a null array skips its step
template <typename T>
__global__ __launch_bounds__(PNT) void scale_apply_kernel(const T* __restrict__ X, long long n, int F, long long ld,
                                                          const double* __restrict__ shift,
                                                          const double* __restrict__ scale, T* __restrict__ out,
                                                          long long ldo, int vec, int mode)
{
    // thread -> one group of CW consecutive columns (its shift / scale live in registers) and a row
    // lane; RU rows in flight per thread.  No per-element index arithmetic.
    constexpr int CW = 16 / sizeof(T);
    constexpr int RU = 4;
    const int tid = threadIdx.x;
    const int ngroups = (F + CW - 1) / CW;
    int cpb = 1;
    while (cpb < ngroups && cpb < PNT) cpb <<= 1;
    const int rl = PNT / cpb;
    const int tc = tid % cpb, tr = tid / cpb;
    for (int g0 = 0; g0 < ngroups; g0 += cpb) {
        const int col = (g0 + tc) * CW;
        if (col >= F) continue;
        double sh[CW], sc[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            const int c = col + e < F ? col + e : F - 1;
            sh[e] = shift ? shift[c] : 0.0;
            sc[e] = scale ? scale[c] : 1.0;
        }
        for (long long r0 = (long long)blockIdx.x * rl * RU + tr; r0 < n; r0 += (long long)gridDim.x * rl * RU) {
            T v[RU][CW];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const long long r = r0 + (long long)u * rl;
                const long long rr = r < n ? r : n - 1;
                if (vec) {
                    *reinterpret_cast<float4*>(&v[u][0]) = *reinterpret_cast<const float4*>(X + rr * ld + col);
                } else {
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[u][e] = (col + e < F) ? X[rr * ld + col + e] : (T)0;
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const long long r = r0 + (long long)u * rl;
                if (r >= n) continue;
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    T x = v[u][e];
                    if (mode == 0) {
                        if (shift) x = (T)((double)x - sh[e]);
                        if (scale) x = (T)((double)x / sc[e]);
                    } else {
                        if (scale) x = (T)((double)x * sc[e]);
                        if (shift) x = (T)((double)x + sh[e]);
                    }
                    v[u][e] = x;
                }
                if (vec) {
                    *reinterpret_cast<float4*>(out + r * ldo + col) = *reinterpret_cast<float4*>(&v[u][0]);
                } else {
#pragma unroll
                    for (int e = 0; e < CW; ++e)
                        if (col + e < F) out[r * ldo + col + e] = v[u][e];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Exact per-column order statistics (RobustScaler: np.nanmedian / np.nanpercentile per feature,
// sklearn/preprocessing/_data.py RobustScaler.fit) by most-significant-digit radix SELECT: values map
// to order-preserving unsigned keys; one pass histograms the next `bits` bits of the keys that match
// the already-known high bits of a target, the host picks the digit holding the wanted rank and the
// next pass refines -- 3 passes for float32, 6 for float64, each one HBM stream over the data, no
// sort and no transposed copy.  Same thread mapping as colstats_kernel (column group x row lane, 8
// rows in flight): the lanes of a wave own different columns, so their atomics never collide, and a
// thread run-length merges its 8 consecutive frames of a column (MD features move slowly: same digit).
// ---------------------------------------------------------------------------
struct DigitArgs {
    const ScanChunk* chunks;
    long long nchunks;
    long long ld;
    int F;
    int R;                            // targets per column (1 in the first pass: the histogram is shared)
    const unsigned long long* prefix; // [R][F] high bits already known (right-aligned), nullptr in the first pass
    int shift;                        // LSB position of the digit
    int bits;                         // digit width
    unsigned long long* hist;         // [R][F][1 << bits]
};

__device__ __forceinline__ unsigned long long order_key(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? (unsigned)~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long order_key(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

template <typename T>
__global__ __launch_bounds__(PNT) void col_digit_kernel(DigitArgs P)
{
    constexpr int CW = 16 / sizeof(T);
    constexpr int RU = 8;
    const int tid = threadIdx.x;
    const int ngroups = (P.F + CW - 1) / CW;
    int cpb = 1;
    while (cpb < ngroups && cpb < PNT) cpb <<= 1;
    const int rl = PNT / cpb;
    const int tc = tid % cpb, tr = tid / cpb;
    const bool vec = (P.F % CW == 0) && (P.ld % CW == 0);
    const unsigned long long mask = (1ull << P.bits) - 1ull;
    const size_t nbin = (size_t)1 << P.bits;
    for (int g0 = 0; g0 < ngroups; g0 += cpb) {
        const int col = (g0 + tc) * CW;
        if (col >= P.F) continue;
        for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
            const ScanChunk ch = P.chunks[c];
            const global_ptr<T> X = as_global<T>(ch.base);
            const bool al = vec && ((((uintptr_t)ch.base) & 15) == 0);
            for (long long k0 = tr; k0 < ch.n; k0 += (long long)rl * RU) {
                T v[RU][CW];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const long long kr = k0 + (long long)u * rl;
                    const long long rr = kr < ch.n ? kr : ch.n - 1;
                    const global_ptr<T> p = X + rr * P.ld + col;
                    if (al) {
                        *reinterpret_cast<float4*>(&v[u][0]) = load16_global<T>(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[u][e] = (col + e < P.F) ? p[e] : (T)0;
                    }
                }
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    if (col + e >= P.F) continue;
                    for (int r = 0; r < P.R; ++r) {
                        unsigned long long* h = P.hist + ((size_t)r * P.F + col + e) * nbin;
                        const unsigned long long want = P.prefix ? P.prefix[(size_t)r * P.F + col + e] : 0ull;
                        long long cur = -1;
                        unsigned long long run = 0;
#pragma unroll
                        for (int u = 0; u < RU; ++u) {
                            const T x = v[u][e];
                            const unsigned long long key = order_key(x);
                            const bool ok = (k0 + (long long)u * rl < ch.n) && (x == x) &&
                                            (!P.prefix || (key >> (P.shift + P.bits)) == want);
                            const long long d = ok ? (long long)((key >> P.shift) & mask) : -1;
                            if (d != cur) {
                                if (cur >= 0) atomicAdd(h + cur, run);
                                cur = d;
                                run = 0;
                            }
                            ++run;
                        }
                        if (cur >= 0) atomicAdd(h + cur, run);
                    }
                }
            }
        }
    }
}

// First pass of the select (no prefix yet: EVERY value is counted): global atomics would be ~5e9 at
// 10M x 512, so this pass keeps workgroup-private histograms in LDS instead -- a workgroup owns a tile
// of DCOLS = 16 columns (one 64-byte segment per row for float32) and all 1 << bits <= 2048 bins of each
// (128 KiB), streams its share of the rows, and adds its counts to the global histogram once at the end.
constexpr int DCOLS = 16;

template <typename T>
__global__ __launch_bounds__(PNT, 1) void col_digit_lds_kernel(DigitArgs P)
{
    extern __shared__ unsigned dh[];  // [DCOLS][1 << bits]
    constexpr int CW = 16 / sizeof(T);
    constexpr int GPT = DCOLS / CW;   // column groups per tile (4 f32 / 8 f64)
    constexpr int RL = PNT / GPT;     // row lanes
    constexpr int RU = 8;
    const int tid = threadIdx.x;
    const int nbin = 1 << P.bits;
    const int ntile = (P.F + DCOLS - 1) / DCOLS;
    const int tile = blockIdx.x % ntile, part = blockIdx.x / ntile, nparts = gridDim.x / ntile;
    const int tc = tid % GPT, tr = tid / GPT;
    const int col = tile * DCOLS + tc * CW;
    const bool vec = (P.F % CW == 0) && (P.ld % CW == 0);
    for (int i = tid; i < DCOLS * nbin; i += PNT) dh[i] = 0u;
    __syncthreads();
    if (col < P.F && part < nparts) {
        for (long long c = part; c < P.nchunks; c += nparts) {
            const ScanChunk ch = P.chunks[c];
            const global_ptr<T> X = as_global<T>(ch.base);
            const bool al = vec && ((((uintptr_t)ch.base) & 15) == 0);
            for (long long k0 = tr; k0 < ch.n; k0 += (long long)RL * RU) {
                T v[RU][CW];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const long long kr = k0 + (long long)u * RL;
                    const long long rr = kr < ch.n ? kr : ch.n - 1;
                    const global_ptr<T> p = X + rr * P.ld + col;
                    if (al) {
                        *reinterpret_cast<float4*>(&v[u][0]) = load16_global<T>(p);
                    } else {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[u][e] = (col + e < P.F) ? p[e] : (T)0;
                    }
                }
#pragma unroll
                for (int e = 0; e < CW; ++e) {
                    if (col + e >= P.F) continue;
                    unsigned* h = dh + (tc * CW + e) * nbin;
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const T x = v[u][e];
                        if ((k0 + (long long)u * RL < ch.n) && (x == x)) atomicAdd(h + (int)(order_key(x) >> P.shift), 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < DCOLS * nbin; i += PNT) {
        const int cc = tile * DCOLS + i / nbin;
        if (cc < P.F && dh[i]) atomicAdd(P.hist + (size_t)cc * nbin + (i % nbin), (unsigned long long)dh[i]);
    }
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_colstats(const void* const* X_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                 msm_idx_t n_features, msm_idx_t ld, int on_device, double* out5F, int* has_inf)
{
    if (n_seq < 0 || (n_seq > 0 && (!X_ptrs || !n_rows)) || !out5F)
        return fail(MSM_ERR_INVALID, "msm_colstats: bad argument");
    if (dtype_bytes != 4 && dtype_bytes != 8) return fail(MSM_ERR_INVALID, "dtype_bytes must be 4 or 8");
    if (n_features < 1 || ld < n_features) return fail(MSM_ERR_INVALID, "msm_colstats: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int F = (int)n_features;
    int rc;
    DevBuf &dPart = pool(PS_PART), &dTab = pool(PS_IDS), &dOut = pool(PS_SUM), &dX = pool(PS_X);
    // host trajectories are staged in groups of <= 1 GiB; statistics of the groups are merged on the host
    std::vector<double> acc((size_t)5 * F, 0.0);
    for (int c = 0; c < F; ++c) {
        acc[(size_t)3 * F + c] = INFINITY;
        acc[(size_t)4 * F + c] = -INFINITY;
    }
    int any_inf = 0;
    const size_t row_bytes = (size_t)F * dtype_bytes;
    msm_idx_t s = 0;
    while (s < n_seq) {
        msm_idx_t e = s;
        std::vector<ScanChunk> tab;
        size_t bytes = 0;
        if (on_device) {
            e = n_seq;
        } else {
            while (e < n_seq && (e == s || bytes + (size_t)n_rows[e] * row_bytes <= ((size_t)1 << 30))) {
                bytes += ((size_t)n_rows[e] * row_bytes + 255) & ~(size_t)255;
                ++e;
            }
            if ((rc = dX.reserve(bytes ? bytes : 256))) return rc;
        }
        size_t off = 0;
        long long group_ld = on_device ? ld : F;
        for (msm_idx_t i = s; i < e; ++i) {
            if (n_rows[i] < 0 || (n_rows[i] > 0 && !X_ptrs[i])) return fail(MSM_ERR_INVALID, "msm_colstats: bad sequence %lld", (long long)i);
            const char* base = (const char*)X_ptrs[i];
            if (!on_device) {
                char* d = dX.as<char>() + off;
                if (n_rows[i] > 0) {
                    if (ld == F) {
                        if ((rc = h2d_bulk(d, X_ptrs[i], (size_t)n_rows[i] * row_bytes))) return rc;
                    } else {
                        MSM_HIP_CHECK(hipMemcpy2DAsync(d, row_bytes, X_ptrs[i], (size_t)ld * dtype_bytes, row_bytes,
                                                       (size_t)n_rows[i], hipMemcpyHostToDevice, stream()));
                    }
                }
                base = d;
                off += ((size_t)n_rows[i] * row_bytes + 255) & ~(size_t)255;
            }
            for (long long r0 = 0; r0 < n_rows[i]; r0 += SCAN_ROWS) {
                ScanChunk ch;
                ch.base = base + (size_t)r0 * (size_t)group_ld * dtype_bytes;
                ch.n = std::min<long long>(SCAN_ROWS, n_rows[i] - r0);
                tab.push_back(ch);
            }
        }
        if (!tab.empty()) {
            // one resident round: with more workgroups than fit (first version: 1024 on 768 slots) the second round runs
            // a third full, and chunks of 4096 rows left 2441 of them on 768 workgroups (3 or 4 each: 25 % tail)
            static int slots[2] = {0, 0};
            int& sl = slots[dtype_bytes == 4 ? 0 : 1];
            if (!sl) {
                int occ = 0;
                if (dtype_bytes == 4)
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, colstats_kernel<float>, PNT, 0);
                else
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, colstats_kernel<double>, PNT, 0);
                sl = std::max(1, std::min(PNB, std::max(1, occ) * num_cus()));
            }
            const int nb = (int)std::min<size_t>(tab.size(), (size_t)sl);
            if ((rc = dTab.reserve(tab.size() * sizeof(ScanChunk) + 16))) return rc;
            if ((rc = dPart.reserve((size_t)nb * 5 * F * sizeof(double)))) return rc;
            if ((rc = dOut.reserve((size_t)5 * F * sizeof(double) + 16))) return rc;
            int* dflag = reinterpret_cast<int*>(dOut.as<double>() + (size_t)5 * F);
            MSM_HIP_CHECK(hipMemcpyAsync(dTab.p, tab.data(), tab.size() * sizeof(ScanChunk), hipMemcpyHostToDevice, stream()));
            MSM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), stream()));
            ScanArgs P;
            P.chunks = dTab.as<ScanChunk>();
            P.nchunks = (long long)tab.size();
            P.ld = group_ld;
            P.F = F;
            P.part = dPart.as<double>();
            P.flag = dflag;
            if (dtype_bytes == 4)
                hipLaunchKernelGGL(colstats_kernel<float>, dim3(nb), dim3(PNT), 0, stream(), P);
            else
                hipLaunchKernelGGL(colstats_kernel<double>, dim3(nb), dim3(PNT), 0, stream(), P);
            MSM_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(colstats_merge_kernel, dim3((unsigned)ceil_div(F, 128)), dim3(128), 0, stream(),
                               P.part, nb, F, dOut.as<double>());
            MSM_HIP_CHECK(hipGetLastError());
            std::vector<double> h((size_t)5 * F + 2);
            MSM_HIP_CHECK(hipMemcpyAsync(h.data(), dOut.p, (size_t)5 * F * sizeof(double) + sizeof(int), hipMemcpyDeviceToHost, stream()));
            MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // also: `tab` and the staging buffer are free again
            int f;
            memcpy(&f, h.data() + (size_t)5 * F, sizeof(int));
            any_inf |= f;
            for (int c = 0; c < F; ++c) {  // Chan merge of this group into the running statistics
                const double nb_ = h[c];
                if (nb_ == 0.0) continue;
                const double na = acc[c], tot = na + nb_;
                const double delta = h[(size_t)F + c] - acc[(size_t)F + c];
                const double w = nb_ / tot;
                acc[(size_t)F + c] += delta * w;
                acc[(size_t)2 * F + c] += h[(size_t)2 * F + c] + delta * delta * na * w;
                acc[c] = tot;
                acc[(size_t)3 * F + c] = std::min(acc[(size_t)3 * F + c], h[(size_t)3 * F + c]);
                acc[(size_t)4 * F + c] = std::max(acc[(size_t)4 * F + c], h[(size_t)4 * F + c]);
            }
        }
        s = e;
    }
    memcpy(out5F, acc.data(), (size_t)5 * F * sizeof(double));
    if (has_inf) *has_inf = any_inf;
    return MSM_OK;
}

int msm_scale_apply(const void* X, int dtype_bytes, msm_idx_t n_rows, msm_idx_t n_features, msm_idx_t ld,
                    const double* shift, const double* scale, int mode, void* out, msm_idx_t ld_out, int on_device)
{
    if (mode != 0 && mode != 1) return fail(MSM_ERR_INVALID, "msm_scale_apply: mode must be 0 or 1");
    if (!X || !out) return fail(MSM_ERR_INVALID, "msm_scale_apply: null pointer");
    if (dtype_bytes != 4 && dtype_bytes != 8) return fail(MSM_ERR_INVALID, "dtype_bytes must be 4 or 8");
    if (n_rows < 0 || n_features < 1 || ld < n_features || ld_out < n_features) return fail(MSM_ERR_INVALID, "msm_scale_apply: bad shape");
    if (n_rows == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int F = (int)n_features;
    int rc;
    DevBuf &dPar = pool(PS_PAR), &dX = pool(PS_X), &dO = pool(PS_OUT);
    if ((rc = dPar.reserve((size_t)2 * F * sizeof(double)))) return rc;
    double* dshift = shift ? dPar.as<double>() : nullptr;
    double* dscale = scale ? dPar.as<double>() + F : nullptr;
    if (shift) MSM_HIP_CHECK(hipMemcpyAsync(dshift, shift, (size_t)F * sizeof(double), hipMemcpyHostToDevice, stream()));
    if (scale) MSM_HIP_CHECK(hipMemcpyAsync(dscale, scale, (size_t)F * sizeof(double), hipMemcpyHostToDevice, stream()));
    const void* xin = X;
    void* xout = out;
    long long ldi = ld, ldo = ld_out;
    const size_t row_bytes = (size_t)F * dtype_bytes;
    if (!on_device) {
        if ((rc = dX.reserve((size_t)n_rows * row_bytes))) return rc;
        if ((rc = dO.reserve((size_t)n_rows * row_bytes))) return rc;
        if (ld == F) {
            if ((rc = h2d_bulk(dX.p, X, (size_t)n_rows * row_bytes))) return rc;
        } else {
            MSM_HIP_CHECK(hipMemcpy2DAsync(dX.p, row_bytes, X, (size_t)ld * dtype_bytes, row_bytes, (size_t)n_rows,
                                           hipMemcpyHostToDevice, stream()));
        }
        xin = dX.p;
        xout = dO.p;
        ldi = ldo = F;
    }
    const int CW = 16 / dtype_bytes;
    const int vec = (F % CW == 0) && (ldi % CW == 0) && (ldo % CW == 0) && ((((uintptr_t)xin | (uintptr_t)xout) & 15) == 0);
    int cpb = 1;
    while (cpb < ceil_div(F, CW) && cpb < PNT) cpb <<= 1;
    const long long rows_per_block = (long long)(PNT / cpb) * 4;
    const unsigned grid = (unsigned)std::min<long long>(ceil_div(n_rows, rows_per_block), 8LL * 2048);
    if (dtype_bytes == 4)
        hipLaunchKernelGGL(scale_apply_kernel<float>, dim3(grid), dim3(PNT), 0, stream(), (const float*)xin, (long long)n_rows, F,
                           ldi, dshift, dscale, (float*)xout, ldo, vec, mode);
    else
        hipLaunchKernelGGL(scale_apply_kernel<double>, dim3(grid), dim3(PNT), 0, stream(), (const double*)xin, (long long)n_rows,
                           F, ldi, dshift, dscale, (double*)xout, ldo, vec, mode);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device) {
        if (ld_out == F) {
            if ((rc = d2h_bulk(out, xout, (size_t)n_rows * row_bytes))) return rc;
        } else {
            MSM_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)ld_out * dtype_bytes, xout, row_bytes, row_bytes, (size_t)n_rows,
                                           hipMemcpyDeviceToHost, stream()));
        }
    }
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // shift/scale staging and scratch are reused
    return MSM_OK;
}

int msm_col_digit_hist(const void* const* X_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                       msm_idx_t n_features, msm_idx_t ld, const uint64_t* prefix, int n_targets, int shift, int bits,
                       int64_t* hist)
{
    if (n_seq < 0 || (n_seq > 0 && (!X_ptrs || !n_rows)) || !hist) return fail(MSM_ERR_INVALID, "msm_col_digit_hist: bad argument");
    if (dtype_bytes != 4 && dtype_bytes != 8) return fail(MSM_ERR_INVALID, "dtype_bytes must be 4 or 8");
    if (n_features < 1 || ld < n_features || n_targets < 1 || bits < 1 || bits > 16 || shift < 0 || shift + bits > dtype_bytes * 8)
        return fail(MSM_ERR_INVALID, "msm_col_digit_hist: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int F = (int)n_features;
    const int R = prefix ? n_targets : 1;
    const size_t nh = (size_t)R * F * ((size_t)1 << bits);
    int rc;
    DevBuf &dTab = pool(PS_IDS), &dHist = pool(PS_OUT), &dPre = pool(PS_PAR);
    std::vector<ScanChunk> tab;
    for (msm_idx_t i = 0; i < n_seq; ++i) {
        if (n_rows[i] < 0 || (n_rows[i] > 0 && !X_ptrs[i])) return fail(MSM_ERR_INVALID, "msm_col_digit_hist: bad sequence %lld", (long long)i);
        for (long long r0 = 0; r0 < n_rows[i]; r0 += 4096) {
            ScanChunk ch;
            ch.base = (const char*)X_ptrs[i] + (size_t)r0 * (size_t)ld * dtype_bytes;
            ch.n = std::min<long long>(4096, n_rows[i] - r0);
            tab.push_back(ch);
        }
    }
    if ((rc = dHist.reserve(nh * sizeof(int64_t)))) return rc;
    MSM_HIP_CHECK(hipMemsetAsync(dHist.p, 0, nh * sizeof(int64_t), stream()));
    if (!tab.empty()) {
        if ((rc = dTab.reserve(tab.size() * sizeof(ScanChunk) + 16))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dTab.p, tab.data(), tab.size() * sizeof(ScanChunk), hipMemcpyHostToDevice, stream()));
        DigitArgs P;
        P.chunks = dTab.as<ScanChunk>();
        P.nchunks = (long long)tab.size();
        P.ld = ld;
        P.F = F;
        P.R = R;
        P.prefix = nullptr;
        if (prefix) {
            if ((rc = dPre.reserve((size_t)R * F * sizeof(uint64_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dPre.p, prefix, (size_t)R * F * sizeof(uint64_t), hipMemcpyHostToDevice, stream()));
            P.prefix = dPre.as<unsigned long long>();
        }
        P.shift = shift;
        P.bits = bits;
        P.hist = dHist.as<unsigned long long>();
        const int nb = (int)std::min<size_t>(tab.size(), (size_t)PNB);
        if (!prefix && bits <= 11 && shift + bits == dtype_bytes * 8) {
            // first pass: workgroup-private LDS histograms (a workgroup's count of one bin stays below 2^32)
            const size_t lds = (size_t)DCOLS * ((size_t)1 << bits) * sizeof(unsigned);
            const int ntile = (int)ceil_div(F, DCOLS);
            const int nparts = (int)std::max<long long>(1, std::min<long long>((long long)tab.size(), (2LL * num_cus()) / ntile));
            static bool attr_set = false;
            if (!attr_set) {
                MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(col_digit_lds_kernel<float>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(col_digit_lds_kernel<double>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                attr_set = true;
            }
            if (dtype_bytes == 4)
                hipLaunchKernelGGL(col_digit_lds_kernel<float>, dim3(ntile * nparts), dim3(PNT), lds, stream(), P);
            else
                hipLaunchKernelGGL(col_digit_lds_kernel<double>, dim3(ntile * nparts), dim3(PNT), lds, stream(), P);
        } else if (dtype_bytes == 4) {
            hipLaunchKernelGGL(col_digit_kernel<float>, dim3(nb), dim3(PNT), 0, stream(), P);
        } else {
            hipLaunchKernelGGL(col_digit_kernel<double>, dim3(nb), dim3(PNT), 0, stream(), P);
        }
        MSM_HIP_CHECK(hipGetLastError());
    }
    MSM_HIP_CHECK(hipMemcpyAsync(hist, dHist.p, nh * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"
