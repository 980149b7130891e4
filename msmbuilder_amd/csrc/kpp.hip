// kpp.hip -- greedy k-means++ seeding of MiniBatchKMeans on the device.
//
// Replaces the host loop that mirrored scikit-learn's `_kmeans_plusplus` (sklearn/cluster/_kmeans.py:163-259, reached from
// msmbuilder/cluster/__init__.py:67-69): K - 1 rounds, each one an inverse-CDF draw of `2 + log K` candidates from the
// current squared distances, their distances to every row, and the choice of the candidate with the lowest potential.
// On the host that is O(K x rows x F) numpy work -- 2.3 of the 2.6 seconds of a batch-65,536 fit on a [10M, 10]
// projection (init_size = 196,608 rows; profiles/r04_mbk65536_host_profile.txt).  Here a round is four small launches
// queued back to back, with no host synchronisation inside the loop: the uniforms of all rounds are drawn up front from
// the SAME RandomState stream (`uniform(size=(K - 1, L))` draws element by element like K - 1 calls of `uniform(size=L)`).
//
// Arithmetic: scikit-learn evaluates float32 rows through `_euclidean_distances_upcast` -- float64 `-2 x.c + ||c||^2 + ||x||^2`,
// clamped at 0, rounded to float32 -- and so does kpp_dist_kernel (a float64 FMA chain over the features; the host loop
// this replaces used a float32 sgemm).  The cumulative sums are float64 like `stable_cumsum`'s, formed blockwise instead
// of sequentially; the potentials are float64 sums (scikit-learn: a float32 dot).  A draw can therefore fall into the
// neighbouring row only when it lands within ~1e-15 of a bin edge, and the arg-min over the candidates can differ only
// between potentials that agree to ~1e-7.  What that means against scikit-learn itself: its current potential is a FLOAT32
// sum whose value depends on the summation order of its BLAS / numpy build, and `rand_vals = uniform * current_pot` moves
// with it -- on samples of a few thousand rows the draws land in the same bins and the seeds are scikit-learn's row for row
// (tests/golden/mbkm_golden.npz: 20,000 x 10, K = 200, captured from scikit-learn 1.7.2; test_kmeans_plusplus_matches_sklearn),
// on the bench's 196,608-row seeding sample they are the seeds of the float64 restatement the tests keep beside it and differ
// from the installed scikit-learn's in some rounds (tests/test_gpu_kmeans.py::test_device_kmeans_plusplus_large_sample
// bounds the potential within 5 %).  INTEGRATION.md section 2 states this where an integrator reads it.
//
// Shapes (round 5): candidate rows that do not fit the 60 KB staging tile (L x F x 4 bytes: e.g. K >= 403 at
// F > 1875) are gathered into a device buffer and read through the L2 instead; seeding samples beyond 7.3M rows keep the
// block prefix of the draw in device memory instead of LDS.  Both used to be refused.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace msm {

namespace {

constexpr int KPP_NT = 256;
constexpr int KPP_RPB = 1024;   // rows per block (4 per thread)
constexpr int KPP_LMAX = 16;    // candidates per round: 2 + log K <= 16 up to K = 1.2 million

template <typename T>   // T: the rows' type.  scikit-learn keeps distances, potentials' inputs and centres in it (float32 rows:
struct KppArgsT {       // float64-upcast arithmetic rounded to float32; float64 rows: float64 throughout)
    const T* X;          // [n][F]
    long long n;
    int F, L;
    T* dbuf;             // [2][L][n] candidate distances (after the min with `closest`), double-buffered over rounds
    double* cuml;        // [n] inclusive scan of `closest` inside each block
    double* bsum;        // [nb] block totals
    double* ppart;       // [nb][L] per-block potentials of the candidates
    const double* u;     // [K - 1][L] uniforms
    long long* cand;     // [L] candidate rows of the round
    int* best;           // [1] index (0 .. L - 1) of the previous round's winner inside dbuf
    double* pot;         // [1] current potential (rounded to T like scikit-learn's)
    T* centers;          // [K][F]
    long long* ids;      // [K]
    T* cglob;            // [L][F] the round's candidate rows when they do not fit the LDS tile (else null)
    double* bpre;        // [nb] prefix of the block totals when nb doubles do not fit LDS (else null)
};

// `closest` of round r = the winning candidate's row of the previous round's buffer
template <typename T>
__device__ __forceinline__ const T* kpp_closest(const KppArgsT<T>& P, int round)
{
    return P.dbuf + ((size_t)((round + 1) & 1) * P.L + (size_t)(*P.best)) * (size_t)P.n;
}

// 1) blockwise float64 inclusive scan of `closest`
template <typename T>
__global__ __launch_bounds__(KPP_NT) void kpp_scan_kernel(KppArgsT<T> P, int round)
{
    __shared__ double wsum[KPP_NT / 64];
    const T* closest = kpp_closest(P, round);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long i0 = (long long)blockIdx.x * KPP_RPB + tid * 4;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q < P.n ? (double)closest[i0 + q] : 0.0;
    v[1] += v[0];
    v[2] += v[1];
    v[3] += v[2];
    double s = v[3];   // inclusive scan of the thread totals over the wave, then over the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double t = __shfl_up(s, d, 64);
        if (lane >= d) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    double off = s - v[3];
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i0 + q < P.n) P.cuml[i0 + q] = off + v[q];
    if (tid == KPP_NT - 1) P.bsum[blockIdx.x] = off + v[3];
}

// 2) the round's candidates: searchsorted(cumsum(closest), u * pot), side = 'left', clipped to n - 1
template <typename T>
__global__ __launch_bounds__(KPP_NT) void kpp_pick_kernel(KppArgsT<T> P, int round, int nb)
{
    extern __shared__ double pre_lds[];   // [nb] inclusive prefix of the block totals (or P.bpre: seeding samples beyond 7.3M rows)
    double* pre = P.bpre ? P.bpre : pre_lds;
    const int tid = threadIdx.x;
    // sequential prefix by chunks: thread t owns blocks [t * per, (t + 1) * per)
    const int per = (nb + KPP_NT - 1) / KPP_NT;
    __shared__ double tsum[KPP_NT];
    double a = 0.0;
    for (int b = tid * per; b < nb && b < (tid + 1) * per; ++b) a += P.bsum[b];
    tsum[tid] = a;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int t = 0; t < KPP_NT; ++t) {
            const double x = tsum[t];
            tsum[t] = run;
            run += x;
        }
    }
    __syncthreads();
    a = tsum[tid];
    for (int b = tid * per; b < nb && b < (tid + 1) * per; ++b) {
        a += P.bsum[b];
        pre[b] = a;
    }
    __syncthreads();
    if (tid < P.L) {
        const double r = P.u[(size_t)(round - 1) * P.L + tid] * *P.pot;
        int lo = 0, hi = nb;   // first block whose inclusive prefix reaches r
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        long long id = P.n - 1;
        if (lo < nb) {
            const double base = lo ? pre[lo - 1] : 0.0;
            const long long b0 = (long long)lo * KPP_RPB;
            const long long cnt = P.n - b0 < KPP_RPB ? P.n - b0 : KPP_RPB;
            long long l = 0, h = cnt;
            while (l < h) {
                const long long mid = (l + h) >> 1;
                if (base + P.cuml[b0 + mid] < r) l = mid + 1;
                else h = mid;
            }
            id = b0 + (l < cnt ? l : cnt - 1);
        }
        P.cand[tid] = id;
    }
}

// 3) distances of every row to the candidates, the min with `closest`, per-block potentials
template <typename T, bool FIRST>
__global__ __launch_bounds__(KPP_NT) void kpp_dist_kernel(KppArgsT<T> P, int round, long long first)
{
    extern __shared__ __attribute__((aligned(16))) char cs_lds_raw[];   // [L][F] candidate rows (or P.cglob, gathered by kpp_gather_kernel: wide rows)
    T* cs_lds = reinterpret_cast<T*>(cs_lds_raw);
    __shared__ double cc[KPP_LMAX];
    __shared__ double red[KPP_LMAX][KPP_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = FIRST ? 1 : P.L, F = P.F;
    const T* cs = P.cglob ? P.cglob : cs_lds;
    if (!P.cglob) {
        for (int e = tid; e < L * F; e += KPP_NT) {
            const int j = e / F, f = e - j * F;
            const long long row = FIRST ? first : P.cand[j];
            cs_lds[e] = P.X[(size_t)row * F + f];
        }
    }
    __syncthreads();
    if (tid < L) {
        double s = 0.0;
        for (int f = 0; f < F; ++f) s = fma((double)cs[tid * F + f], (double)cs[tid * F + f], s);
        cc[tid] = s;
    }
    __syncthreads();
    const T* closest = FIRST ? nullptr : kpp_closest(P, round);
    T* out = P.dbuf + (size_t)(round & 1) * P.L * (size_t)P.n;
    double pot[KPP_LMAX];
#pragma unroll
    for (int j = 0; j < KPP_LMAX; ++j) pot[j] = 0.0;
    for (int q = 0; q < 4; ++q) {
        const long long i = (long long)blockIdx.x * KPP_RPB + q * KPP_NT + tid;   // coalesced over the block
        if (i >= P.n) continue;
        const T* x = P.X + (size_t)i * F;
        double xx = 0.0, dot[KPP_LMAX];
#pragma unroll
        for (int j = 0; j < KPP_LMAX; ++j) dot[j] = 0.0;
        for (int f = 0; f < F; ++f) {
            const double xv = (double)x[f];
            xx = fma(xv, xv, xx);
#pragma unroll
            for (int j = 0; j < KPP_LMAX; ++j)
                if (j < L) dot[j] = fma(xv, (double)cs[j * F + f], dot[j]);
        }
        const T cl = FIRST ? (T)INFINITY : closest[i];
#pragma unroll
        for (int j = 0; j < KPP_LMAX; ++j)
            if (j < L) {
                double d = -2.0 * dot[j];   // scikit-learn: d = -2 X Y^T; d += XX (candidates); d += YY (rows); max(d, 0); (float32 rows: -> float32)
                d += cc[j];
                d += xx;
                T df = (T)(d > 0.0 ? d : 0.0);
                df = df < cl ? df : cl;
                out[(size_t)j * P.n + i] = df;
                pot[j] += (double)df;
            }
    }
#pragma unroll
    for (int j = 0; j < KPP_LMAX; ++j)
        if (j < L) {
            double s = pot[j];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
            if (lane == 0) red[j][wave] = s;
        }
    __syncthreads();
    if (tid < L) P.ppart[(size_t)blockIdx.x * P.L + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// 3w) the same for rows of 32 features and more: ONE WAVE per row, lanes over the features (16-byte loads when the rows allow
// it), 32 rows per workgroup.  kpp_dist_kernel gives a row to a THREAD: its loads are strided by the row length and a
// seeding sample of 3 x batch_size rows is a grid of a few dozen workgroups -- 1.1 ms per round on 24,576 x 512 (50 MB), a
// MiniBatchKMeans(k=1000) fit of 1M x 512 spent 1.1 of its 1.28 s there (scripts/clusterprobe.py).  Same float64 formula;
// the features are summed lane-strided and by a butterfly instead of front to back (scikit-learn's order is its BLAS's).
constexpr int KPP_WROWS = 32;   // rows per workgroup (8 per wave)
template <typename T, bool FIRST>
__global__ __launch_bounds__(KPP_NT) void kpp_dist_wide_kernel(KppArgsT<T> P, int round, long long first)
{
    extern __shared__ __attribute__((aligned(16))) char cs_lds_w_raw[];   // [L][F] candidate rows (or P.cglob)
    T* cs_lds_w = reinterpret_cast<T*>(cs_lds_w_raw);
    constexpr int E = 16 / (int)sizeof(T);   // elements of a 16-byte load
    struct alignas(16) V16 { T e[E]; };
    __shared__ double cc[KPP_LMAX];
    __shared__ double red[KPP_LMAX][KPP_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = FIRST ? 1 : P.L, F = P.F;
    const T* cs = P.cglob ? P.cglob : cs_lds_w;
    if (!P.cglob) {
        for (int e = tid; e < L * F; e += KPP_NT) {
            const int j = e / F, f = e - j * F;
            const long long row = FIRST ? first : P.cand[j];
            cs_lds_w[e] = P.X[(size_t)row * F + f];
        }
    }
    __syncthreads();
    for (int j = wave; j < L; j += KPP_NT / 64) {
        double s = 0.0;
        for (int f = lane; f < F; f += 64) s = fma((double)cs[j * F + f], (double)cs[j * F + f], s);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) cc[j] = s;
    }
    __syncthreads();
    const T* closest = FIRST ? nullptr : kpp_closest(P, round);
    T* out = P.dbuf + (size_t)(round & 1) * P.L * (size_t)P.n;
    const bool vec4 = (F % E) == 0 && ((((uintptr_t)P.X) | ((uintptr_t)cs)) & 15) == 0;
    double pot[KPP_LMAX];
#pragma unroll
    for (int j = 0; j < KPP_LMAX; ++j) pot[j] = 0.0;
    for (int q = 0; q < KPP_WROWS / (KPP_NT / 64); ++q) {
        const long long i = (long long)blockIdx.x * KPP_WROWS + q * (KPP_NT / 64) + wave;   // uniform over the wave
        if (i >= P.n) break;
        const T* x = P.X + (size_t)i * F;
        double xx = 0.0, dot[KPP_LMAX];
#pragma unroll
        for (int j = 0; j < KPP_LMAX; ++j) dot[j] = 0.0;
        if (vec4) {
            for (int f4 = lane; f4 < F / E; f4 += 64) {
                const V16 xv = reinterpret_cast<const V16*>(x)[f4];
#pragma unroll
                for (int e = 0; e < E; ++e) xx = fma((double)xv.e[e], (double)xv.e[e], xx);
#pragma unroll
                for (int j = 0; j < KPP_LMAX; ++j)
                    if (j < L) {
                        const V16 cv = reinterpret_cast<const V16*>(cs + (size_t)j * F)[f4];
#pragma unroll
                        for (int e = 0; e < E; ++e) dot[j] = fma((double)xv.e[e], (double)cv.e[e], dot[j]);
                    }
            }
        } else {
            for (int f = lane; f < F; f += 64) {
                const double xv = (double)x[f];
                xx = fma(xv, xv, xx);
#pragma unroll
                for (int j = 0; j < KPP_LMAX; ++j)
                    if (j < L) dot[j] = fma(xv, (double)cs[(size_t)j * F + f], dot[j]);
            }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) xx += __shfl_xor(xx, m, 64);
        const T cl = FIRST ? (T)INFINITY : closest[i];
#pragma unroll
        for (int j = 0; j < KPP_LMAX; ++j)
            if (j < L) {
                double dj = dot[j];
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) dj += __shfl_xor(dj, m, 64);
                double d = -2.0 * dj;   // scikit-learn's order of the three terms (kpp_dist_kernel)
                d += cc[j];
                d += xx;
                T df = (T)(d > 0.0 ? d : 0.0);
                df = df < cl ? df : cl;
                if (lane == 0) {
                    out[(size_t)j * P.n + i] = df;
                    pot[j] += (double)df;
                }
            }
    }
#pragma unroll
    for (int j = 0; j < KPP_LMAX; ++j)
        if (j < L && lane == 0) red[j][wave] = pot[j];
    __syncthreads();
    if (tid < L) P.ppart[(size_t)blockIdx.x * P.L + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// 3') wide rows: the round's candidate rows into one device buffer (read by every block of kpp_dist_kernel through the L2)
// (`round0`: the one candidate of round 0 is row `first`; every later round reads P.cand -- also when it has ONE candidate:
//  ADVICE r5, the kernel used to take `L == 1` for "round 0")
template <typename T>
__global__ __launch_bounds__(KPP_NT) void kpp_gather_kernel(KppArgsT<T> P, int round0, long long first)
{
    const int j = blockIdx.x;
    const long long row = round0 ? first : P.cand[j];
    for (int f = threadIdx.x; f < P.F; f += KPP_NT) P.cglob[(size_t)j * P.F + f] = P.X[(size_t)row * P.F + f];
}

// 4) potentials -> the winner (first minimum), the new centre, the new current potential
template <typename T>
__global__ __launch_bounds__(KPP_NT) void kpp_best_kernel(KppArgsT<T> P, int round, int nb, long long first)
{
    __shared__ double part[KPP_LMAX][KPP_NT];
    __shared__ int bsel;
    const int tid = threadIdx.x;
    const int L = round == 0 ? 1 : P.L;
    for (int j = 0; j < L; ++j) {
        double a = 0.0;
        for (int b = tid; b < nb; b += KPP_NT) a += P.ppart[(size_t)b * P.L + j];
        part[j][tid] = a;
    }
    __syncthreads();
    for (int s = KPP_NT / 2; s > 0; s >>= 1) {
        if (tid < s)
            for (int j = 0; j < L; ++j) part[j][tid] += part[j][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        int b = 0;
        for (int j = 1; j < L; ++j)
            if ((T)part[j][0] < (T)part[b][0]) b = j;   // potentials in the rows' type like scikit-learn's, first minimum
        bsel = b;
        *P.best = b;
        *P.pot = (double)(T)part[b][0];
        P.ids[round] = round == 0 ? first : P.cand[b];
    }
    __syncthreads();
    const long long row = round == 0 ? first : P.cand[bsel];
    for (int f = tid; f < P.F; f += KPP_NT) P.centers[(size_t)round * P.F + f] = P.X[(size_t)row * P.F + f];
}

}  // namespace

}  // namespace msm

using namespace msm;

namespace {

/* k-means++ seeds of the n x F rows X (host or device per on_device): centre 0 = row `first`, then K - 1 rounds with
 * L candidates each, drawn with the uniforms u[(K - 1) * L] (host, float64, in [0, 1)): scikit-learn's `_kmeans_plusplus`
 * given the same draws.  centers[K * F] (the rows' type) and ids[K] (rows of X) are host arrays. */
template <typename T>
int kmeans_plusplus_t(const T* X, msm_idx_t n, msm_idx_t F, msm_idx_t K, msm_idx_t first, const double* u, int L,
                      T* centers, msm_idx_t* ids, int on_device)
{
    if (!X || !centers || !ids || (K > 1 && !u)) return fail(MSM_ERR_INVALID, "kmeans_plusplus: null pointer");
    if (n < 1 || F < 1 || K < 1 || K > n || first < 0 || first >= n) return fail(MSM_ERR_INVALID, "kmeans_plusplus: bad shape");
    if (L < 1 || L > KPP_LMAX) return fail(MSM_ERR_INVALID, "kmeans_plusplus: 1 <= candidates per round <= %d", KPP_LMAX);
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    if (ceil_div(n, KPP_RPB) > 0x7fffffffLL) return fail(MSM_ERR_INVALID, "kmeans_plusplus: too many rows");
    const int nb = (int)ceil_div(n, KPP_RPB);
    const bool wrows = F >= 32 && ceil_div(n, KPP_WROWS) <= 0x7fffffffLL;   // a wave per row (kpp_dist_wide_kernel)
    const int nbd = wrows ? (int)ceil_div(n, KPP_WROWS) : nb;                // workgroups (= potential partials) of the distance kernel
    const bool wide = (size_t)L * F * sizeof(T) > 60000;   // candidate rows beyond the LDS staging tile: through a device buffer
    const bool longpre = nb > 7168;                            // block prefix beyond LDS: in device memory
    DevBuf &dX = pool(PS_X), &dW = pool(PS_W), &dO = pool(PS_OUT);
    int rc;
    const T* Xd = X;
    if (!on_device) {
        if ((rc = dX.reserve((size_t)n * F * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * F * sizeof(T)))) return rc;
        Xd = dX.as<T>();
    }
    // workspace: dbuf [2][L][n] T | cuml [n] f64 | bsum [nb] | ppart [nb][L] | u [(K-1) L] | pot | cand [L] i64 | best
    const size_t nu = (size_t)(K - 1) * L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_dbuf = take(2 * (size_t)L * n * sizeof(T)), o_cuml = take((size_t)n * sizeof(double)),
                 o_bsum = take((size_t)nb * sizeof(double)), o_pp = take((size_t)nbd * L * sizeof(double)),
                 o_u = take(std::max<size_t>(nu, 1) * sizeof(double)), o_pot = take(sizeof(double)), o_cand = take(L * sizeof(long long)),
                 o_best = take(sizeof(int)), o_cg = take(wide ? (size_t)L * F * sizeof(T) : 0), o_bpre = take(longpre ? (size_t)nb * sizeof(double) : 0);
    if ((rc = dW.reserve(off))) return rc;
    if ((rc = dO.reserve((size_t)K * F * sizeof(T) + (size_t)K * sizeof(long long)))) return rc;
    char* w = dW.as<char>();
    KppArgsT<T> P;
    P.X = Xd;
    P.n = n;
    P.F = (int)F;
    P.L = L;
    P.dbuf = reinterpret_cast<T*>(w + o_dbuf);
    P.cuml = reinterpret_cast<double*>(w + o_cuml);
    P.bsum = reinterpret_cast<double*>(w + o_bsum);
    P.ppart = reinterpret_cast<double*>(w + o_pp);
    P.u = reinterpret_cast<const double*>(w + o_u);
    P.pot = reinterpret_cast<double*>(w + o_pot);
    P.cand = reinterpret_cast<long long*>(w + o_cand);
    P.best = reinterpret_cast<int*>(w + o_best);
    P.centers = dO.as<T>();
    P.ids = reinterpret_cast<long long*>(dO.as<char>() + (size_t)K * F * sizeof(T));
    P.cglob = wide ? reinterpret_cast<T*>(w + o_cg) : nullptr;
    P.bpre = longpre ? reinterpret_cast<double*>(w + o_bpre) : nullptr;
    if (nu) MSM_HIP_CHECK(hipMemcpyAsync(w + o_u, u, nu * sizeof(double), hipMemcpyHostToDevice, stream()));
    const size_t lds_c = wide ? 0 : (size_t)L * F * sizeof(T), lds_pre = longpre ? 0 : (size_t)nb * sizeof(double);
    // round 0: distances to the first centre = `closest`, its potential
    if (wide) hipLaunchKernelGGL(kpp_gather_kernel<T>, dim3(1), dim3(KPP_NT), 0, stream(), P, 1, (long long)first);
    if (wrows) hipLaunchKernelGGL((kpp_dist_wide_kernel<T, true>), dim3(nbd), dim3(KPP_NT), lds_c, stream(), P, 0, (long long)first);
    else hipLaunchKernelGGL((kpp_dist_kernel<T, true>), dim3(nb), dim3(KPP_NT), lds_c, stream(), P, 0, (long long)first);
    hipLaunchKernelGGL(kpp_best_kernel<T>, dim3(1), dim3(KPP_NT), 0, stream(), P, 0, nbd, (long long)first);
    for (int r = 1; r < (int)K; ++r) {
        hipLaunchKernelGGL(kpp_scan_kernel<T>, dim3(nb), dim3(KPP_NT), 0, stream(), P, r);
        hipLaunchKernelGGL(kpp_pick_kernel<T>, dim3(1), dim3(KPP_NT), lds_pre, stream(), P, r, nb);
        if (wide) hipLaunchKernelGGL(kpp_gather_kernel<T>, dim3((unsigned)L), dim3(KPP_NT), 0, stream(), P, 0, (long long)first);
        if (wrows) hipLaunchKernelGGL((kpp_dist_wide_kernel<T, false>), dim3(nbd), dim3(KPP_NT), lds_c, stream(), P, r, (long long)first);
        else hipLaunchKernelGGL((kpp_dist_kernel<T, false>), dim3(nb), dim3(KPP_NT), lds_c, stream(), P, r, (long long)first);
        hipLaunchKernelGGL(kpp_best_kernel<T>, dim3(1), dim3(KPP_NT), 0, stream(), P, r, nbd, (long long)first);
    }
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemcpyAsync(centers, P.centers, (size_t)K * F * sizeof(T), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(ids, P.ids, (size_t)K * sizeof(long long), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // namespace

extern "C" {

int msm_kmeans_plusplus_f32(const float* X, msm_idx_t n, msm_idx_t F, msm_idx_t K, msm_idx_t first, const double* u, int L,
                            float* centers, msm_idx_t* ids, int on_device)
{
    return kmeans_plusplus_t<float>(X, n, F, K, first, u, L, centers, ids, on_device);
}

int msm_kmeans_plusplus_f64(const double* X, msm_idx_t n, msm_idx_t F, msm_idx_t K, msm_idx_t first, const double* u, int L,
                            double* centers, msm_idx_t* ids, int on_device)
{
    return kmeans_plusplus_t<double>(X, n, F, K, first, u, L, centers, ids, on_device);
}

}  // extern "C"
