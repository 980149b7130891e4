// distance.hip -- exact-arithmetic libdistance kernels (dist / cdist / assign_nearest)
// and the fused k-centers pass for gfx950.
//
// Replaces /root/reference/msmbuilder/libdistance/src/{distance_kernels.h:41-293,
// dist.hpp:4-80, cdist.hpp:4-49, assign.hpp:6-91} and the k-pass loop of
// /root/reference/msmbuilder/cluster/kcenters.py:91-97.
//
// Bit-exactness contract (what makes integer labels identical to the CPU path):
// every (row, centre) pair is owned by ONE lane, which visits the features in
// order i = 0..m-1 with ONE fp64 accumulator; for float inputs u-v / u+v are
// fp32 operations widened afterwards; multiply and add are separately rounded
// (this file is compiled with -ffp-contract=off); euclidean takes the sqrt
// before comparing; comparisons are strict `<` in ascending centre order, so
// the lowest index wins.  Parallelism is over rows (and centre tiles), never
// over the feature axis.  This is HBM/L2- and fp64-VALU-bound work: no MFMA.
//
// Tiling: a workgroup stages a [256 rows x FC features] tile of X through LDS
// (coalesced in, row stride FC+1 so that lane-per-row reads are conflict-free)
// and a [CJ centres x FC] tile of Y (wave-uniform broadcast reads).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "distance_dev.h"

#include "distance_pair_dev.h"   // pair_kernel / pair_small_kernel / assign_small2_kernel: assign_nearest, cdist, dist in exact arithmetic
#include "distance_kcpass_dev.h"   // kcenters_pass_kernel: one fused k-centers pass (exact triangle-inequality pruning)
#include "distance_wide_dev.h"   // wide_kernel: the streaming path for wide rows (assign_nearest / cdist / k-centers pass)
#include "distance_kcsel_dev.h"   // kc_finalize / kc_candidate / kc_select kernels: the argmax exchange of the k-centers loops
#include "distance_pdist_dev.h"   // pdist_kernel, sumdist_kernel, sum_partial_kernel
namespace msm {


// ---- host-side dispatch ----------------------------------------------------
// widest aligned per-lane vector load for a [*, m] row-major array, 0 if the fast path does not apply
template <typename T>
static int row_vecw(const void* X, long long m, bool has_indices)
{
    if (has_indices || m > FeatChunk<T>::FC) return 0;
    const size_t rb = (size_t)m * sizeof(T);
    const uintptr_t a = (uintptr_t)X;
    if (a % 16 == 0 && rb % 16 == 0) return 16;
    if (a % 8 == 0 && rb % 8 == 0) return 8;
    return (a % sizeof(T) == 0) ? (int)sizeof(T) : 0;
}

constexpr size_t wide_lds(int ncl) { return (size_t)2 * DT * WP * 4 + (size_t)2 * ncl * 32 * 4 + (size_t)DT * 16; }

// wide-row streaming path applies: long rows, 16-byte aligned vectors, no row gather
template <typename T>
static bool wide_ok(const void* X, const void* Y, long long m, bool has_indices)
{
    constexpr int E = 16 / (int)sizeof(T);
    return !has_indices && m > FeatChunk<T>::FC && (m % E) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0 &&
           (size_t)m * sizeof(T) < ((size_t)1 << 24);
}

static int wide_grid(long long n)
{
    return (int)std::min<long long>(ceil_div(n, DT), 2LL * num_cus());  // one resident round (2 workgroups / CU)
}

template <typename T, int MM, int MODE, int NCT>
static void launch_wide_nc(int grid, const WideArgs& A)
{
    constexpr size_t lds = wide_lds(MODE == 2 ? WNC : NCT);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_kernel<T, MM, MODE, NCT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wide_kernel<T, MM, MODE, NCT>), dim3(grid), dim3(DT), lds, stream(), A);
}

template <typename T, int MM, int MODE>
static void launch_wide(int grid, const WideArgs& A)
{
    // assign_nearest picks its centre-group size by shape
    if (MODE == 0) {
        if (A.pa.K <= 8) return launch_wide_nc<T, MM, 0, 8>(grid, A);
        // (32-centre groups for float64 rows were measured too: 2M x 256 x K = 100 8.74 ms against 8.31 ms with 16 -- the
        //  4 KiB of extra LDS cost the second workgroup of a CU and more than the halved re-streaming of the row tile gains)
    }
    launch_wide_nc<T, MM, MODE, WNC>(grid, A);
}

template <typename T, int MODE>
void launch_pair(int metric, int grid, const PairArgs& P)
{
    const bool wide = wide_ok<T>(P.X, P.Y, P.m, P.X_indices != nullptr);
    WideArgs A;
    memset(&A, 0, sizeof(A));
    A.pa = P;
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (P.vecw > 0 && MODE == 0 && (MM == M_EUCLIDEAN || MM == M_SQEUCLIDEAN) &&              \
            (sizeof(T) == 4 ? launch_small3_f32(MM, grid, P) : launch_small3_f64(MM, grid, P))) { \
        } else if (P.vecw > 0 && MODE == 0)                                                       \
            hipLaunchKernelGGL((assign_small2_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P); /* every block writes its partial */ \
        else if (P.vecw > 0)                                                                      \
            hipLaunchKernelGGL((pair_small_kernel<T, MM, MODE>), dim3(grid), dim3(DT), 0, stream(), P); \
        else if (wide)                                                                            \
            launch_wide<T, MM, MODE>(grid, A);                                                    \
        else                                                                                      \
            hipLaunchKernelGGL((pair_kernel<T, MM, MODE>), dim3(grid), dim3(DT), 0, stream(), P); \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

template <typename T>
void launch_kc(int metric, int grid, const KcArgs& P)
{
    const bool wide = P.vecw == 0 && wide_ok<T>(P.X, P.ycenter ? P.ycenter : P.X, P.m, false);
    WideArgs A;
    memset(&A, 0, sizeof(A));
    A.kc = P;
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (wide)                                                                                 \
            launch_wide<T, MM, 2>(grid, A);                                                       \
        else if (P.vecw > 0)                                                                      \
            hipLaunchKernelGGL((kcenters_pass_kernel<T, MM, true>), dim3(grid), dim3(DT), 0, stream(), P); \
        else                                                                                      \
            hipLaunchKernelGGL((kcenters_pass_kernel<T, MM, false>), dim3(grid), dim3(DT), 0, stream(), P); \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

template <typename T, int WHICH>
void launch_pd(int metric, int grid, const PdArgs& P)
{
#define MSM_CASE(MM)                                                                              \
    case MM:                                                                                      \
        if (WHICH == 0)                                                                           \
            hipLaunchKernelGGL((pdist_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P);      \
        else                                                                                      \
            hipLaunchKernelGGL((sumdist_kernel<T, MM>), dim3(grid), dim3(DT), 0, stream(), P);    \
        break;
    switch (metric) {
        MSM_CASE(M_EUCLIDEAN)
        MSM_CASE(M_SQEUCLIDEAN)
        MSM_CASE(M_CITYBLOCK)
        MSM_CASE(M_CHEBYSHEV)
        MSM_CASE(M_CANBERRA)
        MSM_CASE(M_BRAYCURTIS)
        MSM_CASE(M_HAMMING)
        MSM_CASE(M_JACCARD)
    }
#undef MSM_CASE
}

static int sum_partials_host(const double* dpartial, int n, double* out)
{
    std::vector<double> h((size_t)n);
    MSM_HIP_CHECK(hipMemcpyAsync(h.data(), dpartial, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += h[i];
    *out = s;
    return MSM_OK;
}

// Rows whose length is not a multiple of 16 bytes (171 float32 contact features: SURVEY 8(d)'s C3 stress variant) cannot take
// the wide-row streaming kernels and fell to the scalar-staged pair kernel -- 20x slower per pair-element (280,000 x 171 x
// K = 200: assign_nearest 32.7 ms, a k-centers pass 134 us at 1.4 TB/s; profiles/r04_pmc_all_first.txt).  For the NORM
// metrics a zero column is an exact no-op in the reference's arithmetic (a float difference 0 - 0 = +0, then s + 0*0 = s,
// s + |0| = s, max(s, 0) = s for s >= 0), so such rows are copied once into a buffer zero-padded to the next multiple of 16
// bytes -- one streaming pass, 2 x the input bytes -- and everything downstream sees aligned rows.  Bit-identical outputs.
template <typename T>
static bool pad_rows_pays(int mid, long long m, long long n, bool has_indices)
{
    constexpr int E = 16 / (int)sizeof(T);
    return !has_indices && (mid == M_EUCLIDEAN || mid == M_SQEUCLIDEAN || mid == M_CITYBLOCK || mid == M_CHEBYSHEV) &&
           m > FeatChunk<T>::FC && (m % E) != 0 && n * m >= (1LL << 20);
}

template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ X, long long n, long long m, long long mp, T* __restrict__ out)
{
    constexpr int E = 16 / (int)sizeof(T);
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x, per = mp / E;   // 16-byte groups
    if (g >= n * per) return;
    const long long i = g / per, c = (g - i * per) * E;
    T v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = c + e < m ? X[i * m + c + e] : (T)0;
#pragma unroll
    for (int e = 0; e < E; ++e) out[i * mp + c + e] = v[e];
}

template <typename T>
static int pad_rows(const T* X, long long n, long long m, DevBuf& buf, const T** out, long long* mp_out)
{
    constexpr int E = 16 / (int)sizeof(T);
    const long long mp = (m + E - 1) / E * E;
    int rc = buf.reserve((size_t)std::max<long long>(n, 1) * mp * sizeof(T));
    if (rc) return rc;
    if (n > 0) {
        const long long groups = n * (mp / E);
        hipLaunchKernelGGL(pad_rows_kernel<T>, dim3((unsigned)ceil_div(groups, 256)), dim3(256), 0, stream(), X, n, m, mp, buf.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
    }
    *out = buf.as<T>();
    *mp_out = mp;
    return MSM_OK;
}

template <typename T>
int assign_nearest_impl(const T* X, const T* Y, const char* metric, const msm_idx_t* X_indices,
                        msm_idx_t n_X, msm_idx_t n_Y, msm_idx_t m, msm_idx_t n_idx,
                        msm_idx_t* assignments, double* min_dist, double* inertia, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !Y || !assignments) return fail(MSM_ERR_INVALID, "assign_nearest: null pointer");
    if (n_X < 0 || n_Y < 0 || m < 1) return fail(MSM_ERR_INVALID, "assign_nearest: bad shape");
    const long long n = X_indices ? n_idx : n_X;
    if (inertia) *inertia = 0.0;
    if (n == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dY = pool(PS_Y), &dIdx = pool(PS_IDX), &dLab = pool(PS_LAB), &dMin = pool(PS_MIN),
           &dPart = pool(PS_PART);
    int grid = (int)std::min<long long>(ceil_div(n, DT), 2048);
    if ((rc = dY.reserve((size_t)(n_Y ? n_Y : 1) * m * sizeof(T)))) return rc;
    if (n_Y) MSM_HIP_CHECK(hipMemcpyAsync(dY.p, Y, (size_t)n_Y * m * sizeof(T), hipMemcpyHostToDevice, stream()));
    if ((rc = dPart.reserve((size_t)grid * sizeof(double)))) return rc;
    PairArgs P;
    memset(&P, 0, sizeof(P));
    P.Y = dY.p;
    P.n = n;
    P.K = n_Y;
    P.m = m;
    P.partial = dPart.as<double>();
    if (on_device) {
        P.X = X;
        P.X_indices = X_indices;
        P.labels = assignments;
        P.min_dist = min_dist;
    } else {
        if ((rc = dX.reserve((size_t)n_X * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n_X * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)n * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dLab.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
        P.labels = dLab.as<msm_idx_t>();
        if (min_dist) {
            if ((rc = dMin.reserve((size_t)n * sizeof(double)))) return rc;
            P.min_dist = dMin.as<double>();
        }
    }
    if (pad_rows_pays<T>(mid, m, n_X, P.X_indices != nullptr)) {
        const T *xp = nullptr, *yp = nullptr;
        long long mp = m;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.X), n_X, m, pool(PS_PADX), &xp, &mp))) return rc;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.Y), n_Y, m, pool(PS_PADY), &yp, &mp))) return rc;
        P.X = xp;
        P.Y = yp;
        P.m = m = mp;
    }
    P.vecw = row_vecw<T>(P.X, m, P.X_indices != nullptr);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.Y, m, P.X_indices != nullptr)) grid = wide_grid(n);
    launch_pair<T, 0>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device) {
        MSM_HIP_CHECK(hipMemcpyAsync(assignments, P.labels, (size_t)n * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
        if (min_dist)
            MSM_HIP_CHECK(hipMemcpyAsync(min_dist, P.min_dist, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    double s = 0.0;
    if ((rc = sum_partials_host(P.partial, grid, &s))) return rc;  // synchronises
    if (inertia) *inertia = s;
    return MSM_OK;
}

template <typename T>
int cdist_impl(const T* XA, const T* XB, const char* metric, msm_idx_t na, msm_idx_t nb,
               msm_idx_t m, const msm_idx_t* X_indices, msm_idx_t n_idx, double* out, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!XA || !XB || !out) return fail(MSM_ERR_INVALID, "cdist/dist: null pointer");
    if (na < 0 || nb < 0 || m < 1) return fail(MSM_ERR_INVALID, "cdist/dist: bad shape");
    const long long n = X_indices ? n_idx : na;
    if (n == 0 || nb == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dY = pool(PS_Y), &dIdx = pool(PS_IDX), &dOut = pool(PS_OUT);
    int grid = (int)std::min<long long>(ceil_div(n, DT), 2048);
    if ((rc = dY.reserve((size_t)nb * m * sizeof(T)))) return rc;
    if ((rc = h2d_bulk(dY.p, XB, (size_t)nb * m * sizeof(T)))) return rc;
    PairArgs P;
    memset(&P, 0, sizeof(P));
    P.Y = dY.p;
    P.n = n;
    P.K = nb;
    P.m = m;
    if (on_device) {
        P.X = XA;
        P.X_indices = X_indices;
        P.out = out;
    } else {
        if ((rc = dX.reserve((size_t)na * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, XA, (size_t)na * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)n * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dOut.reserve((size_t)n * nb * sizeof(double)))) return rc;
        P.out = dOut.as<double>();
    }
    P.vecw = row_vecw<T>(P.X, m, P.X_indices != nullptr);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.Y, m, P.X_indices != nullptr)) grid = wide_grid(n);
    launch_pair<T, 1>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device)
        MSM_HIP_CHECK(hipMemcpyAsync(out, P.out, (size_t)n * nb * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // scratch (and the staged Y) die with this frame
    return MSM_OK;
}

}  // namespace msm
#include "distance_screen_dev.h"   // kcenters_screen_pass_kernel, ksc_convert_kernel: k-centers passes screened on a low-precision copy
#include "distance_kcbatch_dev.h"   // kcb_* kernels: several centres per pass (threshold lists), single GPU and row-sharded
#include "distance_wscreen_dev.h"   // kcenters_wscreen_pass_kernel: screened passes of wide rows / float32 rows on a feature-major byte copy
#include "distance_wbatch_dev.h"    // kwb_*: several centres per screened pass of wide rows (threshold lists)

namespace msm {

template <typename T, int JB, int R, int U>
static void launch_kwb(int grid, size_t lds, const KwsArgs& A, KcbState* St)
{
    static bool attr_set = false;
    if (!attr_set) {   // up to 64 KiB of centres beside the kernel's own ~40 KiB
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kwb_pass_kernel<T, JB, R, U>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    hipLaunchKernelGGL((kwb_pass_kernel<T, JB, R, U>), dim3(grid), dim3(DT), lds, stream(), A, St);
}



// what the last k-centers fit streamed (for bench.py's bytes-per-pass figure): pass counts and the bytes a pass reads per row
struct KcStats {
    long long rows = 0, plain_passes = 0, screened_passes = 0, plain_row_bytes = 0, screen_row_bytes = 0, batch_fallbacks = 0;
    long long wide_candidates = 0, wide_updates = 0;   // wide screened passes: rows re-evaluated exactly / rows that changed
};
static KcStats g_kc_stats;

// The screen copy in use is FMT 2 (bytes + a row scale); the float32 / bfloat16 formats of the kernels' FMT parameter were
// measured in round 3 and are no longer instantiated.
constexpr int KSC_FMT = 2;

struct KscBufs {
    DevBuf xf, curf, misc;
};
static KscBufs& ksc_bufs()
{
    static KscBufs b;
    return b;
}

// out[k][:] = X[ids[k]][:] -- the chosen rows themselves (cluster_centers_), gathered on the device so that they travel
// with the ids in the fit's one final synchronisation (torch indexing with a Python list is three round trips)
template <typename T>
__global__ void kc_gather_centres_kernel(const T* __restrict__ X, long long m, const msm_idx_t* __restrict__ ids, T* __restrict__ out)
{
    const msm_idx_t row = ids[blockIdx.x];
    for (long long f = threadIdx.x; f < m; f += blockDim.x) out[(size_t)blockIdx.x * m + f] = X[(size_t)row * m + f];
}

template <typename T>
int kcenters_impl(const T* X, msm_idx_t n, msm_idx_t m, msm_idx_t K, const char* metric,
                  msm_idx_t seed, msm_idx_t* ids, msm_idx_t* labels, double* distances,
                  double* inertia, int on_device, T* centers_out = nullptr)
{
    const msm_idx_t m_rows = m;   // the caller's row length (the fit may run on a zero-padded copy)
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !ids || !labels || !distances) return fail(MSM_ERR_INVALID, "kcenters_fit: null pointer");
    if (n < 1 || m < 1 || K < 1) return fail(MSM_ERR_INVALID, "kcenters_fit: bad shape");
    if (seed < 0 || seed >= n) return fail(MSM_ERR_INVALID, "kcenters_fit: seed_index out of range");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dLab = pool(PS_LAB), &dDist = pool(PS_MIN), &dPart = pool(PS_PART), &dIds = pool(PS_IDS),
           &dSum = pool(PS_SUM);
    int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)2 * nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dIds.reserve((size_t)K * sizeof(msm_idx_t)))) return rc;
    if ((rc = dSum.reserve((size_t)nblk * sizeof(double)))) return rc;
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.m = m;
    P.seed = seed;
    P.nblk = nblk;
    P.ids = dIds.as<msm_idx_t>();
    P.prune = 1;
    if (on_device) {
        P.X = X;
        P.labels = labels;
        P.dist = distances;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = dLab.reserve((size_t)n * sizeof(msm_idx_t)))) return rc;
        if ((rc = dDist.reserve((size_t)n * sizeof(double)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        P.X = dX.p;
        P.labels = dLab.as<msm_idx_t>();
        P.dist = dDist.as<double>();
    }
    if (pad_rows_pays<T>(mid, m, n, false)) {   // odd rows, norm metric: a zero-padded aligned copy (see pad_rows_pays)
        const T* xp = nullptr;
        long long mp = m;
        if ((rc = pad_rows<T>(static_cast<const T*>(P.X), n, m, pool(PS_PADX), &xp, &mp))) return rc;
        P.X = xp;
        P.m = m = mp;
    }
    P.vecw = row_vecw<T>(P.X, m, false);
    const int nblk0 = nblk;   // (the partial buffers hold two arrays of this many entries)
    if (P.vecw == 0 && wide_ok<T>(P.X, P.X, m, false)) P.nblk = nblk = std::min(nblk, wide_grid(n));
    KcPartial* part = dPart.as<KcPartial>();
    const bool screen = sizeof(T) == 8 && mid == M_EUCLIDEAN && P.vecw > 0 && n >= 65536 && K > 8;
    g_kc_stats = KcStats();
    g_kc_stats.rows = n;
    g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);   // the row, distances_, labels_ (pruning test)
    g_kc_stats.plain_passes = K;
    if (screen) {
        // A few plain passes first (in the first passes most rows change, and a candidate costs the screen's bytes on top of
        // the plain pass's), then the screened ones.  Measured on 10M x 10 float64, K = 200 (scripts/kcperf.py, kcblobs.py),
        // plain / screened from pass 16 / from pass 4 / from pass 1: tICA projection 26.6 / 13.1 / 11.8 / 11.8 ms, white
        // noise 33.7 / 15.1 / 13.3 / 17.6 ms, 40 separated blobs (where per-row pruning is at its best) 17.9 / 17.1 / 17.1 /
        // 17.5 ms -- so no decision is needed.
        // (with several centres per pass the early centres are cheap on the copy too -- one pass applies a batch of them to
        //  every row, whatever fraction changes: 2 plain passes, tICA projection 4.53 -> 4.26 ms; 4 for one centre per pass)
        const bool kcb_on = !(getenv("MSM_KC_BATCH") && atoi(getenv("MSM_KC_BATCH")) == 0) && nblk <= 1024;  // read per fit
        const int KSC_PROBE = kcb_on ? 2 : 4;
        KscBufs& B = ksc_bufs();
        const int np = (int)((m + 1) / 2);
        if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;  // [gmax2[2] | ... | c0[16]]
        MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
        msm_idx_t it = 0;
        for (; it < K && it < KSC_PROBE; ++it) {
            P.it = (int)it;
            P.prev = part + (size_t)((it + 1) & 1) * nblk;
            P.next = part + (size_t)(it & 1) * nblk;
            launch_kc<T>(mid, nblk, P);
        }
        const bool use_screen = it < K;
        if (use_screen) {
            constexpr int fmt = KSC_FMT;
            g_kc_stats.plain_passes = it;
            g_kc_stats.screened_passes = K - it;
            g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, fmt) * 4 + 4);   // the screen copy's row + curf
            if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, fmt) + 1) * sizeof(float)))) return rc;
            KscArgs S;
            memset(&S, 0, sizeof(S));
            S.X = reinterpret_cast<const double*>(P.X);
            S.xs = B.xf.p;
            S.gmax2 = B.misc.as<unsigned long long>();
            S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
            hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, P.ids, (long long)m, S.c0, (const double*)nullptr);
            S.n = n;
            S.m = m;
            S.nblk = nblk;
            S.vecw = P.vecw;
            S.seed = seed;
            S.dist = P.dist;
            S.labels = P.labels;
            S.ids = P.ids;
            const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
            switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, fmt>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
            }
            if (kcb_on) {
                // several centres per pass (kcb_select_kernel / kcenters_batch_pass_kernel): rounds of {selector, pass} are
                // queued four at a time -- a round whose selector finds all K centres fixed is two empty launches -- and the
                // host looks at the progress counter between the groups
                DevBuf& SB = pool(PS_W);
                if ((rc = SB.reserve(sizeof(KcbState)))) return rc;
                KcbState* St = SB.as<KcbState>();
                hipLaunchKernelGGL(kcb_init_kernel, dim3(1), dim3(64), 0, stream(), St, (int)it);
                int rounds = 0, done = (int)it;
                int head4[4] = {(int)it, 0, 0, 0};   // k_done, J, rounds, fallbacks: the head of KcbState
                while (done < (int)K) {
                    const int group = kcb_group((int)K, done, rounds, (int)it);
                    for (int r = 0; r < group; ++r, ++rounds) {
                        S.prev = part + (size_t)((it + 1 + rounds) & 1) * nblk;   // partials of the last pass that ran
                        S.next = part + (size_t)((it + rounds) & 1) * nblk;
                        switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcb_select_kernel<NP_>), dim3(1), dim3(1024), 0, stream(), S, St, (int)K); \
                               hipLaunchKernelGGL((kcenters_batch_pass_kernel<NP_>), dim3(nblk), dim3(DT), 0, stream(), S, St); break;
                            MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                        }
                    }
                    MSM_HIP_CHECK(hipGetLastError());
                    MSM_HIP_CHECK(hipMemcpyAsync(head4, St, sizeof(head4), hipMemcpyDeviceToHost, stream()));
                    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                    done = head4[0];
                    if (rounds > 4 * (int)K) return fail(MSM_ERR_HIP, "k-centers: the batched passes made no progress");
                }
                g_kc_stats.screened_passes = head4[2];   // passes that streamed the copy (the empty rounds of the last group are not counted)
                g_kc_stats.batch_fallbacks = head4[3];
                it = K;
            }
            for (; it < K; ++it) {
                S.it = (int)it;
                S.prev = part + (size_t)((it + 1) & 1) * nblk;
                S.next = part + (size_t)(it & 1) * nblk;
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcenters_screen_pass_kernel<NP_, fmt>), dim3(nblk), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
            }
        }
    } else if (mid == M_EUCLIDEAN && n >= 65536 && K > 8 && m <= 4096 && (size_t)m * sizeof(T) > 64 &&
               !(getenv("MSM_KC_WSCREEN") && atoi(getenv("MSM_KC_WSCREEN")) == 0)) {   // (read per fit: the tests' A/B switch)
        // Round 6: everything the register-resident screen above does not take -- float32 rows of any length, float64 rows of
        // more than 16 features -- is screened on a FEATURE-major byte copy (distance_wscreen_dev.h): a few plain passes
        // (most rows still change), then m + 8 bytes per row and pass instead of m sizeof(T) + 16.  Bit-identical results.
        constexpr int KWS_PROBE = 4;
        KscBufs& B = ksc_bufs();
        const int nb4 = (int)((m + 3) / 4);
        msm_idx_t it = 0;
        for (; it < K && it < KWS_PROBE; ++it) {
            P.it = (int)it;
            P.prev = part + (size_t)((it + 1) & 1) * nblk;
            P.next = part + (size_t)(it & 1) * nblk;
            launch_kc<T>(mid, nblk, P);
        }
        if (it < K) {
            const size_t qbytes = (size_t)nb4 * n * sizeof(unsigned);
            if ((rc = B.xf.reserve(qbytes + (size_t)n * sizeof(float) + (size_t)n * sizeof(unsigned short) + 64))) return rc;
            if ((rc = B.misc.reserve(64 + (size_t)m * sizeof(double)))) return rc;   // [gmax2 | stats[2] | ... | c0[m]]
            MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
            KwsArgs S;
            memset(&S, 0, sizeof(S));
            S.X = P.X;
            S.q = B.xf.as<unsigned>();
            S.curf = reinterpret_cast<float*>(static_cast<char*>(B.xf.p) + qbytes);
            S.sf = reinterpret_cast<unsigned short*>(static_cast<char*>(B.xf.p) + qbytes + (size_t)n * sizeof(float));
            S.gmax2 = B.misc.as<unsigned long long>();
            const bool want_stats = getenv("MSM_KC_STATS") && atoi(getenv("MSM_KC_STATS")) == 1;   // diagnostics (scripts/kcwide.py)
            S.stats = want_stats ? S.gmax2 + 1 : nullptr;   // (every candidate lane adds to ONE word: tens of thousands of atomics per pass)
            S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
            S.n = n;
            S.m = m;
            S.nb4 = nb4;
            S.nblk = nblk;
            S.dist = P.dist;
            S.labels = P.labels;
            S.ids = P.ids;
            hipLaunchKernelGGL(kws_origin_kernel<T>, dim3(1), dim3(256), 0, stream(), static_cast<const T*>(P.X), P.ids, (long long)m,
                               const_cast<double*>(S.c0), S.gmax2);
            const int crows = std::max(1, std::min(KWS_CROWS, KWS_CWORDS / (nb4 + 1)));   // rows a workgroup transposes at a time
            const int gconv = (int)std::min<long long>(ceil_div(n, crows), 16LL * num_cus());
            if ((size_t)m * sizeof(T) <= 192)   // short rows: a thread per row (2M x 17 float64: convert 0.99 -> 0.6 ms; from 256-byte rows the strided walks lose: 1M x 64 float32 fit 3.9 -> 4.5 ms)
                hipLaunchKernelGGL(kws_convert_rowthread_kernel<T>, dim3((unsigned)std::min<long long>(ceil_div(n, DT), 16LL * num_cus())), dim3(DT),
                                   (size_t)m * sizeof(double), stream(), S);
            else
                hipLaunchKernelGGL(kws_convert_kernel<T>, dim3(gconv), dim3(DT), (size_t)m * sizeof(double), stream(), S, crows);
            MSM_HIP_CHECK(hipGetLastError());
            g_kc_stats.plain_passes = it;
            g_kc_stats.screened_passes = K - it;
            g_kc_stats.screen_row_bytes = (long long)(4 * nb4 + 2 + 4);   // byte planes + scale + rounded-up distance
            const size_t lds = (size_t)4 * nb4 * sizeof(float) + (size_t)m * sizeof(T);
            const int kr = nb4 <= 4 ? 8 : nb4 <= 16 ? 4 : 2;   // rows per thread (x 4 / 8 / 16 planes per trip: distance_wscreen_dev.h)
            const int gpass = (int)std::min<long long>(ceil_div(n, (long long)kr * DT), nblk0);
            // (the screened passes write `gpass` partials into arrays `nblk0` apart; the plain passes before them wrote `nblk`
            //  partials `nblk` apart: the first screened pass reads those)
            // Several centres per pass (distance_wbatch_dev.h): as many as fit 64 KiB of LDS beside the pass's own 40 KiB --
            // 16 up to 256 float32 / 170 float64 features, 8 up to twice that, one beyond.  MSM_KC_WBATCH=0: one centre per pass.
            int jmax = 0;
            {
                const size_t per = (size_t)4 * nb4 * sizeof(float) + ((size_t)m * sizeof(T) + 15) / 16 * 16;   // float32 copy + the row at a 16-byte pitch
                if (16 * per <= 65536) jmax = 16;
                else if (8 * per <= 65536) jmax = 8;
                // (First version: a round cost ~250 us on top of its streaming and the batches only paid for passes of >= 90 MB.
                //  The cost was the exact re-evaluation -- one lane per candidate walking its row element by element out of LDS,
                //  ~10 us per row and centre -- and the one-workgroup selector; with 16-byte fetches ahead of the chain
                //  (kwb_exact_euclid) and the selector on several workgroups the batches win at every shape of scripts/kcwide.py:
                //  280,000 x 171 K = 200 6.6 -> 4.3 ms, 1M x 171 K = 500 28.2 -> 13.9 ms, 500,000 x 512 27.4 -> 24.4 ms.)
                // Default for rows of up to 1 KiB; MSM_KC_WBATCH=0 / 1 (read per fit) forces one centre per screened pass / the batches.
                const char* be = getenv("MSM_KC_WBATCH");
                const bool pays = (size_t)m * sizeof(T) <= 1024;   // (500,000 x 512 float32: 24.5 ms batched against 19.7 ms one centre per pass -- 16 accumulators per row over 128 words, and the union of 16 centres' candidates at 2 KiB per row)
                if (be ? atoi(be) == 0 : !pays) jmax = 0;
            }
            if (jmax > 0) {
                DevBuf& SB = pool(PS_W);
                if ((rc = SB.reserve(sizeof(KcbState) + sizeof(KwbSync)))) return rc;
                KcbState* St = SB.as<KcbState>();
                KwbSync* Sy = reinterpret_cast<KwbSync*>(static_cast<char*>(SB.p) + sizeof(KcbState));
                hipLaunchKernelGGL(kcb_init_kernel, dim3(1), dim3(64), 0, stream(), St, (int)it);
                MSM_HIP_CHECK(hipMemsetAsync(Sy, 0, sizeof(KwbSync), stream()));
                // the selector on several workgroups (kwb_select_multi_kernel): as many as it takes for a slice of a full list
                // to fit the LDS beside the centre, one row per thread; 0: one workgroup (rows beyond ~9 KiB; MSM_KC_WSELECT=1)
                int nbsel = 0;
                size_t ldssel = 0;
                {
                    const size_t pitch = (size_t)(m | 1), cpad = (size_t)((m + 3) & ~3LL);
                    for (int nb = 8; nb <= 64; nb += 8) {
                        const size_t rpw = (size_t)ceil_div(KCB_CAP, nb);
                        const size_t bytes = (cpad + rpw * pitch) * sizeof(T);
                        if (rpw <= (size_t)DT && bytes <= 150000) {
                            nbsel = nb;
                            ldssel = bytes;
                            break;
                        }
                    }
                    if (getenv("MSM_KC_WSELECT") && atoi(getenv("MSM_KC_WSELECT")) == 1) nbsel = 0;
                    if (nbsel) {
                        static bool attr_set = false;
                        if (!attr_set) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kwb_select_multi_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150016);
                            attr_set = true;
                        }
                    }
                }
                const int it0 = (int)it;
                const int kr2 = nb4 <= 16 ? 4 : 2;   // rows per thread: 16 (8) accumulators each
                const int gp2 = (int)std::min<long long>(ceil_div(n, (long long)kr2 * DT), nblk0);
                const size_t lds2 = (size_t)jmax * ((size_t)4 * nb4 * sizeof(float) + ((size_t)m * sizeof(T) + 15) / 16 * 16);
                int rounds = 0, done = it0;
                const int wcap = KCB_CAP;   // (shorter lists were tried: 512 rows -> three centres per round instead of twelve)
                int head4[4] = {it0, 0, 0, 0};   // k_done, J, rounds, fallbacks: the head of KcbState
                while (done < (int)K) {
                    const int group = kcb_group((int)K, done, rounds, it0);
                    for (int r = 0; r < group; ++r, ++rounds) {
                        S.prev = rounds == 0 ? part + (size_t)((it0 + 1) & 1) * nblk : part + (size_t)((it0 + rounds + 1) & 1) * nblk0;
                        S.next = part + (size_t)((it0 + rounds) & 1) * nblk0;
                        S.nblk = rounds == 0 ? nblk : gp2;
                        if (nbsel) hipLaunchKernelGGL((kwb_select_multi_kernel<T>), dim3(nbsel), dim3(DT), ldssel, stream(), S, St, Sy, (int)K, jmax, wcap);
                        else hipLaunchKernelGGL((kwb_select_kernel<T>), dim3(1), dim3(1024), (size_t)m * sizeof(T), stream(), S, St, (int)K, jmax, wcap);
                        if (jmax == 16) {
                            if (kr2 == 4) launch_kwb<T, 16, 4, 8>(gp2, lds2, S, St);
                            else launch_kwb<T, 16, 2, 24>(gp2, lds2, S, St);
                        } else {
                            if (kr2 == 4) launch_kwb<T, 8, 4, 8>(gp2, lds2, S, St);
                            else launch_kwb<T, 8, 2, 24>(gp2, lds2, S, St);
                        }
                    }
                    MSM_HIP_CHECK(hipGetLastError());
                    unsigned sy4[4] = {0, 0, 0, 0};
                    MSM_HIP_CHECK(hipMemcpyAsync(head4, St, sizeof(head4), hipMemcpyDeviceToHost, stream()));
                    MSM_HIP_CHECK(hipMemcpyAsync(sy4, Sy, sizeof(sy4), hipMemcpyDeviceToHost, stream()));
                    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                    done = head4[0];
                    if (sy4[2]) return fail(MSM_ERR_HIP, "k-centers: the selector's workgroups did not meet at their barrier (MSM_KC_WSELECT=1 runs it on one workgroup)");
                    if (rounds > 4 * (int)K) return fail(MSM_ERR_HIP, "k-centers: the batched wide passes made no progress");
                }
                g_kc_stats.screened_passes = head4[2];
                g_kc_stats.batch_fallbacks = head4[3];
                it = K;
            }
            for (; it < K; ++it) {
                S.it = (int)it;
                S.prev = it == KWS_PROBE ? part + (size_t)((it + 1) & 1) * nblk : part + (size_t)((it + 1) & 1) * nblk0;
                S.next = part + (size_t)(it & 1) * nblk0;
                S.nblk = it == KWS_PROBE ? nblk : gpass;
                if (kr == 8) hipLaunchKernelGGL((kcenters_wscreen_pass_kernel<T, 8, 4>), dim3(gpass), dim3(DT), lds, stream(), S);
                else if (kr == 4) hipLaunchKernelGGL((kcenters_wscreen_pass_kernel<T, 4, 8>), dim3(gpass), dim3(DT), lds, stream(), S);
                else hipLaunchKernelGGL((kcenters_wscreen_pass_kernel<T, 2, 24>), dim3(gpass), dim3(DT), lds, stream(), S);
            }
            if (want_stats) {   // one more round trip
                unsigned long long hs[2] = {0, 0};
                MSM_HIP_CHECK(hipMemcpyAsync(hs, S.stats, sizeof(hs), hipMemcpyDeviceToHost, stream()));
                MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                g_kc_stats.wide_candidates = (long long)hs[0];
                g_kc_stats.wide_updates = (long long)hs[1];
            }
        }
    } else
    for (msm_idx_t it = 0; it < K; ++it) {
        P.it = (int)it;
        P.prev = part + (size_t)((it + 1) & 1) * nblk;
        P.next = part + (size_t)(it & 1) * nblk;
        launch_kc<T>(mid, nblk, P);
    }
    MSM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nblk), dim3(DT), 0, stream(), P.dist, (long long)n, dSum.as<double>());
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemcpyAsync(ids, P.ids, (size_t)K * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
    if (centers_out) {
        DevBuf& dCen = pool(PS_Y);
        if ((rc = dCen.reserve((size_t)K * m_rows * sizeof(T)))) return rc;
        const T* Xsrc = on_device ? X : static_cast<const T*>(dX.p);
        hipLaunchKernelGGL((kc_gather_centres_kernel<T>), dim3((unsigned)K), dim3(64), 0, stream(), Xsrc, (long long)m_rows, P.ids, dCen.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
        MSM_HIP_CHECK(hipMemcpyAsync(centers_out, dCen.p, (size_t)K * m_rows * sizeof(T), hipMemcpyDeviceToHost, stream()));
    }
    if (!on_device) {
        MSM_HIP_CHECK(hipMemcpyAsync(labels, P.labels, (size_t)n * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(distances, P.dist, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    double s = 0.0;
    if ((rc = sum_partials_host(dSum.as<double>(), nblk, &s))) return rc;  // synchronises
    if (inertia) *inertia = s;
    return MSM_OK;
}

template <typename T>
int pdist_impl(const T* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
               msm_idx_t n_idx, double* out, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !out) return fail(MSM_ERR_INVALID, "pdist: null pointer");
    if (n < 0 || m < 1) return fail(MSM_ERR_INVALID, "pdist: bad shape");
    const long long nn = X_indices ? n_idx : n;
    if (nn < 2) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dIdx = pool(PS_IDX), &dOut = pool(PS_OUT);
    const size_t npairs = (size_t)nn * (size_t)(nn - 1) / 2;
    PdArgs P;
    memset(&P, 0, sizeof(P));
    P.n = nn;
    P.m = m;
    if (on_device) {
        P.X = X;
        P.X_indices = X_indices;
        P.out = out;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        P.X = dX.p;
        if (X_indices) {
            if ((rc = dIdx.reserve((size_t)nn * sizeof(msm_idx_t)))) return rc;
            MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, X_indices, (size_t)nn * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            P.X_indices = dIdx.as<msm_idx_t>();
        }
        if ((rc = dOut.reserve(npairs * sizeof(double)))) return rc;
        P.out = dOut.as<double>();
    }
    launch_pd<T, 0>(mid, (int)std::min<long long>(nn - 1, 4096), P);
    MSM_HIP_CHECK(hipGetLastError());
    if (!on_device && (rc = d2h_bulk(out, P.out, npairs * sizeof(double)))) return rc;
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

template <typename T>
int sumdist_impl(const T* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                 msm_idx_t p, double* sum, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !sum || (p > 0 && !pairs)) return fail(MSM_ERR_INVALID, "sumdist: null pointer");
    if (n < 0 || m < 1 || p < 0) return fail(MSM_ERR_INVALID, "sumdist: bad shape");
    *sum = 0.0;
    if (p == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dX = pool(PS_X), &dIdx = pool(PS_IDX), &dPart = pool(PS_PART);
    const int grid = (int)std::min<long long>(ceil_div(p, DT), 1024);
    if ((rc = dPart.reserve((size_t)grid * sizeof(double)))) return rc;
    PdArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.m = m;
    P.p = p;
    P.partial = dPart.as<double>();
    if (on_device) {
        P.X = X;
        P.pairs = pairs;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = dIdx.reserve((size_t)p * 2 * sizeof(msm_idx_t)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, pairs, (size_t)p * 2 * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
        P.X = dX.p;
        P.pairs = dIdx.as<msm_idx_t>();
    }
    launch_pd<T, 1>(mid, grid, P);
    MSM_HIP_CHECK(hipGetLastError());
    return sum_partials_host(P.partial, grid, sum);
}

// One externally driven pass (multi-rank k-centers): centre coordinates come from the host,
// the local (max distance, lowest local row) comes back.
template <typename T>
int kcenters_pass_impl(const T* X, msm_idx_t n, msm_idx_t m, const T* y, msm_idx_t it, const char* metric,
                       msm_idx_t* labels, double* distances, double* max_dist, msm_idx_t* argmax,
                       T* argmax_row, int on_device)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!X || !y || !labels || !distances || !max_dist || !argmax) return fail(MSM_ERR_INVALID, "kcenters_pass: null pointer");
    if (n < 1 || m < 1 || it < 0) return fail(MSM_ERR_INVALID, "kcenters_pass: bad shape");
    if (!on_device) return fail(MSM_ERR_INVALID, "kcenters_pass: per-row arrays must be device resident");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dPart = pool(PS_PART), &dY = pool(PS_Y), &dIds = pool(PS_IDS);
    int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dY.reserve((size_t)m * sizeof(T)))) return rc;
    if ((rc = dIds.reserve(sizeof(msm_idx_t)))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(dY.p, y, (size_t)m * sizeof(T), hipMemcpyHostToDevice, stream()));
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.X = X;
    P.n = n;
    P.m = m;
    P.it = (int)it;
    P.nblk = nblk;
    P.next = dPart.as<KcPartial>();
    P.prev = nullptr;
    P.dist = distances;
    P.labels = labels;
    P.ids = dIds.as<msm_idx_t>();
    P.ycenter = dY.p;
    P.vecw = row_vecw<T>(P.X, m, false);
    if (P.vecw == 0 && wide_ok<T>(P.X, P.ycenter, m, false)) P.nblk = nblk = std::min(nblk, wide_grid(n));
    launch_kc<T>(mid, nblk, P);
    MSM_HIP_CHECK(hipGetLastError());
    // device-side final reduce + fetch of the winning row: ONE small D2H per pass
    DevBuf& dBest = pool(PS_SUM);
    if ((rc = dBest.reserve(sizeof(KcPartial) + (size_t)m * sizeof(T)))) return rc;
    KcPartial* dbest = dBest.as<KcPartial>();
    T* drow = reinterpret_cast<T*>(dbest + 1);
    hipLaunchKernelGGL((kc_finalize_kernel<T>), dim3(1), dim3(DT), 0, stream(), P.next, nblk, X, (long long)m, dbest, drow);
    MSM_HIP_CHECK(hipGetLastError());
    std::vector<char> hb(sizeof(KcPartial) + (size_t)m * sizeof(T));
    MSM_HIP_CHECK(hipMemcpyAsync(hb.data(), dbest, hb.size(), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    const KcPartial* hp = reinterpret_cast<const KcPartial*>(hb.data());
    *max_dist = hp->v;
    *argmax = hp->i;
    if (argmax_row && hp->i >= 0) memcpy(argmax_row, hb.data() + sizeof(KcPartial), (size_t)m * sizeof(T));
    return MSM_OK;
}

// one pass + candidate record, everything on the stream, no synchronisation
template <typename T>
int kcenters_pass_dev_impl(const T* X, msm_idx_t n, msm_idx_t m, const T* y_dev, msm_idx_t it, const char* metric,
                           msm_idx_t* labels, double* distances, msm_idx_t row_offset, double* cand_dev,
                           const T* centers_dev = nullptr)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!y_dev || !cand_dev || (n > 0 && (!X || !labels || !distances))) return fail(MSM_ERR_INVALID, "kcenters_pass_dev: null pointer");
    if (n < 0 || m < 1 || it < 0) return fail(MSM_ERR_INVALID, "kcenters_pass_dev: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    int rc;
    DevBuf &dPart = pool(PS_PART), &dIds = pool(PS_IDS);
    int nblk = (int)std::min<long long>(ceil_div(std::max<long long>(n, 1), DT), KC_MAXBLK);
    if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial)))) return rc;
    if ((rc = dIds.reserve(sizeof(msm_idx_t)))) return rc;
    KcArgs P;
    memset(&P, 0, sizeof(P));
    P.X = X;
    P.n = n;
    P.m = m;
    P.it = (int)it;
    P.next = dPart.as<KcPartial>();
    P.dist = distances;
    P.labels = labels;
    P.ids = dIds.as<msm_idx_t>();
    P.ycenter = y_dev;
    P.centers = centers_dev;                        // centres 0 .. it of the fit (pruning table); null: no pruning
    P.prune = centers_dev ? 1 : 0;
    if (n > 0) {
        P.vecw = row_vecw<T>(P.X, m, false);
        if (P.vecw == 0 && wide_ok<T>(P.X, P.ycenter, m, false)) nblk = std::min(nblk, wide_grid(n));
        P.nblk = nblk;
        launch_kc<T>(mid, nblk, P);
        MSM_HIP_CHECK(hipGetLastError());
    } else {
        nblk = 0;  // empty shard: the candidate kernel reports "none"
    }
    hipLaunchKernelGGL((kc_candidate_kernel<T>), dim3(1), dim3(DT), 0, stream(), P.next, nblk, X, (long long)m,
                       (long long)row_offset, cand_dev);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

template <typename T>
int kcenters_select_impl(const double* cands_dev, msm_idx_t world, msm_idx_t m, T* y_dev, T* centers_dev,
                         msm_idx_t* ids_dev, msm_idx_t slot)
{
    if (!cands_dev || !y_dev || !centers_dev || !ids_dev || world < 1 || m < 1 || slot < 0)
        return fail(MSM_ERR_INVALID, "kcenters_select: bad argument");
    hipLaunchKernelGGL((kc_select_kernel<T>), dim3(1), dim3(DT), 0, stream(), cands_dev, (int)world, (long long)m, y_dev,
                       centers_dev, ids_dev, (long long)slot);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// centre 0 of a sharded fit: the rank that owns the seed row publishes it through the same record exchange
template <typename T>
__global__ __launch_bounds__(DT) void kc_seed_candidate_kernel(const T* __restrict__ X, long long m, long long local_row,
                                                               long long global_row, double* __restrict__ cand)
{
    if (threadIdx.x == 0) {
        cand[0] = local_row >= 0 ? 1.0 : -1.0;
        cand[1] = local_row >= 0 ? (double)global_row : -1.0;
    }
    for (long long f = threadIdx.x; f < m; f += DT) cand[2 + f] = local_row >= 0 ? (double)X[local_row * m + f] : 0.0;
}

// The whole row-sharded fit: K x (pass, candidate record, all-gather over the library communicator, select), all
// queued on the stream -- the host enqueues and synchronises once at the end.
template <typename T>
int kcenters_fit_sharded_impl(const T* X, msm_idx_t n, msm_idx_t m, msm_idx_t K, const char* metric, msm_idx_t seed,
                              msm_idx_t row_offset, msm_idx_t* labels, double* distances, msm_idx_t* ids, T* centers,
                              double* inertia)
{
    const int mid = metric_id(metric);
    if (mid < 0) return fail(MSM_ERR_METRIC, "unknown metric '%s'", metric ? metric : "(null)");
    if (!ids || !centers || (n > 0 && (!X || !labels || !distances))) return fail(MSM_ERR_INVALID, "kcenters_fit_sharded: null pointer");
    if (n < 0 || m < 1 || K < 1 || seed < 0 || row_offset < 0) return fail(MSM_ERR_INVALID, "kcenters_fit_sharded: bad shape");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const int world = comm_world();
    const size_t rec = (size_t)(2 + m);
    // [cand rec | cands world*rec | sums 1024 + 1] doubles, [y m | centers K*m] T, [ids K] int64
    DevBuf& dS = pool(PS_S);
    const size_t nd = rec + (size_t)world * rec + 1032;
    const size_t bytes = nd * sizeof(double) + ((size_t)(K + 1) * m * sizeof(T) + 15) / 16 * 16 + (size_t)K * sizeof(msm_idx_t);
    int rc = dS.reserve(bytes);
    if (rc) return rc;
    double* cand = dS.as<double>();
    // without a communicator (a world of one: bench.py's strong-scaling model, single-process use) the "gathered" records
    // ARE the shard's record: no copy per centre
    double* cands = comm_active() ? cand + rec : cand;
    double* sums = cands + (size_t)world * rec;
    T* y = reinterpret_cast<T*>(sums + 1032);
    T* cen = y + m;
    msm_idx_t* dids = reinterpret_cast<msm_idx_t*>(reinterpret_cast<char*>(y) + ((size_t)(K + 1) * m * sizeof(T) + 15) / 16 * 16);
    const long long local_seed = (seed >= row_offset && seed < row_offset + n) ? (long long)(seed - row_offset) : -1;
    hipLaunchKernelGGL((kc_seed_candidate_kernel<T>), dim3(1), dim3(DT), 0, stream(), X, (long long)m, local_seed, (long long)seed, cand);
    MSM_HIP_CHECK(hipGetLastError());
    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
    const bool fused = n > 0 && row_vecw<T>(X, m, false) > 0;  // register path (m <= FC): one kernel per centre
    // Several centres per exchange (kcb_*: threshold lists, an identical selection on every rank): float64 rows of <= 16
    // features, euclidean.  The decision uses nothing a rank knows alone -- not its shard size, which may be zero --, because
    // it changes the exchange pattern: PROBE all-gathers of one candidate, then one all-gather of a round record per round.
    bool batched = false;
    if constexpr (sizeof(T) == 8) {
        constexpr int probe = 2;
        const char* be = getenv("MSM_KC_BATCH");   // 0: one centre per exchange (A/B switch of the tests; read per fit)
        batched = mid == M_EUCLIDEAN && m <= FeatChunk<T>::FC && K > 8 && K > probe &&
                  !(be && atoi(be) == 0);
        if (batched) {
            const msm_idx_t PROBE = probe;
            const int np = (int)((m + 1) / 2);
            const int nblk = (int)std::min<long long>(std::max<long long>(ceil_div(std::max<long long>(n, 1), DT), 1), KC_MAXBLK);
            DevBuf& dPart = pool(PS_PART);
            if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial) + 16))) return rc;
            unsigned* counter = reinterpret_cast<unsigned*>(dPart.as<KcPartial>() + nblk);
            MSM_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), stream()));
            g_kc_stats = KcStats();
            g_kc_stats.rows = n;
            g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);
            g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, 2) * 4 + 4);
            g_kc_stats.plain_passes = PROBE;
            // ---- the first PROBE centres: one candidate per exchange, plain passes (a rank without rows keeps the pattern) ----
            if (n > 0) {
                KcArgs P;
                memset(&P, 0, sizeof(P));
                P.X = X;
                P.n = n;
                P.m = m;
                P.nblk = nblk;
                P.next = dPart.as<KcPartial>();
                P.dist = distances;
                P.labels = labels;
                P.vecw = row_vecw<T>(X, m, false);
                P.centers = cen;
                P.prune = 1;
                P.sel_cands = cands;
                P.sel_world = world;
                P.sel_centers = cen;
                P.sel_ids = dids;
                P.cand_out = cand;
                P.row_offset = row_offset;
                P.counter = counter;
                for (msm_idx_t it = 0; it < PROBE; ++it) {
                    P.it = (int)it;
                    launch_kc<T>(mid, nblk, P);
                    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                }
            } else {
                if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, 0))) return rc;
                for (msm_idx_t it = 0; it < PROBE; ++it) {
                    if ((rc = kcenters_pass_dev_impl<T>(X, n, m, y, it, metric, labels, distances, row_offset, cand, cen))) return rc;
                    if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                    if (it + 1 < PROBE && (rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, it + 1))) return rc;
                }
            }
            MSM_HIP_CHECK(hipGetLastError());
            // ---- rounds ----
            const size_t RD = kcb_rec_doubles(m);
            DevBuf& W = pool(PS_W);
            const size_t stb = (sizeof(KcbState) + 15) / 16 * 16;
            if ((rc = W.reserve(stb + (size_t)(1 + world) * RD * sizeof(double)))) return rc;
            KcbState* St = W.as<KcbState>();
            double* recL = reinterpret_cast<double*>(static_cast<char*>(W.p) + stb);
            double* recsG = comm_active() ? recL + RD : recL;   // a world of one: the gathered records ARE the rank's record
            hipLaunchKernelGGL(kcb_init_kernel, dim3(1), dim3(64), 0, stream(), St, (int)PROBE);
            hipLaunchKernelGGL(kcb_boot_records_kernel, dim3((unsigned)world), dim3(64), 0, stream(), cands, world, (long long)m, recsG);
            KscArgs S;
            memset(&S, 0, sizeof(S));
            if (n > 0) {
                KscBufs& B = ksc_bufs();
                if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;
                if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, 2) + 1) * sizeof(float)))) return rc;
                MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
                S.X = reinterpret_cast<const double*>(X);
                S.xs = B.xf.p;
                S.gmax2 = B.misc.as<unsigned long long>();
                S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
                S.n = n;
                S.m = m;
                S.nblk = nblk;
                S.vecw = row_vecw<T>(X, m, false);
                S.seed = seed;
                S.dist = distances;
                S.labels = labels;
                S.ids = dids;
                S.next = dPart.as<KcPartial>();
                S.row_offset = row_offset;
                hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, dids, (long long)m, S.c0,
                                   reinterpret_cast<const double*>(cen));
                const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, 2>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
            }
            MSM_HIP_CHECK(hipGetLastError());
            int rounds = 0, done = (int)PROBE;
            int head4[4] = {(int)PROBE, 0, 0, 0};
            while (done < (int)K) {
                const int group = kcb_group((int)K, done, rounds, (int)PROBE);
                for (int r = 0; r < group; ++r, ++rounds) {
                    switch (np) {
#define MSM_KSC(NP_) case NP_: \
                        hipLaunchKernelGGL((kcb_select_sharded_kernel<NP_>), dim3(1), dim3(1024), 0, stream(), recsG, world, (long long)m, St, (int)K, \
                                           reinterpret_cast<double*>(cen), dids); \
                        if (n > 0) { \
                            hipLaunchKernelGGL((kcenters_batch_pass_kernel<NP_>), dim3(nblk), dim3(DT), 0, stream(), S, St); \
                            hipLaunchKernelGGL((kcb_pack_kernel<NP_>), dim3(1), dim3(DT), 0, stream(), S, St, recL); \
                        } \
                        break;
                        MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                    }
                    if (n <= 0) hipLaunchKernelGGL(kcb_empty_record_kernel, dim3(1), dim3(64), 0, stream(), recL);
                    MSM_HIP_CHECK(hipGetLastError());
                    if ((rc = comm_allgather(recL, recsG, RD * sizeof(double)))) return rc;
                }
                MSM_HIP_CHECK(hipMemcpyAsync(head4, St, sizeof(head4), hipMemcpyDeviceToHost, stream()));
                MSM_HIP_CHECK(hipStreamSynchronize(stream()));
                done = head4[0];
                if (rounds > 4 * (int)K) return fail(MSM_ERR_HIP, "k-centers: the batched rounds made no progress");
            }
            g_kc_stats.screened_passes = head4[2];
            g_kc_stats.batch_fallbacks = head4[3];
        }
    }
    if (batched) {
    } else if (fused) {
        // every rank must take the same path or the all-gathers would not match: rows are a property of the data type and
        // width only, except for an EMPTY shard -- which therefore runs the generic kernels but keeps the exchange pattern
        DevBuf &dPart = pool(PS_PART);
        const int nblk = (int)std::min<long long>(ceil_div(n, DT), KC_MAXBLK);
        if ((rc = dPart.reserve((size_t)nblk * sizeof(KcPartial) + 16))) return rc;
        unsigned* counter = reinterpret_cast<unsigned*>(dPart.as<KcPartial>() + nblk);
        MSM_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), stream()));
        KcArgs P;
        memset(&P, 0, sizeof(P));
        P.X = X;
        P.n = n;
        P.m = m;
        P.nblk = nblk;
        P.next = dPart.as<KcPartial>();
        P.dist = distances;
        P.labels = labels;
        P.vecw = row_vecw<T>(X, m, false);
        P.centers = cen;
        P.prune = 1;
        P.sel_cands = cands;
        P.sel_world = world;
        P.sel_centers = cen;
        P.sel_ids = dids;
        P.cand_out = cand;
        P.row_offset = row_offset;
        P.counter = counter;
        // Screened passes (float64 rows, euclidean; see kcenters_screen_pass_kernel) after a few plain ones, as in the
        // single-process fit.  The decision is LOCAL to a rank: the exchange pattern (one all-gather per centre) is the same
        // for both kernels, so a rank with a small shard may keep the plain kernel while its peers screen.
        constexpr int KSC_PROBE = 4;
        constexpr long long KSC_MIN_ROWS = 65536;
        const bool screen = sizeof(T) == 8 && mid == M_EUCLIDEAN && n >= KSC_MIN_ROWS && K > 8 && K > KSC_PROBE;
        KscArgs S;
        memset(&S, 0, sizeof(S));
        const int np = (int)((m + 1) / 2);
        constexpr int fmt = KSC_FMT;
        if (screen) {
            KscBufs& B = ksc_bufs();
            if ((rc = B.misc.reserve(64 + 16 * sizeof(double)))) return rc;
            if ((rc = B.xf.reserve((size_t)n * (ksc_words(np, fmt) + 1) * sizeof(float)))) return rc;
            MSM_HIP_CHECK(hipMemsetAsync(B.misc.p, 0, 64, stream()));
            S.X = reinterpret_cast<const double*>(X);
            S.xs = B.xf.p;
            S.gmax2 = B.misc.as<unsigned long long>();
            S.c0 = reinterpret_cast<double*>(static_cast<char*>(B.misc.p) + 64);
            S.n = n;
            S.m = m;
            S.nblk = nblk;
            S.vecw = P.vecw;
            S.seed = seed;
            S.dist = distances;
            S.labels = labels;
            S.ids = dids;
            S.next = dPart.as<KcPartial>();
            S.sel_cands = cands;
            S.sel_world = world;
            S.sel_centers = reinterpret_cast<double*>(cen);
            S.sel_ids = dids;
            S.cand_out = cand;
            S.row_offset = row_offset;
            S.counter = counter;
        }
        g_kc_stats = KcStats();
        g_kc_stats.rows = n;
        g_kc_stats.plain_row_bytes = (long long)(m * sizeof(T) + 16);
        g_kc_stats.screen_row_bytes = (long long)(ksc_words(np, fmt) * 4 + 4);
        for (msm_idx_t it = 0; it < K; ++it) {
            if (screen && it >= KSC_PROBE) {
                if (it == KSC_PROBE) {
                    // the copy's origin is centre 0 as every rank selected it (possibly another rank's row)
                    hipLaunchKernelGGL(ksc_origin_kernel, dim3(1), dim3(64), 0, stream(), S.X, dids, (long long)m, S.c0,
                                       reinterpret_cast<const double*>(cen));
                    const int gconv = (int)std::min<long long>(ceil_div(n, DT), 8LL * num_cus());
                    switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((ksc_convert_kernel<NP_, fmt>), dim3(gconv), dim3(DT), 0, stream(), S); break;
                        MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                    }
                }
                S.it = (int)it;
                switch (np) {
#define MSM_KSC(NP_) case NP_: hipLaunchKernelGGL((kcenters_screen_pass_kernel<NP_, fmt>), dim3(nblk), dim3(DT), 0, stream(), S); break;
                    MSM_KSC(1) MSM_KSC(2) MSM_KSC(3) MSM_KSC(4) MSM_KSC(5) MSM_KSC(6) MSM_KSC(7) MSM_KSC(8)
#undef MSM_KSC
                }
                ++g_kc_stats.screened_passes;
            } else {
                P.it = (int)it;
                launch_kc<T>(mid, nblk, P);
                ++g_kc_stats.plain_passes;
            }
            if (it + 1 < K && (rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
        }
        MSM_HIP_CHECK(hipGetLastError());
    } else {
        if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, 0))) return rc;
        for (msm_idx_t it = 0; it < K; ++it) {
            if ((rc = kcenters_pass_dev_impl<T>(X, n, m, y, it, metric, labels, distances, row_offset, cand, cen))) return rc;
            if (it + 1 < K) {
                if ((rc = comm_allgather(cand, cands, rec * sizeof(double)))) return rc;
                if ((rc = kcenters_select_impl<T>(cands, world, m, y, cen, dids, it + 1))) return rc;
            }
        }
    }
    // inertia = sum of ALL ranks' distances_: local fp64 tree sum, then one all-reduce of a single double
    const int nblk = (int)std::min<long long>(std::max<long long>(ceil_div(std::max<long long>(n, 1), DT), 1), 1024);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nblk), dim3(DT), 0, stream(), distances, (long long)n, sums);
    hipLaunchKernelGGL(sum_partial_kernel, dim3(1), dim3(DT), 0, stream(), sums, (long long)nblk, sums + 1024);
    MSM_HIP_CHECK(hipGetLastError());
    if ((rc = comm_allreduce_f64(sums + 1024, 1))) return rc;
    double tot = 0.0;
    MSM_HIP_CHECK(hipMemcpyAsync(&tot, sums + 1024, sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(ids, dids, (size_t)K * sizeof(msm_idx_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(centers, cen, (size_t)K * m * sizeof(T), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (inertia) *inertia = tot;
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_kcenters_pass_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, float* argmax_row, int on_device)
{
    return kcenters_pass_impl<float>(X, n, m, y, it, metric, labels, distances, max_dist, argmax, argmax_row, on_device);
}

int msm_kcenters_pass_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, double* argmax_row, int on_device)
{
    return kcenters_pass_impl<double>(X, n, m, y, it, metric, labels, distances, max_dist, argmax, argmax_row, on_device);
}

int msm_kcenters_pass_dev_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev)
{
    return kcenters_pass_dev_impl<float>(X, n, m, y_dev, it, metric, labels, distances, row_offset, cand_dev);
}

int msm_kcenters_pass_dev_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev)
{
    return kcenters_pass_dev_impl<double>(X, n, m, y_dev, it, metric, labels, distances, row_offset, cand_dev);
}

int msm_kcenters_select_f32(const double* cands_dev, msm_idx_t world, msm_idx_t m, float* y_dev, float* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot)
{
    return kcenters_select_impl<float>(cands_dev, world, m, y_dev, centers_dev, ids_dev, slot);
}

int msm_kcenters_select_f64(const double* cands_dev, msm_idx_t world, msm_idx_t m, double* y_dev, double* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot)
{
    return kcenters_select_impl<double>(cands_dev, world, m, y_dev, centers_dev, ids_dev, slot);
}

int msm_pdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device)
{
    return pdist_impl<float>(X, metric, n, m, X_indices, n_X_indices, out, on_device);
}

int msm_pdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device)
{
    return pdist_impl<double>(X, metric, n, m, X_indices, n_X_indices, out, on_device);
}

int msm_sumdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device)
{
    return sumdist_impl<float>(X, metric, n, m, pairs, p, sum, on_device);
}

int msm_sumdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device)
{
    return sumdist_impl<double>(X, metric, n, m, pairs, p, sum, on_device);
}

int msm_dist_f32(const float* X, const float* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device)
{
    return cdist_impl<float>(X, y, metric, n, 1, m, X_indices, n_X_indices, out, on_device);
}

int msm_dist_f64(const double* X, const double* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device)
{
    return cdist_impl<double>(X, y, metric, n, 1, m, X_indices, n_X_indices, out, on_device);
}

int msm_cdist_f32(const float* XA, const float* XB, const char* metric, msm_idx_t na, msm_idx_t nb,
                  msm_idx_t m, double* out, int on_device)
{
    return cdist_impl<float>(XA, XB, metric, na, nb, m, nullptr, 0, out, on_device);
}

int msm_cdist_f64(const double* XA, const double* XB, const char* metric, msm_idx_t na,
                  msm_idx_t nb, msm_idx_t m, double* out, int on_device)
{
    return cdist_impl<double>(XA, XB, metric, na, nb, m, nullptr, 0, out, on_device);
}

int msm_assign_nearest_f32(const float* X, const float* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device)
{
    return assign_nearest_impl<float>(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                      assignments, min_dist, inertia, on_device);
}

int msm_assign_nearest_f64(const double* X, const double* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device)
{
    return assign_nearest_impl<double>(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                       assignments, min_dist, inertia, on_device);
}

int msm_kcenters_last_stats(msm_idx_t* out5)
{
    if (!out5) return fail(MSM_ERR_INVALID, "msm_kcenters_last_stats: null pointer");
    out5[0] = g_kc_stats.rows;
    out5[1] = g_kc_stats.plain_passes;
    out5[2] = g_kc_stats.screened_passes;
    out5[3] = g_kc_stats.plain_row_bytes;
    out5[4] = g_kc_stats.screen_row_bytes;
    return MSM_OK;
}

int msm_kcenters_last_wide_stats(msm_idx_t* out2)
{
    if (!out2) return fail(MSM_ERR_INVALID, "msm_kcenters_last_wide_stats: null pointer");
    out2[0] = g_kc_stats.wide_candidates;
    out2[1] = g_kc_stats.wide_updates;
    return MSM_OK;
}

int msm_kcenters_fit_sharded_f32(const float* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, float* centers, double* inertia)
{
    return kcenters_fit_sharded_impl<float>(X, n_local, m, n_clusters, metric, seed_index, row_offset, labels, distances, ids,
                                            centers, inertia);
}

int msm_kcenters_fit_sharded_f64(const double* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, double* centers, double* inertia)
{
    return kcenters_fit_sharded_impl<double>(X, n_local, m, n_clusters, metric, seed_index, row_offset, labels, distances, ids,
                                             centers, inertia);
}

int msm_kcenters_fit_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device)
{
    return kcenters_impl<float>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device);
}

int msm_kcenters_fit_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device)
{
    return kcenters_impl<double>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device);
}

/* the same fit, also returning the chosen rows: centers[n_clusters][m] (HOST) = X[ids] (kcenters.py:98 cluster_centers_) */
int msm_kcenters_fit2_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric, msm_idx_t seed_index,
                          msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia, int on_device, float* centers)
{
    return kcenters_impl<float>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device, centers);
}

int msm_kcenters_fit2_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric, msm_idx_t seed_index,
                          msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia, int on_device, double* centers)
{
    return kcenters_impl<double>(X, n, m, n_clusters, metric, seed_index, ids, labels, distances, inertia, on_device, centers);
}

}  // extern "C"
