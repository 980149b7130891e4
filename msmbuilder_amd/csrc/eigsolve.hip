// eigsolve.hip -- device-side generalized symmetric-definite eigensolve: the
// `scipy.linalg.eigh(offset_correlation, b=covariance, eigvals=(F-k, F-1))` of
// /root/reference/msmbuilder/decomposition/tica.py:188-194 on the GPU.  LAPACK's dsygv* recipe spelled
// out over rocSOLVER / rocBLAS building blocks: B = L L^T (dpotrf), C = L^-1 A L^-T (two dtrsm --
// rocSOLVER's own dsygvd spends most of its time in a slow dsygst here: 138 vs 60 ms at F = 2048),
// C = Y diag(w) Y^T (dsyevd, divide & conquer), and v = L^-T y for the k requested columns only.
// Eigenvectors come back B-orthonormal (v^T B v = 1) like LAPACK's.
// The library is resolved at first use with dlopen (the copy PyTorch already loaded when there
// is one), so libmsmhip has no link-time dependency on it and every other entry point works
// without it.  Worth it from F ~ 1024: 61 ms vs 185 ms on 8 host threads at F = 2048.
#include "common.h"

#include <dlfcn.h>

#include <mutex>
#include <string>

namespace msm {

typedef void* rb_handle;
typedef int (*fn_create)(rb_handle*);
typedef int (*fn_set_stream)(rb_handle, hipStream_t);
typedef int (*fn_dpotrf)(rb_handle, int, int, double*, int, int*);
typedef int (*fn_dsyevd)(rb_handle, int, int, int, double*, int, double*, double*, int*);
typedef int (*fn_dtrsm)(rb_handle, int, int, int, int, int, int, const double*, const double*, int, double*, int);
typedef int (*fn_dgemm)(rb_handle, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int);

struct Solver {
    void* lib = nullptr;
    fn_create create = nullptr;
    fn_set_stream set_stream = nullptr;
    fn_dpotrf dpotrf = nullptr;
    fn_dsyevd dsyevd = nullptr;
    fn_dtrsm dtrsm = nullptr;
    fn_dgemm dgemm = nullptr;
    rb_handle handle = nullptr;
    std::string error;
};

static Solver& solver()
{
    static Solver s;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librocsolver.so", "librocsolver.so.0", "/opt/rocm/lib/librocsolver.so.0"};
        s.lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);  // PyTorch's copy, if it is in the process
        for (int i = 0; !s.lib && i < 3; ++i) s.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!s.lib) {
            s.error = "librocsolver.so not found";
            return;
        }
        s.create = (fn_create)dlsym(s.lib, "rocblas_create_handle");
        s.set_stream = (fn_set_stream)dlsym(s.lib, "rocblas_set_stream");
        s.dpotrf = (fn_dpotrf)dlsym(s.lib, "rocsolver_dpotrf");
        s.dsyevd = (fn_dsyevd)dlsym(s.lib, "rocsolver_dsyevd");
        s.dtrsm = (fn_dtrsm)dlsym(s.lib, "rocblas_dtrsm");
        s.dgemm = (fn_dgemm)dlsym(s.lib, "rocblas_dgemm");
        if (!s.create || !s.set_stream || !s.dpotrf || !s.dsyevd || !s.dtrsm || !s.dgemm) {
            s.error = "rocsolver_dpotrf / rocsolver_dsyevd / rocblas_dtrsm / rocblas_dgemm not reachable through librocsolver";
            return;
        }
        if (s.create(&s.handle) != 0) s.error = "rocblas_create_handle failed";
    });
    return s;
}

// out[j][i] = Z[(n - 1 - j) * n + i] (eigenvector of the j-th LARGEST eigenvalue), vals[j] = D[n - 1 - j]
__global__ void top_pairs_kernel(const double* __restrict__ Z, const double* __restrict__ D, int n, int k,
                                 double* __restrict__ vecs, double* __restrict__ vals)
{
    const int j = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        vecs[(size_t)j * n + i] = Z[(size_t)(n - 1 - j) * n + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) vals[j] = D[n - 1 - j];
}

// ---- building blocks shared with tica.hip's device-resident solve (declared in common.h) -----------------------
// B = L L^T in place (lower, column-major == upper of the row-major symmetric buffer), then A <- L^-1 A L^-T.
// Nothing is synchronised: *dinfo (device int) receives potrf's info and is read by the caller with its results.
// With `Winv` and `T` (n x n scratch each) and n <= 1024 the library's own factorisation also leaves W = U^-T in Winv and the
// reduction is C = W A W^T by two GEMMs; otherwise two dtrsm against the factor.
int sygv_reduce_device(double* A, double* B, int n, int* dinfo, double* Winv, double* T)
{
    Solver& s = solver();
    if (!s.error.empty()) return fail(MSM_ERR_STATE, "device eigensolver unavailable: %s", s.error.c_str());
    if (s.set_stream(s.handle, stream()) != 0) return fail(MSM_ERR_HIP, "rocblas_set_stream failed");
    const double one = 1.0;
    int st;
    // B = L L^T: the library's own blocked kernel up to n = 1024 (one launch per 32 rows; rocSOLVER's dpotrf is launch-
    // latency bound there: 1.3 ms at n = 512), rocSOLVER beyond.
    const bool own = n <= 1024;
    if (own && Winv && T) {
        int rc = potrf_upper_device(B, n, dinfo, Winv);
        if (rc) return rc;
        // buffers are row-major; rocBLAS reads them column-major, i.e. transposed: T = (Winv buffer)^T A = W A, then
        // C = T (Winv buffer) = W A W^T (symmetric, so the result is the same in either reading)
        const double zero = 0.0;
        st = s.dgemm(s.handle, 112, 111, n, n, n, &one, Winv, n, A, n, &zero, T, n);
        if (st == 0) st = s.dgemm(s.handle, 111, 111, n, n, n, &one, T, n, Winv, n, &zero, A, n);
        if (st != 0) return fail(MSM_ERR_HIP, "rocblas_dgemm failed with rocblas_status %d", st);
        return MSM_OK;
    }
    if (own) {
        int rc = potrf_upper_device(B, n, dinfo, nullptr);
        if (rc) return rc;
    } else {
        st = s.dpotrf(s.handle, 122, n, B, n, dinfo);
        if (st != 0) return fail(MSM_ERR_HIP, "rocsolver_dpotrf failed with rocblas_status %d", st);
    }
    st = s.dtrsm(s.handle, 141, 122, 111, 131, n, n, &one, B, n, A, n);               // X = L^-1 A
    if (st == 0) st = s.dtrsm(s.handle, 142, 122, 112, 131, n, n, &one, B, n, A, n);  // C = X L^-T
    if (st != 0) return fail(MSM_ERR_HIP, "rocblas_dtrsm failed with rocblas_status %d", st);
    return MSM_OK;
}

// Y (n x k, column-major: k eigenvectors of the reduced problem as contiguous columns) <- L^-T Y
int sygv_back_device(const double* L, double* Y, int n, int k)
{
    Solver& s = solver();
    if (!s.error.empty()) return fail(MSM_ERR_STATE, "device eigensolver unavailable: %s", s.error.c_str());
    if (s.set_stream(s.handle, stream()) != 0) return fail(MSM_ERR_HIP, "rocblas_set_stream failed");
    const double one = 1.0;
    const int st = s.dtrsm(s.handle, 141, 122, 112, 131, n, k, &one, L, n, Y, n);
    if (st != 0) return fail(MSM_ERR_HIP, "rocblas_dtrsm failed with rocblas_status %d", st);
    return MSM_OK;
}

// all eigenpairs of the symmetric A (in place: columns = eigenvectors, D ascending); E: n doubles of workspace
int syevd_device(double* A, int n, double* D, double* E, int* dinfo)
{
    Solver& s = solver();
    if (!s.error.empty()) return fail(MSM_ERR_STATE, "device eigensolver unavailable: %s", s.error.c_str());
    if (s.set_stream(s.handle, stream()) != 0) return fail(MSM_ERR_HIP, "rocblas_set_stream failed");
    const int st = s.dsyevd(s.handle, 211, 122, n, A, n, D, E, dinfo);
    if (st != 0) return fail(MSM_ERR_HIP, "rocsolver_dsyevd failed with rocblas_status %d", st);
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_sygv_top(const double* A, const double* B, msm_idx_t n, msm_idx_t k, double* evals, double* evecs,
                 int on_device)
{
    if (!A || !B || !evals || !evecs) return fail(MSM_ERR_INVALID, "msm_sygv_top: null pointer");
    if (n < 1 || n > 32768 || k < 1 || k > n) return fail(MSM_ERR_INVALID, "msm_sygv_top: need 1 <= k <= n <= 32768");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    Solver& s = solver();
    if (!s.error.empty()) return fail(MSM_ERR_STATE, "device eigensolver unavailable: %s", s.error.c_str());
    int rc;
    const size_t nn = (size_t)n * n;
    DevBuf &dA = pool(PS_X), &dB = pool(PS_Y), &dW = pool(PS_W), &dO = pool(PS_OUT);
    if ((rc = dA.reserve(nn * sizeof(double)))) return rc;
    if ((rc = dB.reserve(nn * sizeof(double)))) return rc;
    if ((rc = dW.reserve((size_t)2 * n * sizeof(double) + 16))) return rc;
    if ((rc = dO.reserve(((size_t)k * n + k) * sizeof(double)))) return rc;
    const hipMemcpyKind in = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // symmetric inputs: row-major == column-major
    MSM_HIP_CHECK(hipMemcpyAsync(dA.p, A, nn * sizeof(double), in, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(dB.p, B, nn * sizeof(double), in, stream()));
    double* D = dW.as<double>();
    double* E = D + n;
    int* info = reinterpret_cast<int*>(E + n);
    if (s.set_stream(s.handle, stream()) != 0) return fail(MSM_ERR_HIP, "rocblas_set_stream failed");
    // enums: fill_lower 122, side_left 141 / right 142, op none 111 / transpose 112, non_unit 131, evect_original 211.
    // All matrices are column-major for the library; A and B are symmetric, so the caller's row-major
    // buffers are used as they are.
    const int N = (int)n;
    const double one = 1.0;
    double* L = dB.as<double>();
    double* Cm = dA.as<double>();
    int st = s.dpotrf(s.handle, 122, N, L, N, info);                                   // B = L L^T
    if (st != 0) return fail(MSM_ERR_HIP, "rocsolver_dpotrf failed with rocblas_status %d", st);
    int hinfo = 0;
    MSM_HIP_CHECK(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (hinfo != 0)
        return fail(MSM_ERR_INVALID, "the leading minor of order %d of 'b' is not positive definite. The factorization of 'b' "
                    "could not be completed and no eigenvalues or eigenvectors were computed.", hinfo);
    st = s.dtrsm(s.handle, 141, 122, 111, 131, N, N, &one, L, N, Cm, N);               // X = L^-1 A
    if (st == 0) st = s.dtrsm(s.handle, 142, 122, 112, 131, N, N, &one, L, N, Cm, N);  // C = X L^-T
    if (st != 0) return fail(MSM_ERR_HIP, "rocblas_dtrsm failed with rocblas_status %d", st);
    st = s.dsyevd(s.handle, 211, 122, N, Cm, N, D, E, info);                           // C = Y diag(D) Y^T, D ascending
    if (st != 0) return fail(MSM_ERR_HIP, "rocsolver_dsyevd failed with rocblas_status %d", st);
    // back-transform only the k columns that are returned: v = L^-T y
    st = s.dtrsm(s.handle, 141, 122, 112, 131, N, (int)k, &one, L, N, Cm + (size_t)(n - k) * n, N);
    if (st != 0) return fail(MSM_ERR_HIP, "rocblas_dtrsm failed with rocblas_status %d", st);
    double* ovecs = dO.as<double>();
    double* ovals = ovecs + (size_t)k * n;
    hipLaunchKernelGGL(top_pairs_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)k), dim3(256), 0, stream(),
                       dA.as<double>(), D, (int)n, (int)k, ovecs, ovals);
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, stream()));
    const hipMemcpyKind out = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    MSM_HIP_CHECK(hipMemcpyAsync(evecs, ovecs, (size_t)k * n * sizeof(double), out, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(evals, ovals, (size_t)k * sizeof(double), out, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (hinfo != 0) return fail(MSM_ERR_INVALID, "eigenvalue iteration did not converge (info = %d)", hinfo);
    return MSM_OK;
}

}  // extern "C"
