// common.h -- shared host-side plumbing of libmsmhip (error state, stream, scratch buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/msmhip.h"

namespace msm {

// ---- error state (thread-local message, C-ABI status codes) --------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define MSM_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::msm::fail(MSM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                            \
    } while (0)

hipStream_t stream();
int num_cus();

// Bulk upload of caller-owned PAGEABLE host memory, ordered like a hipMemcpyAsync on stream().  A plain
// hipMemcpyAsync from pageable memory goes through the runtime's single-threaded bounce buffer (measured
// 13.7 GB/s); this one has a few worker threads memcpy slices into a ring of pinned buffers while the DMA
// engine ships the previous slice on a copy stream.  Small copies fall through to hipMemcpyAsync.
// after_compute = false: the copy does NOT wait for the work queued on stream() so far (the caller knows the destination is idle);
// what is queued on stream() afterwards still waits for the copy.
int h2d_bulk(void* dst_device, const void* src_host, size_t bytes, bool after_compute = true);
// The other direction, BLOCKING: returns when dst_host holds the data (results produced on stream()).
int d2h_bulk(void* dst_host, const void* src_device, size_t bytes);

// RAII device scratch used when the caller hands over host pointers.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // grows, never shrinks; returns status
    void release();
    ~DevBuf() { release(); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

// Process-lifetime scratch slots (grow-only, never freed: hipFree synchronises the device and
// a per-call malloc/free pair costs more than the small kernels it serves).  Every entry point
// that uses them synchronises the stream before returning, so slots can be shared.
DevBuf& pool(int slot);
enum PoolSlot { PS_X = 0, PS_Y, PS_IDX, PS_LAB, PS_MIN, PS_PART, PS_OUT, PS_IDS, PS_SUM, PS_PAR, PS_W, PS_S, PS_PADX, PS_PADY, PS_COUNT };

// runtime.hip: the bf16 image path's process-lifetime ring (nullptr + msm_last_error on failure); MSM_TICA_IMG_RING_MB, 2 GB
struct ImgRing {
    char* p = nullptr;
    size_t bytes = 0;
};
ImgRing* img_ring();

// metric ids shared by host dispatch and device kernels
enum Metric : int {
    M_EUCLIDEAN = 0,
    M_SQEUCLIDEAN,
    M_CITYBLOCK,
    M_CHEBYSHEV,
    M_CANBERRA,
    M_BRAYCURTIS,
    M_HAMMING,
    M_JACCARD,
    M_COUNT
};
int metric_id(const char* name);  // -1 if unknown

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// comm.hip: the library's communicator (RCCL over xGMI, or a host callback in gloo test runs).  Device buffers,
// ordered on stream(); no-ops (all-gather: a copy) without a communicator.
bool comm_active();
int comm_rank();
int comm_world();
int comm_allreduce_f64(double* dbuf, size_t n);                     // sum, in place
int comm_allgather(const void* dsend, void* drecv, size_t bytes);   // drecv[r * bytes ...] = rank r's dsend

// eigsolve.hip: rocSOLVER / rocBLAS building blocks (dlopen'ed at first use) on stream(), nothing synchronised
int sygv_reduce_device(double* A, double* B, int n, int* dinfo, double* Winv, double* T);   // B = L L^T (lower, col-major), A <- L^-1 A L^-T; Winv, T: optional n x n (own route: Winv <- L^-1)
int sygv_back_device(const double* L, double* Y, int n, int k);    // Y[n x k col-major] <- L^-T Y
int syevd_device(double* A, int n, double* D, double* E, int* dinfo);
// toppairs.hip: own Cholesky and the residual check of the solve, queued on stream(), nothing synchronised
int potrf_upper_device(double* B, int n, int* dinfo, double* Winv);   // B = U^T U on the row-major upper triangle (== dpotrf 'L', col-major); Winv (or null) <- U^-T
int winv_back_device(const double* W, int n, const double* Y, int k, double* V);   // V (k rows) = W^T Y
int pair_residual_device(const double* Cm, int n, const double* Y, const double* vals, int k, double* res);   // res: 2k + k*k doubles

// subspace.hip: k largest eigenpairs of a symmetric matrix with spectrum in [lower, inf) by Chebyshev-filtered subspace
// iteration (block of 32, Rayleigh-Ritz on the host); synchronises; *converged = 0 -> outputs meaningless, use the fallback
int subspace_topk_device(const double* Cm, int n, int k, double lower, double tol, int degree, int max_outer, double* lam,
                         double* Yk, double* work, double* pin /* subspace_pin_doubles() of pinned host memory */, int* converged,
                         int* outer_used, double first_cut /* prior for the first filter, NaN: Rayleigh-Ritz first */, double first_top);
size_t subspace_work_doubles(int n);
size_t subspace_pin_doubles();

// A data pointer that was itself LOADED from memory (e.g. out of a descriptor table) is a
// generic pointer to the compiler, which then emits flat_load: slower, and because FLAT counts on
// both vmcnt and lgkmcnt every counted wait degenerates to vmcnt(0).  Re-type it as global.
template <typename T>
using global_ptr = const T __attribute__((address_space(1)))*;
template <typename T>
__device__ __forceinline__ global_ptr<T> as_global(const void* p)
{
    return (global_ptr<T>)(uintptr_t)p;
}
typedef float raw_f32x4 __attribute__((ext_vector_type(4)));  // 16-byte load unit (no class ctor)
template <typename T>
__device__ __forceinline__ float4 load16_global(global_ptr<T> p)
{
    const raw_f32x4 v = *(global_ptr<raw_f32x4>)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

}  // namespace msm
