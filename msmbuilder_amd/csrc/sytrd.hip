// sytrd.hip -- Householder tridiagonalisation of a symmetric matrix on the device, n <= 1024 (LAPACK dsytd2, lower).
//
// Why it exists.  tICA._solve reduces the generalized problem to a standard one on the device (tica.hip,
// msm_tica_reduce) and needs the k largest eigenpairs of the reduced F x F matrix (tica.py:188-194 asks LAPACK's dsygvx
// for them).  On the host that is dsyevr, and at F = 512 dsyevr IS its tridiagonalisation: 5.7 of 5.9 ms on one thread, a
// memory-bound symv per column.  rocSOLVER's dsytrd / dsyevd take 10-11 ms at this size: a few launches per column, 512
// columns, latency-bound.  The F x F matrix is 2 MB: it fits the REGISTERS of a few CUs.
//
// The kernel.  n / 16 workgroups (32 at n = 512) are launched once and stay resident for all n - 2 columns.  Workgroup g
// owns the columns c = g, g + G, ... (cyclic, so the shrinking trailing matrix stays balanced), whole columns, in
// registers: thread t holds rows t, t + 256, ... of its workgroup's 16 columns.  With whole columns local, the symmetric
// matrix-vector product p = tau A v needs NO cross-workgroup reduction -- p[c] = tau A[:, c] . v -- and the rank-2 update
// A -= v w^T + w v^T is local once v and w are known everywhere.  Per column the only exchange is therefore an all-gather
// of p (n doubles) plus the not-yet-updated NEXT pivot column from its owner; every workgroup then forms w, the updated
// pivot column, its norm and the next reflector REDUNDANTLY (identical arithmetic, identical order: identical bits), so
// there is ONE grid barrier per column instead of LAPACK's sequence of dependent BLAS-2 calls.  The barrier is a
// flag array, not a counter: workgroup g release-stores the step number to its own word after publishing its part,
// and one wave of every workgroup polls all G words in parallel (one lane each) with acquire loads -- no serialised
// read-modify-writes.  Every spin is bounded (clock64): a workgroup that never arrives (the GPU was not idle enough to
// keep all of them resident) makes the others give up, and the host falls back to LAPACK -- the kernel cannot hang.
//
// Output in LAPACK's dsytrd(lower) convention so that the host can finish with dstemr (selected eigenpairs of the
// tridiagonal, O(kn)) and dormqr (apply the reflectors to k vectors): d, e, tau, and the reflectors as the packed
// column-major block A(2:n, 1:n-1) that dormtr hands to dormqr (V[i][r-1] = v_i[r]; v_i[i+1] = 1, zeros above).
#include "common.h"

#include <algorithm>

namespace msm {

constexpr int TRD_NT = 256;     // threads per workgroup
constexpr int TRD_CPW = 16;     // columns per workgroup
constexpr int TRD_MAXN = 1024;
constexpr long long TRD_SPIN_CYCLES = 400LL * 1000 * 1000;  // ~0.2 s: give up, never hang

struct TrdArgs {
    const double* A;   // n x n symmetric (row-major == column-major)
    int n, G;
    double* d;         // [n]
    double* e;         // [n - 1]
    double* tau;       // [n - 1]
    double* V;         // [n-1][n-1]: row i = reflector i without its first entry = column i of LAPACK's A(2:n, 1:n-1)
    double* pbuf;      // [2][n]   p = tau A v of the current column, all-gathered
    double* cbuf;      // [2][n]   the next pivot column before this column's update
    int* flags;        // [G] step stamps (barrier), zero before the launch
    int* status;       // [1]: 0 ok, 1 a workgroup timed out
};

// deterministic block sum: every thread gets the same value
__device__ __forceinline__ double trd_block_sum(double x, double* red, int tid)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Exchange buffers are written and read with agent-scope relaxed atomics (`global_store/load_dwordx2 ... sc1`: write-through
// stores, L1-bypassing loads), which keeps the release fence of the grid barrier below cheap -- the L2 holds nothing dirty
// of this kernel's for it to write back.  Since round 4 the barrier itself is a release fence / acquire fence pair at
// agent scope (see the loop): the sc1 accesses alone are a form MI355X_MICROARCH.md lists as valid on gfx950, but not one
// the memory model promises.  The kernel sits on the solve's FALLBACK route only (subspace.hip is the default).
__device__ __forceinline__ void trd_store(double* p, double x)
{
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double trd_load(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Thread layout: 256 threads = 32 row lanes x 8 column groups; a thread holds rows rq, rq + 32, ... (RPT of them) of the
// TWO columns of its group, so a column's dot product with v is a 5-step butterfly over the 32 row lanes of a half-wave.
template <int RPT>
__global__ __launch_bounds__(TRD_NT, 1) void sytrd_coop_kernel(TrdArgs P)
{
    __shared__ double v[TRD_MAXN], w[TRD_MAXN], col[TRD_MAXN];
    __shared__ double red[8];
    __shared__ int bail;
    const int tid = threadIdx.x, g = blockIdx.x, n = P.n, G = P.G;
    const int lane = tid & 63, wave = tid >> 6;
    const int rq = tid & 31, cs = tid >> 5;          // row lane, column group
    const int c0 = g + G * (2 * cs), c1 = g + G * (2 * cs + 1);   // this thread's two columns (cyclic over workgroups)

    double a0[RPT], a1[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rq + 32 * k;
        a0[k] = (r < n && c0 < n) ? P.A[(size_t)r * n + c0] : 0.0;
        a1[k] = (r < n && c1 < n) ? P.A[(size_t)r * n + c1] : 0.0;
    }
    if (tid == 0) bail = 0;
    for (int r = tid; r < n; r += TRD_NT) col[r] = P.A[(size_t)r * n];   // pivot column 0
    for (int r = n + tid; r < TRD_MAXN; r += TRD_NT) {                     // rows beyond n: harmless zeros
        col[r] = 0.0;
        v[r] = 0.0;
        w[r] = 0.0;
    }
    __syncthreads();

    for (int i = 0; i + 2 < n; ++i) {
        // ---- reflector i from the current pivot column (redundantly in every workgroup): LAPACK dlarfg
        double part = 0.0;
        for (int r = tid; r < n; r += TRD_NT)
            if (r >= i + 2) part += col[r] * col[r];
        const double xn2 = trd_block_sum(part, red, tid);
        const double alpha = col[i + 1];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (xn2 != 0.0) {
            const double nrm = sqrt(alpha * alpha + xn2);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        for (int r = tid; r < n; r += TRD_NT) v[r] = r > i + 1 ? col[r] * scale : (r == i + 1 ? 1.0 : 0.0);
        const bool rec = g == i % G;   // the owner of column i records d, e, tau and the reflector
        if (rec && tid == 0) {
            P.d[i] = col[i];
            P.e[i] = beta;
            P.tau[i] = tau;
        }
        __syncthreads();
        if (rec)   // LAPACK's A(2:n, 1:n-1) as a packed (n-1) x (n-1) column-major block: column i = v_i[1 .. n-1]
            for (int r = 1 + tid; r < n; r += TRD_NT) P.V[(size_t)i * (n - 1) + (r - 1)] = v[r];

        // ---- p[c] = tau A[:, c] . v for the owned columns c > i; the next pivot column as it is NOW
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const double vr = v[rq + 32 * k];
            d0 += a0[k] * vr;
            d1 += a1[k] * vr;
        }
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) {
            d0 += __shfl_xor(d0, m, 64);
            d1 += __shfl_xor(d1, m, 64);
        }
        double* pb = P.pbuf + (size_t)(i & 1) * n;
        if (rq == 0) {
            if (c0 < n) trd_store(pb + c0, c0 > i ? tau * d0 : 0.0);
            if (c1 < n) trd_store(pb + c1, c1 > i ? tau * d1 : 0.0);
        }
        {
            const int nx = i + 1;              // next pivot column: owner workgroup nx % G, slot nx / G
            if (g == nx % G && cs == (nx / G) / 2) {
                double* cb = P.cbuf + (size_t)(i & 1) * n;
                const bool second = ((nx / G) & 1) != 0;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int r = rq + 32 * k;
                    if (r < n) trd_store(cb + r, second ? a1[k] : a0[k]);
                }
            }
        }
        // ---- grid barrier i: stamp = i + 1 (flag per workgroup; one wave polls all flags, one lane each).
        // Publication order (round 4): the workgroup's exchange stores -> workgroup barrier -> ONE agent-scope
        // RELEASE fence (lane 0) -> drained -> flag store; on the other side relaxed polls -> ONE agent-scope ACQUIRE fence
        // by the polling wave -> workgroup barrier -> the exchange loads.  (Rounds 2-3 published the flag behind a
        // workgroup-scope fence only and relied on the sc1 form of the data accesses.)
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-back is complete before the flag can be seen
            __hip_atomic_store(P.flags + g, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {
            const long long t0 = clock64();
            for (;;) {
                int seen = i + 1;
                for (int q = lane; q < G; q += 64) {
                    const int f = __hip_atomic_load(P.flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (f < i + 1) seen = f;
                }
                if (__all(seen >= i + 1)) break;
                if (clock64() - t0 > TRD_SPIN_CYCLES) {
                    if (lane == 0) {
                        bail = 1;
                        __hip_atomic_store(P.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (bail) return;

        // ---- everyone: w = p + alpha2 v, rank-2 update of the owned columns, next pivot column
        const double* cbr = P.cbuf + (size_t)(i & 1) * n;
        double pv = 0.0;
        {
            // all of the thread's exchange loads first (clamped indices), then the arithmetic: a load that is consumed in
            // the same loop iteration is a round trip per iteration, and this sits on the critical path of every column
            constexpr int NL = TRD_MAXN / TRD_NT;
            double pr[NL], cr[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int r = tid + k * TRD_NT;
                const int rc = r < n ? r : n - 1;
                pr[k] = trd_load(pb + rc);
                cr[k] = trd_load(cbr + rc);
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int r = tid + k * TRD_NT;
                if (r < n) {
                    w[r] = pr[k];
                    pv += pr[k] * v[r];
                    col[r] = r > i ? cr[k] : 0.0;
                }
            }
        }
        const double ptv = trd_block_sum(pv, red, tid);
        const double alpha2 = -0.5 * tau * ptv;
        for (int r = tid; r < n; r += TRD_NT) w[r] = w[r] + alpha2 * v[r];
        __syncthreads();
        const double wn = w[i + 1];                  // v[i + 1] = 1
        for (int r = tid; r < n; r += TRD_NT)
            if (r > i) col[r] = col[r] - v[r] * wn - w[r];
        {
            const double w0 = c0 < n ? w[c0] : 0.0, v0 = c0 < n ? v[c0] : 0.0;
            const double w1 = c1 < n ? w[c1] : 0.0, v1 = c1 < n ? v[c1] : 0.0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const double vr = v[rq + 32 * k], wr = w[rq + 32 * k];
                a0[k] -= vr * w0 + wr * v0;    // columns <= i: w = v = 0 there, nothing changes
                a1[k] -= vr * w1 + wr * v1;
            }
        }
        __syncthreads();
    }
    // the last 2 x 2 block: col = column n - 2 of the final matrix (rows n - 2, n - 1); d[n - 1] from its owner
    if (n >= 2) {
        if (g == (n - 2) % G && tid == 0) {
            P.d[n - 2] = col[n - 2];
            P.e[n - 2] = col[n - 1];
            P.tau[n - 2] = 0.0;
        }
        {
            const int c = n - 1, r = n - 1;
            if (g == c % G && cs == (c / G) / 2 && rq == (r & 31)) {
                const bool second = ((c / G) & 1) != 0;
                double val = 0.0;
#pragma unroll
                for (int k = 0; k < RPT; ++k)
                    if (k == r / 32) val = second ? a1[k] : a0[k];
                P.d[n - 1] = val;
            }
        }
        if (g == (n - 2) % G)
            for (int r = 1 + tid; r < n; r += TRD_NT) P.V[(size_t)(n - 2) * (n - 1) + (r - 1)] = r == n - 1 ? 1.0 : 0.0;
    } else if (g == 0 && tid == 0) {
        P.d[0] = P.A[0];
    }
}

// A (device, n x n symmetric) -> d, e, tau, V (device; V row i = reflector i).  Queued on stream(); *status (device int)
// is 1 if the cooperative kernel gave up (not all workgroups resident) -- the caller then uses LAPACK on the host.
int sytrd_device(const double* A, int n, double* d, double* e, double* tau, double* V, double* work8n, int* status)
{
    if (n < 1 || n > TRD_MAXN) return fail(MSM_ERR_INVALID, "sytrd_device: n = %d out of range (1 .. %d)", n, TRD_MAXN);
    TrdArgs P;
    P.A = A;
    P.n = n;
    P.G = (int)ceil_div(n, TRD_CPW);
    P.d = d;
    P.e = e;
    P.tau = tau;
    P.V = V;
    P.pbuf = work8n;                          // 2n
    P.cbuf = work8n + 2 * (size_t)n;          // 2n
    P.flags = reinterpret_cast<int*>(work8n + 4 * (size_t)n);   // G <= 64 ints inside the remaining 4n doubles
    P.status = status;
    if (P.G > num_cus()) return fail(MSM_ERR_INVALID, "sytrd_device: %d workgroups do not fit %d CUs", P.G, num_cus());
    MSM_HIP_CHECK(hipMemsetAsync(work8n, 0, 8 * (size_t)n * sizeof(double), stream()));   // flags 0
    MSM_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int), stream()));
    const int rpt = (int)ceil_div(n, 32);   // rows per thread (32 row lanes)
    if (rpt <= 4)
        hipLaunchKernelGGL(sytrd_coop_kernel<4>, dim3(P.G), dim3(TRD_NT), 0, stream(), P);
    else if (rpt <= 8)
        hipLaunchKernelGGL(sytrd_coop_kernel<8>, dim3(P.G), dim3(TRD_NT), 0, stream(), P);
    else if (rpt <= 16)
        hipLaunchKernelGGL(sytrd_coop_kernel<16>, dim3(P.G), dim3(TRD_NT), 0, stream(), P);
    else
        hipLaunchKernelGGL(sytrd_coop_kernel<32>, dim3(P.G), dim3(TRD_NT), 0, stream(), P);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

/* Householder tridiagonalisation of a symmetric n x n matrix (host or device per on_device), LAPACK dsytrd(lower) outputs:
 * d[n], e[n-1], tau[n-1] and V[(n-1)*(n-1)] = the reflector block A(2:n, 1:n-1), column-major (what dormqr takes).  *status = 1 when the
 * cooperative kernel could not keep all its workgroups resident (nothing usable was produced). */
int msm_sytrd(const double* A, msm_idx_t n, double* d, double* e, double* tau, double* V, int* status, int on_device)
{
    if (!A || !d || !e || !tau || !V || !status) return fail(MSM_ERR_INVALID, "msm_sytrd: null pointer");
    if (n < 1 || n > TRD_MAXN) return fail(MSM_ERR_INVALID, "msm_sytrd: need 1 <= n <= %d", TRD_MAXN);
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    const size_t nn = (size_t)n * n;
    DevBuf& buf = pool(PS_W);
    int rc = buf.reserve((2 * nn + 11 * (size_t)n + 64) * sizeof(double) + 4096);
    if (rc) return rc;
    double* dA = buf.as<double>();
    double* dV = dA + nn;
    double* dd = dV + nn;
    double* de = dd + n;
    double* dt = de + n;
    double* dw = dt + n;  // 8n: stamped exchange records
    int* dflags = reinterpret_cast<int*>(dw + 8 * (size_t)n);
    const double* src = A;
    if (!on_device) {
        MSM_HIP_CHECK(hipMemcpyAsync(dA, A, nn * sizeof(double), hipMemcpyHostToDevice, stream()));
        src = dA;
    }
    if ((rc = sytrd_device(src, (int)n, dd, de, dt, dV, dw, dflags))) return rc;
    const hipMemcpyKind k = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    int st = 0;
    MSM_HIP_CHECK(hipMemcpyAsync(d, dd, n * sizeof(double), k, stream()));
    if (n > 1) MSM_HIP_CHECK(hipMemcpyAsync(e, de, (n - 1) * sizeof(double), k, stream()));
    if (n > 1) MSM_HIP_CHECK(hipMemcpyAsync(tau, dt, (n - 1) * sizeof(double), k, stream()));
    if (n > 1) MSM_HIP_CHECK(hipMemcpyAsync(V, dV, (size_t)(n - 1) * (n - 1) * sizeof(double), k, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(&st, dflags, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    *status = st;
    return MSM_OK;
}

}  // extern "C"
