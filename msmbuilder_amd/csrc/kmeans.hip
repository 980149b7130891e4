// kmeans.hip -- k-means labelling (GEMM form on fp32 MFMA) and the MiniBatchKMeans
// step for gfx950.
//
// msmbuilder.cluster.MiniBatchKMeans is a 3-line subclass of scikit-learn's
// (/root/reference/msmbuilder/cluster/__init__.py:67-69); the arithmetic restated
// here is scikit-learn's (third-party, unpinned by the reference -- DESIGN.md):
//   labels  = argmin_j ( ||c_j||^2 - 2 x.c_j )   fp32, first minimum wins
//             (sklearn/cluster/_k_means_lloyd.pyx chunked sgemm + argmin)
//   inertia = sum_i ||x_i - c_label(i)||^2       (sklearn _k_means_common.pyx _inertia_dense)
//   update  : c <- (c*w + sum_{i in batch, label=j} x_i) / (w + n_j), w += n_j,
//             samples visited in batch order (sklearn _k_means_minibatch.pyx:59-109)
// The x.c term is a [rows x F] . [F x K] contraction: v_mfma_f32_32x32x2_f32 with
// LDS-staged [128 x 32] row/centre tiles (pitch 33: conflict-free ds_read_b32 for
// the row-strided fragment reads), a running per-lane argmin over centre tiles and
// one wavefront min-reduction (value, lowest index) per row at the end.
#include "common.h"

#include <algorithm>
#include <vector>

namespace msm {

constexpr int KR = 128;   // rows per workgroup
constexpr int KCT = 128;  // centres per tile
constexpr int KBK = 32;   // features per K-step
constexpr int KP = KBK + 1;
constexpr int KNT = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct KmArgs {
    const float* X;         // [n, m] (or gathered batch)
    const msm_idx_t* rows;  // optional row gather (batch indices), else nullptr
    long long n, m, K;
    const float* C;         // device [K, m]
    const float* cnorm;     // device [K]
    int32_t* labels;        // [n]
};

__device__ __forceinline__ void km_load(float4 (&xa)[4], float4 (&ca)[4], const KmArgs& P,
                                        long long row0, long long j0, int k0, int tid)
{
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = r0 + 32 * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
        const long long i = row0 + rr;
        if (i < P.n) {
            const long long r = P.rows ? P.rows[i] : i;
            const float* p = P.X + r * P.m + k0 + c4;
            if (k0 + c4 + 3 < P.m && ((P.m & 3) == 0)) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k0 + c4 + 0 < P.m) v.x = p[0];
                if (k0 + c4 + 1 < P.m) v.y = p[1];
                if (k0 + c4 + 2 < P.m) v.z = p[2];
                if (k0 + c4 + 3 < P.m) v.w = p[3];
            }
        }
        const long long jc = j0 + rr;
        if (jc < P.K) {
            const float* p = P.C + jc * P.m + k0 + c4;
            if (k0 + c4 + 3 < P.m && ((P.m & 3) == 0)) {
                w = *reinterpret_cast<const float4*>(p);
            } else {
                if (k0 + c4 + 0 < P.m) w.x = p[0];
                if (k0 + c4 + 1 < P.m) w.y = p[1];
                if (k0 + c4 + 2 < P.m) w.z = p[2];
                if (k0 + c4 + 3 < P.m) w.w = p[3];
            }
        }
        xa[j] = v;
        ca[j] = w;
    }
}

__device__ __forceinline__ void km_store(const float4 (&xa)[4], const float4 (&ca)[4], float* Xs,
                                         float* Cs, int tid)
{
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float* px = Xs + (r0 + 32 * j) * KP + c4;
        float* pc = Cs + (r0 + 32 * j) * KP + c4;
        px[0] = xa[j].x; px[1] = xa[j].y; px[2] = xa[j].z; px[3] = xa[j].w;
        pc[0] = ca[j].x; pc[1] = ca[j].y; pc[2] = ca[j].z; pc[3] = ca[j].w;
    }
}

__global__ __launch_bounds__(KNT, 2) void kmeans_label_kernel(KmArgs P)
{
    __shared__ float Xs[2][KR * KP];
    __shared__ float Cs[2][KCT * KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    const long long row0 = (long long)blockIdx.x * KR;
    const int nk = (int)((P.m + KBK - 1) / KBK);

    float best[2][16];
    int bidx[2][16];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            best[bi][r] = INFINITY;
            bidx[bi][r] = 0x7fffffff;
        }

    for (long long j0 = 0; j0 < P.K; j0 += KCT) {
        f32x16 acc[2][2];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
        float4 xa[4], ca[4];
        km_load(xa, ca, P, row0, j0, 0, tid);
        __syncthreads();  // previous centre tile's last fragment reads are done
        km_store(xa, ca, Xs[0], Cs[0], tid);
        __syncthreads();
        for (int s = 0; s < nk; ++s) {
            const int buf = s & 1;
            if (s + 1 < nk) km_load(xa, ca, P, row0, j0, (s + 1) * KBK, tid);
            const float* Ab = Xs[buf] + (wr * 64 + cl) * KP + kl;
            const float* Bb = Cs[buf] + (wc * 64 + cl) * KP + kl;
#pragma unroll 4
            for (int kk = 0; kk < KBK / 2; ++kk) {
                const float a0 = Ab[2 * kk], a1 = Ab[32 * KP + 2 * kk];
                const float b0 = Bb[2 * kk], b1 = Bb[32 * KP + 2 * kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (s + 1 < nk) km_store(xa, ca, Xs[buf ^ 1], Cs[buf ^ 1], tid);
            __syncthreads();
        }
        // running argmin over this centre tile (ascending j per lane, strict <)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
            const long long j = j0 + wc * 64 + bj * 32 + cl;
            if (j < P.K) {
                const float cn = P.cnorm[j];
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = cn - 2.f * acc[bi][bj][r];
                        if (v < best[bi][r]) {
                            best[bi][r] = v;
                            bidx[bi][r] = (int)j;
                        }
                    }
            }
        }
    }
    // wavefront min-reduction over the 32 lanes that share a row (value, lowest index)
    float* redv = Xs[0];                            // [2 (wc)][128 rows]
    int* redi = reinterpret_cast<int*>(Cs[0]);      // [2][128]
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = best[bi][r];
            int ix = bidx[bi][r];
#pragma unroll
            for (int msk = 1; msk < 32; msk <<= 1) {
                const float ov = __shfl_xor(v, msk, 64);
                const int oi = __shfl_xor(ix, msk, 64);
                if (ov < v || (ov == v && oi < ix)) {
                    v = ov;
                    ix = oi;
                }
            }
            if (cl == 0) {
                const int row = wr * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                redv[wc * KR + row] = v;
                redi[wc * KR + row] = ix;
            }
        }
    __syncthreads();
    if (tid < KR) {
        const long long i = row0 + tid;
        if (i < P.n) {
            float v0 = redv[tid], v1 = redv[KR + tid];
            int i0 = redi[tid], i1 = redi[KR + tid];
            int lab = (v1 < v0 || (v1 == v0 && i1 < i0)) ? i1 : i0;
            if (lab == 0x7fffffff) lab = 0;  // all-NaN row: sklearn's argmin returns 0
            P.labels[i] = lab;
        }
    }
}

// per-row ||x - c_label||^2 (fp32 difference, fp64 accumulate), one wave per row;
// per-block fp64 partial sums for the inertia.
__global__ __launch_bounds__(KNT) void kmeans_inertia_kernel(KmArgs P, double* __restrict__ partial)
{
    __shared__ double red[KNT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double tot = 0.0;
    for (long long i = (long long)blockIdx.x * 4 + wave; i < P.n; i += (long long)gridDim.x * 4) {
        const long long r = P.rows ? P.rows[i] : i;
        const float* x = P.X + r * P.m;
        const float* c = P.C + (long long)P.labels[i] * P.m;
        double s = 0.0;
        for (long long k = lane; k < P.m; k += 64) {
            const float d = x[k] - c[k];
            s += (double)d * (double)d;
        }
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) s += __shfl_xor(s, msk, 64);
        tot += s;
    }
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// One workgroup per centre: scan the batch labels, visit members in batch order.
// apply != 0: sklearn's streaming-mean update in fp32, in place on centers/counts.
// sums/cnts (nullable): fp64 batch sums and counts for the multi-GPU all-reduce.
__global__ __launch_bounds__(KNT) void mbk_update_kernel(KmArgs P, float* __restrict__ centers,
                                                         float* __restrict__ counts,
                                                         double* __restrict__ sums,
                                                         double* __restrict__ cnts, int apply)
{
    extern __shared__ int members[];  // compacted member positions, chunked
    __shared__ int nmem;
    const int j = blockIdx.x, tid = threadIdx.x;
    const int CH = 4096;
    const float w_old = counts[j];
    // per-thread feature accumulators live in a loop over feature blocks of KNT
    long long total = 0;
    for (long long f0 = 0; f0 < P.m; f0 += KNT) {
        const long long f = f0 + tid;
        float acc32 = (f < P.m) ? centers[(long long)j * P.m + f] * w_old : 0.f;
        double acc64 = 0.0;
        long long cnt = 0;
        for (long long b0 = 0; b0 < P.n; b0 += CH) {
            __syncthreads();
            if (tid == 0) {
                int k = 0;
                const long long be = std::min<long long>(P.n, b0 + CH);
                for (long long b = b0; b < be; ++b)
                    if (P.labels[b] == j) members[k++] = (int)(b - b0);
                nmem = k;
            }
            __syncthreads();
            cnt += nmem;
            if (f < P.m) {
                for (int k = 0; k < nmem; ++k) {
                    const long long b = b0 + members[k];
                    const long long r = P.rows ? P.rows[b] : b;
                    const float x = P.X[r * P.m + f];
                    acc32 += x;
                    acc64 += (double)x;
                }
            }
        }
        total = cnt;
        if (f < P.m) {
            if (sums) sums[(long long)j * P.m + f] = acc64;
            if (apply && cnt > 0) {
                const float w_new = w_old + (float)cnt;
                const float alpha = 1.0f / w_new;
                centers[(long long)j * P.m + f] = acc32 * alpha;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (cnts) cnts[j] = (double)total;
        if (apply && total > 0) counts[j] = w_old + (float)total;
    }
}

static int km_prepare(const float* centers, msm_idx_t K, msm_idx_t m, DevBuf& dC, float** dCent, float** dNorm)
{
    int rc = dC.reserve(((size_t)K * m + (size_t)K) * sizeof(float));
    if (rc) return rc;
    std::vector<float> cn((size_t)K);
    for (msm_idx_t j = 0; j < K; ++j) {
        double s = 0.0;
        for (msm_idx_t f = 0; f < m; ++f) s += (double)centers[j * m + f] * (double)centers[j * m + f];
        cn[(size_t)j] = (float)s;
    }
    *dCent = dC.as<float>();
    *dNorm = *dCent + (size_t)K * m;
    MSM_HIP_CHECK(hipMemcpyAsync(*dCent, centers, (size_t)K * m * sizeof(float), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(*dNorm, cn.data(), (size_t)K * sizeof(float), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // `cn` is a stack-frame vector
    return MSM_OK;
}

static int km_label_and_inertia(KmArgs& P, double* inertia)
{
    const unsigned grid = (unsigned)ceil_div(P.n, KR);
    hipLaunchKernelGGL(kmeans_label_kernel, dim3(grid), dim3(KNT), 0, stream(), P);
    MSM_HIP_CHECK(hipGetLastError());
    if (inertia) {
        const int nb = (int)std::min<long long>(ceil_div(P.n, 4), 1024);
        DevBuf& dPart = pool(PS_PART);
        int rc = dPart.reserve((size_t)nb * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(kmeans_inertia_kernel, dim3(nb), dim3(KNT), 0, stream(), P, dPart.as<double>());
        MSM_HIP_CHECK(hipGetLastError());
        std::vector<double> h((size_t)nb);
        MSM_HIP_CHECK(hipMemcpyAsync(h.data(), dPart.p, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += h[(size_t)i];
        *inertia = s;
    }
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_kmeans_label_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* centers,
                         msm_idx_t K, int32_t* labels, double* inertia, int on_device)
{
    if (!X || !centers || !labels) return fail(MSM_ERR_INVALID, "kmeans_label: null pointer");
    if (n < 0 || m < 1 || K < 1) return fail(MSM_ERR_INVALID, "kmeans_label: bad shape");
    if (inertia) *inertia = 0.0;
    if (n == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf &dC = pool(PS_Y), &dX = pool(PS_X), &dL = pool(PS_LAB);
    float *dCent, *dNorm;
    int rc = km_prepare(centers, K, m, dC, &dCent, &dNorm);
    if (rc) return rc;
    KmArgs P;
    memset(&P, 0, sizeof(P));
    P.n = n;
    P.m = m;
    P.K = K;
    P.C = dCent;
    P.cnorm = dNorm;
    if (on_device) {
        P.X = X;
        P.labels = labels;
    } else {
        if ((rc = dX.reserve((size_t)n * m * sizeof(float)))) return rc;
        if ((rc = dL.reserve((size_t)n * sizeof(int32_t)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dX.p, X, (size_t)n * m * sizeof(float), hipMemcpyHostToDevice, stream()));
        P.X = dX.as<float>();
        P.labels = dL.as<int32_t>();
    }
    if ((rc = km_label_and_inertia(P, inertia))) return rc;
    if (!on_device)
        MSM_HIP_CHECK(hipMemcpyAsync(labels, P.labels, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_step_f32(const float* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx,
                     msm_idx_t B, float* centers, float* counts, msm_idx_t K,
                     double* batch_inertia, double* batch_sums, double* batch_counts,
                     int apply_update, int on_device)
{
    if (!X || !batch_idx || !centers || !counts) return fail(MSM_ERR_INVALID, "mbk_step: null pointer");
    if (n < 1 || m < 1 || K < 1 || B < 1) return fail(MSM_ERR_INVALID, "mbk_step: bad shape");
    for (msm_idx_t b = 0; b < B; ++b)
        if (batch_idx[b] < 0 || batch_idx[b] >= n) return fail(MSM_ERR_INVALID, "mbk_step: batch index out of range");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf &dC = pool(PS_Y), &dXb = pool(PS_X), &dIdx = pool(PS_IDX), &dL = pool(PS_LAB), &dW = pool(PS_W), &dS = pool(PS_S);
    float *dCent, *dNorm;
    int rc = km_prepare(centers, K, m, dC, &dCent, &dNorm);
    if (rc) return rc;
    KmArgs P;
    memset(&P, 0, sizeof(P));
    P.n = B;
    P.m = m;
    P.K = K;
    P.C = dCent;
    P.cnorm = dNorm;
    if ((rc = dL.reserve((size_t)B * sizeof(int32_t)))) return rc;
    P.labels = dL.as<int32_t>();
    if (on_device) {
        if ((rc = dIdx.reserve((size_t)B * sizeof(msm_idx_t)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dIdx.p, batch_idx, (size_t)B * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
        P.X = X;
        P.rows = dIdx.as<msm_idx_t>();
    } else {
        // gather the batch on the host, ship only B rows
        std::vector<float> xb((size_t)B * m);
        for (msm_idx_t b = 0; b < B; ++b)
            memcpy(xb.data() + (size_t)b * m, X + batch_idx[b] * m, (size_t)m * sizeof(float));
        if ((rc = dXb.reserve(xb.size() * sizeof(float)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dXb.p, xb.data(), xb.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        P.X = dXb.as<float>();
        P.rows = nullptr;
    }
    if ((rc = km_label_and_inertia(P, batch_inertia))) return rc;
    if ((rc = dW.reserve((size_t)K * sizeof(float)))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(dW.p, counts, (size_t)K * sizeof(float), hipMemcpyHostToDevice, stream()));
    double *dSums = nullptr, *dCnts = nullptr;
    if (batch_sums || batch_counts) {
        if ((rc = dS.reserve(((size_t)K * m + (size_t)K) * sizeof(double)))) return rc;
        dSums = dS.as<double>();
        dCnts = dSums + (size_t)K * m;
    }
    hipLaunchKernelGGL(mbk_update_kernel, dim3((unsigned)K), dim3(KNT), 4096 * sizeof(int), stream(), P,
                       dCent, dW.as<float>(), dSums, dCnts, apply_update);
    MSM_HIP_CHECK(hipGetLastError());
    if (apply_update) {
        MSM_HIP_CHECK(hipMemcpyAsync(centers, dCent, (size_t)K * m * sizeof(float), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(counts, dW.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, stream()));
    }
    if (batch_sums)
        MSM_HIP_CHECK(hipMemcpyAsync(batch_sums, dSums, (size_t)K * m * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (batch_counts)
        MSM_HIP_CHECK(hipMemcpyAsync(batch_counts, dCnts, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"
