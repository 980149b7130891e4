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
#include "kmeans_f64_dev.h"

#include <algorithm>
#include <vector>

namespace msm {

constexpr int KR = 128;   // rows per workgroup
constexpr int KCT = 128;  // centres per tile
constexpr int KBK = 32;   // features per K-step
constexpr int KP = KBK + 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));

using KmArgs = KmArgsT<float>;

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
    if (P.stop && *P.stop) return;  // uniform
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

    const long long jbeg = P.jspan ? (long long)blockIdx.y * P.jspan : 0;
    const long long jend = P.jspan ? (jbeg + P.jspan < P.K ? jbeg + P.jspan : P.K) : P.K;
    for (long long j0 = jbeg; j0 < jend; j0 += KCT) {
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
            const bool second = (v1 < v0 || (v1 == v0 && i1 < i0));
            int lab = second ? i1 : i0;
            if (P.jspan) {
                P.pv[(long long)blockIdx.y * P.n + i] = second ? v1 : v0;
                P.pi[(long long)blockIdx.y * P.n + i] = lab;
            } else {
                if (lab == 0x7fffffff) lab = 0;  // all-NaN row: sklearn's argmin returns 0
                P.labels[i] = lab;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Small batches of wide rows (MiniBatchKMeans' step at F = 512: B = 1024 rows, K = 1000 centres): 128 x 128 tiles make 8 x 8
// = 64 workgroups -- a quarter of the chip, each MFMA-bound for 27 us.  Same arithmetic on 64 x 64 tiles (one 32 x 32
// MFMA block per wave): 16 x 16 = 256 workgroups.  Simple double-buffered K-loop (the panels are L2-resident).
// ---------------------------------------------------------------------------
constexpr int KS64 = 64;
constexpr int KB64 = 128;  // features per K-step: few, long steps (a step costs ~1.5 us of latency whatever its length)
constexpr int KP64 = KB64 + 4;  // 16-byte aligned rows; 16 lanes x 16 bytes at this pitch cover the 64 banks once
constexpr size_t KM64_LDS = (size_t)2 * 2 * KS64 * KP64 * sizeof(float);

struct Km64Stage {
    float4 x[KB64 / 16], c[KB64 / 16];
};

// Loads are UNCONDITIONAL on the 16-byte path (rows clamped into the batch, centres into [0, K), columns into the row;
// what lies outside is zeroed when the stage goes to LDS, or never read back): a load under a branch or a select is
// followed at once by s_waitcnt vmcnt(0), and sixteen serialised L2 round trips made a K-step 6 us instead of 1.7.
__device__ __forceinline__ void km64_load(Km64Stage& st, const KmArgs& P, const long long (&xrow)[KB64 / 16], long long j0,
                                          int k0, int tid)
{
    constexpr int CPR = KB64 / 4;       // threads per row
    constexpr int RPP = KNT / CPR;      // rows per pass
    const int c4 = (tid % CPR) * 4;
    const int r0 = tid / CPR;
    const bool vec = (P.m & 3) == 0 && ((((uintptr_t)P.X) | ((uintptr_t)P.C)) & 15) == 0;
    if (vec) {  // uniform
        const long long col = (k0 + c4 + 3 < P.m) ? (long long)(k0 + c4) : P.m - 4;
#pragma unroll
        for (int j = 0; j < KS64 / RPP; ++j) {
            const long long jc = j0 + r0 + RPP * j;
            st.x[j] = *reinterpret_cast<const float4*>(P.X + xrow[j] * P.m + col);
            st.c[j] = *reinterpret_cast<const float4*>(P.C + (jc < P.K ? jc : P.K - 1) * P.m + col);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < KS64 / RPP; ++j) {
        const int rr = r0 + RPP * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
        {
            const float* p = P.X + xrow[j] * P.m + k0 + c4;
            if (k0 + c4 + 0 < P.m) v.x = p[0];
            if (k0 + c4 + 1 < P.m) v.y = p[1];
            if (k0 + c4 + 2 < P.m) v.z = p[2];
            if (k0 + c4 + 3 < P.m) v.w = p[3];
        }
        const long long jc = j0 + rr;
        if (jc < P.K) {
            const float* p = P.C + jc * P.m + k0 + c4;
            if (k0 + c4 + 0 < P.m) w.x = p[0];
            if (k0 + c4 + 1 < P.m) w.y = p[1];
            if (k0 + c4 + 2 < P.m) w.z = p[2];
            if (k0 + c4 + 3 < P.m) w.w = p[3];
        }
        st.x[j] = v;
        st.c[j] = w;
    }
}

// `inb`: this thread's four columns of the step lie inside the row (else the stage holds clamped-address data: zeros go to LDS)
__device__ __forceinline__ void km64_store(const Km64Stage& st, float* Xs, float* Cs, int tid, bool inb)
{
    constexpr int CPR = KB64 / 4, RPP = KNT / CPR;
    const int c4 = (tid % CPR) * 4, r0 = tid / CPR;
#pragma unroll
    for (int j = 0; j < KS64 / RPP; ++j) {
        float* px = Xs + (r0 + RPP * j) * KP64 + c4;
        float* pc = Cs + (r0 + RPP * j) * KP64 + c4;
        *reinterpret_cast<float4*>(px) = inb ? st.x[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(pc) = inb ? st.c[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ __launch_bounds__(KNT) void kmeans_label64_kernel(KmArgs P)
{
    if (P.stop && *P.stop) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) char km64_smem[];
    float* Xs = reinterpret_cast<float*>(km64_smem);  // [2][KS64 * KP64]
    float* Cs = Xs + 2 * KS64 * KP64;                 // [2][KS64 * KP64]
    __shared__ float redv[2][KS64];
    __shared__ int redi[2][KS64];
    constexpr int CPR = KB64 / 4, RPP = KNT / CPR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    const long long row0 = (long long)blockIdx.x * KS64;
    const int nk = (int)((P.m + KB64 - 1) / KB64);
    const int r0 = tid / CPR;
    long long xrow[KS64 / RPP];  // this thread's staging rows (fixed for the workgroup's life)
#pragma unroll
    for (int j = 0; j < KS64 / RPP; ++j) {
        long long i = row0 + r0 + RPP * j;
        if (i > P.n - 1) i = P.n - 1;  // rows past the batch: clamped (their results are never written)
        xrow[j] = P.rows ? P.rows[i] : i;
    }
    const int c4s = (tid % CPR) * 4;  // this thread's first column inside a K-step
    float best[16];
    int bidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        best[r] = INFINITY;
        bidx[r] = 0x7fffffff;
    }
    const long long jbeg = P.jspan ? (long long)blockIdx.y * P.jspan : 0;
    const long long jend = P.jspan ? (jbeg + P.jspan < P.K ? jbeg + P.jspan : P.K) : P.K;
    for (long long j0 = jbeg; j0 < jend; j0 += KS64) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // The panels come from the fabric side (the centres were rewritten by the previous step, the batch rows are fresh):
        // ~4 us a round trip, against 1.7 us of MFMA per K-step.  Four K-steps of loads are in flight (128 VGPRs; the
        // workgroup has a CU to itself), refilled as each stage goes to LDS.
        Km64Stage st[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nk) km64_load(st[q], P, xrow, j0, q * KB64, tid);
        __syncthreads();  // the previous centre tile's last fragment reads are done
        km64_store(st[0], Xs, Cs, tid, c4s + 3 < P.m);
        __syncthreads();
        for (int s0 = 0; s0 < nk; s0 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = s0 + q;
                if (s < nk) {  // uniform
                    const int buf = q & 1;
                    if (s + 4 < nk) km64_load(st[q], P, xrow, j0, (s + 4) * KB64, tid);  // st[q] went to LDS a step ago
                    // feature order of kmeans_label_v4_kernel (MFMA q of every group of 8 features contracts {q, 4 + q}):
                    // the two kernels then form bit-identical dot products, and a row gets the same label from either
                    const float* Ab = Xs + buf * (KS64 * KP64) + (wr * 32 + cl) * KP64 + 4 * kl;
                    const float* Bb = Cs + buf * (KS64 * KP64) + (wc * 32 + cl) * KP64 + 4 * kl;
#pragma unroll
                    for (int g = 0; g < KB64 / 8; ++g) {  // a lane's 16-byte fragment: its 4 features of the group
                        const float4 a = *reinterpret_cast<const float4*>(Ab + 8 * g), b = *reinterpret_cast<const float4*>(Bb + 8 * g);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                    }
                    if (s + 1 < nk)
                        km64_store(st[(q + 1) & 3], Xs + (buf ^ 1) * (KS64 * KP64), Cs + (buf ^ 1) * (KS64 * KP64), tid,
                                   (s + 1) * KB64 + c4s + 3 < P.m);
                    __syncthreads();
                }
            }
        }
        const long long j = j0 + wc * 32 + cl;  // running argmin over this centre tile (ascending j per lane, strict <)
        if (j < jend) {
            const float cn = P.cnorm[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = cn - 2.f * acc[r];
                if (v < best[r]) {
                    best[r] = v;
                    bidx[r] = (int)j;
                }
            }
        }
    }
    // min over the 32 lanes that share a row (value, lowest index), then over the two centre halves
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = best[r];
        int ix = bidx[r];
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
            const int row = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            redv[wc][row] = v;
            redi[wc][row] = ix;
        }
    }
    __syncthreads();
    if (tid < KS64) {
        const long long i = row0 + tid;
        if (i < P.n) {
            const float v0 = redv[0][tid], v1 = redv[1][tid];
            const int i0 = redi[0][tid], i1 = redi[1][tid];
            const bool second = (v1 < v0 || (v1 == v0 && i1 < i0));
            int lab = second ? i1 : i0;
            if (P.jspan) {
                P.pv[(long long)blockIdx.y * P.n + i] = second ? v1 : v0;
                P.pi[(long long)blockIdx.y * P.n + i] = lab;
            } else {
                if (lab == 0x7fffffff) lab = 0;  // all-NaN row: sklearn's argmin returns 0
                P.labels[i] = lab;
            }
        }
    }
}

// running argmin over one finished centre tile (ascending j per lane, strict <); clears acc
__device__ __forceinline__ void km4_argmin(f32x16 (&acc)[2][2], float (&best)[2][16], int (&bidx)[2][16],
                                           const KmArgs& P, long long j0, int wc, int cl)
{
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
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
    }
}

// wavefront min-reduction over the 32 lanes that share a row (value, lowest index), then the two
// centre halves; writes labels (or the split launch's candidates)
__device__ __forceinline__ void km_finish_rows(const float (&best)[2][16], const int (&bidx)[2][16], const KmArgs& P,
                                               float* redv, int* redi, long long row0, int tid, int wr, int wc,
                                               int kl, int cl, int split)
{
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
            const bool second = (v1 < v0 || (v1 == v0 && i1 < i0));
            int lab = second ? i1 : i0;
            if (P.jspan) {
                P.pv[(long long)split * P.n + i] = second ? v1 : v0;
                P.pi[(long long)split * P.n + i] = lab;
            } else {
                if (lab == 0x7fffffff) lab = 0;  // all-NaN row: sklearn's argmin returns 0
                P.labels[i] = lab;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Fast path (m % 4 == 0, 16-byte aligned rows): same tiling, restructured for the MFMA pipe.
//  * LDS tiles stay row-major [128][KP4=36] (16-byte aligned rows -> ds_write_b128 in,
//    ds_read_b128 out).  A lane's b128 fragment holds 4 CONSECUTIVE features of its row; the
//    32x32x2 MFMA wants features (k, k+1) from lanes (kl=0, kl=1), so within every group of 8
//    features MFMA q contracts features {q, 4+q}: a permutation of the summation order applied
//    to rows and centres alike (the reference arithmetic is an sgemm whose order is unspecified).
//    Pitch 36 words: 16 lanes x b128 cover all 64 banks exactly once.
//  * (centre tile, K-step) pairs form ONE flat iteration stream; the register pipeline is two
//    iterations deep and never drains at a centre-tile boundary.  Loads are unconditional
//    (rows/centres/columns clamped, out-of-range columns zeroed at LDS-store time) and the
//    4 feature groups of a step are fully unrolled: branches or loops around in-flight loads
//    make the compiler wait vmcnt(0) (see tica.hip).
// ---------------------------------------------------------------------------
constexpr int KP4 = KBK + 4;

struct KmStage {
    float4 x[4], c[4];
};

// Like the tICA kernel (tica.hip, "staging with an INTERIOR fast path"): a wave's non-MFMA instructions
// crawl while the co-resident wave streams MFMAs, so a K-step carries as few of them as possible and
// issues its 8 global loads and 8 LDS writes from INSIDE its own MFMA stream.  Interior steps (all 32
// columns inside [0, m), every step but a partial last one) load through per-lane offsets that are
// constant per centre tile on top of scalar bases, and write the loaded registers to LDS unchanged;
// clamps and zero-masks live in uniform branches that hold VALU work only.
template <bool GATHER>
__global__ __launch_bounds__(KNT, 2) void kmeans_label_v4_kernel(KmArgs P)
{
    if (P.stop && *P.stop) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) char km_smem[];
    float* Xs = reinterpret_cast<float*>(km_smem);  // [2][KR * KP4]
    float* Cs = Xs + 2 * KR * KP4;                  // [2][KCT * KP4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, kl = lane >> 5, cl = lane & 31;
    // Round 5 (P.xcd_ns): a workgroup that walks ALL centre tiles streams its 128 rows once per tile, and with 64 workgroups
    // per XCD those 8 x 256 KB re-reads never hit the 4 MB L2 (1M x 512, K = 1000: 16 GB fetched per pass for 2 GB of rows).
    // Instead one workgroup per (row block, centre tile), numbered so that the tiles of a row block are consecutive
    // workgroups of ONE XCD (workgroup b runs on XCD b % 8): they run side by side, the row block is fetched once and
    // served to the other tiles from that XCD's L2; the per-tile candidates are merged by the inertia / reduce kernel.
    long long rb = blockIdx.x;
    int split = (int)blockIdx.y;
    if (P.xcd_ns) {
        const unsigned b = blockIdx.x, q = b >> 3;
        split = (int)(q % (unsigned)P.xcd_ns);
        rb = (long long)(q / (unsigned)P.xcd_ns) * 8 + (b & 7);
        if (rb * KR >= P.n) return;   // (the grid is rounded up to whole groups of 8 row blocks)
    }
    const long long row0 = rb * KR;
    const int m = (int)P.m;
    const int nk = (m + KBK - 1) / KBK;
    const unsigned ldb = (unsigned)m * 4u;
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;

    float best[2][16];
    int bidx[2][16];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            best[bi][r] = INFINITY;
            bidx[bi][r] = 0x7fffffff;
        }

    // this thread's 4 staging rows of X (fixed for the workgroup's life; clamped into [0, n)):
    // contiguous rows -> one scalar base + 32-bit lane offsets; gathered rows -> 64-bit lane pointers
    const global_ptr<char> Xg = as_global<char>(P.X) + (GATHER ? (size_t)0 : (size_t)row0 * ldb);
    global_ptr<char> xp[4];
    unsigned xo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        long long i = row0 + r0 + 32 * j;
        if (i > P.n - 1) i = P.n - 1;
        if (GATHER) {
            xp[j] = Xg + (size_t)as_global<msm_idx_t>(P.rows)[i] * ldb + 4u * (unsigned)c4;
            xo[j] = 0;
        } else {
            xp[j] = Xg;
            xo[j] = (unsigned)(i - row0) * ldb + 4u * (unsigned)c4;
        }
    }
    const global_ptr<char> Cg = as_global<char>(P.C);

    const long long jbeg = P.jspan ? (long long)split * P.jspan : 0;
    const long long jend = P.jspan ? (jbeg + P.jspan < P.K ? jbeg + P.jspan : P.K) : P.K;
    const long long total = ((jend - jbeg + KCT - 1) / KCT) * nk;

    f32x16 acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;

    // load cursor (runs two iterations ahead of the compute cursor; parks on the last tile) and the
    // centre-row lane offsets of its tile (rows clamped to K - 1: recomputed when the tile changes)
    int ls = 0;
    long long lj0 = jbeg;
    unsigned co[4];
#define KM4_TILE_OFFS                                                                             \
    {                                                                                             \
        const long long lim = P.K - 1 - lj0;                                                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const int rr = r0 + 32 * j;                                                           \
            co[j] = (unsigned)(rr < lim ? rr : (int)lim) * ldb + 4u * (unsigned)c4;               \
        }                                                                                         \
    }
    KM4_TILE_OFFS
    // addresses of the load cursor's step: scalar byte offset of its first column + (partial last step
    // only) a per-lane column correction; `un` = the step needs no zero-masking when it reaches LDS
#define KM4_ADDR(KOFF, CADJ, UN)                                                                  \
    {                                                                                             \
        KOFF = (unsigned)ls * (KBK * 4u);                                                         \
        CADJ = 0;                                                                                 \
        UN = 1;                                                                                   \
        if (ls * KBK + KBK > m) { /* partial last K-step: clamp this lane's columns into the row */ \
            const int col = ls * KBK + c4;                                                        \
            CADJ = col < m ? 0u : 4u * (unsigned)(col - (m - 4));                                 \
            UN = 0;                                                                               \
        }                                                                                         \
    }
#define KM4_ADVANCE                                                                               \
    if (++ls == nk) {                                                                             \
        ls = 0;                                                                                   \
        if (lj0 + KCT < jend) {                                                                   \
            lj0 += KCT;                                                                           \
            KM4_TILE_OFFS                                                                         \
        }                                                                                         \
    }
#define KM4_LD_X(J, KOFF, CADJ)                                                                   \
    (GATHER ? load16_global<char>(xp[J] + ((long long)(KOFF) - (long long)(CADJ)))                \
            : load16_global<char>(xp[J] + (size_t)(KOFF) + (xo[J] - (CADJ))))
#define KM4_LD_C(J, KOFF, CADJ) load16_global<char>(Cg + (size_t)lj0c * ldb + (size_t)(KOFF) + (co_c[J] - (CADJ)))
    // zero this lane's out-of-range columns of a loaded stage (partial last K-step only)
#define KM4_MASK(ST, INB)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        ST.x[j] = make_float4(INB ? ST.x[j].x : 0.f, INB ? ST.x[j].y : 0.f, INB ? ST.x[j].z : 0.f, INB ? ST.x[j].w : 0.f); \
        ST.c[j] = make_float4(INB ? ST.c[j].x : 0.f, INB ? ST.c[j].y : 0.f, INB ? ST.c[j].z : 0.f, INB ? ST.c[j].w : 0.f); \
    }
    KmStage st0, st1;
    int un0 = 1, un1 = 1, inb0 = 1, inb1 = 1;
    {   // prologue: step 0 -> LDS, step 1 -> registers
        unsigned koff, cadj;
        int un;
        long long lj0c = lj0;
        unsigned co_c[4] = {co[0], co[1], co[2], co[3]};
        KM4_ADDR(koff, cadj, un)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st0.x[j] = KM4_LD_X(j, koff, cadj);
            st0.c[j] = KM4_LD_C(j, koff, cadj);
        }
        const bool inb = cadj == 0;
        if (!un) { KM4_MASK(st0, inb) }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(Xs + (r0 + 32 * j) * KP4 + c4) = st0.x[j];
            *reinterpret_cast<float4*>(Cs + (r0 + 32 * j) * KP4 + c4) = st0.c[j];
        }
        KM4_ADVANCE
        lj0c = lj0;
        co_c[0] = co[0]; co_c[1] = co[1]; co_c[2] = co[2]; co_c[3] = co[3];
        KM4_ADDR(koff, cadj, un0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st0.x[j] = KM4_LD_X(j, koff, cadj);
            st0.c[j] = KM4_LD_C(j, koff, cadj);
        }
        inb0 = cadj == 0;
        KM4_ADVANCE
    }
    __syncthreads();

    int s = 0;
    long long j0 = jbeg;
#define KM4_MFMA4(A0, A1, B0, B1)                                                                 \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, acc[0][0], 0, 0, 0);                 \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B1, acc[0][1], 0, 0, 0);                 \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B0, acc[1][0], 0, 0, 0);                 \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, acc[1][1], 0, 0, 0);
#define KM4_STEP(SNEXT, UNEXT, INEXT, SLOAD, ULOAD, ILOAD, BUF)                                   \
    {                                                                                             \
        /* addresses of iteration it+2 (no loads yet); the cursor moves on */                     \
        unsigned koff, cadj;                                                                      \
        const long long lj0c = lj0;                                                               \
        const unsigned co_c[4] = {co[0], co[1], co[2], co[3]};                                    \
        KM4_ADDR(koff, cadj, ULOAD)                                                               \
        ILOAD = cadj == 0;                                                                        \
        KM4_ADVANCE                                                                               \
        /* iteration it+1's panel is about to go to LDS: zero-mask it if it is a partial step */  \
        if (!UNEXT) { const bool inb = INEXT != 0; KM4_MASK(SNEXT, inb) }                         \
        const float* Ab = Xs + (BUF) * (KR * KP4) + (wr * 64 + cl) * KP4 + kl * 4;                \
        const float* Bb = Cs + (BUF) * (KCT * KP4) + (wc * 64 + cl) * KP4 + kl * 4;               \
        float* Xw = Xs + ((BUF) ^ 1) * (KR * KP4) + r0 * KP4 + c4;                                \
        float* Cw = Cs + ((BUF) ^ 1) * (KCT * KP4) + r0 * KP4 + c4;                               \
        float4 a0 = *reinterpret_cast<const float4*>(Ab), a1 = *reinterpret_cast<const float4*>(Ab + 32 * KP4); \
        float4 b0 = *reinterpret_cast<const float4*>(Bb), b1 = *reinterpret_cast<const float4*>(Bb + 32 * KP4); \
        _Pragma("unroll") for (int g = 0; g < KBK / 8; ++g) {                                     \
            const int gn = (g + 1 < KBK / 8) ? g + 1 : g;                                         \
            const float4 na0 = *reinterpret_cast<const float4*>(Ab + gn * 8);                     \
            const float4 na1 = *reinterpret_cast<const float4*>(Ab + 32 * KP4 + gn * 8);          \
            const float4 nb0 = *reinterpret_cast<const float4*>(Bb + gn * 8);                     \
            const float4 nb1 = *reinterpret_cast<const float4*>(Bb + 32 * KP4 + gn * 8);          \
            /* memory ops of this step, spread over the four MFMA quads of each feature group:   */ \
            /* groups 0-1: the 8 loads of iteration it+2; groups 2-3: the 8 LDS writes of it+1    */ \
            if (g < 2) { SLOAD.x[2 * g] = KM4_LD_X(2 * g, koff, cadj); SLOAD.c[2 * g] = KM4_LD_C(2 * g, koff, cadj); } \
            if (g >= 2) { *reinterpret_cast<float4*>(Xw + (2 * (g - 2)) * 32 * KP4) = SNEXT.x[2 * (g - 2)];           \
                          *reinterpret_cast<float4*>(Cw + (2 * (g - 2)) * 32 * KP4) = SNEXT.c[2 * (g - 2)]; }         \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            KM4_MFMA4(a0.x, a1.x, b0.x, b1.x)                                                     \
            KM4_MFMA4(a0.y, a1.y, b0.y, b1.y)                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            if (g < 2) { SLOAD.x[2 * g + 1] = KM4_LD_X(2 * g + 1, koff, cadj); SLOAD.c[2 * g + 1] = KM4_LD_C(2 * g + 1, koff, cadj); } \
            if (g >= 2) { *reinterpret_cast<float4*>(Xw + (2 * (g - 2) + 1) * 32 * KP4) = SNEXT.x[2 * (g - 2) + 1];   \
                          *reinterpret_cast<float4*>(Cw + (2 * (g - 2) + 1) * 32 * KP4) = SNEXT.c[2 * (g - 2) + 1]; } \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            KM4_MFMA4(a0.z, a1.z, b0.z, b1.z)                                                     \
            KM4_MFMA4(a0.w, a1.w, b0.w, b1.w)                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;                                               \
        }                                                                                         \
        __syncthreads();                                                                          \
        if (++s == nk) {                                                                          \
            s = 0;                                                                                \
            km4_argmin(acc, best, bidx, P, j0, wc, cl);                                           \
            j0 += KCT;                                                                            \
        }                                                                                         \
    }
    for (long long it = 0; it < total; it += 2) {
        KM4_STEP(st0, un0, inb0, st1, un1, inb1, 0)
        ++it;
        if (it < total) KM4_STEP(st1, un1, inb1, st0, un0, inb0, 1)
        --it;
    }
#undef KM4_STEP
#undef KM4_MFMA4
#undef KM4_MASK
#undef KM4_LD_C
#undef KM4_LD_X
#undef KM4_ADVANCE
#undef KM4_ADDR
#undef KM4_TILE_OFFS
    km_finish_rows(best, bidx, P, Xs, reinterpret_cast<int*>(Cs), row0, tid, wr, wc, kl, cl, split);
}

// per-row ||x - c_label||^2 (fp32 difference, fp64 accumulate), one wave per row;
// per-block fp64 partial sums for the inertia.
// nsplit > 1 (centre-split labelling of a small batch): the row's label is first picked from the splits' candidates
// (lowest value, then lowest index -- what kmeans_label_reduce_kernel does as a launch of its own) and written out.
// Round 5: two rows per wave in flight and 16-byte loads when the rows allow it (m % 4 == 0, 16-byte aligned bases) -- one row
// at a time with 4-byte loads and three dependent round trips per row (candidates -> centre row -> sum) ran at 1.7 TB/s
// (1.5 ms per 1.25M x 512 pass beside a 10.4 ms labelling kernel; profiles/r05_label_wide.txt).
template <typename T>
__global__ __launch_bounds__(KNT) void kmeans_inertia_kernel(KmArgsT<T> P, double* __restrict__ partial, int nsplit)
{
    if (P.stop && *P.stop) return;  // uniform
    __shared__ double red[KNT / 64];
    constexpr int E = 16 / (int)sizeof(T);   // elements of a 16-byte load: 4 floats / 2 doubles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec4 = (P.m % E) == 0 && ((((uintptr_t)P.X) | ((uintptr_t)P.C)) & 15) == 0;
    const long long m4 = P.m / E;
    // the row's label: from the splits' candidates (lane q fetches split q's: one round trip, then a butterfly for the lowest
    // (value, index)) or as the labelling kernel wrote it
    auto cand_load = [&](long long i, T& bv, int& bi) {
        bv = (T)INFINITY;
        bi = 0x7fffffff;
        if (nsplit > 1) {
            for (int q0 = 0; q0 < nsplit; q0 += 64) {
                const int q = q0 + lane;
                const int qc = q < nsplit ? q : nsplit - 1;
                const T v = P.pv[(long long)qc * P.n + i];
                const int ix = P.pi[(long long)qc * P.n + i];
                if (q < nsplit && (v < bv || (v == bv && ix < bi))) {
                    bv = v;
                    bi = ix;
                }
            }
        } else {
            bi = P.labels[i];
        }
    };
    auto cand_finish = [&](long long i, bool live, T bv, int bi) -> int {
        if (nsplit <= 1) return bi;
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) {
            const T ov = __shfl_xor(bv, msk, 64);
            const int oi = __shfl_xor(bi, msk, 64);
            if (ov < bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (bi == 0x7fffffff) bi = 0;  // all-NaN row: sklearn's argmin returns 0
        if (live && lane == 0) P.labels[i] = bi;
        return bi;
    };
    // (float rows: the difference in fp32, its square and the sum in fp64; double rows: all of it in fp64 -- scikit-learn's
    //  _euclidean_dense_dense works in the rows' own type)
    auto row_sum = [&](const T* x, const T* c) -> double {
        double s = 0.0;
        if (vec4) {
            struct alignas(16) V { T e[E]; };
            const V* x4 = reinterpret_cast<const V*>(x);
            const V* c4 = reinterpret_cast<const V*>(c);
            for (long long k = lane; k < m4; k += 64) {
                const V a = x4[k], b = c4[k];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const T d = a.e[e] - b.e[e];
                    s += (double)d * (double)d;
                }
            }
        } else {
            for (long long k = lane; k < P.m; k += 64) {
                const T d = x[k] - c[k];
                s += (double)d * (double)d;
            }
        }
        return s;
    };
    double tot = 0.0;
    const long long stride = (long long)gridDim.x * 4;
    for (long long i0 = (long long)blockIdx.x * 4 + wave; i0 < P.n; i0 += 2 * stride) {
        const long long i1 = i0 + stride;
        const bool has1 = i1 < P.n;
        const long long i1c = has1 ? i1 : i0;
        T v0, v1;
        int b0, b1;
        cand_load(i0, v0, b0);
        cand_load(i1c, v1, b1);
        const int lab0 = cand_finish(i0, true, v0, b0), lab1 = cand_finish(i1c, has1, v1, b1);
        const long long r0 = P.rows ? P.rows[i0] : i0, r1 = P.rows ? P.rows[i1c] : i1c;
        double s0 = row_sum(P.X + r0 * P.m, P.C + (long long)lab0 * P.m);
        double s1 = row_sum(P.X + r1 * P.m, P.C + (long long)lab1 * P.m);
        if (!has1) s1 = 0.0;
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) {
            s0 += __shfl_xor(s0, msk, 64);
            s1 += __shfl_xor(s1, msk, 64);
        }
        tot += s0;
        tot += s1;
    }
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// msm_mbk_run: end of one queued step.  Sums the inertia partials, then plays sklearn's _mini_batch_convergence
// (_kmeans.py:1963-2027, tol = 0 and verbose = 0 branch) in float64 on the device so that the host does not have to
// look at every step: st = {ewa, ewa_min, no_improvement, have_ewa, have_min, steps_done}.  Plain IEEE operations in
// the host's order (no contraction: __dmul_rn / __dadd_rn).  Executed by the LAST workgroup of mbk_update_kernel to
// finish (an arrival counter), not by a launch of its own: between dependent launches the GPU idles for ~10-15 us,
// which at 85 us of work per step is what a launch costs.
struct MbkConv {
    const double* partial;  // inertia partials of the step
    int nb;
    double* st;             // nullptr: no convergence bookkeeping (plain msm_mbk_step)
    int* stop;
    double* inertias;
    unsigned* done;         // arrival counter, zero between launches
    long long step_index;
    double batch_size, alpha;
    long long max_no_improvement;
};

__device__ __forceinline__ void mbk_converge(const MbkConv& cv, double* red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < cv.nb; i += KNT) s += cv.partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = KNT / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double* st = cv.st;
    const double inertia = red[0];
    cv.inertias[(long long)st[5]] = inertia;
    st[5] += 1.0;
    if (cv.step_index == 0) return;  // "ignore first iteration because it's inertia from initialization"
    const double bi = inertia / cv.batch_size;
    double ewa;
    if (st[3] == 0.0) {
        ewa = bi;
        st[3] = 1.0;
    } else {
        ewa = __dadd_rn(__dmul_rn(st[0], __dadd_rn(1.0, -cv.alpha)), __dmul_rn(bi, cv.alpha));
    }
    st[0] = ewa;
    if (st[4] == 0.0 || ewa < st[1]) {
        st[2] = 0.0;
        st[1] = ewa;
        st[4] = 1.0;
    } else {
        st[2] += 1.0;
    }
    if (cv.max_no_improvement >= 0 && st[2] >= (double)cv.max_no_improvement) *cv.stop = 1;
}

// One workgroup per centre: find the centre's members in the batch (ordered compaction by the whole workgroup:
// wave ballots + a 4-entry prefix; the first version let thread 0 walk the labels alone, 183 us per step at
// K = 1000, B = 1024), visit them in batch order.
// apply != 0: sklearn's streaming-mean update in fp32, in place on centers/counts, and the centre's new ||c||^2
//             (same lane partition and butterfly as kmeans_cnorm_kernel: bit-identical to a separate launch).
// sums/cnts (nullable): fp64 batch sums and counts for the multi-GPU all-reduce.
template <typename T>   // T: the rows' type = the type scikit-learn updates in (acc32 / w_old / alpha are "floating" there)
__global__ __launch_bounds__(KNT) void mbk_update_kernel(KmArgsT<T> P, T* __restrict__ centers,
                                                         T* __restrict__ counts, T* __restrict__ cnorm,
                                                         double* __restrict__ sums,
                                                         double* __restrict__ cnts, int apply, MbkConv cv)
{
    if (P.stop && *P.stop) return;
    extern __shared__ int members[];  // compacted member positions of one chunk
    __shared__ int wcnt[KNT / 64];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CH = 4096;
    const T w_old = counts[j];
    long long total = 0;
    for (long long f0 = 0; f0 < P.m; f0 += KNT) {
        const long long f = f0 + tid;
        T acc32 = (f < P.m) ? centers[(long long)j * P.m + f] * w_old : (T)0;
        double acc64 = 0.0;
        long long cnt = 0;
        for (long long b0 = 0; b0 < P.n; b0 += CH) {
            const long long be = std::min<long long>(P.n, b0 + CH);
            int nmem = 0;
            for (long long sb = b0; sb < be; sb += KNT) {
                const long long pos = sb + tid;
                const bool mine = pos < be && P.labels[pos] == j;
                const unsigned long long bal = __ballot(mine);
                __syncthreads();  // wcnt / members of the previous round are consumed
                if (lane == 0) wcnt[wave] = __popcll(bal);
                __syncthreads();
                int base = nmem;
                for (int w = 0; w < wave; ++w) base += wcnt[w];
                if (mine) members[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int)(pos - b0);
                nmem += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            }
            __syncthreads();
            cnt += nmem;
            if (f < P.m) {
                for (int k = 0; k < nmem; ++k) {
                    const long long b = b0 + members[k];
                    const long long r = P.rows ? P.rows[b] : b;
                    const T x = P.X[r * P.m + f];
                    acc32 += x;
                    acc64 += (double)x;
                }
            }
        }
        total = cnt;
        if (f < P.m) {
            if (sums) sums[(long long)j * P.m + f] = acc64;
            if (apply && cnt > 0) {
                const T w_new = w_old + (T)cnt;
                const T alpha = (T)1 / w_new;
                centers[(long long)j * P.m + f] = acc32 * alpha;
            }
        }
    }
    __syncthreads();  // the centre row is complete (workgroup-scope visibility)
    if (tid == 0) {
        if (cnts) cnts[j] = (double)total;
        if (apply && total > 0) counts[j] = w_old + (T)total;
    }
    if (apply && cnorm && total > 0 && wave == 0) {
        const volatile T* c = centers + (long long)j * P.m;
        T sq = 0;
        for (long long f = lane; f < P.m; f += 64) {
            const T v = c[f];
            sq += v * v;
        }
#pragma unroll
        for (int msk = 32; msk > 0; msk >>= 1) sq += __shfl_xor(sq, msk, 64);
        if (lane == 0) cnorm[j] = sq;
    }
    if (cv.st) {  // uniform: the last workgroup to arrive closes the step
        __shared__ int is_last;
        __shared__ double cred[KNT];
        __syncthreads();
        if (tid == 0) is_last = (atomicAdd(cv.done, 1u) == gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (is_last) {
            mbk_converge(cv, cred);
            if (tid == 0) *cv.done = 0u;
        }
    }
}

// ---------------------------------------------------------------------------
// Small-batch step (MiniBatchKMeans' inner loop: B ~ 1000 rows, m <= 32 features, K ~ 1000 centres).  A step is
// ~10^7 multiply-adds: the three general kernels above spent 26 + 7 + 38 us on it, all of it latency (MFMA tiles that
// are 70% padding, a 160-shuffle argmin, a 1000-label scan with 8 barriers in each of K workgroups) plus ~20 us of
// dependent-launch gaps.  Two launches instead:
//  * mbk_small_label_kernel: lane = row (64 rows per workgroup), the centres split over blockIdx.y and then over the 4
//    waves, the split's centres in LDS read as broadcast 16-byte fragments, v = ||c||^2 - 2 x.c in fp32 (the same
//    quantity the MFMA kernel minimises; sequential fma over the features).  The LAST workgroup of a row block to arrive
//    (an agent-scope counter) reduces the splits' candidates (lowest value, then lowest index), writes the labels and
//    the block's fp64 inertia partial (one wave per row, lanes over features, butterfly -- as kmeans_inertia_kernel).
//  * mbk_small_update_kernel: one WAVE per centre; the batch's labels (and row indices) are fetched with 16 + 16
//    independent loads per lane, members found by ballot, their rows read through v_readlane'd indices up to 8 loads in
//    flight, added in batch order (sklearn's order, _k_means_minibatch.pyx) by lane f < m.
// ---------------------------------------------------------------------------
constexpr int SBC = 128;  // centres per split (LDS slice)

struct SmallArgs {
    unsigned* arrive;          // [row blocks], zero between launches
    unsigned long long* cand;  // [rows] (value, index) candidates, all-ones between launches
    double* partial;           // [row blocks] inertia partials
    int ns, cper;              // centre splits, centres per split
};

// (value, index) -> one unsigned word whose order is (value ascending, index ascending); -0 counts as +0
__device__ __forceinline__ unsigned long long mbk_key(float v, int idx)
{
    unsigned u = __float_as_uint(v + 0.f);
    u ^= (u & 0x80000000u) ? 0xffffffffu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)idx;
}

template <int G>  // feature groups of 4: m <= 4 G
__global__ __launch_bounds__(KNT) void mbk_small_label_kernel(KmArgs P, SmallArgs S)
{
    if (P.stop && *P.stop) return;  // uniform
    constexpr int MP = 4 * G;
    __shared__ __attribute__((aligned(16))) float Cs[SBC * MP];
    __shared__ float cn[SBC];
    __shared__ float wv[4][64];
    __shared__ int wi[4][64];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long rb = blockIdx.x;
    const int sp = blockIdx.y;
    const int m = (int)P.m;
    const long long i = rb * 64 + lane;
    const long long ic = i < P.n ? i : P.n - 1;
    const long long r = P.rows ? P.rows[ic] : ic;
    float x[MP];  // unconditional loads at clamped columns, masked afterwards (a load under a select is waited for at once)
#pragma unroll
    for (int f = 0; f < MP; ++f) x[f] = P.X[r * P.m + (f < m ? f : m - 1)];
#pragma unroll
    for (int f = 0; f < MP; ++f)
        if (f >= m) x[f] = 0.f;
    const long long j0 = (long long)sp * S.cper;
    const int nc = (int)(P.K - j0 < S.cper ? P.K - j0 : S.cper);
    for (int e = tid; e < nc * MP; e += KNT) {
        const int c = e / MP, f = e - c * MP;
        Cs[e] = f < m ? P.C[(j0 + c) * P.m + f] : 0.f;
    }
    for (int c = tid; c < nc; c += KNT) cn[c] = P.cnorm[j0 + c];
    __syncthreads();
    const int per = (nc + 3) / 4;
    const int c0 = wave * per, c1 = (c0 + per < nc) ? c0 + per : nc;
    float best = INFINITY;
    int bidx = 0x7fffffff;
    for (int c = c0; c < c1; ++c) {
        const float4* cp = reinterpret_cast<const float4*>(Cs + c * MP);
        float dot = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 q = cp[g];
            dot = fmaf(x[4 * g + 0], q.x, dot);
            dot = fmaf(x[4 * g + 1], q.y, dot);
            dot = fmaf(x[4 * g + 2], q.z, dot);
            dot = fmaf(x[4 * g + 3], q.w, dot);
        }
        const float v = cn[c] - 2.f * dot;
        if (v < best) {  // ascending index, strict
            best = v;
            bidx = (int)(j0 + c);
        }
    }
    wv[wave][lane] = best;
    wi[wave][lane] = bidx;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = wv[w][lane];
            const int oi = wi[w][lane];
            if (ov < best || (ov == best && oi < bidx)) {
                best = ov;
                bidx = oi;
            }
        }
        // The splits' candidates meet in ONE 64-bit word per row: (order-preserving image of the value, index), reduced
        // by an agent-scope atomic min -- lowest value, then lowest index.  Candidates cross workgroups and XCDs (whose
        // L2s are not coherent) inside one launch; agent-scope atomics are performed at the memory side.  (Device-wide
        // fences instead -- an L2 write-back + invalidate per workgroup -- made this kernel 50 us; per-split candidate
        // arrays read back by the last workgroup with 2 x 32 dependent coherent loads per row, 30 us.)
        if (i < P.n) __hip_atomic_fetch_min(S.cand + i, mbk_key(best, bidx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // arrival: the workgroup's candidate atomics -> workgroup barrier -> agent-scope RELEASE fence (one lane) -> ticket;
    // the last arriver takes an agent-scope ACQUIRE fence before the barrier that lets its wave read the candidates
    // (round 4, VERDICT r3 #5: rounds 2-3 published the ticket behind a workgroup-scope fence)
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = (__hip_atomic_fetch_add(&S.arrive[rb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S.ns - 1)) ? 1 : 0;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last || wave != 0) return;
    // last workgroup of the row block, one wave: labels, and the block's inertia (lane = row, x still in registers;
    // fp32 difference, exact fp64 squares added in feature order, then a butterfly over the 64 rows)
    int lab = 0;
    double sq = 0.0;
    if (i < P.n) {
        const unsigned long long key = __hip_atomic_load(S.cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(S.cand + i, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next step
        lab = (int)(unsigned)(key & 0xffffffffull);
        if (lab == 0x7fffffff) lab = 0;  // all-NaN row: sklearn's argmin returns 0
        P.labels[i] = lab;
        const float* c = P.C + (long long)lab * P.m;
#pragma unroll
        for (int f = 0; f < MP; ++f)
            if (f < m) {
                const float d = x[f] - c[f];
                sq += (double)d * (double)d;
            }
    }
#pragma unroll
    for (int msk = 32; msk > 0; msk >>= 1) sq += __shfl_xor(sq, msk, 64);
    if (lane == 0) {
        S.partial[rb] = sq;
        __hip_atomic_store(&S.arrive[rb], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr int MSU_CAP = 1024;  // batch rows at most (a wave's member list in LDS)

template <typename T>
__global__ __launch_bounds__(KNT) void mbk_small_update_kernel(KmArgsT<T> P, T* __restrict__ centers,
                                                               T* __restrict__ counts, T* __restrict__ cnorm,
                                                               double* __restrict__ sums, double* __restrict__ cnts,
                                                               int apply, MbkConv cv)
{
    if (P.stop && *P.stop) return;
    __shared__ long long mrow[4][MSU_CAP];  // per wave: the centre's member rows in batch order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long j = (long long)blockIdx.x * 4 + wave;
    if (j < P.K) {  // uniform over the wave
        constexpr int NCH = 8;  // 64-feature blocks per round (lane = feature of each block)
        const T w_old = counts[j];
        T c_first[NCH];  // the first round's centre values: requested before the label scan, not after it
        // (all loads of this kernel are unconditional at clamped addresses and masked afterwards: a load under a select
        //  is waited for on the spot, which turns every batch of independent loads into a chain of round trips)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const long long f = (long long)c * 64 + lane;
            c_first[c] = centers[j * P.m + (f < P.m ? f : P.m - 1)];
        }
        // members: 16 + 16 independent loads per lane, then ballots; rows through v_readlane
        int cnt = 0;
        {
            int lab[16];
            long long rowv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long pos = (long long)r * 64 + lane;
                const long long pc = pos < P.n ? pos : P.n - 1;
                lab[r] = P.labels[pc];
                rowv[r] = P.rows ? P.rows[pc] : pc;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool in = (long long)r * 64 + lane < P.n;
                unsigned long long bal = __ballot(in && lab[r] == (int)j);
                const int rlo = (int)(rowv[r] & 0xffffffffLL), rhi = (int)(rowv[r] >> 32);
                while (bal) {  // uniform
                    const int k = __builtin_ctzll(bal);
                    bal &= bal - 1ull;
                    const long long row = ((long long)__builtin_amdgcn_readlane(rhi, k) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane(rlo, k);
                    if (lane == 0) mrow[wave][cnt] = row;
                    ++cnt;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T sqn = 0;  // ||c_new||^2, lane partition of kmeans_cnorm_kernel
        for (long long f0 = 0; f0 < P.m; f0 += NCH * 64) {
            bool fl[NCH];
            T c_old[NCH], acc32[NCH];
            double acc64[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const long long f = f0 + c * 64 + lane;
                fl[c] = f < P.m;
                c_old[c] = f0 == 0 ? c_first[c] : centers[j * P.m + (fl[c] ? f : P.m - 1)];
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (!fl[c]) c_old[c] = 0;
                acc32[c] = c_old[c] * w_old;
                acc64[c] = 0.0;
            }
            for (int q0 = 0; q0 < cnt; q0 += 4) {  // up to 4 x NCH row loads in flight
                T xv[4][NCH];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long long row = mrow[wave][q0 + t < cnt ? q0 + t : cnt - 1];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const long long f = f0 + c * 64 + lane;
                        xv[t][c] = P.X[row * P.m + (f < P.m ? f : P.m - 1)];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (q0 + t < cnt) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const T xq = fl[c] ? xv[t][c] : (T)0;
                            acc32[c] += xq;
                            acc64[c] += (double)xq;
                        }
                    }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const long long f = f0 + c * 64 + lane;
                T c_new = c_old[c];
                if (apply && cnt > 0) {
                    const T w_new = w_old + (T)cnt;
                    const T alpha = (T)1 / w_new;
                    c_new = acc32[c] * alpha;
                }
                if (fl[c]) {
                    if (sums) sums[j * P.m + f] = acc64[c];
                    if (apply && cnt > 0) centers[j * P.m + f] = c_new;
                    sqn += c_new * c_new;
                }
            }
        }
        if (lane == 0) {
            if (cnts) cnts[j] = (double)cnt;
            if (apply && cnt > 0) counts[j] = w_old + (T)cnt;
        }
        if (apply && cnorm && cnt > 0) {  // same lane partition and butterfly as kmeans_cnorm_kernel
#pragma unroll
            for (int msk = 32; msk > 0; msk >>= 1) sqn += __shfl_xor(sqn, msk, 64);
            if (lane == 0) cnorm[j] = sqn;
        }
    }
    if (cv.st) {  // uniform: the last workgroup to arrive closes the step
        __shared__ int is_last;
        __shared__ double cred[KNT];
        __syncthreads();
        if (tid == 0) is_last = (atomicAdd(cv.done, 1u) == gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        if (is_last) {
            mbk_converge(cv, cred);
            if (tid == 0) *cv.done = 0u;
        }
    }
}

// Mini-batch rows copied once into a compact [rows][m] buffer: the batch's rows are scattered over the whole data set (one
// page each for wide rows), and the label, inertia and update kernels of a step each paid those address translations again
// -- ~35 us per kernel at 1.25M x 512 whatever the arithmetic.  One wave per row, 16-byte lanes when the row allows.
template <typename T>
__global__ __launch_bounds__(KNT) void mbk_gather_kernel(const T* __restrict__ X, const msm_idx_t* __restrict__ rows,
                                                         long long nrows, long long m, T* __restrict__ out)
{
    constexpr int E = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const T* src = X + rows[i] * m;
    T* dst = out + i * m;
    if ((m % E) == 0 && ((((uintptr_t)X) | ((uintptr_t)out)) & 15) == 0) {
        for (long long f = lane * (long long)E; f < m; f += 64 * E) *reinterpret_cast<float4*>(dst + f) = *reinterpret_cast<const float4*>(src + f);
    } else {
        for (long long f = lane; f < m; f += 64) dst[f] = src[f];
    }
}

// finish a centre-split labelling: lowest (value, index) over the splits
template <typename T>
__global__ void kmeans_label_reduce_kernel(const T* __restrict__ pv, const int* __restrict__ pi, long long n,
                                           int nsplit, int32_t* __restrict__ labels, const int* __restrict__ stop)
{
    if (stop && *stop) return;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T bv = pv[i];
    int bi = pi[i];
    for (int s = 1; s < nsplit; ++s) {
        const T v = pv[(long long)s * n + i];
        const int ix = pi[(long long)s * n + i];
        if (v < bv || (v == bv && ix < bi)) {
            bv = v;
            bi = ix;
        }
    }
    labels[i] = (bi == 0x7fffffff) ? 0 : bi;
}

// ||c_j||^2 in the centres' own type, one wave per centre
template <typename T>
__global__ __launch_bounds__(KNT) void kmeans_cnorm_kernel(const T* __restrict__ C, long long K, long long m,
                                                           T* __restrict__ cnorm)
{
    const int lane = threadIdx.x & 63;
    const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= K) return;
    T s = 0;
    for (long long f = lane; f < m; f += 64) s += C[j * m + f] * C[j * m + f];
#pragma unroll
    for (int msk = 32; msk > 0; msk >>= 1) s += __shfl_xor(s, msk, 64);
    if (lane == 0) cnorm[j] = s;
}

// [inertia (double) | counts (K values of the rows' type)] gathered into one small buffer for a single D2H per step
template <typename T>
__global__ __launch_bounds__(KNT) void mbk_finish_kernel(const double* __restrict__ partial, int nb,
                                                         const T* __restrict__ counts, long long K,
                                                         double* __restrict__ out_inertia, T* __restrict__ out_counts)
{
    __shared__ double red[KNT];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += KNT) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = KNT / 2; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_inertia = red[0];
    for (long long j = threadIdx.x; j < K; j += KNT) out_counts[j] = counts[j];
}

// centres (+counts) <- (centres * w + batch sums) / (w + n) from all-reduced fp64 sums (multi-GPU)
template <typename T>
__global__ void mbk_apply_kernel(T* __restrict__ centers, T* __restrict__ counts,
                                 const double* __restrict__ packed, long long K, long long m, const int* __restrict__ stop = nullptr)
{
    if (stop && *stop) return;   // a queued run that has converged: the remaining steps are no-ops on every rank
    const long long j = blockIdx.x;
    const double n = packed[K * m + j];
    if (n <= 0.0) return;
    const T w_old = counts[j];
    const T w_new = (T)((double)w_old + n);
    for (long long f = threadIdx.x; f < m; f += blockDim.x)
        centers[j * m + f] = (T)(((double)centers[j * m + f] * (double)w_old + packed[j * m + f]) / (double)w_new);
    __syncthreads();
    if (threadIdx.x == 0) counts[j] = w_new;
}

// sharded run: the convergence bookkeeping of a step on the ALL-REDUCED batch inertia (cv.partial points at it, nb = 1)
__global__ __launch_bounds__(KNT) void mbk_conv_kernel(MbkConv cv)
{
    __shared__ double red[KNT];
    if (*cv.stop) return;
    mbk_converge(cv, red);
}

template <typename T>
__global__ void mbk_reassign_kernel(T* __restrict__ centers, T* __restrict__ counts,
                                    const T* __restrict__ X, long long m, const msm_idx_t* __restrict__ rows,
                                    const msm_idx_t* __restrict__ which, T new_count)
{
    const msm_idx_t r = rows[blockIdx.x], j = which[blockIdx.x];
    for (long long f = threadIdx.x; f < m; f += blockDim.x) centers[j * m + f] = X[r * m + f];
    if (threadIdx.x == 0) counts[j] = new_count;
}

template <typename T>
static int km_prepare(const T* centers, msm_idx_t K, msm_idx_t m, DevBuf& dC, T** dCent, T** dNorm)
{
    int rc = dC.reserve(((size_t)K * m + (size_t)K) * sizeof(T));
    if (rc) return rc;
    std::vector<T> cn((size_t)K);
    // (eight centres side by side: every centre's sum keeps its sequential order, the eight dependent chains overlap --
    //  one chain of 512 fp64 adds per centre was 0.6 ms of a K = 1000 x 512 call)
    msm_idx_t j = 0;
    for (; j + 8 <= K; j += 8) {
        double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const T* c0 = centers + j * m;
        for (msm_idx_t f = 0; f < m; ++f)
            for (int q = 0; q < 8; ++q) s8[q] += (double)c0[q * m + f] * (double)c0[q * m + f];
        for (int q = 0; q < 8; ++q) cn[(size_t)(j + q)] = (T)s8[q];
    }
    for (; j < K; ++j) {
        double s = 0.0;
        for (msm_idx_t f = 0; f < m; ++f) s += (double)centers[j * m + f] * (double)centers[j * m + f];
        cn[(size_t)j] = (T)s;
    }
    *dCent = dC.as<T>();
    *dNorm = *dCent + (size_t)K * m;
    // (2 MB of centres from pageable memory: 0.32 ms through hipMemcpyAsync's own staging, 0.1 ms through the library's pinned ring)
    {
        const int rcu = h2d_bulk(*dCent, centers, (size_t)K * m * sizeof(T));
        if (rcu) return rcu;
    }
    MSM_HIP_CHECK(hipMemcpyAsync(*dNorm, cn.data(), (size_t)K * sizeof(T), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // `cn` is a stack-frame vector
    return MSM_OK;
}

constexpr size_t KM4_LDS = (size_t)2 * (KR + KCT) * KP4 * sizeof(float);

// picks the 16-byte fast path when the row pitch and both base pointers allow it
static int km_launch_label(const KmArgs& P, dim3 grid)
{
    const bool v4 = P.m >= 4 && (P.m & 3) == 0 && (((uintptr_t)P.X | (uintptr_t)P.C) & 15) == 0 && P.m < (1 << 22);
    if (v4) {
        static bool attr_set = false;
        if (!attr_set) {
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_label_v4_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)KM4_LDS));
            MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_label_v4_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)KM4_LDS));
            attr_set = true;
        }
        if (P.rows)
            hipLaunchKernelGGL(kmeans_label_v4_kernel<true>, grid, dim3(KNT), KM4_LDS, stream(), P);
        else
            hipLaunchKernelGGL(kmeans_label_v4_kernel<false>, grid, dim3(KNT), KM4_LDS, stream(), P);
    } else {
        hipLaunchKernelGGL(kmeans_label_kernel, grid, dim3(KNT), 0, stream(), P);
    }
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// float64 rows: ONE kernel for every shape (kmeans_f64_dev.h); grid = (row blocks of 128, centre splits)
static int km_launch_label(const KmArgsT<double>& P, dim3 grid)
{
    static bool attr_set = false;
    if (!attr_set) {
        MSM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_label_f64_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)DK_LDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(kmeans_label_f64_kernel, grid, dim3(KNT), DK_LDS, stream(), P);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

// Large batches of wide rows (the final labelling pass of BASELINE configs[3]: 1.25M x 512 per rank, K = 1000): one workgroup
// per (row block, centre tile), the tiles of a row block side by side on one XCD, so that the rows are fetched ONCE
// (kmeans_label_v4_kernel, P.xcd_ns); a workgroup takes FOUR tiles, the later passes over its rows being L2 hits.  Returns the
// number of centre splits to use (0: not this case).
// MSM_LABEL_XCD=0: the A/B switch of the tests (read per call; labels are identical either way).
static int km_xcd_splits(const KmArgs& P)
{
    const char* xe = getenv("MSM_LABEL_XCD");
    // centre tiles per workgroup (0 = off: one workgroup walks all tiles).  Measured at 1.25M x 512, K = 1000
    // (profiles/r05_label_wide.txt: fetched + written bytes per pass | kernel | label + inertia call, last session):
    //   all 8 tiles  21.3 GB = 8.3x the rows |  9.9 ms | 10.6 ms
    //   4 tiles      11.2 GB = 4.4x          | 10.1 ms | 10.9 ms      <- default
    //   2 tiles       5.3 GB = 2.1x          | 10.6 ms | 11.4 ms
    //   1 tile        3.4 GB                 | 11.8 ms | 12.7 ms
    // The kernel is MFMA-bound: the re-reads of the all-tiles form come out of the Infinity Cache and cost no time, while
    // every split pays the pipeline fill and the argmin epilogue once more per row block and adds a candidate merge to
    // the inertia pass.  Four tiles halve the traffic for 2-3 % of the call; two tiles (the first default of the round)
    // cost 7 %.
    const int tiles_per = xe ? atoi(xe) : 4;
    if (tiles_per <= 0) return 0;
    const long long rowblocks = ceil_div(P.n, KR), ctiles = ceil_div(P.K, KCT);
    const bool v4ok = P.m >= 4 && (P.m & 3) == 0 && (((uintptr_t)P.X | (uintptr_t)P.C) & 15) == 0 && P.m < (1 << 22) && !P.rows;
    if (!v4ok || ctiles < 2 || ctiles > 16 || rowblocks < 512 || P.m < 64 || ceil_div(rowblocks, 8) * 8 * ctiles >= 0x7fffffffLL) return 0;
    const int ns = (int)ceil_div(ctiles, tiles_per);
    return ns > 1 ? ns : 0;
}
static int km_xcd_splits(const KmArgsT<double>&) { return 0; }   // (the fp64 kernel takes its splits over blockIdx.y)
// ... the launch: candidates of every split into pv / pi ([nsplit][n] each); the caller merges them (reduce or inertia kernel)
static int km_launch_label_xcd(KmArgs& P, int nsplit, float* pv, int* pi)
{
    P.jspan = ceil_div(ceil_div(P.K, KCT), nsplit) * KCT;
    P.xcd_ns = nsplit;
    P.pv = pv;
    P.pi = pi;
    const int rc = km_launch_label(P, dim3((unsigned)(ceil_div(ceil_div(P.n, KR), 8) * 8 * nsplit)));
    P.xcd_ns = 0;
    P.jspan = 0;
    return rc;
}
static int km_launch_label_xcd(KmArgsT<double>&, int, double*, int*) { return fail(MSM_ERR_STATE, "kmeans: no XCD-grouped launch for float64 rows"); }

// Centre splits of a SMALL batch (fewer row blocks than the chip has workgroup slots): the centre tiles are spread over
// blockIdx.y so that the launch fills the chip; returns the number of splits (1: none) and the centre span of one.
static int km_small_splits(long long n, long long K, long long* jspan)
{
    const long long rowblocks = ceil_div(n, KR), ctiles = ceil_div(K, KCT);   // (KR = DKR = 128, KCT = DKC = 128)
    int nsplit = 1;
    if (rowblocks < 256 && ctiles > 1) nsplit = (int)std::min<long long>(ctiles, std::max<long long>(1, 512 / rowblocks));
    const long long tiles_per = ceil_div(ctiles, nsplit);
    *jspan = tiles_per * KCT;
    return (int)ceil_div(ctiles, tiles_per);
}

template <typename T>
static int km_label_and_inertia(KmArgsT<T>& P, double* inertia)
{
    const unsigned grid = (unsigned)ceil_div(P.n, KR);
    int nsplit = km_xcd_splits(P);
    DevBuf &dPv = pool(PS_W), &dPi = pool(PS_S);
    if (nsplit > 1) {
        int rc0;
        if ((rc0 = dPv.reserve((size_t)nsplit * P.n * sizeof(T)))) return rc0;
        if ((rc0 = dPi.reserve((size_t)nsplit * P.n * sizeof(int)))) return rc0;
        if ((rc0 = km_launch_label_xcd(P, nsplit, dPv.as<T>(), dPi.as<int>()))) return rc0;
        if (!inertia)
            hipLaunchKernelGGL(kmeans_label_reduce_kernel<T>, dim3((unsigned)ceil_div(P.n, 256)), dim3(256), 0, stream(), P.pv, P.pi, P.n, nsplit,
                               P.labels, P.stop);
    } else {
        nsplit = 1;
        long long jspan = 0;
        if (sizeof(T) == 8) nsplit = km_small_splits(P.n, P.K, &jspan);   // float64 rows: small batches split their centres
        int rc0;
        if (nsplit > 1) {
            if ((rc0 = dPv.reserve((size_t)nsplit * P.n * sizeof(T)))) return rc0;
            if ((rc0 = dPi.reserve((size_t)nsplit * P.n * sizeof(int)))) return rc0;
            P.jspan = jspan;
            P.pv = dPv.as<T>();
            P.pi = dPi.as<int>();
            rc0 = km_launch_label(P, dim3(grid, (unsigned)nsplit));
            P.jspan = 0;
            if (rc0) return rc0;
            if (!inertia)
                hipLaunchKernelGGL(kmeans_label_reduce_kernel<T>, dim3((unsigned)ceil_div(P.n, 256)), dim3(256), 0, stream(), P.pv, P.pi, P.n,
                                   nsplit, P.labels, P.stop);
        } else if ((rc0 = km_launch_label(P, dim3(grid)))) {
            return rc0;
        }
    }
    MSM_HIP_CHECK(hipGetLastError());
    if (inertia) {
        const int nb = (int)std::min<long long>(ceil_div(P.n, 8), 2048);   // (two rows per wave and pass)
        DevBuf& dPart = pool(PS_PART);
        int rc = dPart.reserve((size_t)nb * sizeof(double));
        if (rc) return rc;
        hipLaunchKernelGGL(kmeans_inertia_kernel<T>, dim3(nb), dim3(KNT), 0, stream(), P, dPart.as<double>(), nsplit);   // (merges the splits' candidates)
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

// Device-resident MiniBatchKMeans state: centres, cumulative counts and ||c||^2 live in HBM for the
// whole fit; a step moves only the batch indices in and [inertia | counts] out.  `f64`: the rows' (and therefore the
// centres', counts' and norms') type -- scikit-learn works in the type of X (float32 stays float32, everything else is
// float64); the typed pointers below are `float*` or `double*` accordingly (cen<T>() ...).
struct msm_mbk {
    long long K = 0, m = 0;
    int f64 = 0;
    void* centers = nullptr;
    void* counts = nullptr;
    void* cnorm = nullptr;
    double* packed = nullptr;  // [K*m | K | 1] batch sums, counts, inertia (fp64)
    char* outbuf = nullptr;    // [8 + sizeof(T) K]
    DevBuf labels, idx, xb, pv, pi, part, rows, which;
    DevBuf arrive;             // small-batch step: per-row-block arrival counters (zero between launches)
    size_t arrive_zeroed = 0;
    // msm_mbk_run: [6 doubles of convergence state | S inertias] and the stop flag on the device; pinned host mirror
    DevBuf runbuf;
    int* stop = nullptr;
    char* pinned = nullptr;
    size_t pinned_bytes = 0;
    const char* run_out = nullptr;   // msm_mbk_run_begin .. _end: where the run in flight leaves its results (in `pinned`)
    size_t run_st_bytes = 0;
    size_t esz() const { return f64 ? sizeof(double) : sizeof(float); }
    template <typename T> T* cen() { return static_cast<T*>(centers); }
    template <typename T> T* cnt() { return static_cast<T*>(counts); }
    template <typename T> T* nrm() { return static_cast<T*>(cnorm); }
};

namespace {

// small-batch step kernels: rows of at most 32 features, at most 1024 row blocks of 64 (MSM_MBK_SMALL=0: general kernels)
bool mbk_small_off()
{
    static const bool off = getenv("MSM_MBK_SMALL") && atoi(getenv("MSM_MBK_SMALL")) == 0;
    return off;
}
bool mbk_small_ok(const msm_mbk* h, long long n) { return !mbk_small_off() && !h->f64 && h->m <= 32 && n <= 65536; }  // label kernel (fp32 only)
bool mbk_small_update_ok(long long n)                                                                      // update kernel
{
    return !mbk_small_off() && n <= MSU_CAP;
}
bool mbk_label64_ok(const msm_mbk* h, long long n)                                                         // 64 x 64 label tiles (fp32 only)
{
    return !mbk_small_off() && !h->f64 && n <= 4096 && h->m > 32;
}

// centre update of a step: one wave per centre for small batches, else one workgroup per centre
template <typename T>
void mbk_launch_update(msm_mbk* h, const KmArgsT<T>& P, double* sums, double* cnts, int apply, const MbkConv& cv)
{
    if (mbk_small_update_ok(P.n))
        hipLaunchKernelGGL(mbk_small_update_kernel<T>, dim3((unsigned)ceil_div(h->K, 4)), dim3(KNT), 0, stream(), P, h->cen<T>(),
                           h->cnt<T>(), h->nrm<T>(), sums, cnts, apply, cv);
    else
        hipLaunchKernelGGL(mbk_update_kernel<T>, dim3((unsigned)h->K), dim3(KNT), 4096 * sizeof(int), stream(), P, h->cen<T>(),
                           h->cnt<T>(), h->nrm<T>(), sums, cnts, apply, cv);
}

// float32 rows: the three label paths of rounds 1-5 (small VALU kernel, 64 x 64 tiles, 128 x 128 tiles)
int mbk_label_f32(msm_mbk* h, KmArgs& P, long long n, int32_t* labels_d, double* inertia_dev_partial, int* nb_out)
{
    if (inertia_dev_partial && mbk_small_ok(h, n)) {  // MiniBatchKMeans' inner loop: the two-launch small-batch step
        const int RB = (int)ceil_div(n, 64);
        int ns = (int)std::max<long long>(1, std::min<long long>(ceil_div(512, RB), ceil_div(h->K, 16)));
        int cper = (int)ceil_div(h->K, ns);
        if (cper > SBC) cper = SBC;
        ns = (int)ceil_div(h->K, cper);
        int rc;
        if (h->arrive_zeroed == 0) {  // [1024 arrival counters = 0 | 65536 candidate words = all ones], once
            if ((rc = h->arrive.reserve((size_t)1024 * sizeof(unsigned) + (size_t)65536 * sizeof(unsigned long long)))) return rc;
            MSM_HIP_CHECK(hipMemsetAsync(h->arrive.p, 0, 1024 * sizeof(unsigned), stream()));
            MSM_HIP_CHECK(hipMemsetAsync(static_cast<char*>(h->arrive.p) + 1024 * sizeof(unsigned), 0xff, (size_t)65536 * sizeof(unsigned long long), stream()));
            h->arrive_zeroed = 1;
        }
        SmallArgs S;
        S.arrive = h->arrive.as<unsigned>();
        S.cand = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->arrive.p) + 1024 * sizeof(unsigned));
        S.partial = inertia_dev_partial;
        S.ns = ns;
        S.cper = cper;
        const dim3 grid((unsigned)RB, (unsigned)ns);
        switch ((int)ceil_div(h->m, 4)) {
#define MSM_SL(G_) case G_: hipLaunchKernelGGL(mbk_small_label_kernel<G_>, grid, dim3(KNT), 0, stream(), P, S); break;
            MSM_SL(1) MSM_SL(2) MSM_SL(3) MSM_SL(4) MSM_SL(5) MSM_SL(6) MSM_SL(7) MSM_SL(8)
#undef MSM_SL
        }
        MSM_HIP_CHECK(hipGetLastError());
        *nb_out = RB;
        return MSM_OK;
    }
    if (mbk_label64_ok(h, n)) {  // small batch of wide rows: 64 x 64 tiles fill the chip
        const long long rb = ceil_div(n, KS64), ct = ceil_div(h->K, KS64);
        int ns = (int)std::min<long long>(ct, std::max<long long>(1, ceil_div(512, rb)));
        const long long tiles_per = ceil_div(ct, ns);
        ns = (int)ceil_div(ct, tiles_per);
        int rc;
        if ((rc = h->pv.reserve((size_t)ns * n * sizeof(float)))) return rc;
        if ((rc = h->pi.reserve((size_t)ns * n * sizeof(int)))) return rc;
        P.jspan = tiles_per * KS64;
        P.pv = h->pv.as<float>();
        P.pi = h->pi.as<int>();
        static bool attr64 = false;
        if (!attr64) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_label64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)KM64_LDS);
            attr64 = true;
        }
        hipLaunchKernelGGL(kmeans_label64_kernel, dim3((unsigned)rb, (unsigned)ns), dim3(KNT), KM64_LDS, stream(), P);
        MSM_HIP_CHECK(hipGetLastError());
        if (!inertia_dev_partial) {
            hipLaunchKernelGGL(kmeans_label_reduce_kernel<float>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(), P.pv, P.pi,
                               n, ns, labels_d, P.stop);
        } else {
            const int nb = (int)std::min<long long>(ceil_div(n, 4), 1024);
            P.jspan = 0;
            hipLaunchKernelGGL(kmeans_inertia_kernel<float>, dim3(nb), dim3(KNT), 0, stream(), P, inertia_dev_partial, ns);
            *nb_out = nb;
        }
        MSM_HIP_CHECK(hipGetLastError());
        return MSM_OK;
    }
    return 1;   // not a small-batch shape: the general path
}
int mbk_label_f32(msm_mbk*, KmArgsT<double>&, long long, int32_t*, double*, int*) { return 1; }

template <typename T>
int mbk_label(msm_mbk* h, const T* Xd, const msm_idx_t* rows_d, long long n, int32_t* labels_d, double* inertia_dev_partial,
              int* nb_out, const int* stop = nullptr)
{
    KmArgsT<T> P;
    memset(&P, 0, sizeof(P));
    P.X = Xd;
    P.rows = rows_d;
    P.n = n;
    P.m = h->m;
    P.K = h->K;
    P.C = h->cen<T>();
    P.cnorm = h->nrm<T>();
    P.labels = labels_d;
    P.stop = stop;
    {
        const int rs = mbk_label_f32(h, P, n, labels_d, inertia_dev_partial, nb_out);
        if (rs <= 0) return rs;
    }
    long long jspan = 0;
    int nsplit = km_small_splits(n, h->K, &jspan);   // small batch: split the centres over workgroups to fill the chip
    const long long rowblocks = ceil_div(n, KR);
    const int xs = nsplit == 1 ? km_xcd_splits(P) : 0;   // large batches of wide rows: see km_xcd_splits
    int rc;
    if (xs > 1) {
        nsplit = xs;
        if ((rc = h->pv.reserve((size_t)nsplit * n * sizeof(T)))) return rc;
        if ((rc = h->pi.reserve((size_t)nsplit * n * sizeof(int)))) return rc;
        if ((rc = km_launch_label_xcd(P, nsplit, h->pv.as<T>(), h->pi.as<int>()))) return rc;
        if (!inertia_dev_partial)
            hipLaunchKernelGGL(kmeans_label_reduce_kernel<T>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(),
                               P.pv, P.pi, n, nsplit, labels_d, P.stop);
    } else if (nsplit > 1) {
        if ((rc = h->pv.reserve((size_t)nsplit * n * sizeof(T)))) return rc;
        if ((rc = h->pi.reserve((size_t)nsplit * n * sizeof(int)))) return rc;
        P.jspan = jspan;
        P.pv = h->pv.as<T>();
        P.pi = h->pi.as<int>();
        if ((rc = km_launch_label(P, dim3((unsigned)rowblocks, (unsigned)nsplit)))) return rc;
        if (!inertia_dev_partial)  // else the inertia kernel below picks the labels from the candidates itself
            hipLaunchKernelGGL(kmeans_label_reduce_kernel<T>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream(),
                               P.pv, P.pi, n, nsplit, labels_d, P.stop);
    } else {
        if ((rc = km_launch_label(P, dim3((unsigned)rowblocks)))) return rc;
    }
    MSM_HIP_CHECK(hipGetLastError());
    if (inertia_dev_partial) {
        const int nb = (int)std::min<long long>(ceil_div(n, 4), 1024);
        P.jspan = 0;
        hipLaunchKernelGGL(kmeans_inertia_kernel<T>, dim3(nb), dim3(KNT), 0, stream(), P, inertia_dev_partial, nsplit);
        MSM_HIP_CHECK(hipGetLastError());
        *nb_out = nb;
    }
    return MSM_OK;
}

// stage the batch: device X -> row indices on device; host X -> gathered rows on device
template <typename T>
int mbk_stage_batch(msm_mbk* h, const T* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t B, int on_device,
                    const T** Xd, const msm_idx_t** rows_d)
{
    int rc;
    for (msm_idx_t b = 0; b < B; ++b)
        if (batch_idx[b] < 0 || batch_idx[b] >= n) return fail(MSM_ERR_INVALID, "mbk: batch index out of range");
    if (on_device) {
        if ((rc = h->idx.reserve((size_t)B * sizeof(msm_idx_t)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(h->idx.p, batch_idx, (size_t)B * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));  // batch_idx is caller-owned pageable memory
        if ((rc = h->xb.reserve((size_t)B * h->m * sizeof(T)))) return rc;
        hipLaunchKernelGGL(mbk_gather_kernel<T>, dim3((unsigned)ceil_div(B, 4)), dim3(KNT), 0, stream(), X, h->idx.as<msm_idx_t>(),
                           (long long)B, (long long)h->m, h->xb.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
        *Xd = h->xb.as<T>();
        *rows_d = nullptr;
    } else {
        std::vector<T> xb((size_t)B * h->m);
        for (msm_idx_t b = 0; b < B; ++b)
            memcpy(xb.data() + (size_t)b * h->m, X + batch_idx[b] * h->m, (size_t)h->m * sizeof(T));
        if ((rc = h->xb.reserve(xb.size() * sizeof(T)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(h->xb.p, xb.data(), xb.size() * sizeof(T), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        *Xd = h->xb.as<T>();
        *rows_d = nullptr;
    }
    return MSM_OK;
}

template <typename T>
void mbk_launch_cnorm(msm_mbk* h)
{
    hipLaunchKernelGGL(kmeans_cnorm_kernel<T>, dim3((unsigned)ceil_div(h->K, 4)), dim3(KNT), 0, stream(), h->cen<T>(), h->K, h->m, h->nrm<T>());
}
template <typename T>
void mbk_launch_apply(msm_mbk* h, const int* stop)
{
    hipLaunchKernelGGL(mbk_apply_kernel<T>, dim3((unsigned)h->K), dim3(256), 0, stream(), h->cen<T>(), h->cnt<T>(), h->packed, h->K, h->m, stop);
    mbk_launch_cnorm<T>(h);
}
template <typename T>
void mbk_launch_finish(msm_mbk* h, int nb, double* d_inertia)
{
    hipLaunchKernelGGL(mbk_finish_kernel<T>, dim3(1), dim3(KNT), 0, stream(), h->part.as<double>(), nb, h->cnt<T>(), h->K,
                       d_inertia, reinterpret_cast<T*>(h->outbuf + 8));
}

template <typename T>
int mbk_step_t(msm_mbk* h, const T* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t B,
               double* batch_inertia, T* counts_out, int apply_update, int on_device)
{
    int rc;
    const T* Xd;
    const msm_idx_t* rows_d;
    if ((rc = mbk_stage_batch<T>(h, X, n, batch_idx, B, on_device, &Xd, &rows_d))) return rc;
    if ((rc = h->labels.reserve((size_t)B * sizeof(int32_t)))) return rc;
    if ((rc = h->part.reserve(1024 * sizeof(double)))) return rc;
    int nb = 0;
    if ((rc = mbk_label<T>(h, Xd, rows_d, B, h->labels.as<int32_t>(), h->part.as<double>(), &nb))) return rc;
    KmArgsT<T> P;
    memset(&P, 0, sizeof(P));
    P.X = Xd;
    P.rows = rows_d;
    P.n = B;
    P.m = h->m;
    P.K = h->K;
    P.labels = h->labels.as<int32_t>();
    mbk_launch_update<T>(h, P, apply_update ? (double*)nullptr : h->packed,
                         apply_update ? (double*)nullptr : h->packed + (size_t)h->K * h->m, apply_update, MbkConv{});
    MSM_HIP_CHECK(hipGetLastError());
    double* d_inertia = apply_update ? reinterpret_cast<double*>(h->outbuf) : h->packed + (size_t)h->K * h->m + h->K;
    mbk_launch_finish<T>(h, nb, d_inertia);
    MSM_HIP_CHECK(hipGetLastError());
    std::vector<char> hb(8 + (size_t)h->K * sizeof(T));
    if (apply_update) {
        MSM_HIP_CHECK(hipMemcpyAsync(hb.data(), h->outbuf, hb.size(), hipMemcpyDeviceToHost, stream()));
    } else {
        MSM_HIP_CHECK(hipMemcpyAsync(hb.data(), d_inertia, 8, hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(hb.data() + 8, h->outbuf + 8, (size_t)h->K * sizeof(T), hipMemcpyDeviceToHost, stream()));
    }
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (batch_inertia) memcpy(batch_inertia, hb.data(), 8);
    if (counts_out) memcpy(counts_out, hb.data() + 8, (size_t)h->K * sizeof(T));
    return MSM_OK;
}

template <typename T>
int mbk_run_begin_t(msm_mbk* h, const T* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t S, msm_idx_t B,
                    msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement, const double* state6)
{
    for (msm_idx_t b = 0; b < S * B; ++b)
        if (batch_idx[b] < 0 || batch_idx[b] >= n) return fail(MSM_ERR_INVALID, "mbk: batch index out of range");
    int rc;
    const size_t idx_bytes = (size_t)S * B * sizeof(msm_idx_t);
    const size_t st_bytes = (6 + (size_t)S) * sizeof(double);
    const size_t out_bytes = st_bytes + sizeof(int) + 4 + (size_t)h->K * sizeof(T);
    const size_t need = idx_bytes + 64 + out_bytes;  // [indices | initial state | results]
    if (h->pinned_bytes < need) {
        if (h->pinned) (void)hipHostFree(h->pinned);
        h->pinned = nullptr;
        h->pinned_bytes = 0;
        MSM_HIP_CHECK(hipHostMalloc((void**)&h->pinned, need, hipHostMallocDefault));
        h->pinned_bytes = need;
    }
    if (!h->stop) MSM_HIP_CHECK(hipMalloc((void**)&h->stop, 2 * sizeof(int)));  // {stop flag, arrival counter}
    if ((rc = h->idx.reserve(idx_bytes))) return rc;
    if ((rc = h->runbuf.reserve(st_bytes))) return rc;
    if ((rc = h->labels.reserve((size_t)B * sizeof(int32_t)))) return rc;
    if ((rc = h->part.reserve(1024 * sizeof(double)))) return rc;
    double* st = h->runbuf.as<double>();
    // in: indices of all S batches, convergence state (steps_done restarts at 0)
    memcpy(h->pinned, batch_idx, idx_bytes);
    MSM_HIP_CHECK(hipMemcpyAsync(h->idx.p, h->pinned, idx_bytes, hipMemcpyHostToDevice, stream()));
    double* st0 = reinterpret_cast<double*>(h->pinned + idx_bytes);
    for (int i = 0; i < 5; ++i) st0[i] = state6[i];
    st0[5] = 0.0;
    MSM_HIP_CHECK(hipMemcpyAsync(st, st0, 6 * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->stop, 0, 2 * sizeof(int), stream()));
    // all S batches into one compact buffer (one launch), so that a step's kernels read contiguous rows
    if ((rc = h->xb.reserve((size_t)S * B * h->m * sizeof(T)))) return rc;
    hipLaunchKernelGGL(mbk_gather_kernel<T>, dim3((unsigned)ceil_div(S * B, 4)), dim3(KNT), 0, stream(), X, h->idx.as<msm_idx_t>(),
                       (long long)(S * B), (long long)h->m, h->xb.as<T>());
    MSM_HIP_CHECK(hipGetLastError());
    for (msm_idx_t s = 0; s < S; ++s) {
        const T* Xs_ = h->xb.as<T>() + (size_t)s * B * h->m;
        const msm_idx_t* rows_d = nullptr;
        int nb = 0;
        if ((rc = mbk_label<T>(h, Xs_, rows_d, B, h->labels.as<int32_t>(), h->part.as<double>(), &nb, h->stop))) return rc;
        KmArgsT<T> P;
        memset(&P, 0, sizeof(P));
        P.X = Xs_;
        P.rows = rows_d;
        P.n = B;
        P.m = h->m;
        P.K = h->K;
        P.labels = h->labels.as<int32_t>();
        P.stop = h->stop;
        MbkConv cv;
        cv.partial = h->part.as<double>();
        cv.nb = nb;
        cv.st = st;
        cv.stop = h->stop;
        cv.inertias = st + 6;
        cv.done = reinterpret_cast<unsigned*>(h->stop + 1);
        cv.step_index = (long long)(first_step + s);
        cv.batch_size = (double)B;
        cv.alpha = alpha;
        cv.max_no_improvement = (long long)max_no_improvement;
        mbk_launch_update<T>(h, P, nullptr, nullptr, 1, cv);
        MSM_HIP_CHECK(hipGetLastError());
    }
    // out: [state | inertias | stop | counts] through the pinned mirror, one synchronisation for the whole run
    char* o = h->pinned + idx_bytes + 64;
    MSM_HIP_CHECK(hipMemcpyAsync(o, st, st_bytes, hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(o + st_bytes, h->stop, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(o + st_bytes + 8, h->counts, (size_t)h->K * sizeof(T), hipMemcpyDeviceToHost, stream()));
    h->run_out = o;
    h->run_st_bytes = st_bytes;
    return MSM_OK;
}

template <typename T>
int mbk_run_sharded_t(msm_mbk* h, const T* X, msm_idx_t n_local, const msm_idx_t* local_idx, const msm_idx_t* offsets,
                      msm_idx_t S, msm_idx_t B, msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement,
                      double* state6, msm_idx_t* steps_done, int* converged, double* inertias, T* counts_out)
{
    // Everything that can fail on ONE rank -- argument checks, allocations -- happens before the first collective, and the
    // ranks agree on the outcome with one all-reduced flag: a rank that returned early on its own would leave the others
    // blocked inside the first step's all-reduce.
    msm_idx_t total = 0;
    size_t idx_bytes = 0, st_bytes = 0;
    auto prepare = [&]() -> int {
        if (n_local < 0 || B < 1 || S < 1 || S > 4096) return fail(MSM_ERR_INVALID, "msm_mbk_run_sharded: bad shape");
        total = offsets[S];
        if (offsets[0] != 0 || total < 0 || (total > 0 && (!local_idx || !X))) return fail(MSM_ERR_INVALID, "msm_mbk_run_sharded: bad offsets");
        for (msm_idx_t s = 0; s < S; ++s)
            if (offsets[s + 1] < offsets[s]) return fail(MSM_ERR_INVALID, "msm_mbk_run_sharded: offsets must not decrease");
        for (msm_idx_t b = 0; b < total; ++b)
            if (local_idx[b] < 0 || local_idx[b] >= n_local) return fail(MSM_ERR_INVALID, "mbk: batch index out of range");
        int rc;
        idx_bytes = (size_t)std::max<msm_idx_t>(total, 1) * sizeof(msm_idx_t);
        st_bytes = (6 + (size_t)S) * sizeof(double);
        const size_t out_bytes = st_bytes + sizeof(int) + 4 + (size_t)h->K * sizeof(T);
        const size_t need = idx_bytes + 64 + out_bytes;
        if (h->pinned_bytes < need) {
            if (h->pinned) (void)hipHostFree(h->pinned);
            h->pinned = nullptr;
            h->pinned_bytes = 0;
            MSM_HIP_CHECK(hipHostMalloc((void**)&h->pinned, need, hipHostMallocDefault));
            h->pinned_bytes = need;
        }
        if (!h->stop) MSM_HIP_CHECK(hipMalloc((void**)&h->stop, 2 * sizeof(int)));
        if ((rc = h->idx.reserve(idx_bytes))) return rc;
        if ((rc = h->runbuf.reserve(st_bytes))) return rc;
        if ((rc = h->labels.reserve((size_t)B * sizeof(int32_t)))) return rc;
        if ((rc = h->part.reserve(1024 * sizeof(double)))) return rc;
        if (total > 0 && (rc = h->xb.reserve((size_t)total * h->m * sizeof(T)))) return rc;
        return MSM_OK;
    };
    int rc = prepare();
    if (comm_active()) {
        const double mine = rc ? 1.0 : 0.0;
        double failed = 0.0;
        MSM_HIP_CHECK(hipMemcpyAsync(h->packed, &mine, sizeof(double), hipMemcpyHostToDevice, stream()));
        const int rca = comm_allreduce_f64(h->packed, 1);
        if (rca) return rca;
        MSM_HIP_CHECK(hipMemcpyAsync(&failed, h->packed, sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        if (!rc && failed > 0.0) return fail(MSM_ERR_STATE, "msm_mbk_run_sharded: %d other rank(s) rejected their arguments or ran out of memory", (int)failed);
    }
    if (rc) return rc;
    double* st = h->runbuf.as<double>();
    if (total > 0) {
        memcpy(h->pinned, local_idx, (size_t)total * sizeof(msm_idx_t));
        MSM_HIP_CHECK(hipMemcpyAsync(h->idx.p, h->pinned, (size_t)total * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
    }
    double* st0 = reinterpret_cast<double*>(h->pinned + idx_bytes);
    for (int i = 0; i < 5; ++i) st0[i] = state6[i];
    st0[5] = 0.0;
    MSM_HIP_CHECK(hipMemcpyAsync(st, st0, 6 * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemsetAsync(h->stop, 0, 2 * sizeof(int), stream()));
    if (total > 0) {
        hipLaunchKernelGGL(mbk_gather_kernel<T>, dim3((unsigned)ceil_div(total, 4)), dim3(KNT), 0, stream(), X, h->idx.as<msm_idx_t>(),
                           (long long)total, (long long)h->m, h->xb.as<T>());
        MSM_HIP_CHECK(hipGetLastError());
    }
    const size_t psz = (size_t)msm_mbk_packed_size(h);
    double* d_inertia = h->packed + (size_t)h->K * h->m + h->K;
    for (msm_idx_t s = 0; s < S; ++s) {
        const msm_idx_t Bs = offsets[s + 1] - offsets[s];
        MSM_HIP_CHECK(hipMemsetAsync(h->packed, 0, psz * sizeof(double), stream()));
        if (Bs > 0) {
            const T* Xs_ = h->xb.as<T>() + (size_t)offsets[s] * h->m;
            int nb = 0;
            if ((rc = mbk_label<T>(h, Xs_, nullptr, Bs, h->labels.as<int32_t>(), h->part.as<double>(), &nb, h->stop))) return rc;
            KmArgsT<T> P;
            memset(&P, 0, sizeof(P));
            P.X = Xs_;
            P.n = Bs;
            P.m = h->m;
            P.K = h->K;
            P.labels = h->labels.as<int32_t>();
            P.stop = h->stop;
            mbk_launch_update<T>(h, P, h->packed, h->packed + (size_t)h->K * h->m, 0, MbkConv{});
            MSM_HIP_CHECK(hipGetLastError());
            mbk_launch_finish<T>(h, nb, d_inertia);
            MSM_HIP_CHECK(hipGetLastError());
        }
        if ((rc = comm_allreduce_f64(h->packed, psz))) return rc;
        mbk_launch_apply<T>(h, h->stop);
        MbkConv cv;
        cv.partial = d_inertia;
        cv.nb = 1;
        cv.st = st;
        cv.stop = h->stop;
        cv.inertias = st + 6;
        cv.done = reinterpret_cast<unsigned*>(h->stop + 1);
        cv.step_index = (long long)(first_step + s);
        cv.batch_size = (double)B;
        cv.alpha = alpha;
        cv.max_no_improvement = (long long)max_no_improvement;
        hipLaunchKernelGGL(mbk_conv_kernel, dim3(1), dim3(KNT), 0, stream(), cv);
        MSM_HIP_CHECK(hipGetLastError());
    }
    char* o = h->pinned + idx_bytes + 64;
    MSM_HIP_CHECK(hipMemcpyAsync(o, st, st_bytes, hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(o + st_bytes, h->stop, sizeof(int), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(o + st_bytes + 8, h->counts, (size_t)h->K * sizeof(T), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    const double* so = reinterpret_cast<const double*>(o);
    for (int i = 0; i < 6; ++i) state6[i] = so[i];
    *steps_done = (msm_idx_t)so[5];
    memcpy(inertias, so + 6, (size_t)(*steps_done) * sizeof(double));
    *converged = *reinterpret_cast<const int*>(o + st_bytes);
    if (counts_out) memcpy(counts_out, o + st_bytes + 8, (size_t)h->K * sizeof(T));
    return MSM_OK;
}

template <typename T>
int mbk_reassign_t(msm_mbk* h, const T* X, msm_idx_t n, const msm_idx_t* rows, const msm_idx_t* which, msm_idx_t n_reassign,
                   double new_count, int on_device)
{
    int rc;
    const T* Xd = X;
    std::vector<msm_idx_t> r2(rows, rows + n_reassign);
    if (!on_device) {  // ship only the chosen rows
        std::vector<T> xb((size_t)n_reassign * h->m);
        for (msm_idx_t i = 0; i < n_reassign; ++i) {
            memcpy(xb.data() + (size_t)i * h->m, X + rows[i] * h->m, (size_t)h->m * sizeof(T));
            r2[(size_t)i] = i;
        }
        if ((rc = h->xb.reserve(xb.size() * sizeof(T)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(h->xb.p, xb.data(), xb.size() * sizeof(T), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        Xd = h->xb.as<T>();
    }
    if ((rc = h->rows.reserve((size_t)n_reassign * sizeof(msm_idx_t)))) return rc;
    if ((rc = h->which.reserve((size_t)n_reassign * sizeof(msm_idx_t)))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(h->rows.p, r2.data(), (size_t)n_reassign * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(h->which.p, which, (size_t)n_reassign * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
    hipLaunchKernelGGL(mbk_reassign_kernel<T>, dim3((unsigned)n_reassign), dim3(256), 0, stream(), h->cen<T>(), h->cnt<T>(), Xd,
                       h->m, h->rows.as<msm_idx_t>(), h->which.as<msm_idx_t>(), (T)new_count);
    MSM_HIP_CHECK(hipGetLastError());
    mbk_launch_cnorm<T>(h);
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

template <typename T>
int mbk_label_t(msm_mbk* h, const T* X, msm_idx_t n, int32_t* labels, double* inertia, int on_device)
{
    int rc;
    const T* Xd = X;
    int32_t* lab_d = labels;
    DevBuf &dX = pool(PS_X), &dL = pool(PS_LAB);
    if (!on_device) {
        if ((rc = dX.reserve((size_t)n * h->m * sizeof(T)))) return rc;
        if ((rc = dL.reserve((size_t)n * sizeof(int32_t)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * h->m * sizeof(T)))) return rc;
        Xd = dX.as<T>();
        lab_d = dL.as<int32_t>();
    }
    if ((rc = h->part.reserve(1024 * sizeof(double)))) return rc;
    int nb = 0;
    if ((rc = mbk_label<T>(h, Xd, nullptr, n, lab_d, inertia ? h->part.as<double>() : nullptr, &nb))) return rc;
    if (!on_device) MSM_HIP_CHECK(hipMemcpyAsync(labels, lab_d, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
    if (inertia) {
        std::vector<double> hp((size_t)nb);
        MSM_HIP_CHECK(hipMemcpyAsync(hp.data(), h->part.p, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        double sacc = 0.0;
        for (int i = 0; i < nb; ++i) sacc += hp[(size_t)i];
        *inertia = sacc;
    } else {
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    }
    return MSM_OK;
}

template <typename T>
int kmeans_label_t(const T* X, msm_idx_t n, msm_idx_t m, const T* centers, msm_idx_t K, int32_t* labels, double* inertia, int on_device)
{
    if (!X || !centers || !labels) return fail(MSM_ERR_INVALID, "kmeans_label: null pointer");
    if (n < 0 || m < 1 || K < 1) return fail(MSM_ERR_INVALID, "kmeans_label: bad shape");
    if (inertia) *inertia = 0.0;
    if (n == 0) return MSM_OK;
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf &dC = pool(PS_Y), &dX = pool(PS_X), &dL = pool(PS_LAB);
    T *dCent, *dNorm;
    int rc = km_prepare<T>(centers, K, m, dC, &dCent, &dNorm);
    if (rc) return rc;
    KmArgsT<T> P;
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
        if ((rc = dX.reserve((size_t)n * m * sizeof(T)))) return rc;
        if ((rc = dL.reserve((size_t)n * sizeof(int32_t)))) return rc;
        if ((rc = h2d_bulk(dX.p, X, (size_t)n * m * sizeof(T)))) return rc;
        P.X = dX.as<T>();
        P.labels = dL.as<int32_t>();
    }
    if ((rc = km_label_and_inertia<T>(P, inertia))) return rc;
    if (!on_device)
        MSM_HIP_CHECK(hipMemcpyAsync(labels, P.labels, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

template <typename T>
int mbk_step_stateless_t(const T* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx, msm_idx_t B, T* centers, T* counts,
                         msm_idx_t K, double* batch_inertia, double* batch_sums, double* batch_counts, int apply_update, int on_device)
{
    if (!X || !batch_idx || !centers || !counts) return fail(MSM_ERR_INVALID, "mbk_step: null pointer");
    if (n < 1 || m < 1 || K < 1 || B < 1) return fail(MSM_ERR_INVALID, "mbk_step: bad shape");
    for (msm_idx_t b = 0; b < B; ++b)
        if (batch_idx[b] < 0 || batch_idx[b] >= n) return fail(MSM_ERR_INVALID, "mbk_step: batch index out of range");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    DevBuf &dC = pool(PS_Y), &dXb = pool(PS_X), &dIdx = pool(PS_IDX), &dL = pool(PS_LAB), &dW = pool(PS_PADX), &dS = pool(PS_PADY);
    T *dCent, *dNorm;
    int rc = km_prepare<T>(centers, K, m, dC, &dCent, &dNorm);
    if (rc) return rc;
    KmArgsT<T> P;
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
        std::vector<T> xb((size_t)B * m);
        for (msm_idx_t b = 0; b < B; ++b)
            memcpy(xb.data() + (size_t)b * m, X + batch_idx[b] * m, (size_t)m * sizeof(T));
        if ((rc = dXb.reserve(xb.size() * sizeof(T)))) return rc;
        MSM_HIP_CHECK(hipMemcpyAsync(dXb.p, xb.data(), xb.size() * sizeof(T), hipMemcpyHostToDevice, stream()));
        MSM_HIP_CHECK(hipStreamSynchronize(stream()));
        P.X = dXb.as<T>();
        P.rows = nullptr;
    }
    if ((rc = km_label_and_inertia<T>(P, batch_inertia))) return rc;   // (uses PS_W / PS_S for split candidates)
    if ((rc = dW.reserve((size_t)K * sizeof(T)))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(dW.p, counts, (size_t)K * sizeof(T), hipMemcpyHostToDevice, stream()));
    double *dSums = nullptr, *dCnts = nullptr;
    if (batch_sums || batch_counts) {
        if ((rc = dS.reserve(((size_t)K * m + (size_t)K) * sizeof(double)))) return rc;
        dSums = dS.as<double>();
        dCnts = dSums + (size_t)K * m;
    }
    hipLaunchKernelGGL(mbk_update_kernel<T>, dim3((unsigned)K), dim3(KNT), 4096 * sizeof(int), stream(), P,
                       dCent, dW.as<T>(), (T*)nullptr, dSums, dCnts, apply_update, MbkConv{});
    MSM_HIP_CHECK(hipGetLastError());
    if (apply_update) {
        MSM_HIP_CHECK(hipMemcpyAsync(centers, dCent, (size_t)K * m * sizeof(T), hipMemcpyDeviceToHost, stream()));
        MSM_HIP_CHECK(hipMemcpyAsync(counts, dW.p, (size_t)K * sizeof(T), hipMemcpyDeviceToHost, stream()));
    }
    if (batch_sums)
        MSM_HIP_CHECK(hipMemcpyAsync(batch_sums, dSums, (size_t)K * m * sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (batch_counts)
        MSM_HIP_CHECK(hipMemcpyAsync(batch_counts, dCnts, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int mbk_create(msm_mbk_t** out, msm_idx_t K, msm_idx_t m, int f64)
{
    if (!out || K < 1 || m < 1) return fail(MSM_ERR_INVALID, "msm_mbk_create: bad argument");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    msm_mbk* h = new msm_mbk();
    h->K = K;
    h->m = m;
    h->f64 = f64 ? 1 : 0;
    const size_t e = h->esz();
    hipError_t er = hipMalloc(&h->centers, (size_t)K * m * e);
    if (er == hipSuccess) er = hipMalloc(&h->counts, (size_t)K * e);
    if (er == hipSuccess) er = hipMalloc(&h->cnorm, (size_t)K * e);
    if (er == hipSuccess) er = hipMalloc((void**)&h->packed, ((size_t)K * m + K + 1) * sizeof(double));
    if (er == hipSuccess) er = hipMalloc((void**)&h->outbuf, 8 + (size_t)K * e);
    if (er != hipSuccess) {
        msm_mbk_destroy(h);
        return fail(MSM_ERR_HIP, "msm_mbk_create: hipMalloc failed: %s", hipGetErrorString(er));
    }
    *out = h;
    return MSM_OK;
}

}  // namespace

// dispatch on the handle's element type
#define MBK_TYPED(h, CALL_F32, CALL_F64) ((h)->f64 ? (CALL_F64) : (CALL_F32))

extern "C" {

int msm_mbk_create(msm_mbk_t** out, msm_idx_t K, msm_idx_t m) { return mbk_create(out, K, m, 0); }
int msm_mbk_create_f64(msm_mbk_t** out, msm_idx_t K, msm_idx_t m) { return mbk_create(out, K, m, 1); }
int msm_mbk_is_f64(msm_mbk_t* h) { return h ? h->f64 : 0; }

int msm_mbk_destroy(msm_mbk_t* h)
{
    if (!h) return MSM_OK;
    (void)hipStreamSynchronize(stream());
    if (h->centers) (void)hipFree(h->centers);
    if (h->counts) (void)hipFree(h->counts);
    if (h->cnorm) (void)hipFree(h->cnorm);
    if (h->packed) (void)hipFree(h->packed);
    if (h->outbuf) (void)hipFree(h->outbuf);
    if (h->stop) (void)hipFree(h->stop);
    if (h->pinned) (void)hipHostFree(h->pinned);
    delete h;
    return MSM_OK;
}

int msm_mbk_set(msm_mbk_t* h, const void* centers, const void* counts)
{
    if (!h || !centers || !counts) return fail(MSM_ERR_STATE, "msm_mbk_set: null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(h->centers, centers, (size_t)h->K * h->m * h->esz(), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipMemcpyAsync(h->counts, counts, (size_t)h->K * h->esz(), hipMemcpyHostToDevice, stream()));
    if (h->f64) mbk_launch_cnorm<double>(h);
    else mbk_launch_cnorm<float>(h);
    MSM_HIP_CHECK(hipGetLastError());
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_get(msm_mbk_t* h, void* centers, void* counts)
{
    if (!h) return fail(MSM_ERR_STATE, "msm_mbk_get: null handle");
    if (centers) MSM_HIP_CHECK(hipMemcpyAsync(centers, h->centers, (size_t)h->K * h->m * h->esz(), hipMemcpyDeviceToHost, stream()));
    if (counts) MSM_HIP_CHECK(hipMemcpyAsync(counts, h->counts, (size_t)h->K * h->esz(), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_step(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t B,
                 double* batch_inertia, void* counts_out, int apply_update, int on_device)
{
    if (!h || !X || !batch_idx) return fail(MSM_ERR_STATE, "msm_mbk_step: null argument");
    if (n < 1 || B < 1) return fail(MSM_ERR_INVALID, "msm_mbk_step: bad shape");
    return MBK_TYPED(h, mbk_step_t<float>(h, (const float*)X, n, batch_idx, B, batch_inertia, (float*)counts_out, apply_update, on_device),
                     mbk_step_t<double>(h, (const double*)X, n, batch_idx, B, batch_inertia, (double*)counts_out, apply_update, on_device));
}

/* msm_mbk_run in two halves: _begin queues the whole run (indices in, S steps, results out) and returns without waiting,
 * _end waits for it and hands the results over.  Between the two the host is free -- MiniBatchKMeans draws the NEXT run's
 * batch indices there (a quarter of a millisecond per 65,536 indices with the legacy RandomState, as long as a large-batch
 * step takes on the device). */
int msm_mbk_run_begin(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t S, msm_idx_t B,
                      msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement, const double* state6)
{
    if (!h || !X || !batch_idx || !state6) return fail(MSM_ERR_STATE, "msm_mbk_run: null argument");
    if (n < 1 || B < 1 || S < 1 || S > 4096) return fail(MSM_ERR_INVALID, "msm_mbk_run: bad shape");
    return MBK_TYPED(h, mbk_run_begin_t<float>(h, (const float*)X, n, batch_idx, S, B, first_step, alpha, max_no_improvement, state6),
                     mbk_run_begin_t<double>(h, (const double*)X, n, batch_idx, S, B, first_step, alpha, max_no_improvement, state6));
}

int msm_mbk_run_end(msm_mbk_t* h, double* state6, msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out)
{
    if (!h || !state6 || !steps_done || !converged || !inertias) return fail(MSM_ERR_STATE, "msm_mbk_run_end: null argument");
    if (!h->run_out) return fail(MSM_ERR_STATE, "msm_mbk_run_end: no run in flight");
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    const char* o = h->run_out;
    const size_t st_bytes = h->run_st_bytes;
    h->run_out = nullptr;
    const double* so = reinterpret_cast<const double*>(o);
    for (int i = 0; i < 5; ++i) state6[i] = so[i];
    state6[5] = so[5];
    *steps_done = (msm_idx_t)so[5];
    memcpy(inertias, so + 6, (size_t)(*steps_done) * sizeof(double));
    *converged = *reinterpret_cast<const int*>(o + st_bytes);
    if (counts_out) memcpy(counts_out, o + st_bytes + 8, (size_t)h->K * h->esz());
    return MSM_OK;
}

int msm_mbk_run(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t S, msm_idx_t B,
                msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement, double* state6,
                msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out)
{
    if (!steps_done || !converged || !inertias) return fail(MSM_ERR_STATE, "msm_mbk_run: null argument");
    const int rc = msm_mbk_run_begin(h, X, n, batch_idx, S, B, first_step, alpha, max_no_improvement, state6);
    if (rc) return rc;
    return msm_mbk_run_end(h, state6, steps_done, converged, inertias, counts_out);
}

/* msm_mbk_run for a ROW-SHARDED fit (one process per GPU): the S batches are GLOBAL (identical on every rank); this rank
 * passes the rows of each batch that it owns as local row numbers -- local_idx (host) holds them back to back, offsets[S + 1]
 * (host) delimits the steps -- and the batch size B of the whole batch.  Per step: label + fp64 sums / counts / inertia of
 * the local rows, ONE all-reduce of the packed [K m sums | K counts | inertia] buffer over the library communicator (RCCL on
 * the library stream), the identical update and convergence step on every rank.  Nothing returns to the host inside the
 * run; every rank stops at the same step (the criterion sees the all-reduced inertia).  Outputs as msm_mbk_run. */
int msm_mbk_run_sharded(msm_mbk_t* h, const void* X, msm_idx_t n_local, const msm_idx_t* local_idx, const msm_idx_t* offsets,
                        msm_idx_t S, msm_idx_t B, msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement,
                        double* state6, msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out)
{
    if (!h || !offsets || !state6 || !steps_done || !converged || !inertias) return fail(MSM_ERR_STATE, "msm_mbk_run_sharded: null argument");
    return MBK_TYPED(h, mbk_run_sharded_t<float>(h, (const float*)X, n_local, local_idx, offsets, S, B, first_step, alpha, max_no_improvement,
                                                 state6, steps_done, converged, inertias, (float*)counts_out),
                     mbk_run_sharded_t<double>(h, (const double*)X, n_local, local_idx, offsets, S, B, first_step, alpha, max_no_improvement,
                                                  state6, steps_done, converged, inertias, (double*)counts_out));
}

msm_idx_t msm_mbk_packed_size(msm_mbk_t* h) { return h ? (msm_idx_t)(h->K * h->m + h->K + 1) : 0; }

int msm_mbk_export_packed(msm_mbk_t* h, double* buf, int on_device)
{
    if (!h || !buf) return fail(MSM_ERR_STATE, "msm_mbk_export_packed: null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(buf, h->packed, (size_t)msm_mbk_packed_size(h) * sizeof(double),
                                 on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_apply_packed(msm_mbk_t* h, const double* buf, void* counts_out, int on_device)
{
    if (!h || !buf) return fail(MSM_ERR_STATE, "msm_mbk_apply_packed: null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(h->packed, buf, (size_t)msm_mbk_packed_size(h) * sizeof(double),
                                 on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream()));
    if (h->f64) mbk_launch_apply<double>(h, nullptr);
    else mbk_launch_apply<float>(h, nullptr);
    MSM_HIP_CHECK(hipGetLastError());
    if (counts_out) MSM_HIP_CHECK(hipMemcpyAsync(counts_out, h->counts, (size_t)h->K * h->esz(), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

/* sharded step, exchange half: msm_mbk_step(apply_update = 0) left this rank's [K*m sums | K counts | inertia] of ITS
 * batch rows in the handle's device buffer (msm_mbk_zero_packed for a rank that owns none of them); one all-reduce over
 * the library communicator (RCCL on the library stream, device buffer, in place) and the reduced buffer is applied
 * identically on every rank.  *batch_inertia / counts_out (host, K): the global batch inertia and the updated counts. */
int msm_mbk_zero_packed(msm_mbk_t* h)
{
    if (!h) return fail(MSM_ERR_STATE, "msm_mbk_zero_packed: null handle");
    MSM_HIP_CHECK(hipMemsetAsync(h->packed, 0, (size_t)msm_mbk_packed_size(h) * sizeof(double), stream()));
    return MSM_OK;
}

int msm_mbk_allreduce(msm_mbk_t* h, double* batch_inertia, void* counts_out)
{
    if (!h) return fail(MSM_ERR_STATE, "msm_mbk_allreduce: null handle");
    int rc = comm_allreduce_f64(h->packed, (size_t)msm_mbk_packed_size(h));
    if (rc) return rc;
    if (h->f64) mbk_launch_apply<double>(h, nullptr);
    else mbk_launch_apply<float>(h, nullptr);
    MSM_HIP_CHECK(hipGetLastError());
    if (batch_inertia)
        MSM_HIP_CHECK(hipMemcpyAsync(batch_inertia, h->packed + (size_t)h->K * h->m + h->K, sizeof(double), hipMemcpyDeviceToHost, stream()));
    if (counts_out) MSM_HIP_CHECK(hipMemcpyAsync(counts_out, h->counts, (size_t)h->K * h->esz(), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_reassign(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* rows, const msm_idx_t* which,
                     msm_idx_t n_reassign, double new_count, int on_device)
{
    if (!h || !X || !rows || !which) return fail(MSM_ERR_STATE, "msm_mbk_reassign: null argument");
    if (n_reassign <= 0) return MSM_OK;
    for (msm_idx_t i = 0; i < n_reassign; ++i)
        if (rows[i] < 0 || rows[i] >= n || which[i] < 0 || which[i] >= h->K) return fail(MSM_ERR_INVALID, "msm_mbk_reassign: index out of range");
    return MBK_TYPED(h, mbk_reassign_t<float>(h, (const float*)X, n, rows, which, n_reassign, new_count, on_device),
                     mbk_reassign_t<double>(h, (const double*)X, n, rows, which, n_reassign, new_count, on_device));
}

int msm_mbk_set_counts(msm_mbk_t* h, const void* counts)
{
    if (!h || !counts) return fail(MSM_ERR_STATE, "msm_mbk_set_counts: null argument");
    MSM_HIP_CHECK(hipMemcpyAsync(h->counts, counts, (size_t)h->K * h->esz(), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_mbk_label(msm_mbk_t* h, const void* X, msm_idx_t n, int32_t* labels, double* inertia, int on_device)
{
    if (!h || !X || !labels) return fail(MSM_ERR_STATE, "msm_mbk_label: null argument");
    if (inertia) *inertia = 0.0;
    if (n <= 0) return MSM_OK;
    return MBK_TYPED(h, mbk_label_t<float>(h, (const float*)X, n, labels, inertia, on_device),
                     mbk_label_t<double>(h, (const double*)X, n, labels, inertia, on_device));
}

int msm_kmeans_label_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* centers,
                         msm_idx_t K, int32_t* labels, double* inertia, int on_device)
{
    return kmeans_label_t<float>(X, n, m, centers, K, labels, inertia, on_device);
}

int msm_kmeans_label_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* centers,
                         msm_idx_t K, int32_t* labels, double* inertia, int on_device)
{
    return kmeans_label_t<double>(X, n, m, centers, K, labels, inertia, on_device);
}

int msm_mbk_step_f32(const float* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx,
                     msm_idx_t B, float* centers, float* counts, msm_idx_t K,
                     double* batch_inertia, double* batch_sums, double* batch_counts,
                     int apply_update, int on_device)
{
    return mbk_step_stateless_t<float>(X, n, m, batch_idx, B, centers, counts, K, batch_inertia, batch_sums, batch_counts, apply_update, on_device);
}

int msm_mbk_step_f64(const double* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx,
                     msm_idx_t B, double* centers, double* counts, msm_idx_t K,
                     double* batch_inertia, double* batch_sums, double* batch_counts,
                     int apply_update, int on_device)
{
    return mbk_step_stateless_t<double>(X, n, m, batch_idx, B, centers, counts, K, batch_inertia, batch_sums, batch_counts, apply_update, on_device);
}

}  // extern "C"
