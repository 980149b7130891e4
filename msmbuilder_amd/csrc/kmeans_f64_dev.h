// kmeans_f64_dev.h -- kmeans_label_f64_kernel: nearest-centre labelling of float64 rows on the fp64 matrix pipe.
// (included by kmeans.hip; also declares KmArgsT, the argument block shared by the fp32 and fp64 kernels)
//
// Why it exists (round 6): msmbuilder.cluster.MiniBatchKMeans is scikit-learn's estimator
// (/root/reference/msmbuilder/cluster/__init__.py:67-69), scikit-learn keeps float64 input in float64 (labels from a dgemm,
// float64 centres and counts), and the reference pipeline feeds it the float64 output of tICA.transform
// (decomposition/tica.py:329-352).  Rounds 1-5 narrowed such input to fp32.  This kernel is the float64 twin of
// kmeans_label_kernel:
//   label_i = argmin_j ( ||c_j||^2 - 2 x_i . c_j )   in float64, first minimum wins
// x.c on v_mfma_f64_16x16x4_f64 (64 cycles per 2,048 flop and SIMD: 78.6 TF), one workgroup (2 x 2 waves) per 128 rows x 128
// centres, a wave owning 64 x 64 of it as 4 x 4 MFMA blocks (64 accumulators in 128 registers), K-step 8 features through a
// double-buffered LDS pair of [128][10] float64 panels (pitch 10: the 16 rows x 2 feature columns of a half-wave's
// ds_read_b64 fragment cover the 64 banks exactly once; a step of 16 features spilled registers under the 256 of two waves
// per SIMD).  The fp64 MFMA is slow enough that nothing else matters: a K-step
// is 32 MFMAs = 2,048 cycles per wave beside 8 8-byte loads and 8 ds_write_b64 per thread, so the staging is plain
// (unconditional loads at clamped addresses, masked when they go to LDS -- a load under a branch is waited for on the spot) and
// rows of ANY feature count run at the same per-flop rate: 10 features pad to 12 (the K-step of the instruction is 4), where the
// fp32 kernel's 32-feature step is 70 % padding.  One kernel therefore serves every shape: MiniBatchKMeans' small steps
// (centres split over blockIdx.y so that the chip is filled; candidates merged by the inertia / reduce kernel), the
// final labelling pass, wide rows.
#pragma once
#include "common.h"

namespace msm {

constexpr int KNT = 256;   // threads per workgroup of every k-means kernel

template <typename T>   // T = float (fp32 MFMA labelling) or double (fp64 MFMA labelling: scikit-learn keeps float64 input in float64)
struct KmArgsT {
    const T* X;             // [n, m] (or gathered batch)
    const msm_idx_t* rows;  // optional row gather (batch indices), else nullptr
    long long n, m, K;
    const T* C;             // device [K, m]
    const T* cnorm;         // device [K]
    int32_t* labels;        // [n]
    // centre-split launch (small batches): blockIdx.y owns centre tiles [y*jspan, (y+1)*jspan) and
    // writes its (min value, index) candidates to pv/pi [gridDim.y][n]; a reduce kernel finishes
    long long jspan;        // 0 = all centres in one workgroup
    int xcd_ns;             // > 0 (kmeans_label_v4_kernel, large n): a 1-D grid of ceil(rowblocks / 8) x 8 x xcd_ns workgroups in
                            // which the xcd_ns centre splits of a row block are CONSECUTIVE workgroups of one XCD (see the kernel)
    T* pv;
    int* pi;
    const int* stop;        // optional device flag: non-zero -> the launch does nothing (msm_mbk_run: steps queued
                            // behind the one at which the convergence criterion fired)
};

constexpr int DKR = 128;   // rows per workgroup
constexpr int DKC = 128;   // centres per tile
constexpr int DKB = 8;     // features per K-step
constexpr int DKP = 10;    // LDS row pitch in doubles
constexpr size_t DK_LDS = (size_t)2 * (DKR + DKC) * DKP * sizeof(double);   // 40,960 B

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct DkStage {
    double x[4], c[4];
};

__global__ __launch_bounds__(KNT, 2) void kmeans_label_f64_kernel(KmArgsT<double> P)
{
    if (P.stop && *P.stop) return;  // uniform
    extern __shared__ __attribute__((aligned(16))) char dk_smem[];
    double* Xs = reinterpret_cast<double*>(dk_smem);   // [2][DKR * DKP]
    double* Cs = Xs + 2 * DKR * DKP;                    // [2][DKC * DKP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 4, cl = lane & 15;   // A[i = cl][k = kl], B[k = kl][j = cl]; D[i = kl + 4 r][j = cl]
    const long long row0 = (long long)blockIdx.x * DKR;
    const int split = (int)blockIdx.y;
    const long long m = P.m;
    const int nk = (int)((m + DKB - 1) / DKB);
    // staging: thread (fc = tid & 7, r0 = tid >> 3) moves feature column fc of rows r0 + 32 j, j = 0..3, of both panels
    const int fc = tid & 7, r0 = tid >> 3;
    const global_ptr<double> Xg = as_global<double>(P.X), Cg = as_global<double>(P.C);
    // element offsets of the workgroup's 128 staging rows of X (clamped into [0, n); gathered through P.rows): kept in LDS and
    // re-read every K-step -- eight 64-bit offsets per thread are 16 registers the accumulators need
    __shared__ long long xoff[DKR];
    if (tid < DKR) {
        long long i = row0 + tid;
        if (i > P.n - 1) i = P.n - 1;
        xoff[tid] = (P.rows ? (long long)as_global<msm_idx_t>(P.rows)[i] : i) * m;
    }
    __syncthreads();
    double best[16];
    int bidx[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        best[q] = INFINITY;
        bidx[q] = 0x7fffffff;
    }
    const long long jbeg = P.jspan ? (long long)split * P.jspan : 0;
    const long long jend = P.jspan ? (jbeg + P.jspan < P.K ? jbeg + P.jspan : P.K) : P.K;

    for (long long j0 = jbeg; j0 < jend; j0 += DKC) {
        f64x4 acc[4][4];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[bi][bj][r] = 0.0;
        DkStage st;
#define DK_LOAD(K0)                                                                               \
        {                                                                                         \
            const long long col_ = (K0) + fc < m ? (K0) + fc : m - 1;                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
                long long jc_ = j0 + r0 + 32 * j;   /* centre rows clamped to K - 1: never used */ \
                if (jc_ > P.K - 1) jc_ = P.K - 1;                                                 \
                st.x[j] = Xg[xoff[r0 + 32 * j] + col_];                                           \
                st.c[j] = Cg[jc_ * m + col_];                                                     \
            }                                                                                     \
        }
#define DK_STORE(K0, BUF)                                                                         \
        {                                                                                         \
            const bool in_ = (K0) + fc < m;                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                       \
                Xs[(BUF) * (DKR * DKP) + (r0 + 32 * j) * DKP + fc] = in_ ? st.x[j] : 0.0;         \
                Cs[(BUF) * (DKC * DKP) + (r0 + 32 * j) * DKP + fc] = in_ ? st.c[j] : 0.0;         \
            }                                                                                     \
        }
        DK_LOAD(0)
        __syncthreads();   // the previous centre tile's last fragment reads are done
        DK_STORE(0, 0)
        __syncthreads();
        for (int s = 0; s < nk; ++s) {
            const int buf = s & 1;
            if (s + 1 < nk) DK_LOAD((long long)(s + 1) * DKB)
            const double* Ab = Xs + buf * (DKR * DKP) + (wr * 64 + cl) * DKP + kl;
            const double* Bb = Cs + buf * (DKC * DKP) + (wc * 64 + cl) * DKP + kl;
            // (K-steps past the row's last feature hold zeros: skip whole instruction steps of a partial last K-step)
            const int kkn = (m - (long long)s * DKB >= DKB) ? DKB / 4 : (int)((m - (long long)s * DKB + 3) / 4);
#pragma unroll
            for (int kk = 0; kk < DKB / 4; ++kk) {
                if (kk < kkn) {   // uniform
                    double a[4], b[4];
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi) a[bi] = Ab[bi * 16 * DKP + kk * 4];
#pragma unroll
                    for (int bj = 0; bj < 4; ++bj) b[bj] = Bb[bj * 16 * DKP + kk * 4];
#pragma unroll
                    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                        for (int bj = 0; bj < 4; ++bj)
                            acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], b[bj], acc[bi][bj], 0, 0, 0);
                }
            }
            if (s + 1 < nk) DK_STORE((long long)(s + 1) * DKB, buf ^ 1)
            __syncthreads();
        }
#undef DK_LOAD
#undef DK_STORE
        // running argmin over this centre tile (ascending j per lane, strict <)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            const long long j = j0 + wc * 64 + bj * 16 + cl;
            if (j < jend) {
                const double cn = P.cnorm[j];
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = fma(-2.0, acc[bi][bj][r], cn);
                        if (v < best[bi * 4 + r]) {
                            best[bi * 4 + r] = v;
                            bidx[bi * 4 + r] = (int)j;
                        }
                    }
            }
        }
    }
    // min over the 16 lanes that share a row (value, lowest index), then over the two centre halves
    __syncthreads();
    double* redv = Xs;                                  // [2 (wc)][128 rows]
    int* redi = reinterpret_cast<int*>(Xs + 2 * DKR);   // [2][128]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        double v = best[q];
        int ix = bidx[q];
#pragma unroll
        for (int msk = 1; msk < 16; msk <<= 1) {
            const double ov = __shfl_xor(v, msk, 64);
            const int oi = __shfl_xor(ix, msk, 64);
            if (ov < v || (ov == v && oi < ix)) {
                v = ov;
                ix = oi;
            }
        }
        if (cl == 0) {
            const int row = wr * 64 + (q >> 2) * 16 + kl + 4 * (q & 3);
            redv[wc * DKR + row] = v;
            redi[wc * DKR + row] = ix;
        }
    }
    __syncthreads();
    if (tid < DKR) {
        const long long i = row0 + tid;
        if (i < P.n) {
            const double v0 = redv[tid], v1 = redv[DKR + tid];
            const int i0 = redi[tid], i1 = redi[DKR + tid];
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

}  // namespace msm
