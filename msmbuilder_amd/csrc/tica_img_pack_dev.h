// tica_img_pack_dev.h -- tica_img_kernel (packing pre-pass of the bf16 image path) and tica_img_steps_kernel
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "tica_common_dev.h"
#include "tica_img_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// bf16 image path (modes `bf16` / `bf16x2`, BASELINE configs[4]): the sum/difference form of 3.1b on the bf16 matrix
// pipe, in two kernels.
//
//  1. tica_img_kernel: ONE streaming pass turns the frame-major trajectories into a packed bf16 IMAGE of the
//     pair frames u = (x_t - r) + (x_{t+tau} - r) and d = x_t - x_{t+tau} (formed in fp32; r = mean shift row), laid out as
//     the bf16 MFMA wants its operands: 16-byte packets [8 consecutive pairs] of one feature, [pair group][feature][8].
//     The lag, the trajectory edges (a trajectory's pairs are padded with zero packets to a whole K-step), the shift, the
//     fp32 -> bf16 rounding (RNE; bf16x2: hi + mid = 16 significant bits, mid image alongside) and partial feature tiles
//     (the image is zero-padded to a multiple of 256 features) are all handled HERE, once per element.  bf16-STORED
//     trajectories (dtype_bytes = 2) enter through the same kernel.  Traffic: F sizeof(T) read (+ the lagged row, an
//     L2 hit) and 4 B (bf16x2: 8 B) written per pair and feature.
//  2. tica_img_mfma_kernel: H = sum u u^T and D = sum d d^T on the upper tiles, v_mfma_f32_32x32x16_bf16, straight from
//     the image: both operands of a product come from the SAME image at the SAME pair index, so there is no lag, no
//     edge, no mask and no conversion left in the hot loop -- 16-byte packets go global -> (registers) -> LDS unchanged and
//     come back as conflict-free ds_read_b128 fragments.  A workgroup is 8 waves and owns a 256 x 256 tile of H or of D
//     (wave: 64 x 128 outputs, 128 fp32 accumulators): per 32-pair K-step it stages 32 KiB for 128 MFMAs, HALF the
//     L2 -> LDS bytes per flop of the 128 x 128 tiles of round 1 (whose bf16 kernel sat at 0.11 of the bf16 peak, bound by
//     exactly that traffic plus the in-register transpose).  bf16x2 forms hi.hi + hi.mid + mid.hi + mid.mid per 16 pairs.
//     fp32 partials go to the fp64 slabs of the sum/difference layout every <= 8192 pairs; export and un-shift are 3.1b's.
// ---------------------------------------------------------------------------
struct ImgArgs {
    const TicaChunk* chunks;
    long long nchunks;
    long long ld;
    int F, Fp, lag, dtype_bytes;
    const float* shift;
    long long g_off; // first 8-pair group of the super-chunk being packed: the ring slot holds groups [g_off, g_off + G)
    bf16x8* u_hi;   // [G][Fp] packets
    bf16x8* d_hi;
    bf16x8* u_mid;  // bf16x2 only
    bf16x8* d_mid;
    double* colA;   // folded column sums: [nchunks][F] fp64 sums of the chunk's LEFT frames (nullptr: a separate pass made them)
};

// thread -> 4 consecutive features (one 16-byte load per row for float32, 8 bytes for bfloat16) x the 8 pairs of one group:
// 16 row loads in flight, an 8 x 4 transpose in registers, four 16-byte packets per image written back to back (a wave
// writes 4 KiB contiguous).  A workgroup handles one K-step (4 groups) of a 256-feature block per iteration.
// (nontemporal stores of the image were tried here: no change, 11.6 -> 11.6 ms per 1M x 2048 fit)
#define IMG_PACK_STORE(PTR, V) (*(PTR) = (V))
template <bool X2>
__global__ __launch_bounds__(256) void tica_img_kernel(ImgArgs P)
{
    const TicaChunk ch = P.chunks[blockIdx.x];
    const int fq = threadIdx.x & 63, gq = threadIdx.x >> 6;   // feature quad, group within the K-step
    const int f0 = blockIdx.y * 256 + fq * 4;                 // < Fp
    const bool vec = (P.F % 4 == 0) && (P.ld % 4 == 0) && ((((uintptr_t)ch.base) & 15) == 0);
    const bool vec2 = (P.F % 4 == 0) && (P.ld % 4 == 0) && ((((uintptr_t)ch.base) & 7) == 0);   // bfloat16 rows: 8-byte loads
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (P.shift && f0 + q < P.F) ? P.shift[f0 + q] : 0.f;
    long long nv = ch.len - P.lag - ch.row0;   // valid pairs of this chunk: left frames row0 .. with t < len - lag
    if (nv > ch.n) nv = ch.n;
    if (nv < 0) nv = 0;
    const long long nsteps = (nv + 31) / 32;   // whole K-steps of 32 pairs, zero padded
    const size_t esz = (size_t)P.dtype_bytes;
    const global_ptr<char> base = as_global<char>(ch.base);
    const int fc = f0 < P.F ? f0 : (P.F >= 4 ? P.F - 4 : 0);  // clamped column of the vector loads
    double cs[4] = {0.0, 0.0, 0.0, 0.0};   // P.colA: fp64 sums of the left frames this thread loads (its four features)
    __shared__ bf16x8 img_stage[4][X2 ? 4 : 2][256];   // per wave: the packets of its group, one tile per image
    for (long long st = 0; st < nsteps; ++st) {
        const long long gi = st * 4 + gq;
        float a[8][4], b[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long pidx = gi * 8 + e;
            const long long t = ch.row0 + (pidx < nv ? pidx : 0);
            const global_ptr<char> rowa = base + (size_t)t * (size_t)P.ld * esz;
            const global_ptr<char> rowb = rowa + (size_t)P.lag * (size_t)P.ld * esz;
            if (P.dtype_bytes == 4 && vec) {
                const raw_f32x4 va = *(global_ptr<raw_f32x4>)(rowa + (size_t)fc * 4), vb = *(global_ptr<raw_f32x4>)(rowb + (size_t)fc * 4);
                a[e][0] = va.x; a[e][1] = va.y; a[e][2] = va.z; a[e][3] = va.w;
                b[e][0] = vb.x; b[e][1] = vb.y; b[e][2] = vb.z; b[e][3] = vb.w;
            } else if (P.dtype_bytes == 2 && vec2) {
                // four bfloat16 = one 8-byte load (element-wise 2-byte loads made this pre-pass 2.1 TB/s on bfloat16-stored
                // input against 3.8 TB/s on float32)
                typedef unsigned raw_u32x2 __attribute__((ext_vector_type(2)));
                const raw_u32x2 va = *(global_ptr<raw_u32x2>)(rowa + (size_t)fc * 2), vb = *(global_ptr<raw_u32x2>)(rowb + (size_t)fc * 2);
                a[e][0] = __uint_as_float(va.x << 16); a[e][1] = __uint_as_float(va.x & 0xffff0000u);
                a[e][2] = __uint_as_float(va.y << 16); a[e][3] = __uint_as_float(va.y & 0xffff0000u);
                b[e][0] = __uint_as_float(vb.x << 16); b[e][1] = __uint_as_float(vb.x & 0xffff0000u);
                b[e][2] = __uint_as_float(vb.y << 16); b[e][3] = __uint_as_float(vb.y & 0xffff0000u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t col = (size_t)(f0 + q < P.F ? f0 + q : P.F - 1);   // clamped: masked below
                    if (P.dtype_bytes == 4) {
                        a[e][q] = *(global_ptr<float>)(rowa + col * 4);
                        b[e][q] = *(global_ptr<float>)(rowb + col * 4);
                    } else {
                        a[e][q] = (float)*(global_ptr<__bf16>)(rowa + col * 2);
                        b[e][q] = (float)*(global_ptr<__bf16>)(rowb + col * 2);
                    }
                }
            }
        }
        if (P.colA) {   // this pre-pass is bandwidth-bound: the 64 widenings and adds per thread and step are free
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool okp = (gi * 8 + e) < nv;
#pragma unroll
                for (int q = 0; q < 4; ++q) cs[q] += okp ? (double)a[e][q] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 uh, dh, um, dm;
            const bool inF = f0 + q < P.F;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = inF && (gi * 8 + e) < nv;
                float u, d;
                if (P.dtype_bytes == 2) {
                    // bfloat16 rows: a + b and a - b are exact in fp32 (8-bit significands), so ONE rounding each -- and the
                    // arithmetic of the fused kernel (tica_img_dev.h), which this path must match bit for bit
                    u = ok ? (a[e][q] + b[e][q]) - 2.f * r[q] : 0.f;
                    d = ok ? a[e][q] - b[e][q] : 0.f;
                } else {
                    // float32 rows: x - r first (exact by Sterbenz when |mean| >> std, the case the shift exists for)
                    const float ya = ok ? a[e][q] - r[q] : 0.f, yb = ok ? b[e][q] - r[q] : 0.f;
                    u = ya + yb;
                    d = ya - yb;
                }
                const __bf16 u1 = (__bf16)u, d1 = (__bf16)d;
                uh[e] = u1;
                dh[e] = d1;
                if (X2) {
                    um[e] = (__bf16)(u - (float)u1);
                    dm[e] = (__bf16)(d - (float)d1);
                }
            }
            // Round 5: the lane's four packets (64 contiguous bytes per image) go through an LDS staging tile and leave as
            // 1 KiB-contiguous wave stores.  Stored straight from the lane, a store instruction wrote 16 bytes of every 64
            // (lane stride 64 B): four partial passes over each cache line.  XOR swizzle: conflict-free both ways.
            const int sl = 4 * fq + q;
            img_stage[gq][0][sl ^ ((sl >> 3) & 7)] = uh;
            img_stage[gq][1][sl ^ ((sl >> 3) & 7)] = dh;
            if (X2) {
                img_stage[gq][2][sl ^ ((sl >> 3) & 7)] = um;
                img_stage[gq][3][sl ^ ((sl >> 3) & 7)] = dm;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave's LDS operations complete in order: its own packets are all there)
        __builtin_amdgcn_wave_barrier();
        {
            const size_t o0 = (size_t)(ch.g0 - P.g_off + gi) * (size_t)P.Fp + (size_t)blockIdx.y * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sr = 64 * i + fq;
                const int sp = sr ^ ((sr >> 3) & 7);
                IMG_PACK_STORE(P.u_hi + o0 + sr, img_stage[gq][0][sp]);
                IMG_PACK_STORE(P.d_hi + o0 + sr, img_stage[gq][1][sp]);
                if (X2) {
                    IMG_PACK_STORE(P.u_mid + o0 + sr, img_stage[gq][2][sp]);
                    IMG_PACK_STORE(P.d_mid + o0 + sr, img_stage[gq][3][sp]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read out before the next step's packets overwrite it
        __builtin_amdgcn_wave_barrier();
    }
    if (P.colA) {   // the four groups of a feature quad -> one sum per (chunk, feature): plain stores, one writer each
        __shared__ double red[4][64][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) red[gq][fq][q] = cs[q];
        __syncthreads();
        if (gq == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (f0 + q < P.F)
                    P.colA[(size_t)blockIdx.x * P.F + f0 + q] = (red[0][fq][q] + red[1][fq][q]) + (red[2][fq][q] + red[3][fq][q]);
        }
    }
}

// (ImgMfmaArgs and the MFMA kernels of the image path: tica_img_dev.h)

// Round 5, fused kernel (tica_img_dev.h): the K-step records {row of the step's first pair, valid pairs} from the chunk table.
// A chunk's pairs are padded to whole 32-pair steps exactly as tica_img_kernel padded the image (g0 = the chunk's first
// 8-pair group); bf16x2 splits a 32-pair step into two 16-pair steps (the second may hold no pair: nvalid 0).
// (round 6: `row_bytes` = ld x the element size -- the carried pack's 32-pair step records, x2 = 0, are the same table for
//  float32 rows)
__global__ void tica_img_steps_kernel(const TicaChunk* __restrict__ chunks, long long row_bytes, int lag, int x2, ImgStep* __restrict__ steps)
{
    const TicaChunk ch = chunks[blockIdx.x];
    long long nv = ch.len - lag - ch.row0;
    if (nv > ch.n) nv = ch.n;
    if (nv < 0) nv = 0;
    const long long n32 = (nv + 31) / 32, first = ch.g0 / 4;
    for (long long j = threadIdx.x; j < n32; j += blockDim.x) {
        const char* base = (const char*)ch.base + (size_t)(ch.row0 + j * 32) * (size_t)row_bytes;
        const int n = (int)(nv - j * 32 < 32 ? nv - j * 32 : 32);
        if (!x2) {
            steps[first + j] = ImgStep{base, n, 0};
        } else {
            steps[2 * (first + j)] = ImgStep{base, n < 16 ? n : 16, 0};
            steps[2 * (first + j) + 1] = n > 16 ? ImgStep{base + (size_t)16 * (size_t)row_bytes, n - 16, 0} : ImgStep{base, 0, 0};
        }
    }
}

// Carried pack (tica_img_dev.h): a chunk's folded column sums = the sum of its pack steps' (fp64, in step order).
// chunks: the carried super-chunk's entries of the chunk table; its groups are [g_start, g_end).
__global__ void tica_img_colsum_steps_kernel(const TicaChunk* __restrict__ chunks, long long nchunks, long long g_start, long long g_end,
                                             const double* __restrict__ colS, int F, int Fp, double* __restrict__ colA)
{
    const long long cix = blockIdx.x;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= F) return;
    const long long s0 = (chunks[cix].g0 - g_start) / 4;
    const long long s1 = ((cix + 1 < nchunks ? chunks[cix + 1].g0 : g_end) - g_start) / 4;
    double a = 0.0;
    for (long long s = s0; s < s1; ++s) a += colS[(size_t)s * (size_t)Fp + f];
    colA[(size_t)cix * F + f] = a;
}

}  // namespace msm
