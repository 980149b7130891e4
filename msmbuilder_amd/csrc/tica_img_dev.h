// tica_img_dev.h -- device side of the bf16 image path's MFMA pass (modes `bf16` / `bf16x2`, BASELINE configs[4]).
//
// H = sum u u^T and D = sum d d^T on the upper 256 x 256 tiles, straight from the packed image that tica_img_kernel
// writes: 16-byte packets [8 consecutive pairs] of one feature, [pair group][feature][8] (tica.hip).  Included by
// tica.hip (the product) and by scripts/micro/img_mfma.hip (the kernel alone, next to round 3's, on one box).
//
// Round 4: the ping-pong kernel.  Round 3's kernel ran its eight waves in lockstep -- all of them read fragments, then all
// of them multiply -- and moved every packet global -> registers -> ds_write_b128 -> LDS; its matrix pipes were busy 0.53
// of the time (profiles/r03_pmc_bf16_kernel.txt: 44 % of the wave cycles stalled behind the co-resident wave's MFMAs,
// 32 % at barriers and waitcnts).  Here
//   * packets go global -> LDS directly (`global_load_lds_dwordx4`: the image layout IS the LDS layout, a wave-instruction
//     moves 64 packets = one packet row of 64 features, 1 KiB contiguous on both sides): no staging registers, no
//     ds_write pass, a ring of NS slots of 32 KiB;
//   * the two waves of a SIMD alternate roles (MI355X_MICROARCH.md, "Two waves per SIMD"): while waves 0-3 multiply
//     K-step s, waves 4-7 read their fragments of step s and issue a quarter each of a panel's loads; after a barrier they
//     swap.  One wave per SIMD is in its MFMA cluster at any time, at raised priority, and its partner's LDS / VMEM
//     instructions issue beside it.
// The accumulation order per accumulator is the lockstep kernel's (bit-identical slabs; the micro-benchmark checks it).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace msm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float img_f32x16 __attribute__((ext_vector_type(16)));
typedef float img_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned img_u32x2 __attribute__((ext_vector_type(2)));

constexpr int IMG_TM = 128;                       // slab sub-tile (the sum/difference slab layout of tica.hip)
constexpr int IMG_NT = 512;                       // 8 waves: 4 (rows) x 2 (columns), each 64 x 128 outputs
constexpr int IMG_SLOT = 2 * 4 * 256 * 16;        // one K-step in LDS: [A, B][4 packet rows][256 features] 16-byte packets = 32 KiB

// a K-step of raw rows (fused kernel, carried pack): where its first pair's x_t row is and how many of its pairs exist
struct ImgStep {
    const void* rowa;   // x_t row of the step's first pair (trajectory-relative)
    int nvalid;         // pairs of this step that exist (0 .. pairs per step); the rest are zero packets
    int pad;
};

// Round 6: the CARRIED pack (see img_carry_* below): what the multiply of super-chunk k packs for super-chunk k + 1
struct ImgCarry {
    const ImgStep* psteps;   // [np] 32-pair pack steps of the carried super-chunk
    const float* shift;      // [F] reference row r, or nullptr
    bf16x8* u_hi;            // [4 np][Fp] packets: the carried super-chunk's images (the OTHER half of the ring)
    bf16x8* d_hi;
    bf16x8* u_mid;           // bf16x2 only
    bf16x8* d_mid;
    double* colS;            // [np][Fp] fp64 sums of each step's left frames (folded column sums), or nullptr
    long long row_bytes;     // ld * sizeof(element)
    long long lag_bytes;     // lag * row_bytes
    int np;                  // pack steps to carry; 0: nothing
    int nb;                  // items per step: Fp / 64 (bfloat16 rows) or Fp / 32 (float32 rows)
    int stride;              // K-steps of the multiply between two item quads of a workgroup (>= 2)
    int pad;
};

struct ImgMfmaArgs {
    const bf16x8* u_hi;
    const bf16x8* d_hi;
    const bf16x8* u_mid;
    const bf16x8* d_mid;
    long long nsteps;  // K-steps in the image (bf16: 4 groups = 32 pairs each; bf16x2: 2 groups = 16 pairs each)
    int Fp, T, T2, ntiles_sym, ntile2, S, kflush_steps;
    int main_steps;    // ping-pong kernel: K-steps of each full cohort (see img_main_steps); the remainder cohort takes the rest
    double* slabs;     // sum/difference layout: [(S + 1) * ntiles_sym][2][TM * TM]
    long long wrap;    // micro-benchmark only (WRAP kernels): K-step s is READ from step s % wrap -- a cache-resident image
    ImgCarry cy;       // CARRY kernels only
};

// persistent block id -> XCD-contiguous linear id (blocks land on XCD blockIdx % 8); bijective for any grid
__device__ __forceinline__ int img_xcd_linear_id()
{
    const int G = gridDim.x, b = blockIdx.x;
    const int q = G / 8, r = G % 8, xcd = b % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

// K-steps per full cohort for `nsteps` steps over `grid` workgroups and `units` units (host and device agree through the
// kernel argument): rounds x the remainder cohort's share, or an even split when the grid is a whole number of cohorts
inline int img_main_steps(long long nsteps, int grid, int units)
{
    const int S = grid / units, R = grid - S * units;
    if (S == 0) return 0;
    if (R == 0) return (int)(nsteps / S);
    const int rounds = (units + R - 1) / R;
    return (int)(nsteps / ((long long)S * rounds + 1)) * rounds;
}

// unit id of a cohort -> (which matrix, I <= J).  H units first, then D units: an XCD's share of a cohort then reads
// (mostly) ONE image, and consecutive units share their row panel I.
__device__ __forceinline__ void img_decode_unit(int unit, int T2, int& which, int& I, int& J)
{
    const int half = T2 * (T2 + 1) / 2;
    which = unit >= half ? 1 : 0;
    int uix = unit - which * half;
    I = 0;
    while (uix >= T2 - I) {
        uix -= T2 - I;
        ++I;
    }
    J = I + uix;
}

// fp64 merge of a wave's 64 x 128 accumulators into the private slabs of the 128 x 128 sub-tiles (upper ones only)
__device__ __forceinline__ void img_flush(img_f32x16 (&acc)[2][4], const ImgMfmaArgs& P, int cohort, int which, int I, int J,
                                          int wr, int wc, int kl, int cl)
{
    const int ti = 2 * I + (wr >> 1), tj = 2 * J + wc;   // 128-blocks of this wave's outputs
    if (ti <= tj && tj < P.T) {
        const int st = ti * P.T - ti * (ti - 1) / 2 + (tj - ti);
        double* slab = P.slabs + ((size_t)cohort * P.ntiles_sym + st) * (2 * IMG_TM * IMG_TM) + (size_t)which * (IMG_TM * IMG_TM);
        unsigned toff = (unsigned)(((wr & 1) * 64 + 4 * kl) * IMG_TM + cl);
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                double old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = (slab + (bi * 32 + (r & 3) + 8 * (r >> 2)) * IMG_TM + bj * 32)[toff];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    (slab + (bi * 32 + (r & 3) + 8 * (r >> 2)) * IMG_TM + bj * 32)[toff] = old[r] + (double)acc[bi][bj][r];
            }
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
}


// =====================================================================================================================
// Round 6: the CARRIED pack (VERDICT r5 #3b).
//
// The packing pre-pass (tica_img_kernel, tica_img_pack_dev.h) is HBM-bound and ran serialised in front of every multiply:
// 8 x 0.58 ms of a 12.1 ms fit of 1M x 2048, with the matrix pipes idle (running it beside the multiply on other CUs was
// measured slower in round 5: it needs the whole chip's memory pipelines).  But the pack is little work per MULTIPLY step:
// a super-chunk of N pairs is packed once and multiplied by U = 72 units, i.e. a workgroup that multiplies a K-step would
// have to pack 1/18 of a K-step's worth of rows beside it.  So the multiply of super-chunk k packs super-chunk k + 1 into
// the other half of the ring, in the load role of waves 4-7 (the role that already issues the LDS-DMA loads and reads the
// fragments beside the partner wave's MFMAs); only the first super-chunk of a launch still takes the pre-pass kernel.
//
//   * An ITEM = one 32-pair pack step x one 128-byte column block (64 bfloat16 / 32 float32 features): 64 row pieces, the
//     step's x_t rows and its x_{t+tau} rows, brought into a wave-private 8 KiB of LDS by 8 LDS-DMA instructions (8 lanes per
//     piece; a padding pair re-reads the step's last valid pair).  One load phase later (the wave's own vmcnt wait at the
//     end of its MFMA phase covers the pieces: they are older than that phase's image loads) lane (c, g) reads features
//     NF c .. NF c + NF - 1 of pairs 8 g .. 8 g + 7 (ds_read_b64; rows of odd g sit at the neighbouring position, so the two
//     g of a lane group use different bank halves), forms u and d with tica_img_kernel's arithmetic -- the packets are the
//     pre-pass kernel's bit for bit -- writes them XOR-swizzled into the same 8 KiB and stores them as 1 KiB-contiguous
//     wave stores.  The step's fp64 column sums of the left frames (the folded sums) are reduced over the four g by
//     shuffles and stored per (step, feature); tica_img_colsum_steps_kernel adds a chunk's steps afterwards.
//   * The four carrier waves take the four items of a QUAD in the same phases; workgroup p (XCD-linear) takes quads p,
//     p + G, ...: `stride` multiply steps apart, and whatever is left when its segments end is packed in a drain loop, so
//     every item is packed whatever the step counts are.
//   * LDS: the ring's 128 KiB + 4 x 8 KiB = all 160 KiB (LDS-DMA reaches every byte of it: scripts/micro/lds_dma_high.hip).
// =====================================================================================================================
struct ImgCarryState {
    int next;      // next pack step of this workgroup's column blocks
    int pending;   // pack step whose pieces are in the private areas (-1: none)
    int wait;      // multiply steps until the next quad may be issued
    int step;      // distance between this workgroup's pack steps: the workgroups that share its column blocks
    int blk;       // this wave's column block (constant: the lane's reference-row entries stay in registers)
};
// Order inside the carrier's load phase: the carried operations AHEAD of the step's image loads, the item converted one
// multiply step after its pieces were issued (the wave's plain end-of-phase wait, vmcnt in order, covers them).  The other
// order was built and measured -- pieces and stores BEHIND the image loads, two end-of-phase waits with an allowance for
// them, conversion three steps later, so that no wait ever stands on an HBM read: 1M x 2048 bfloat16 rows 12.3 -> 12.9 ms
// against 9.5 -> 10.1 ms for this order (scripts/carryabl.py, profiles/r06_carry.txt): what the carried pack costs is
// not the latency of its pieces but its instructions, which a load-role wave issues in the gaps of its partner's MFMAs,
// and its 12 GB of traffic beside the multiply's image loads.

template <int ESZ>
__device__ __forceinline__ void img_carry_issue(const ImgCarry& C, char* priv, int ps, int blk)
{
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // (the lane's offsets are computed HERE, per item: hoisted out of the K loop they are 30 registers spilled for its whole length)
    typedef const __attribute__((address_space(4))) ImgStep* step_cptr;
    const step_cptr dp = (step_cptr)(uintptr_t)(C.psteps + ps);
    const unsigned long long rowa = (unsigned long long)(uintptr_t)dp->rowa;
    const int last = dp->nvalid - 1;   // >= 0: a pack step holds at least one pair
    const unsigned coloff = (unsigned)blk * 128u + (unsigned)(lane & 7) * 16u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // LDS position 8 j + (lane >> 3) of [x_t rows | x_{t+tau} rows] holds pair (position ^ bit 3 of it)
        int pr = (8 * j + (lane >> 3)) & 31;
        pr ^= (pr >> 3) & 1;
        const int pe = pr < last ? pr : last;
        const unsigned long long ga = rowa + (unsigned long long)(unsigned)pe * (unsigned long long)C.row_bytes +
                                      (j >= 4 ? (unsigned long long)C.lag_bytes : 0ull) + coloff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(global_ptr<char>)ga,
                                         (__attribute__((address_space(3))) void*)(priv + j * 1024), 16, 0, 0);
    }
}

template <bool X2, int ESZ>
struct ImgCarryGeom {
    static constexpr int NF = ESZ == 2 ? 4 : 2;      // features per lane
    static constexpr int W = 16 * NF;                // features per item
    static constexpr int RB = W * 16;                // bytes of one 8-pair group of one image: W packets
    static constexpr int IMGB = 4 * RB;              // ... of the item's four groups
    static constexpr int NSETS = (X2 && ESZ == 2) ? 2 : 1;   // bf16x2 on bfloat16 rows: four images of 4 KiB: u images, then d images
    static constexpr int NI = X2 ? (ESZ == 2 ? 2 : 4) : 2;   // images per set
};

template <bool X2, int ESZ, bool TAIL>
__device__ __forceinline__ void img_carry_convert_body(const ImgCarry& C, char* priv, int ps, int blk, int nvalid, int Fp, const float (&r)[4])
{
    typedef ImgCarryGeom<X2, ESZ> GM;
    constexpr int NF = GM::NF, W = GM::W, RB = GM::RB, IMGB = GM::IMGB, NSETS = GM::NSETS, NI = GM::NI;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // (see img_carry_issue)
    const int c = lane & 15, g = lane >> 4;
    img_u32x2 ra[8], rb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int pos = 8 * g + (e ^ (g & 1));
        ra[e] = *reinterpret_cast<const img_u32x2*>(priv + pos * 128 + 8 * c);
        rb[e] = *reinterpret_cast<const img_u32x2*>(priv + 4096 + pos * 128 + 8 * c);
    }
    auto raw = [&](const img_u32x2& w, int q) -> float {
        if (ESZ == 4) return __uint_as_float(q == 0 ? w.x : w.y);
        const unsigned v = q < 2 ? w.x : w.y;
        return (q & 1) ? __uint_as_float(v & 0xffff0000u) : __uint_as_float(v << 16);
    };
    // the folded column sums: fp64 sums of the left frames (raw values), over the lane's 8 pairs, then over the four g
    double cs[NF];
    if (C.colS) {
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            cs[q] = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double v = (double)raw(ra[e], q);
                cs[q] += (!TAIL || 8 * g + e < nvalid) ? v : 0.0;
            }
            cs[q] += __shfl_xor(cs[q], 16, 64);
            cs[q] += __shfl_xor(cs[q], 32, 64);
        }
    }
    // packets, written XOR-swizzled (conflict-free ds_write_b128 and read-back), tica_img_kernel's arithmetic
    auto swz = [&](int sl) -> int { return NF == 4 ? (sl ^ ((sl >> 3) & 7)) : (sl ^ ((sl >> 3) & 1)); };
#pragma unroll
    for (int set = 0; set < NSETS; ++set) {
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            bf16x8 uh, dh, um, dm;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = !TAIL || 8 * g + e < nvalid;
                const float a = raw(ra[e], q), b = raw(rb[e], q);
                float u, d;
                if (ESZ == 2) {
                    u = ok ? (a + b) - 2.f * r[q] : 0.f;
                    d = ok ? a - b : 0.f;
                } else {
                    const float ya = ok ? a - r[q] : 0.f, yb = ok ? b - r[q] : 0.f;
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
            char* w = priv + g * RB + swz(NF * c + q) * 16;
            if (!X2) {
                *reinterpret_cast<bf16x8*>(w) = uh;
                *reinterpret_cast<bf16x8*>(w + IMGB) = dh;
            } else if (ESZ == 4) {
                *reinterpret_cast<bf16x8*>(w) = uh;
                *reinterpret_cast<bf16x8*>(w + IMGB) = dh;
                *reinterpret_cast<bf16x8*>(w + 2 * IMGB) = um;
                *reinterpret_cast<bf16x8*>(w + 3 * IMGB) = dm;
            } else if (set == 0) {
                *reinterpret_cast<bf16x8*>(w) = uh;
                *reinterpret_cast<bf16x8*>(w + IMGB) = um;
            } else {
                *reinterpret_cast<bf16x8*>(w) = dh;
                *reinterpret_cast<bf16x8*>(w + IMGB) = dm;
            }
        }
        // read back by packet row, store 1 KiB per wave-instruction
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            bf16x8* img = !X2 ? (i == 0 ? C.u_hi : C.d_hi)
                        : ESZ == 4 ? (i == 0 ? C.u_hi : i == 1 ? C.d_hi : i == 2 ? C.u_mid : C.d_mid)
                        : set == 0 ? (i == 0 ? C.u_hi : C.u_mid) : (i == 0 ? C.d_hi : C.d_mid);
#pragma unroll
            for (int k = 0; k < IMGB / 1024; ++k) {
                const int gg = NF == 4 ? k : 2 * k + (lane >> 5);
                const int sr = NF == 4 ? lane : (lane & 31);
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(priv + i * IMGB + gg * RB + swz(sr) * 16);
                img[(size_t)(4 * (long long)ps + gg) * (size_t)Fp + (size_t)(blk * W + sr)] = v;
            }
        }
    }
    // one store instruction: the shuffles left all four g with the same sums, lane (c, g) stores the sum of feature NF c + g
    if (C.colS) {
        double* dst = C.colS + (size_t)ps * (size_t)Fp + (size_t)(blk * W + NF * c);
        if (NF == 4) dst[g] = g == 0 ? cs[0] : g == 1 ? cs[1] : g == 2 ? cs[NF - 2] : cs[NF - 1];
        else if (g < 2) dst[g] = g == 0 ? cs[0] : cs[1];
    }
}

// one load phase of a carrier wave, ahead of the step's fragments and image loads: convert the pending item (its pieces were
// covered by the wave's last end-of-phase wait), then issue the next one when it is due
template <bool X2, int ESZ>
__device__ __forceinline__ void img_carry_phase(const ImgCarry& C, ImgCarryState& cs, char* priv, int Fp, const float (&r)[4], bool drain)
{
    if (cs.pending >= 0) {
        typedef const __attribute__((address_space(4))) ImgStep* step_cptr;
        const int nvalid = ((step_cptr)(uintptr_t)(C.psteps + cs.pending))->nvalid;
        if (!(C.pad & 2)) {   // (pad: timing ablations of scripts/carryabl.py -- 1 no pieces, 2 no conversion; the results are then wrong)
            if (__builtin_expect(nvalid >= 32, 1)) img_carry_convert_body<X2, ESZ, false>(C, priv, cs.pending, cs.blk, nvalid, Fp, r);
            else img_carry_convert_body<X2, ESZ, true>(C, priv, cs.pending, cs.blk, nvalid, Fp, r);
        }
        cs.pending = -1;
    }
    --cs.wait;
    if (cs.next < C.np && (drain || cs.wait <= 0)) {
        if (!(C.pad & 1)) img_carry_issue<ESZ>(C, priv, cs.next, cs.blk);
        cs.pending = cs.next;
        cs.next += cs.step;
        cs.wait = C.stride;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The ping-pong kernel.  Phases are separated by workgroup barriers; K-step s of a cohort's share occupies phases 2n and
// 2n + 1 (n = s - s0):
//     phase 2n    : waves 0-3  MFMA(s)                      | waves 4-7  fragments(s),     issue panel A of step s + D
//     phase 2n + 1: waves 4-7  MFMA(s)                      | waves 0-3  fragments(s + 1), issue panel B of step s + D
// A wave waits for its own loads at the END of its MFMA phase, allowing the newest LAG batches (of 4 loads) to stay in
// flight; with D = 2 + LAG and NS = D + 1 ring slots
//   * panel A of step t is issued in phase 2 (t - D), complete (waited for + barrier) by the end of phase 2 (t - D) + 1 + 2 LAG,
//     panel B one phase later: both <= 2 t - 2, and the first reader of step t is phase 2 t - 1;                      (RAW)
//   * the slot of step t was last read in phase 2 (t - NS) + 0 ... it is rewritten from phase 2 (t - D) > 2 (t - NS). (WAR)
// LAG = 0: a load has one phase (~600 cycles) to land before its issuer waits for it; LAG = 1: three phases.
// ---------------------------------------------------------------------------------------------------------------------
// One SEGMENT of a workgroup's work: K-steps [s0, s1) of one unit, merged into slab row `cohort`.
template <bool X2, int LAG, bool WRAP, int ABL, int CARRY = 0>   // CARRY: 0, or the element size of the rows the load role packs (2 / 4)
__device__ __forceinline__ void img_pp_segment(const ImgMfmaArgs& P, char* smem, int unit, int cohort, int s0, int s1,
                                               ImgCarryState* cst = nullptr, const float (*cr)[4] = nullptr)
{
    constexpr int D = 2 + LAG, NS = D + 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: scalar branches on the role
    const int grp = wave >> 2, wi = wave & 3;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 5, cl = lane & 31;
    int which, I, J;
    img_decode_unit(unit, P.T2, which, I, J);
    const bf16x8* hi = which ? P.d_hi : P.u_hi;
    const bf16x8* mid = which ? P.d_mid : P.u_mid;

    img_f32x16 acc[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;

    // (s0, s1 are scalars: the step clamp of the loads must not cost VALU instructions; 32-bit: the scalar unit has no
    //  64-bit ordered compare, the host keeps nsteps per launch far below 2^31)
    if (s1 <= s0) return;   // uniform over the workgroup

    // ---- loads: this wave moves packet row `wi` of ONE panel per step (waves 4-7: panel A = columns I; waves 0-3: panel B =
    //      columns J), four 64-feature segments = four wave-instructions.  Address = a scalar base per step + ONE per-lane
    //      32-bit offset + an immediate per segment: while the partner wave streams MFMAs a VALU instruction of this wave
    //      waits for a gap in that stream, so the load phase must not need any (first version: two 64-bit VALU adds per
    //      load and 24 register moves per step -- the phase took 890 cycles against 512 of MFMA).
    const int mypanel = grp == 1 ? 0 : 1;
    const bf16x8* rowimg = (X2 && wi >= 2) ? mid : hi;                  // bf16x2: packet rows 0-1 = hi groups, 2-3 = mid groups
    const long long gmul = X2 ? 2 : 4;
    const int grow = X2 ? (wi & 1) : wi;
    const size_t colbase = (size_t)(mypanel == 0 ? I : J) * 256;       // first packet column of the panel
    const unsigned lds_my = (unsigned)(mypanel * 16384 + wi * 4096);    // byte offset inside a slot
    const unsigned lane16 = (unsigned)lane * 16u;
    // (a diagonal unit, I == J, could load ONE panel: measured 4 % SLOWER -- its waves 0-3 then idle through their load phase
    //  and the unit drifts away from the cohort whose panels it shares in L2)
    auto issue = [&](int step, int slot) {
        if (ABL & 1) return;
        int sc = step < s1 ? step : s1 - 1;                             // beyond the share: a harmless re-load (counts stay uniform)
        if (WRAP) sc %= (int)P.wrap;
        const char* sbase = reinterpret_cast<const char*>(rowimg + (size_t)((long long)sc * gmul + grow) * (size_t)P.Fp + colbase);   // uniform
        char* dst = smem + (unsigned)slot * IMG_SLOT + lds_my;
        // the instruction's immediate offset advances BOTH addresses (global: vaddr + offset; LDS: M0 + offset + 16 lane):
        // one address computation and one M0 write serve the four segments
        const __attribute__((address_space(1))) void* g = (const __attribute__((address_space(1))) void*)(sbase + lane16);
        __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)dst;
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    };
    // ---- fragments of a step: set 0 is what the step's first MFMAs take (bf16: pairs 0-15; bf16x2: the mid images),
    //      set 1 the rest (pairs 16-31; the hi images)
    const unsigned fragA = (unsigned)((kl * 256 + wr * 64 + cl) * 16);
    const unsigned fragB = (unsigned)(16384 + (kl * 256 + wc * 128 + cl) * 16);
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    auto frags = [&](int slot, bool force = false) {
        if ((ABL & 2) && !force) return;
        const char* base = smem + (unsigned)slot * IMG_SLOT;
        constexpr int k0 = X2 ? 2 : 0, k1 = X2 ? 0 : 2;   // first packet row of set 0 / set 1
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            fa0[bi] = *reinterpret_cast<const bf16x8*>(base + fragA + (k0 * 256 + bi * 32) * 16);
            fa1[bi] = *reinterpret_cast<const bf16x8*>(base + fragA + (k1 * 256 + bi * 32) * 16);
        }
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            fb0[bj] = *reinterpret_cast<const bf16x8*>(base + fragB + (k0 * 256 + bj * 32) * 16);
            fb1[bj] = *reinterpret_cast<const bf16x8*>(base + fragB + (k1 * 256 + bj * 32) * 16);
        }
    };
    auto mfmas = [&]() {
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
        if (X2) {   // per accumulator: mid.mid, hi.mid, mid.hi, hi.hi -- round 3's order
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj) {
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
                }
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
    };
    auto wait_loads = [&]() {
        if (LAG == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (LAG == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
#define IMG_PP_BARRIER() do { if (!(ABL & 4)) __builtin_amdgcn_s_barrier(); } while (0)

    // ---- prologue: steps s0 .. s0 + D - 1 into slots 0 .. D - 1
#pragma unroll
    for (int q = 0; q < D; ++q) issue(s0 + q, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IMG_PP_BARRIER();

    // Two role-specific loops (same barrier count): one code path per wave, so a step's fragments are read straight into
    // the registers its MFMAs take -- a shared loop body made the compiler read into temporaries and move them.
    int slot = 0, slot_ld = D;   // slot of step s; slot of step s + D (NS = D + 1: the one "behind" slot)
    int steps_acc = 0;
    if (grp == 0) {
        frags(0, true);
        for (int s = s0; s < s1; ++s) {
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            mfmas();                                   // phase 2n
            wait_loads();
            IMG_PP_BARRIER();
            if (s + 1 < s1) frags(slot1);              // phase 2n + 1
            issue(s + D, slot_ld);
            IMG_PP_BARRIER();
            slot = slot1;
            slot_ld = slot_ld + 1 == NS ? 0 : slot_ld + 1;
            if (++steps_acc >= P.kflush_steps || s + 1 == s1) {
                steps_acc = 0;
                img_flush(acc, P, cohort, which, I, J, wr, wc, kl, cl);
            }
        }
    } else {
        if (ABL & 2) frags(0, true);
        for (int s = s0; s < s1; ++s) {
            // (carried pack first: the fragments' 48 registers are not live beside the item's)
            if (CARRY) img_carry_phase<X2, CARRY ? CARRY : 2>(P.cy, *cst, smem + NS * IMG_SLOT + wi * 8192, P.Fp, *cr, false);
            frags(slot);                               // phase 2n
            issue(s + D, slot_ld);
            IMG_PP_BARRIER();
            mfmas();                                   // phase 2n + 1
            wait_loads();
            IMG_PP_BARRIER();
            slot = slot + 1 == NS ? 0 : slot + 1;
            slot_ld = slot_ld + 1 == NS ? 0 : slot_ld + 1;
            if (++steps_acc >= P.kflush_steps || s + 1 == s1) {
                steps_acc = 0;
                img_flush(acc, P, cohort, which, I, J, wr, wc, kl, cl);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail loads: nothing may still be writing LDS ...
    IMG_PP_BARRIER();                                  // ... when the next segment's prologue refills the ring (or at exit)
#undef IMG_PP_BARRIER
}

// Work split (round 4): with U = ntile2 units and G workgroups (one per CU of the stream), S = G / U COHORTS take a share of
// `main_steps` K-steps each, all units side by side -- workgroups that read the same panels at the same time, next to each
// other in the XCD-linear order.  The other R = G % U workgroups (40 of 256 at 2,048 features, which round 3 left idle)
// form a REMAINDER cohort that takes the rest of the steps in ceil(U / R) rounds of R units (slab row S): with
// main_steps = rounds x (remainder's steps) every workgroup multiplies for the same time.
template <bool X2, int LAG, bool WRAP = false, int ABL = 0, int CARRY = 0>   // ABL: micro-benchmark ablations (1: no loads, 2: no fragment reads, 4: no barriers, 8: no priority)
__global__ __launch_bounds__(IMG_NT, 1) void tica_img_pp_kernel(ImgMfmaArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS][A, B][4][256] packets (+ CARRY: 4 x 8 KiB) -- the ONLY LDS object
    const int U = P.ntile2, G = (int)gridDim.x;
    const int S = G / U, R = G - S * U;
    const int p = img_xcd_linear_id();
    // carried pack: workgroup p takes the column blocks 4 bg .. 4 bg + 3 (one per carrier wave), bg = p mod (blocks / 4), of every
    // J-th pack step, J = the workgroups that share bg
    ImgCarryState cst;
    float cr[4] = {0.f, 0.f, 0.f, 0.f};
    if (CARRY) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nbg = P.cy.nb >> 2, bg = p % nbg;
        cst.next = p / nbg;
        cst.step = (G - bg + nbg - 1) / nbg;
        cst.blk = 4 * bg + (wave & 3);
        cst.pending = -1;
        cst.wait = 1;
        if (wave >= 4 && P.cy.shift) {
            constexpr int NF = CARRY == 4 ? 2 : 4;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int q = 0; q < NF; ++q) cr[q] = P.cy.shift[cst.blk * 16 * NF + NF * (lane & 15) + q];
#pragma unroll
            for (int q = 0; q < NF; ++q) asm volatile("" : "+v"(cr[q]));   // (the compiler's wait for these loads is HERE, not at every use in the K loop)
        }
    }
    if (p < S * U) {
        const int cohort = p / U;
        const int s1 = (R == 0 && cohort == S - 1) ? (int)P.nsteps : (cohort + 1) * P.main_steps;   // (no remainder cohort: the last one takes the odd steps)
        img_pp_segment<X2, LAG, WRAP, ABL, CARRY>(P, smem, p - cohort * U, cohort, cohort * P.main_steps, s1, &cst, &cr);
    } else {
        const int r = p - S * U;
        for (int unit = r; unit < U; unit += R)
            img_pp_segment<X2, LAG, WRAP, ABL, CARRY>(P, smem, unit, S, S * P.main_steps, (int)P.nsteps, &cst, &cr);
    }
    if (CARRY) {
        // drain: the quads this workgroup's steps did not reach (every segment ended with vmcnt(0) + a barrier)
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (wave >= 4) {
            constexpr int NS = 3 + LAG;
            while (cst.pending >= 0 || cst.next < P.cy.np) {
                img_carry_phase<X2, CARRY ? CARRY : 2>(P.cy, cst, smem + NS * IMG_SLOT + (wave & 3) * 8192, P.Fp, cr, true);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
}


// =====================================================================================================================
// Round 5: the FUSED kernel -- bfloat16-stored trajectories, no image.
//
// The two-kernel path writes the pair frames u = x_t + x_{t+tau} - 2 r and d = x_t - x_{t+tau} as a packed image (2x the
// bfloat16 input) and reads it back 1.8x through the L2 fabric: 7.9x the algorithmic bytes per fit, with the packing pass
// serialised in front of the multiply (profiles/r04_pmc_config5.txt).  Here the load role of the ping-pong kernel reads
// the RAW rows x_t, x_{t+tau} (8-byte loads: four bfloat16 features of one pair per lane, a wave-instruction = one
// 512-byte row segment of a 256-feature panel), forms u (H units) or d (D units) in fp32, rounds to bf16
// (bf16x2: hi, or mid = bf16(v - hi)) and writes the four 16-byte packets of its 4 features x 8 pairs straight into the
// slot ring.  The 8 x 4 transpose costs nothing: a packet is the eight pairs of ONE feature, which the lane already
// holds in eight registers.  No image, no second kernel, no ring in HBM; the lag, the trajectory edges and the shift are
// handled once per (unit, element) in the load role, beside the partner wave's MFMAs.
//
//   * Packet placement.  A lane owns features 4 l .. 4 l + 3 of its panel; writing packet q of every lane at its natural
//     position (4 l + q) 16 would put the 8 lanes of a ds_write_b128 group on 2 of the 8 bank groups (4-way conflict).
//     The slot therefore holds each 32-feature block PERMUTED: feature 4 c + q sits at position 8 q + c, so 8 consecutive
//     lanes write 128 contiguous bytes.  The fragment reads are unchanged (position-linear, conflict-free); the MFMA
//     tile's row / column p of a block is feature beta(p) = 4 (p & 7) + (p >> 3), and only the slab merge (img_flush,
//     PERM) needs to know.  Every accumulator still adds the same products in the same order as the image path's, so the
//     slabs are bit-identical to the two-kernel path's on the same bfloat16 input (tests/test_gpu_configs.py).
//   * Steps.  K-step s of the launch is described by a 16-byte record {row of pair 0, valid pairs} built on the device
//     from the chunk table (tica.hip, tica_img_steps_kernel): trajectories are padded to whole 32-pair steps exactly as
//     the image was; invalid pairs of a trajectory's last step read a zero row and subtract no shift, i.e. give the
//     zero packets the image held.
//   * Ring: D = 2 steps ahead, NS = 3 slots (96 KiB).  The raw rows of step s + 3 are in registers (16 x 8 bytes per
//     lane) while step s is multiplied: one full step (two phases) to land, waited for by the compiler's own vmcnt.
//       phase 2n    : waves 0-3  MFMA(s)   | waves 4-7  fragments(s),     convert panel A of step s + 2, load A raw of s + 3
//       phase 2n + 1: waves 4-7  MFMA(s)   | waves 0-3  fragments(s + 1), convert panel B of step s + 2, load B raw of s + 3
//     A(t) is written in phase 2 (t - 2), B(t) in 2 (t - 2) + 1; first read in phase 2 t - 1 (RAW: two barriers between);
//     the slot of step t is last read in phase 2 t and rewritten from phase 2 (t + 1) (WAR: one barrier between).
// =====================================================================================================================
struct ImgFusedArgs {
    const ImgStep* steps;   // [nsteps]
    const float* shift;     // [F] reference row r, or nullptr
    long long row_bytes;    // ld * 2
    long long lag_bytes;    // lag * ld * 2
    int nsteps;
    int T, T2, ntiles_sym, ntile2, S, kflush_steps, main_steps;
    double* slabs;
};

// img_flush for the permuted slot layout: tile position p of a 32-block is feature beta(p) = 4 (p & 7) + (p >> 3)
__device__ __forceinline__ void img_flush_perm(img_f32x16 (&acc)[2][4], double* slabs, int ntiles_sym, int T, int cohort, int which, int I, int J,
                                               int wr, int wc, int kl, int cl)
{
    const int ti = 2 * I + (wr >> 1), tj = 2 * J + wc;
    if (ti <= tj && tj < T) {
        const int st = ti * T - ti * (ti - 1) / 2 + (tj - ti);
        double* slab = slabs + ((size_t)cohort * ntiles_sym + st) * (2 * IMG_TM * IMG_TM) + (size_t)which * (IMG_TM * IMG_TM);
        // accumulator r of block (bi, bj): position row (r & 3) + 8 (r >> 2) + 4 kl -> feature 4 (r & 3) + 16 kl + (r >> 2)
        unsigned toff = (unsigned)(((wr & 1) * 64 + 16 * kl) * IMG_TM + 4 * (cl & 7) + (cl >> 3));
        asm volatile("" : "+v"(toff));
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // eight words at a time: the merge must not push the loop's registers out
                    double old[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = 8 * h + r8;
                        old[r8] = (slab + (bi * 32 + 4 * (r & 3) + (r >> 2)) * IMG_TM + bj * 32)[toff];
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = 8 * h + r8;
                        (slab + (bi * 32 + 4 * (r & 3) + (r >> 2)) * IMG_TM + bj * 32)[toff] = old[r8] + (double)acc[bi][bj][r];
                    }
                }
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
}

template <bool X2, int WHICH, int ABL>
__device__ __forceinline__ void img_fused_segment(const ImgFusedArgs& P, char* smem, int I, int J, int cohort, int s0, int s1)
{
    constexpr int D = 2, NS = 3;
    constexpr int PS = X2 ? 16 : 32;                              // pairs per K-step
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 5, cl = lane & 31;

    img_f32x16 acc[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
    if (s1 <= s0) return;   // uniform over the workgroup
    // LDS: [8 waves][x_t rows | x_{t+tau} rows][8 rows][512 bytes] raw staging, then the packet ring [3][A, B][4][256].  (Staging
    // FIRST: the LDS-DMA destination is M0-addressed; the ring is only ever touched by ds_read / ds_write.)
    char* const rawbuf = smem + (unsigned)wave * 8192u;
    char* const ring = smem + 65536u;

    // ---- the load role: packet row `wi` of ONE panel per step (waves 4-7: panel A = columns I; waves 0-3: panel B = columns J)
    const int mypanel = grp == 1 ? 0 : 1;
    const int kr = X2 ? (wi & 1) : wi;                            // 8-pair group of the step this wave converts
    const bool midrow = X2 && wi >= 2;                            // bf16x2: packet rows 2-3 hold mid = bf16(v - hi)
    const int col0 = (mypanel == 0 ? I : J) * 256;
    float r2[4] = {0.f, 0.f, 0.f, 0.f};
    if (WHICH == 0 && P.shift) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r2[q] = 2.f * P.shift[col0 + 4 * lane + q];
    }
    // feature 4 c + q of a 32-block sits at position 8 q + c: lane l = 8 block + c writes packet q at (32 block + 8 q + c) 16
    const unsigned wr_off = (unsigned)(mypanel * 16384 + wi * 4096 + ((lane >> 3) * 32 + (lane & 7)) * 16);
    const unsigned long long krb = (unsigned long long)(kr * 8) * (unsigned long long)P.row_bytes;   // this wave's first pair row inside a step

    // ---- raw rows: global -> LDS directly (global_load_lds, 16 bytes per lane: lanes 0-31 one 512-byte panel row, lanes 32-63 the
    //      next) into the wave's PRIVATE staging area, read back as ds_read_b64 one step later.  Only the issuing wave reads its
    //      staging area: its own vmcnt(0) orders the data, no barrier is involved.  Measured (scripts/micro/img_fused.hip,
    //      profiles/r05_img_fused_micro.txt): loads that RETURN TO REGISTERS cost the CU's texture path ~24 (8-byte) / ~28
    //      (16-byte) cycles per wave-instruction, serialised per CU -- 128 / 64 of them per step made the step 3.4x / 2.4x its
    //      MFMA time whether the input was cache-resident or not; LDS-direct pieces are cheaper and need no registers.
    unsigned voffD[4];   // piece i of an operand: rows 2 i (lanes 0-31) and 2 i + 1 (lanes 32-63) of the wave's group
#pragma unroll
    for (int i = 0; i < 4; ++i)   // (- 1024 i: the instruction's immediate offset, which places piece i in LDS, advances the GLOBAL address too)
        voffD[i] = (unsigned)(col0 * 2 + (lane & 31) * 16) + (unsigned)(2 * i + (lane >> 5)) * (unsigned)P.row_bytes - (unsigned)(1024 * i);
    int nv_raw = PS;                       // valid pairs of the STAGED step (set when its pieces were issued)
    int nv_next = PS;                      // ... of the step being issued
    unsigned long long base_a = 0;         // its x_t rows (scalar)
    auto dma_record = [&](int step) {      // the step's record through the CONSTANT address space: uniform address -> s_load
        const int sc = step < s1 ? step : s1 - 1;                 // beyond the share: a harmless re-load, converted into a dead slot
        typedef const __attribute__((address_space(4))) ImgStep* step_cptr;
        const step_cptr dp = (step_cptr)(uintptr_t)(P.steps + sc);
        base_a = (unsigned long long)(uintptr_t)dp->rowa;
        nv_next = dp->nvalid;
    };
#define IMG_FU_PIECE(I_)                                                                                                       \
    do {                                                                                                                       \
        if (!(ABL & 1)) {                                                                                                      \
            unsigned off_ = voffD[I_];                                                                                         \
            asm volatile("" : "+v"(off_)); /* keeps the zero-extension in this block: SGPR base + 32-bit VGPR offset form */    \
            const unsigned long long ba_ = base_a + krb, bb_ = ba_ + (unsigned long long)P.lag_bytes;                          \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((global_ptr<char>)ba_ + off_),    \
                                             (__attribute__((address_space(3))) void*)rawbuf, 16, (I_) * 1024, 0);             \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((global_ptr<char>)bb_ + off_),    \
                                             (__attribute__((address_space(3))) void*)(rawbuf + 4096), 16, (I_) * 1024, 0);    \
        }                                                                                                                      \
    } while (0)
    auto dma_generic = [&]() {             // any step: a padding pair re-reads the step's last valid pair (its packet entries are zeroed)
        if (ABL & 1) return;
        const int last = nv_next > 0 ? nv_next - 1 : 0;   // (bf16x2: the second 16-pair step of a padded 32-pair step may hold no pair at all)
        const unsigned coloff = (unsigned)(col0 * 2 + (lane & 31) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = kr * 8 + 2 * i + (lane >> 5);
            const int pe = p < last ? p : last;
            const global_ptr<char> ga = (global_ptr<char>)(base_a + (unsigned long long)(unsigned)pe * (unsigned long long)P.row_bytes) + coloff;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)(rawbuf + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (size_t)P.lag_bytes),
                                             (__attribute__((address_space(3))) void*)(rawbuf + 4096 + i * 1024), 16, 0, 0);
        }
    };
    img_u32x2 ra[8], rb[8];
    auto read_staged = [&]() {   // the wave's staged rows -> registers (ds_read_b64 of 512-byte rows: conflict-free)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of one step ago (the loop has no other VMEM)
        if (ABL & 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ra[e] = img_u32x2{0x3f803f80u, 0x3f803f80u}; rb[e] = img_u32x2{0x3f003f00u, 0x3f003f00u}; }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ra[e] = *reinterpret_cast<const img_u32x2*>(rawbuf + e * 512 + lane * 8);
            rb[e] = *reinterpret_cast<const img_u32x2*>(rawbuf + 4096 + e * 512 + lane * 8);
        }
    };
    // packet q of the lane's four features: 8 pairs x (unpack a, unpack b, u or d, round) -> one ds_write_b128
    auto convert_q = [&](char* dst, int q, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (ABL & 16) return;
        bf16x8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned wa = q < 2 ? ra[e].x : ra[e].y, wb = q < 2 ? rb[e].x : rb[e].y;
            const float a = (q & 1) ? __uint_as_float(wa & 0xffff0000u) : __uint_as_float(wa << 16);
            const float b = (q & 1) ? __uint_as_float(wb & 0xffff0000u) : __uint_as_float(wb << 16);
            float v = WHICH == 0 ? (a + b) - r2[q] : a - b;
            if (TAIL && !(kr * 8 + e < nv_raw)) v = 0.f;   // (uniform) a padding pair: the zero packet entries the image held
            const __bf16 hi = (__bf16)v;
            pk[e] = midrow ? (__bf16)(v - (float)hi) : hi;
        }
        *reinterpret_cast<bf16x8*>(dst + q * 128) = pk;
    };
#define IMG_FU_LDS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    // One load phase's raw-row work: the staged step (valid pairs nv_raw) -> packets of ring slot `slot`; the pieces of `next`
    // issued BETWEEN the packets, so that the texture path works while the VALU converts (issued in one burst at the end of
    // the phase -- the first version -- the eight pieces sat on the phase's critical path: 14.4 ms per 1M x 2048).
    auto convert_and_issue = [&](int slot, int next) {
        read_staged();
        dma_record(next);
        IMG_FU_LDS_DONE();                             // the staged rows are in registers (and the record in SGPRs) before new pieces land
        __builtin_amdgcn_sched_barrier(0);
        char* dst = ring + (unsigned)slot * IMG_SLOT + wr_off;
        if (__builtin_expect(nv_raw >= PS && nv_next >= PS, 1)) {   // (uniform) every pair of both steps exists
            IMG_FU_PIECE(0);
            __builtin_amdgcn_sched_barrier(0);
            convert_q(dst, 0, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            IMG_FU_PIECE(1);
            __builtin_amdgcn_sched_barrier(0);
            convert_q(dst, 1, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            IMG_FU_PIECE(2);
            __builtin_amdgcn_sched_barrier(0);
            convert_q(dst, 2, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            IMG_FU_PIECE(3);
            __builtin_amdgcn_sched_barrier(0);
            convert_q(dst, 3, std::false_type{});
        } else {                                       // a trajectory's last step on either side
            dma_generic();
#pragma unroll
            for (int q = 0; q < 4; ++q) convert_q(dst, q, std::true_type{});
            asm volatile("" ::: "memory");             // keeps the two bodies apart (if-converted, every element pays a v_cndmask)
        }
        nv_raw = nv_next;
    };
    // ---- fragments and MFMAs: tica_img_pp_kernel's, on positions.  Only fragment set 0 is read in the load phase; set 1 is read
    //      by the multiplying wave itself, ahead of its first MFMA (the first eight MFMAs, 256 cycles, cover the reads): 24
    //      registers fewer live beside the raw rows and their packets -- without it the kernel spilled.
    const unsigned fragA = (unsigned)((kl * 256 + wr * 64 + cl) * 16);
    const unsigned fragB = (unsigned)(16384 + (kl * 256 + wc * 128 + cl) * 16);
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    auto frags_set = [&](int slot, int set) {
        if (ABL & 2) return;
        const char* base = ring + (unsigned)slot * IMG_SLOT;
        constexpr int k0 = X2 ? 2 : 0, k1 = X2 ? 0 : 2;
        if (set == 0) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) fa0[bi] = *reinterpret_cast<const bf16x8*>(base + fragA + (k0 * 256 + bi * 32) * 16);
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) fb0[bj] = *reinterpret_cast<const bf16x8*>(base + fragB + (k0 * 256 + bj * 32) * 16);
        } else {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) fa1[bi] = *reinterpret_cast<const bf16x8*>(base + fragA + (k1 * 256 + bi * 32) * 16);
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) fb1[bj] = *reinterpret_cast<const bf16x8*>(base + fragB + (k1 * 256 + bj * 32) * 16);
        }
    };
    auto mfmas = [&](int slot) {
        frags_set(slot, 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
        if (X2) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj) {
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
                }
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
                acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define IMG_FU_BARRIER() do { if (!(ABL & 4)) __builtin_amdgcn_s_barrier(); } while (0)

    // ---- prologue: every wave stages and converts its rows of steps s0, s0 + 1 into slots 0, 1 and stages those of s0 + 2
    dma_record(s0);
    dma_generic();
    nv_raw = nv_next;
    convert_and_issue(0, s0 + 1);
    convert_and_issue(1, s0 + 2);
    IMG_FU_LDS_DONE();
    IMG_FU_BARRIER();

    int slot = 0, slot_ld = D;
    int steps_acc = 0;
    if (grp == 0) {
        frags_set(0, 0);
        for (int s = s0; s < s1; ++s) {
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            mfmas(slot);                               // phase 2n
            IMG_FU_BARRIER();
            if (++steps_acc >= P.kflush_steps || s + 1 == s1) {   // (the merge BEFORE the next fragments are read: fewer registers live across it)
                steps_acc = 0;
                img_flush_perm(acc, P.slabs, P.ntiles_sym, P.T, cohort, WHICH, I, J, wr, wc, kl, cl);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < s1) frags_set(slot1, 0);       // phase 2n + 1
            __builtin_amdgcn_sched_barrier(0);
            convert_and_issue(slot_ld, s + D + 1);     //   panel B of step s + 2 (staged one step ago); stage step s + 3
            IMG_FU_LDS_DONE();
            IMG_FU_BARRIER();
            slot = slot1;
            slot_ld = slot_ld + 1 == NS ? 0 : slot_ld + 1;
        }
    } else {
        for (int s = s0; s < s1; ++s) {
            frags_set(slot, 0);                        // phase 2n
            __builtin_amdgcn_sched_barrier(0);
            convert_and_issue(slot_ld, s + D + 1);     //   panel A of step s + 2
            IMG_FU_LDS_DONE();
            IMG_FU_BARRIER();
            mfmas(slot);                               // phase 2n + 1
            IMG_FU_BARRIER();
            slot = slot + 1 == NS ? 0 : slot + 1;
            slot_ld = slot_ld + 1 == NS ? 0 : slot_ld + 1;
            if (++steps_acc >= P.kflush_steps || s + 1 == s1) {
                steps_acc = 0;
                img_flush_perm(acc, P.slabs, P.ntiles_sym, P.T, cohort, WHICH, I, J, wr, wc, kl, cl);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail pieces: nothing may still be writing LDS ...
    IMG_FU_BARRIER();                                  // ... or reading the ring when the next segment's prologue refills it (or at exit)
#undef IMG_FU_BARRIER
#undef IMG_FU_LDS_DONE
#undef IMG_FU_PIECE
}

template <bool X2, int ABL>
__device__ __forceinline__ void img_fused_unit(const ImgFusedArgs& P, char* smem, int unit, int cohort, int s0, int s1)
{
    int which, I, J;
    img_decode_unit(unit, P.T2, which, I, J);
    if (which == 0) img_fused_segment<X2, 0, ABL>(P, smem, I, J, cohort, s0, s1);
    else img_fused_segment<X2, 1, ABL>(P, smem, I, J, cohort, s0, s1);
}

// work split: tica_img_pp_kernel's (whole cohorts + a remainder cohort)
template <bool X2, int ABL = 0>   // ABL (micro-benchmark): 1 no global loads, 2 no fragment reads, 4 no barriers, 16 no conversion
__global__ __launch_bounds__(IMG_NT, 1) void tica_img_fused_kernel(ImgFusedArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [8 waves][x_t | x_{t+tau}][8 rows][512 bytes], then [3][A, B][4][256] packets
    const int U = P.ntile2, G = (int)gridDim.x;
    const int S = G / U, R = G - S * U;
    const int p = img_xcd_linear_id();
    if (p < S * U) {
        const int cohort = p / U;
        const int s1 = (R == 0 && cohort == S - 1) ? P.nsteps : (cohort + 1) * P.main_steps;
        img_fused_unit<X2, ABL>(P, smem, p - cohort * U, cohort, cohort * P.main_steps, s1);
    } else {
        const int r = p - S * U;
        for (int unit = r; unit < U; unit += R) img_fused_unit<X2, ABL>(P, smem, unit, S, S * P.main_steps, P.nsteps);
    }
}
constexpr size_t IMG_FUSED_LDS = 8 * 8192 + (size_t)3 * IMG_SLOT;   // 64 KiB of raw staging + 96 KiB of packet ring = all 160 KiB of the CU

}  // namespace msm
