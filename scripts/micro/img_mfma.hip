// img_mfma.hip -- the bf16 image path's MFMA pass alone: round 3's lockstep kernel against the ping-pong kernel(s) of
// tica_img_dev.h on the same synthetic image, slabs compared bit for bit, times interleaved on one box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I msmbuilder_amd/csrc scripts/micro/img_mfma.hip -o scripts/micro/img_mfma
//   scripts/micro/img_mfma [F=2048] [pairs=1048576] [reps=5] [kflush_pairs=8192]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tica_img_dev.h"

namespace msm {
// ---------------------------------------------------------------------------------------------------------------------
// Round 3's kernel (register-staged packets, all eight waves in lockstep), kept ONLY as the A/B baseline of
// scripts/micro/img_mfma.hip; the product launches tica_img_pp_kernel.
constexpr int IMG_SLOTS = 3;                      // LDS ring: K-steps s (being multiplied), s + 1 (complete), s + 2 (being written)
constexpr size_t IMG_LOCKSTEP_LDS = (size_t)IMG_SLOTS * IMG_SLOT;

// Round 3: fragments PREFETCHED ACROSS THE BARRIER.  With two LDS buffers every K-step began, for all eight waves at once,
// with its fragment reads behind the barrier (and the second k-half's reads behind the first half's MFMAs): the matrix
// pipe idled for two LDS round trips per step (MFMA busy 0.43, 2,500 cycles per step against 1,024 of MFMA work).  With a
// ring of three slots step s + 1 is complete in LDS while step s is multiplied, so a wave reads the FIRST fragment set of
// step s + 1 during step s and starts its MFMAs right behind the barrier; the second set is read at the top of the step,
// under those MFMAs.  The accumulation order per accumulator is unchanged (bit-identical sums).
template <bool X2, bool WRAP = false>
__global__ __launch_bounds__(IMG_NT, 1) void tica_img_lockstep_kernel(ImgMfmaArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8* L = reinterpret_cast<bf16x8*>(smem);   // [3][2][4][256]
    constexpr int PAN = 4 * 256;                   // packets per panel
    const int tid = threadIdx.x;
    const int p = img_xcd_linear_id();
    const int cohort = p / P.ntile2, tile = p % P.ntile2;
    const int which = tile & 1;                    // 0: H = sum u u^T, 1: D = sum d d^T
    int I = 0, uix = tile >> 1;
    while (uix >= P.T2 - I) {
        uix -= P.T2 - I;
        ++I;
    }
    const int J = I + uix;
    const bf16x8* hi = which ? P.d_hi : P.u_hi;
    const bf16x8* mid = which ? P.d_mid : P.u_mid;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int kl = lane >> 5, cl = lane & 31;

    img_f32x16 acc[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;

    // this cohort's contiguous share of the K-steps
    const long long s0 = P.nsteps * cohort / P.S, s1 = P.nsteps * (cohort + 1) / P.S;
    // staging: panel = 4 packet rows x 256 features; thread -> packets tid and tid + 512 of A and of B
    //   bf16  : packet rows = pair groups 4 s .. 4 s + 3 of the hi image
    //   bf16x2: rows 0-1 = groups 2 s, 2 s + 1 of the hi image, rows 2-3 = the same groups of the mid image
    const int c0 = tid & 255, q0 = tid >> 8;  // q0 in {0, 1}: packet rows q0 and q0 + 2
    img_f32x4 ra[2], rb[2], na[2], nb[2];
    // (steps beyond the share are clamped to its last one: never out of the image, loaded and stored but not multiplied)
#define MSM_IMG_LOAD(RA, RB, S_)                                                                   \
    {                                                                                              \
        long long sc_ = (S_) < s1 ? (S_) : s1 - 1;                                                 \
        if (WRAP) sc_ %= P.wrap;                                                                   \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                            \
            const int row = q0 + 2 * h;                                                            \
            const bf16x8* src = (X2 && row >= 2) ? mid : hi;                                       \
            const long long g = X2 ? sc_ * 2 + (row & 1) : sc_ * 4 + row;                          \
            const global_ptr<img_f32x4> base = as_global<img_f32x4>(src + (size_t)g * (size_t)P.Fp); \
            RA[h] = base[I * 256 + c0];                                                            \
            RB[h] = base[J * 256 + c0];                                                            \
        }                                                                                          \
    }
#define MSM_IMG_STORE(RA, RB, SLOT)                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                            \
            const int row = q0 + 2 * h;                                                            \
            *reinterpret_cast<img_f32x4*>(L + (SLOT) * 2 * PAN + row * 256 + c0) = RA[h];          \
            *reinterpret_cast<img_f32x4*>(L + (SLOT) * 2 * PAN + PAN + row * 256 + c0) = RB[h];    \
        }                                                                                          \
    }
    // fragment sets of a K-step in slot SLOT: set 0 is what the step's first MFMAs need (bf16: the k-half of pairs 0-15;
    // bf16x2: the mid images), set 1 the rest (pairs 16-31; the hi images)
#define MSM_IMG_FRAGS(FA, FB, SLOT, SET)                                                           \
    {                                                                                              \
        const bf16x8* Ah_ = L + (SLOT) * 2 * PAN;                                                  \
        const bf16x8* Bh_ = Ah_ + PAN;                                                             \
        const int kg_ = X2 ? ((SET) == 0 ? 2 + kl : kl) : 2 * (SET) + kl;                          \
        _Pragma("unroll") for (int bi = 0; bi < 2; ++bi) FA[bi] = Ah_[kg_ * 256 + wr * 64 + bi * 32 + cl];   \
        _Pragma("unroll") for (int bj = 0; bj < 4; ++bj) FB[bj] = Bh_[kg_ * 256 + wc * 128 + bj * 32 + cl];  \
    }
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    if (s1 > s0) {
        MSM_IMG_LOAD(ra, rb, s0)
        MSM_IMG_STORE(ra, rb, 0)
        MSM_IMG_LOAD(ra, rb, s0 + 1)
        MSM_IMG_STORE(ra, rb, 1)
        MSM_IMG_LOAD(ra, rb, s0 + 2)
    }
    __syncthreads();
    if (s1 > s0) MSM_IMG_FRAGS(fa0, fb0, 0, 0)
    int steps_acc = 0;
    int slot = 0;   // slot of step s; s + 1 -> slot + 1, s + 2 -> slot + 2 (mod 3)
    for (long long s = s0; s < s1; ++s) {
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        MSM_IMG_LOAD(na, nb, s + 3)            // K-step s + 3 -> the other register set
        MSM_IMG_FRAGS(fa1, fb1, slot, 1)       // this step's second fragment set: lands under the first set's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (!X2) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 na0[2], nb0[4];
            MSM_IMG_FRAGS(na0, nb0, slot1, 0)  // the NEXT step's first set (its slot has been complete since the last barrier)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) fa0[bi] = na0[bi];
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) fb0[bj] = nb0[bj];
        } else {
            // set 0 = (am, bm), set 1 = (ah, bh); per accumulator the products come in the order mm, hm, mh, hh as before
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj) {
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb0[bj], acc[bi][bj], 0, 0, 0);
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 na0[2], nb0[4];
            MSM_IMG_FRAGS(na0, nb0, slot1, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj)
                    acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[bi], fb1[bj], acc[bi][bj], 0, 0, 0);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) fa0[bi] = na0[bi];
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) fb0[bj] = nb0[bj];
        }
        MSM_IMG_STORE(ra, rb, slot2)           // K-step s + 2 (loaded one step ago) -> the free slot
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ra[h] = na[h];
            rb[h] = nb[h];
        }
        slot = slot1;
        // fp64 merge into the private slabs of the four 128 x 128 sub-tiles (upper ones only)
        if (++steps_acc >= P.kflush_steps || s + 1 == s1) {
            steps_acc = 0;
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
    }
#undef MSM_IMG_LOAD
#undef MSM_IMG_STORE
#undef MSM_IMG_FRAGS
}


}  // namespace msm

using namespace msm;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        // two bf16 in [-2, 2): sign, exponent 0x3f / 0x3e..., 7 mantissa bits each -- full-range signs (DVFS-honest data)
        const unsigned lo = (x & 0x807fu) | 0x3f00u | ((x >> 3) & 0x0080u);
        const unsigned hi = ((x >> 16) & 0x807fu) | 0x3f00u | ((x >> 19) & 0x0080u);
        p[i] = lo | (hi << 16);
    }
}

struct Variant {
    const char* name;
    void (*launch)(const ImgMfmaArgs&, unsigned grid, hipStream_t);
};

template <bool X2> static void launch_lockstep(const ImgMfmaArgs& a, unsigned g, hipStream_t st)
{
    if (a.wrap) hipLaunchKernelGGL((tica_img_lockstep_kernel<X2, true>), dim3(g), dim3(IMG_NT), IMG_LOCKSTEP_LDS, st, a);
    else hipLaunchKernelGGL((tica_img_lockstep_kernel<X2, false>), dim3(g), dim3(IMG_NT), IMG_LOCKSTEP_LDS, st, a);
}
template <bool X2, int LAG> static void launch_pp(const ImgMfmaArgs& a, unsigned g, hipStream_t st)
{
    if (a.wrap) hipLaunchKernelGGL((tica_img_pp_kernel<X2, LAG, true>), dim3(g), dim3(IMG_NT), (size_t)(3 + LAG) * IMG_SLOT, st, a);
    else hipLaunchKernelGGL((tica_img_pp_kernel<X2, LAG, false>), dim3(g), dim3(IMG_NT), (size_t)(3 + LAG) * IMG_SLOT, st, a);
}

template <int ABL> static void launch_abl(const ImgMfmaArgs& a, unsigned g, hipStream_t st)
{
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_pp_kernel<false, 0, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * IMG_SLOT);
    }
    hipLaunchKernelGGL((tica_img_pp_kernel<false, 0, false, ABL>), dim3(g), dim3(IMG_NT), (size_t)3 * IMG_SLOT, st, a);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int F = argc > 1 ? atoi(argv[1]) : 2048;
    const long long pairs = argc > 2 ? atoll(argv[2]) : 1048576;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int kflush_pairs = argc > 4 ? atoi(argv[4]) : 8192;
    const int vmask = argc > 5 ? atoi(argv[5]) : 7;   // bit v: run variant v (0 = lockstep, the reference)
    const long long wrap = argc > 6 ? atoll(argv[6]) : 0;   // > 0: every kernel reads K-step s from step s % wrap (cache-resident image)
    const int ppgrid = argc > 7 ? atoi(argv[7]) : 0;       // ping-pong kernels: 0 = whole cohorts only (bit-exact check), n = n workgroups (remainder cohort)
    const int T2 = (F + 255) / 256, Fp = T2 * 256, T = (F + 127) / 128;
    const int ntile2 = T2 * (T2 + 1), ntiles_sym = T * (T + 1) / 2;
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int S = cus / ntile2 > 0 ? cus / ntile2 : 1;
    printf("F=%d Fp=%d pairs=%lld  CUs=%d units=%d cohorts=%d grid=%d  kflush=%d pairs\n", F, Fp, pairs, cus, ntile2, S, S * ntile2, kflush_pairs);
#define SETLDS(K, B) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B)))
    SETLDS((tica_img_lockstep_kernel<false, false>), IMG_LOCKSTEP_LDS);
    SETLDS((tica_img_lockstep_kernel<true, false>), IMG_LOCKSTEP_LDS);
    SETLDS((tica_img_lockstep_kernel<false, true>), IMG_LOCKSTEP_LDS);
    SETLDS((tica_img_lockstep_kernel<true, true>), IMG_LOCKSTEP_LDS);
    SETLDS((tica_img_pp_kernel<false, 0, false>), 3 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<false, 1, false>), 4 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 0, false>), 3 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 1, false>), 4 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<false, 0, true>), 3 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<false, 1, true>), 4 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 0, true>), 3 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 1, true>), 4 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<false, 2, false>), 5 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 2, false>), 5 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<false, 2, true>), 5 * IMG_SLOT);
    SETLDS((tica_img_pp_kernel<true, 2, true>), 5 * IMG_SLOT);

    const long long groups = (pairs + 31) / 32 * 4;
    const size_t one = (size_t)groups * Fp * 16;
    char* img = nullptr;
    CK(hipMalloc(&img, 4 * one));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned*>(img), 4 * one / 4, 12345u);
    CK(hipDeviceSynchronize());
    printf("image filled (%zu MB)\n", 4 * one >> 20);
    const size_t slab_n = (size_t)(S + 1) * ntiles_sym * 2 * IMG_TM * IMG_TM;   // (+ 1: the remainder cohort's row)
    double *slabs = nullptr, *ref = nullptr;
    CK(hipMalloc(&slabs, slab_n * sizeof(double)));
    CK(hipMalloc(&ref, slab_n * sizeof(double)));
    std::vector<double> hs(slab_n), hr(slab_n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    for (int x2 = 0; x2 < 2; ++x2) {
        ImgMfmaArgs a;
        memset(&a, 0, sizeof(a));
        a.u_hi = reinterpret_cast<bf16x8*>(img);
        a.d_hi = reinterpret_cast<bf16x8*>(img + one);
        a.u_mid = reinterpret_cast<bf16x8*>(img + 2 * one);
        a.d_mid = reinterpret_cast<bf16x8*>(img + 3 * one);
        a.nsteps = x2 ? groups / 2 : groups / 4;
        a.Fp = Fp; a.T = T; a.T2 = T2; a.ntiles_sym = ntiles_sym; a.ntile2 = ntile2; a.S = S;
        a.wrap = wrap;
        a.kflush_steps = kflush_pairs / (x2 ? 16 : 32);
        if (a.kflush_steps < 1) a.kflush_steps = 1;
        std::vector<Variant> vs;
        if (x2) {
            vs.push_back({"lockstep<x2>", launch_lockstep<true>});
            vs.push_back({"pingpong<x2,lag0>", launch_pp<true, 0>});
            vs.push_back({"pingpong<x2,lag1>", launch_pp<true, 1>});
            if (vmask & 0x200) vs.push_back({"pingpong<x2,lag2>", launch_pp<true, 2>});
        } else {
            vs.push_back({"lockstep", launch_lockstep<false>});
            vs.push_back({"pingpong<lag0>", launch_pp<false, 0>});
            vs.push_back({"pingpong<lag1>", launch_pp<false, 1>});
            if (vmask & 0x200) vs.push_back({"pingpong<lag2>", launch_pp<false, 2>});
            if (vmask & 0x100) {   // ablations of pingpong<lag0> (results are garbage by construction, except "no priority")
                vs.push_back({"pp lag0, no priority", launch_abl<8>});
                vs.push_back({"abl: no loads", launch_abl<1>});
                vs.push_back({"abl: no frag reads", launch_abl<2>});
                vs.push_back({"abl: no loads/frags", launch_abl<3>});
                vs.push_back({"abl: MFMA only", launch_abl<7>});
            }
        }
        const unsigned grid0 = (unsigned)(S * ntile2);
        // executed flop: every unit multiplies a full 256 x 256 tile per pair (bf16x2: four products)
        const double flop = (double)ntile2 * 2.0 * 256 * 256 * (double)(x2 ? a.nsteps * 16 : a.nsteps * 32) * (x2 ? 4 : 1);
        std::vector<std::vector<float>> ms(vs.size());
        for (size_t v = 0; v < vs.size(); ++v) {   // correctness: slabs vs the lockstep kernel's, bit for bit
            if (v < 3 && !((vmask >> v) & 1)) continue;
            printf("  running %s ...\n", vs[v].name);
            double* out = v == 0 ? ref : slabs;
            CK(hipMemset(out, 0, slab_n * sizeof(double)));
            a.slabs = out;
            const unsigned grid = (v > 0 && ppgrid > 0) ? (unsigned)ppgrid : grid0;
            a.main_steps = img_main_steps(a.nsteps, (int)grid, ntile2);
            vs[v].launch(a, grid, 0);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            if (v > 0) {
                CK(hipMemcpy(hs.data(), slabs, slab_n * sizeof(double), hipMemcpyDeviceToHost));
                CK(hipMemcpy(hr.data(), ref, slab_n * sizeof(double), hipMemcpyDeviceToHost));
                size_t bad = 0, nz = 0;
                double worst = 0, big = 0;
                const size_t row = (size_t)ntiles_sym * 2 * IMG_TM * IMG_TM;
                for (size_t i = 0; i < row; ++i) {   // cohort rows summed: the split of the K-steps over rows may differ
                    double x = 0, y = 0;
                    bool same = true;
                    for (int c = 0; c <= S; ++c) {
                        x += hs[c * row + i];
                        y += hr[c * row + i];
                        same = same && hs[c * row + i] == hr[c * row + i];
                    }
                    if (y != 0.0) ++nz;
                    if ((y < 0 ? -y : y) > big) big = y < 0 ? -y : y;
                    if (!same) ++bad;
                    const double d = x - y;
                    if ((d < 0 ? -d : d) > worst) worst = d < 0 ? -d : d;
                }
                printf("  check %-20s: %zu of %zu slab words differ bit for bit; summed over cohorts worst |diff| %.3g of max %.3g (non-zero %zu)\n", vs[v].name, bad, row, worst, big, nz);
            }
        }
        a.slabs = slabs;
        for (int r = 0; r < reps; ++r)
            for (size_t v = 0; v < vs.size(); ++v) {   // interleaved rounds
                if (v < 3 && !((vmask >> v) & 1)) continue;
                const unsigned grid = (v > 0 && ppgrid > 0) ? (unsigned)ppgrid : grid0;
                a.main_steps = img_main_steps(a.nsteps, (int)grid, ntile2);
                CK(hipEventRecord(e0, 0));
                vs[v].launch(a, grid, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t = 0;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms[v].push_back(t);
            }
        for (size_t v = 0; v < vs.size(); ++v) {
            if (v < 3 && !((vmask >> v) & 1)) continue;
            float mn = 1e30f, sum = 0;
            for (float t : ms[v]) { mn = t < mn ? t : mn; sum += t; }
            printf("  %-20s min %.3f ms  mean %.3f ms   %.0f TF executed = %.3f of 2.5 PF  (%.1fM pairs/s)\n", vs[v].name, mn, sum / ms[v].size(),
                   flop / (mn * 1e-3) / 1e12, flop / (mn * 1e-3) / 2.5e15, (double)(x2 ? a.nsteps * 16 : a.nsteps * 32) / (mn * 1e-3) / 1e6);
        }
    }
    return 0;
}
