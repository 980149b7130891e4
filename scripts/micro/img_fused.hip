// img_fused.hip -- round 5: the FUSED bf16 kernel (tica_img_dev.h, tica_img_fused_kernel: raw bfloat16 rows -> packets in
// the load role, no image) against the two-kernel path (a plain packing kernel + tica_img_pp_kernel) on the same synthetic
// bfloat16 trajectories: slabs compared bit for bit, times interleaved on one box, ablations of the fused kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I msmbuilder_amd/csrc scripts/micro/img_fused.hip -o scripts/micro/img_fused
//   scripts/micro/img_fused [F=2048] [n_traj=100] [traj_len=10000] [lag=100] [reps=5] [ablations=1]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tica_img_dev.h"

using namespace msm;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

// bfloat16 pairs in [-2, 2) with full-range signs and mantissas
__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        const unsigned lo = (x & 0x807fu) | 0x3f00u | ((x >> 3) & 0x0080u);
        const unsigned hi = ((x >> 16) & 0x807fu) | 0x3f00u | ((x >> 19) & 0x0080u);
        p[i] = lo | (hi << 16);
    }
}

// the reference packer: one thread per (group, feature) packet, the arithmetic of tica_img_kernel on bfloat16 rows
// (u = (a + b) - 2 r, d = a - b in fp32, RNE to bf16; mid = bf16(v - hi)); padding pairs are zero
struct PackArgs {
    const ImgStep* steps32;   // 32-pair steps
    long long nsteps32;
    long long row_bytes, lag_bytes;
    const float* shift;
    int F;
    bf16x8 *u_hi, *d_hi, *u_mid, *d_mid;
};
__global__ void ref_pack_kernel(PackArgs P)
{
    const long long g = blockIdx.x;          // 8-pair group
    const long long st = g >> 2;
    const int kr = (int)(g & 3);
    const ImgStep d = P.steps32[st];
    for (int f = threadIdx.x; f < P.F; f += blockDim.x) {
        bf16x8 uh, dh, um, dm;
        const float r2 = 2.f * P.shift[f];
        for (int e = 0; e < 8; ++e) {
            const int p = kr * 8 + e;
            float u = 0.f, dd = 0.f;
            if (p < d.nvalid) {
                const char* ra = (const char*)d.rowa + (size_t)p * P.row_bytes;
                const float a = (float)*(const __bf16*)(ra + (size_t)f * 2);
                const float b = (float)*(const __bf16*)(ra + P.lag_bytes + (size_t)f * 2);
                u = (a + b) - r2;
                dd = a - b;
            }
            const __bf16 u1 = (__bf16)u, d1 = (__bf16)dd;
            uh[e] = u1; dh[e] = d1;
            um[e] = (__bf16)(u - (float)u1);
            dm[e] = (__bf16)(dd - (float)d1);
        }
        const size_t o = (size_t)g * P.F + f;
        P.u_hi[o] = uh; P.d_hi[o] = dh; P.u_mid[o] = um; P.d_mid[o] = dm;
    }
}

template <bool X2, int ABL> static void launch_fused(const ImgFusedArgs& a, unsigned g, hipStream_t st)
{
    static bool once = false;
    if (!once) {
        once = true;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_fused_kernel<X2, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_FUSED_LDS));
    }
    hipLaunchKernelGGL((tica_img_fused_kernel<X2, ABL>), dim3(g), dim3(IMG_NT), IMG_FUSED_LDS, st, a);
}
template <bool X2> static void launch_pp(const ImgMfmaArgs& a, unsigned g, hipStream_t st)
{
    static bool once = false;
    if (!once) {
        once = true;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tica_img_pp_kernel<X2, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * IMG_SLOT));
    }
    hipLaunchKernelGGL((tica_img_pp_kernel<X2, 1, false>), dim3(g), dim3(IMG_NT), (size_t)4 * IMG_SLOT, st, a);
}

struct FusedVariant {
    const char* name;
    void (*launch)(const ImgFusedArgs&, unsigned, hipStream_t);
    bool exact;
};

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int F = argc > 1 ? atoi(argv[1]) : 2048;
    const int n_traj = argc > 2 ? atoi(argv[2]) : 100;
    const long long L = argc > 3 ? atoll(argv[3]) : 10000;
    const int lag = argc > 4 ? atoi(argv[4]) : 100;
    const int reps = argc > 5 ? atoi(argv[5]) : 5;
    const int abl = argc > 6 ? atoi(argv[6]) : 1;
    const int wrap = argc > 7 ? atoi(argv[7]) : 0;    // > 0: the FUSED kernels' step s reads the rows of step s % wrap (cache-resident input; results differ)
    if (F % 256 || L <= lag) { fprintf(stderr, "F must be a multiple of 256, traj_len > lag\n"); return 2; }
    const int T2 = F / 256, T = F / 128;
    const int ntile2 = T2 * (T2 + 1), ntiles_sym = T * (T + 1) / 2;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int grid = cus > ntile2 ? cus : ntile2;
    const int S = grid / ntile2;
    // trajectories: ragged on purpose (every third one 13 rows shorter, so its last step has padding pairs)
    std::vector<long long> len(n_traj), row0(n_traj);
    long long rows = 0;
    for (int t = 0; t < n_traj; ++t) { len[t] = L - (t % 3 == 1 ? 13 : 0); row0[t] = rows; rows += len[t]; }
    __bf16* X = nullptr;
    CK(hipMalloc(&X, (size_t)rows * F * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned*>(X), (size_t)rows * F / 2, 4242u);
    std::vector<float> hshift(F);
    for (int f = 0; f < F; ++f) hshift[f] = 0.37f * (float)((f * 37) % 11 - 5) / 5.f;
    float* shift = nullptr;
    CK(hipMalloc(&shift, F * sizeof(float)));
    CK(hipMemcpy(shift, hshift.data(), F * sizeof(float), hipMemcpyHostToDevice));
    // step tables: 32-pair steps (bf16) and 16-pair steps (bf16x2); a trajectory is padded to whole 32-pair steps in both
    std::vector<ImgStep> s32, s16;
    long long pairs = 0;
    for (int t = 0; t < n_traj; ++t) {
        const long long nv = len[t] - lag;
        pairs += nv;
        for (long long j = 0; j * 32 < nv; ++j) {
            const char* base = (const char*)X + (size_t)(row0[t] + j * 32) * F * 2;
            const int n = (int)(nv - j * 32 < 32 ? nv - j * 32 : 32);
            s32.push_back({base, n, 0});
            s16.push_back({base, n < 16 ? n : 16, 0});
            if (n > 16) s16.push_back({base + (size_t)16 * F * 2, n - 16, 0});
            else s16.push_back({base, 0, 0});   // a whole 16-pair step of padding (nvalid 0): any addressable row, every packet zero
        }
    }
    std::vector<ImgStep> w32 = s32, w16 = s16;
    if (wrap > 0) {
        for (size_t i = 0; i < w32.size(); ++i) w32[i] = s32[i % wrap];
        for (size_t i = 0; i < w16.size(); ++i) w16[i] = s16[i % (2 * wrap)];
    }
    ImgStep *d32 = nullptr, *d16 = nullptr, *dw32 = nullptr, *dw16 = nullptr;
    CK(hipMalloc(&dw32, s32.size() * sizeof(ImgStep)));
    CK(hipMalloc(&dw16, s16.size() * sizeof(ImgStep)));
    CK(hipMemcpy(dw32, w32.data(), s32.size() * sizeof(ImgStep), hipMemcpyHostToDevice));
    CK(hipMemcpy(dw16, w16.data(), s16.size() * sizeof(ImgStep), hipMemcpyHostToDevice));
    CK(hipMalloc(&d32, s32.size() * sizeof(ImgStep)));
    CK(hipMalloc(&d16, s16.size() * sizeof(ImgStep)));
    CK(hipMemcpy(d32, s32.data(), s32.size() * sizeof(ImgStep), hipMemcpyHostToDevice));
    CK(hipMemcpy(d16, s16.data(), s16.size() * sizeof(ImgStep), hipMemcpyHostToDevice));
    const long long nsteps32 = (long long)s32.size(), groups = nsteps32 * 4;
    printf("F=%d  %d trajectories x %lld (ragged), lag %d: %lld pairs, %lld 32-pair steps; CUs=%d units=%d grid=%d cohorts=%d\n", F, n_traj, L, lag, pairs,
           nsteps32, cus, ntile2, grid, S);
    const size_t one = (size_t)groups * F * 16;
    char* img = nullptr;
    CK(hipMalloc(&img, 4 * one));
    const size_t slab_n = (size_t)(S + 1) * ntiles_sym * 2 * IMG_TM * IMG_TM;
    double *slabs = nullptr, *ref = nullptr;
    CK(hipMalloc(&slabs, slab_n * sizeof(double)));
    CK(hipMalloc(&ref, slab_n * sizeof(double)));
    std::vector<double> hs(slab_n), hr(slab_n);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    PackArgs PA;
    PA.steps32 = d32; PA.nsteps32 = nsteps32; PA.row_bytes = (long long)F * 2; PA.lag_bytes = (long long)lag * F * 2; PA.shift = shift; PA.F = F;
    PA.u_hi = reinterpret_cast<bf16x8*>(img); PA.d_hi = reinterpret_cast<bf16x8*>(img + one);
    PA.u_mid = reinterpret_cast<bf16x8*>(img + 2 * one); PA.d_mid = reinterpret_cast<bf16x8*>(img + 3 * one);
    hipLaunchKernelGGL(ref_pack_kernel, dim3((unsigned)groups), dim3(256), 0, 0, PA);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    printf("reference image packed (%zu MB)\n", 4 * one >> 20);

    for (int x2 = 0; x2 < 2; ++x2) {
        ImgMfmaArgs MA;
        memset(&MA, 0, sizeof(MA));
        MA.u_hi = PA.u_hi; MA.d_hi = PA.d_hi; MA.u_mid = PA.u_mid; MA.d_mid = PA.d_mid;
        MA.nsteps = x2 ? groups / 2 : groups / 4;
        MA.Fp = F; MA.T = T; MA.T2 = T2; MA.ntiles_sym = ntiles_sym; MA.ntile2 = ntile2; MA.S = S;
        MA.kflush_steps = x2 ? 8192 / 16 : 65536 / 32;
        MA.main_steps = img_main_steps(MA.nsteps, grid, ntile2);
        ImgFusedArgs FA;
        memset(&FA, 0, sizeof(FA));
        FA.steps = wrap > 0 ? (x2 ? dw16 : dw32) : (x2 ? d16 : d32);
        FA.shift = shift;
        FA.row_bytes = PA.row_bytes; FA.lag_bytes = PA.lag_bytes;
        FA.nsteps = (int)MA.nsteps;
        FA.T = T; FA.T2 = T2; FA.ntiles_sym = ntiles_sym; FA.ntile2 = ntile2; FA.S = S;
        FA.kflush_steps = MA.kflush_steps; FA.main_steps = MA.main_steps;
        std::vector<FusedVariant> vs;
        if (x2) {
            vs.push_back({"fused<x2>", launch_fused<true, 0>, true});
        } else {
            vs.push_back({"fused", launch_fused<false, 0>, true});
            if (abl) {
                vs.push_back({"abl: no global loads", launch_fused<false, 1>, false});
                vs.push_back({"abl: no conversion", launch_fused<false, 16>, false});
                vs.push_back({"abl: no loads, no conv", launch_fused<false, 17>, false});
                vs.push_back({"abl: no frag reads", launch_fused<false, 2>, false});
                vs.push_back({"abl: MFMA + barriers", launch_fused<false, 19>, false});
            }
        }
        // reference: the two-kernel path's multiply on the reference image
        CK(hipMemset(ref, 0, slab_n * sizeof(double)));
        MA.slabs = ref;
        if (x2) launch_pp<true>(MA, (unsigned)grid, 0); else launch_pp<false>(MA, (unsigned)grid, 0);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hr.data(), ref, slab_n * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t v = 0; v < vs.size(); ++v) {
            if (!vs[v].exact) continue;
            CK(hipMemset(slabs, 0, slab_n * sizeof(double)));
            FA.slabs = slabs;
            vs[v].launch(FA, (unsigned)grid, 0);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hs.data(), slabs, slab_n * sizeof(double), hipMemcpyDeviceToHost));
            size_t bad = 0, nz = 0;
            double worst = 0, big = 0;
            for (size_t i = 0; i < slab_n; ++i) {
                if (hr[i] != 0.0) ++nz;
                const double m = hr[i] < 0 ? -hr[i] : hr[i];
                if (m > big) big = m;
                if (hs[i] != hr[i]) {
                    ++bad;
                    const double d = hs[i] - hr[i];
                    if ((d < 0 ? -d : d) > worst) worst = d < 0 ? -d : d;
                }
            }
            printf("  check %-22s: %zu of %zu slab words differ bit for bit from pack + ping-pong (non-zero %zu, worst |diff| %.3g of max %.3g)\n", vs[v].name, bad,
                   slab_n, nz, worst, big);
        }
        // times: pack (the PRODUCT's packing kernel is in tica.hip; here the reference packer is not timed) -- multiply alone
        // vs fused; the two-kernel pipeline's pack time comes from the product run (profiles/)
        const double flop = (double)ntile2 * 2.0 * 256 * 256 * (double)(x2 ? MA.nsteps * 16 : MA.nsteps * 32) * (x2 ? 4 : 1);
        std::vector<std::vector<float>> ms(vs.size() + 1);
        FA.slabs = slabs;
        MA.slabs = slabs;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            if (x2) launch_pp<true>(MA, (unsigned)grid, 0); else launch_pp<false>(MA, (unsigned)grid, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms[vs.size()].push_back(t);
            for (size_t v = 0; v < vs.size(); ++v) {
                CK(hipEventRecord(e0, 0));
                vs[v].launch(FA, (unsigned)grid, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t, e0, e1));
                ms[v].push_back(t);
            }
        }
        for (size_t v = 0; v <= vs.size(); ++v) {
            float mn = 1e30f, sum = 0;
            for (float t : ms[v]) { mn = t < mn ? t : mn; sum += t; }
            printf("  %-26s min %.3f ms  mean %.3f ms   %.0f TF executed = %.3f of 2.5 PF  (%.1fM pairs/s)\n",
                   v == vs.size() ? (x2 ? "ping-pong<x2> (image only)" : "ping-pong (image only)") : vs[v].name, mn, sum / ms[v].size(),
                   flop / (mn * 1e-3) / 1e12, flop / (mn * 1e-3) / 2.5e15, (double)pairs / (mn * 1e-3) / 1e6);
        }
    }
    return 0;
}
