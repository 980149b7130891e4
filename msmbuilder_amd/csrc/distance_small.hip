// distance_small.hip -- exact assign_nearest for short rows (at most 128 bytes per row), euclidean family, one kernel
// per row length in 16-byte groups.  A separate translation unit only because its 32 instantiations dominate the build time.
#include "distance_dev.h"

namespace msm {

// assign_small2_kernel with the row length as a COMPILE-TIME number of 16-byte groups (NG = ceil(m / GS)).  In the
// generic kernel every group is a uniform branch -- ds_read, s_waitcnt lgkmcnt(0), 12 fp64 operations, branch -- so each 48
// cycles of arithmetic exposed a full LDS round trip (ISA of the round-2 kernel; 0.54 of the fp64-VALU bound measured by
// scripts/micro/valu_f64.hip: 12.3 cycles per float64 pair-element, 14.7 per float32 one).  Here a centre is straight-line
// code: its NG fragments are fetched while the previous centre computes, nothing branches inside a centre.
// Instantiated for the euclidean family only (the default metric of every clusterer; 2 x 8 x 2 kernels).
// one centre against the two rows of a lane: straight-line arithmetic, then the (lazy-sqrt) record update
template <typename T, int M, int NG>
__device__ __forceinline__ void small3_centre(const raw_f32x4 (&y)[NG], const T (&x0)[NG * (16 / (int)sizeof(T))],
                                              const T (&x1)[NG * (16 / (int)sizeof(T))], int j, long long m, double& best0,
                                              double& best1, double& thr0, double& thr1, int& lab0, int& lab1)
{
    constexpr int GS = 16 / (int)sizeof(T);
    __builtin_amdgcn_sched_barrier(0);
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if constexpr (sizeof(T) == 4 && (M == M_EUCLIDEAN || M == M_SQEUCLIDEAN)) {
        // float32 rows: the two differences of a feature pair in one packed subtract (v_pk_add_f32), then cvt + exact fma each
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x2* yv = reinterpret_cast<const f32x2*>(&y[g]);
#pragma unroll
            for (int e = 0; e < GS; e += 2) {
                const f32x2 u0 = {x0[g * GS + e], x0[g * GS + e + 1]}, u1 = {x1[g * GS + e], x1[g * GS + e + 1]};
                const f32x2 d0 = u0 - yv[e / 2], d1 = u1 - yv[e / 2];
                a0 = __builtin_fma((double)d0.x, (double)d0.x, a0);
                a1 = __builtin_fma((double)d1.x, (double)d1.x, a1);
                a0 = __builtin_fma((double)d0.y, (double)d0.y, a0);
                a1 = __builtin_fma((double)d1.y, (double)d1.y, a1);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const T* ye = reinterpret_cast<const T*>(&y[g]);
#pragma unroll
            for (int e = 0; e < GS; ++e) {
                m_update<T, M>(a0, b0, x0[g * GS + e], ye[e]);
                m_update<T, M>(a1, b1, x1[g * GS + e], ye[e]);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (M == M_EUCLIDEAN) {
        if (a0 < best0) {
            if (a0 < thr0 || sqrt(a0) < sqrt(best0)) {
                best0 = a0;
                thr0 = a0 * (1.0 - 0x1p-48);
                lab0 = j;
            }
        }
        if (a1 < best1) {
            if (a1 < thr1 || sqrt(a1) < sqrt(best1)) {
                best1 = a1;
                thr1 = a1 * (1.0 - 0x1p-48);
                lab1 = j;
            }
        }
    } else {
        const double d0 = m_final<M>(a0, b0, m), d1 = m_final<M>(a1, b1, m);
        if (d0 < best0) {
            best0 = d0;
            lab0 = j;
        }
        if (d1 < best1) {
            best1 = d1;
            lab1 = j;
        }
    }
}

template <typename T, int M, int NG>
__global__ __launch_bounds__(DT) __attribute__((amdgpu_waves_per_eu(4, 4))) void assign_small3_kernel(PairArgs P)
{
    constexpr int GS = 16 / (int)sizeof(T);
    constexpr int MP = NG * GS;               // padded row length (zero padding is exact)
    constexpr int YCAP = 32768 / (int)sizeof(T);
    constexpr int KT = YCAP / MP;
    __shared__ __attribute__((aligned(16))) T Ys[YCAP];
    __shared__ double red[DT];
    const T* X = static_cast<const T*>(P.X);
    const T* Y = static_cast<const T*>(P.Y);
    const int tid = threadIdx.x;
    const int m = (int)P.m;
    double inertia = 0.0;
    bool staged = false;
    const long long ntile = (P.n + 2 * DT - 1) / (2 * DT);
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long i0 = t * (2 * DT) + tid, i1 = i0 + DT;
        T x0[MP], x1[MP];
        {
            const T* p0 = X + (i0 < P.n ? i0 : P.n - 1) * P.m;
            const T* p1 = X + (i1 < P.n ? i1 : P.n - 1) * P.m;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (P.vecw == 16 && (g + 1) * GS <= m) {
                    const raw_f32x4 q0 = *reinterpret_cast<const raw_f32x4*>(p0 + g * GS), q1 = *reinterpret_cast<const raw_f32x4*>(p1 + g * GS);
#pragma unroll
                    for (int e = 0; e < GS; ++e) {
                        x0[g * GS + e] = reinterpret_cast<const T*>(&q0)[e];
                        x1[g * GS + e] = reinterpret_cast<const T*>(&q1)[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < GS; ++e) {
                        const int f = g * GS + e;
                        x0[f] = f < m ? p0[f] : (T)0;
                        x1[f] = f < m ? p1[f] : (T)0;
                    }
                }
            }
        }
        double best0 = INFINITY, best1 = INFINITY, thr0 = INFINITY, thr1 = INFINITY;
        if (M != M_EUCLIDEAN) best0 = best1 = 1.7976931348623157e308;
        int lab0 = -1, lab1 = -1;
        for (long long j0 = 0; j0 < P.K; j0 += KT) {
            const int kt = (int)((P.K - j0) < KT ? (P.K - j0) : KT);
            if (!staged) {  // all centres fit one tile (the usual case): staged once per workgroup, not once per row tile
                __syncthreads();
                for (int e = tid; e < kt * MP; e += DT) {
                    const int c = e / MP, ff = e - c * MP;
                    Ys[e] = ff < m ? Y[(j0 + c) * P.m + ff] : (T)0;
                }
                __syncthreads();
                staged = P.K <= KT;
            }
            raw_f32x4 ya[NG], yb[NG];  // ping-pong: one centre computes from ya while the next one lands in yb
#pragma unroll
            for (int g = 0; g < NG; ++g) ya[g] = *reinterpret_cast<const raw_f32x4*>(Ys + g * GS);
            int c = 0;
            for (; c + 1 < kt; c += 2) {
                const T* y1 = Ys + (c + 1) * MP;
#pragma unroll
                for (int g = 0; g < NG; ++g) yb[g] = *reinterpret_cast<const raw_f32x4*>(y1 + g * GS);
                small3_centre<T, M, NG>(ya, x0, x1, (int)(j0 + c), P.m, best0, best1, thr0, thr1, lab0, lab1);
                const T* y2 = Ys + (c + 2 < kt ? c + 2 : c + 1) * MP;
#pragma unroll
                for (int g = 0; g < NG; ++g) ya[g] = *reinterpret_cast<const raw_f32x4*>(y2 + g * GS);
                small3_centre<T, M, NG>(yb, x0, x1, (int)(j0 + c + 1), P.m, best0, best1, thr0, thr1, lab0, lab1);
            }
            if (c < kt) small3_centre<T, M, NG>(ya, x0, x1, (int)(j0 + c), P.m, best0, best1, thr0, thr1, lab0, lab1);
        }
        double d0 = 1.7976931348623157e308, d1 = 1.7976931348623157e308;
        if (lab0 >= 0) d0 = (M == M_EUCLIDEAN) ? sqrt(best0) : best0;
        if (lab1 >= 0) d1 = (M == M_EUCLIDEAN) ? sqrt(best1) : best1;
        if (i0 < P.n) {
            P.labels[i0] = lab0 < 0 ? 0 : lab0;
            if (P.min_dist) P.min_dist[i0] = d0;
            inertia += d0;
        }
        if (i1 < P.n) {
            P.labels[i1] = lab1 < 0 ? 0 : lab1;
            if (P.min_dist) P.min_dist[i1] = d1;
            inertia += d1;
        }
    }
    red[tid] = inertia;
    __syncthreads();
    for (int s = DT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) P.partial[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// Screened assign_nearest: float64 rows of at most 16 features, euclidean, 2 <= K <= 1024 (KCenters.predict on a tICA
// projection: 10M x 10 against 200 centres).  The exact kernel above spends 3 fp64 instructions per feature and centre;
// but to know which centre is nearest an approximate distance is enough for all centres but one or two.  Per row:
//   1. one float32 sweep over the centres, rows and centres CENTRED on centre 0 (translation invariant; the rounding of
//      the float copies then scales with the spread of the data, not with its offset), in the expanded form
//      w_k = |c~_k|^2 - 2 x~.c~_k + (|x~|^2 + 2 Eh): five packed FMAs, an add, an FMA for ten features.  The two smallest
//      w are kept as integer keys (w >= 0, so its bits order like unsigned integers) whose low IB bits hold the centre
//      index: v_med3_u32 + v_min_u32;
//   2. the exact squared distance a1 of the first key's centre k1, in the reference's arithmetic (assign.hpp:6-91 via
//      distance_kernels.h: (x - y)^2 summed in feature order, every operation rounded separately);
//   3. |w_k - 2 Eh - D_k^2| <= Eh for EVERY centre (D_k the real distance; bound below), and every k != k1 has
//      w_k >= v2 = the second key with its index bits cleared, so D_k^2 >= v2 - 3 Eh.  If that exceeds a1 (by more
//      than the float64 rounding of the exact sums) no other centre can reach, let alone beat, k1's exact distance: k1 IS
//      the reference's label and sqrt(a1) its distance, bit for bit;
//   4. otherwise (two centres within ~1e-5 relative of each other, duplicates, ties, non-finite or huge rows) the row is
//      "slow": its wave runs the exact sweep of the kernel above for it -- same code, same tie rule (first index).
// Error bound.  x~_e = fl32(fl64(x_e - o_e)), c~_e likewise: |x~ - X| <= u'|X| with X = x - o, u' = 2^-24 (1 + 2^-28), so
// | |x~ - c~| - D | <= u'(|X| + |C|) and | |x~ - c~|^2 - D^2 | <= 2.01 u'(|X| + |C|)^2.  The float32 evaluation of the
// expanded form (m + 3 roundings on terms bounded by (|x~| + |c~|)^2) adds at most (m + 4) u (|x~| + |c~|)^2.  Together
// < (m + 8) 2^-24 (|x~| + |c~|_max)^2 =: Eh / 1.01 (S is rounded up by the 1.01).  Rows or centres beyond 3e37 in squared
// norm, or not finite, never take the fast path.  A wave that finds slow rows in more than half of its tiles stops
// screening (lattices, heavily duplicated data: the exact kernel's cost, not 1.5x of it).
// ---------------------------------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(DT) void assign_screen_kernel(PairArgs P)
{
    constexpr int MP = 2 * NG;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char sm[];
    f32x2* Cs = reinterpret_cast<f32x2*>(sm);                              // [K][NG] centred float32 centres
    f32x2* ncs = Cs + (size_t)P.K * NG;                                    // [K] {their squared norm, the centre's index as bits}
    __shared__ double red[DT];
    __shared__ float s_nc[DT];
    const double* X = static_cast<const double*>(P.X);
    const double* Y = static_cast<const double*>(P.Y);
    const int tid = threadIdx.x, m = (int)P.m, K = (int)P.K;
    const unsigned IM = K <= 256 ? 0xffu : 0x3ffu;

    double o[MP];
#pragma unroll
    for (int e = 0; e < MP; ++e) o[e] = e < m ? Y[e] : 0.0;
    {
        float ncl = 0.f;
        for (int k = tid; k < K; k += DT) {
            float nc = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double c0 = 2 * g < m ? Y[(size_t)k * m + 2 * g] : 0.0, c1 = 2 * g + 1 < m ? Y[(size_t)k * m + 2 * g + 1] : 0.0;
                const f32x2 c = {(float)(c0 - o[2 * g]), (float)(c1 - o[2 * g + 1])};
                Cs[(size_t)k * NG + g] = c;
                nc = __builtin_fmaf(c.x, c.x, nc);
                nc = __builtin_fmaf(c.y, c.y, nc);
            }
            ncs[k] = f32x2{nc, __uint_as_float((unsigned)k)};
            ncl = (nc <= 3e37f) ? (nc > ncl ? nc : ncl) : INFINITY;   // NaN or huge: no screening at all
        }
        s_nc[tid] = ncl;
        __syncthreads();
        for (int s = DT / 2; s > 0; s >>= 1) {
            if (tid < s) s_nc[tid] = s_nc[tid] > s_nc[tid + s] ? s_nc[tid] : s_nc[tid + s];
            __syncthreads();
        }
    }
    const float ncmax = s_nc[0];
    const float rnc = sqrtf(ncmax);
    const float cE = (float)(MP + 8) * 0x1p-24f;
    bool screening = ncmax <= 3e37f;   // uniform
    int tiles_done = 0, tiles_slow = 0;

    double inertia = 0.0;
    const long long ntile = (P.n + 2 * DT - 1) / (2 * DT);
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long i0 = t * (2 * DT) + tid, i1 = i0 + DT;
        double x0[MP], x1[MP];
        {
            const double* p0 = X + (i0 < P.n ? i0 : P.n - 1) * P.m;
            const double* p1 = X + (i1 < P.n ? i1 : P.n - 1) * P.m;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (P.vecw == 16 && (g + 1) * 2 <= m) {
                    const raw_f32x4 q0 = *reinterpret_cast<const raw_f32x4*>(p0 + g * 2), q1 = *reinterpret_cast<const raw_f32x4*>(p1 + g * 2);
                    x0[2 * g] = reinterpret_cast<const double*>(&q0)[0];
                    x0[2 * g + 1] = reinterpret_cast<const double*>(&q0)[1];
                    x1[2 * g] = reinterpret_cast<const double*>(&q1)[0];
                    x1[2 * g + 1] = reinterpret_cast<const double*>(&q1)[1];
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int f = g * 2 + e;
                        x0[f] = f < m ? p0[f] : 0.0;
                        x1[f] = f < m ? p1[f] : 0.0;
                    }
                }
            }
        }
        double best0 = INFINITY, best1 = INFINITY;
        int lab0 = -1, lab1 = -1;
        bool need0 = true, need1 = true;   // the row still needs the exact sweep
        if (screening) {
            f32x2 xt0[NG], xt1[NG];
            float nx0 = 0.f, nx1 = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                xt0[g] = f32x2{(float)(x0[2 * g] - o[2 * g]), (float)(x0[2 * g + 1] - o[2 * g + 1])};
                xt1[g] = f32x2{(float)(x1[2 * g] - o[2 * g]), (float)(x1[2 * g + 1] - o[2 * g + 1])};
                nx0 = __builtin_fmaf(xt0[g].x, xt0[g].x, nx0);
                nx0 = __builtin_fmaf(xt0[g].y, xt0[g].y, nx0);
                nx1 = __builtin_fmaf(xt1[g].x, xt1[g].x, nx1);
                nx1 = __builtin_fmaf(xt1[g].y, xt1[g].y, nx1);
            }
            const float S0 = (sqrtf(nx0) + rnc) * 1.001f, S1 = (sqrtf(nx1) + rnc) * 1.001f;
            const float Eh0 = cE * S0 * S0 * 1.01f, Eh1 = cE * S1 * S1 * 1.01f;
            const float off0 = nx0 + 2.f * Eh0, off1 = nx1 + 2.f * Eh1;
            const f32x2 init0 = {-0.5f * off0, 0.f}, init1 = {-0.5f * off1, 0.f};
            unsigned a1k = 0xffffffffu, a2k = 0xffffffffu, b1k = 0xffffffffu, b2k = 0xffffffffu;
            for (int k = 0; k < K; ++k) {
                f32x2 c[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) c[g] = Cs[(size_t)k * NG + g];
                const f32x2 nck = ncs[k];   // (the index comes from LDS too: as a VGPR operand it lets `and` + `or` fuse)
                const float nc = nck.x;
                const unsigned kv = __float_as_uint(nck.y);
                f32x2 s0 = init0, s1 = init1;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    s0 = __builtin_elementwise_fma(xt0[g], c[g], s0);
                    s1 = __builtin_elementwise_fma(xt1[g], c[g], s1);
                }
                float w0 = __builtin_fmaf(-2.f, s0.x + s0.y, nc), w1 = __builtin_fmaf(-2.f, s1.x + s1.y, nc);
                w0 = w0 > 0.f ? w0 : 0.f;
                w1 = w1 > 0.f ? w1 : 0.f;
                const unsigned k0 = (__float_as_uint(w0) & ~IM) | kv, k1 = (__float_as_uint(w1) & ~IM) | kv;
                // second smallest: median of (smallest, second smallest, new); then the smallest
                a2k = max(min(a1k, a2k), min(max(a1k, a2k), k0));
                a1k = min(a1k, k0);
                b2k = max(min(b1k, b2k), min(max(b1k, b2k), k1));
                b1k = min(b1k, k1);
            }
            const int c0i = (int)(a1k & IM), c1i = (int)(b1k & IM);
            const double* y0 = Y + (size_t)c0i * m;
            const double* y1 = Y + (size_t)c1i * m;
            double e0 = 0.0, e1 = 0.0;
#pragma unroll
            for (int e = 0; e < MP; ++e) {
                const double v0 = e < m ? y0[e] : 0.0, v1 = e < m ? y1[e] : 0.0;
                const double d0 = x0[e] - v0, d1 = x1[e] - v1;
                e0 = e0 + d0 * d0;
                e1 = e1 + d1 * d1;
            }
            // w_k = D_k^2 + 2 Eh up to Eh (the |x~|^2 inside `off` is the third term of the expanded square): D_k^2 >= v2 - 3 Eh
            const double lb0 = (double)__uint_as_float(a2k & ~IM) - 4.0 * (double)Eh0;
            const double lb1 = (double)__uint_as_float(b2k & ~IM) - 4.0 * (double)Eh1;
            if (nx0 <= 3e37f && lb0 > e0 * (1.0 + 0x1p-40)) {
                need0 = false;
                best0 = e0;
                lab0 = c0i;
            }
            if (nx1 <= 3e37f && lb1 > e1 * (1.0 + 0x1p-40)) {
                need1 = false;
                best1 = e1;
                lab1 = c1i;
            }
        }
        const unsigned long long sm0 = __builtin_amdgcn_ballot_w64(need0 && i0 < P.n), sm1 = __builtin_amdgcn_ballot_w64(need1 && i1 < P.n);
        const bool any_slow = (sm0 | sm1) != 0;
        const int nslow = __builtin_popcountll(sm0) + __builtin_popcountll(sm1);
        if (any_slow && screening && nslow <= 12) {
            // a few undecided rows: the WAVE takes each of them in turn, lane l evaluating centres l, l + 64, ... exactly (a
            // lane's centres in index order with the sequential rule), then the smallest sqrt wins and among equal ones the
            // lowest index -- what the reference's scan over j = 0 .. K-1 with `d < best` returns
            const int lane = tid & 63;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                unsigned long long mask = r ? sm1 : sm0;
                while (mask) {
                    const int src = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    double xr[MP];
#pragma unroll
                    for (int e = 0; e < MP; ++e) {
                        const double v = r ? x1[e] : x0[e];
                        const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src), hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
                        xr[e] = __hiloint2double(hi, lo);
                    }
                    double b = INFINITY, thr = INFINITY;
                    int l = -1;
                    for (int j = lane; j < K; j += 64) {
                        const double* yj = Y + (size_t)j * m;
                        double a = 0.0;
#pragma unroll
                        for (int e = 0; e < MP; ++e) {
                            const double d = xr[e] - (e < m ? yj[e] : 0.0);
                            a = a + d * d;
                        }
                        if (a < b) {
                            if (a < thr || sqrt(a) < sqrt(b)) {
                                b = a;
                                thr = a * (1.0 - 0x1p-48);
                                l = j;
                            }
                        }
                    }
                    double d = l >= 0 ? sqrt(b) : INFINITY;
                    int li = l >= 0 ? l : 0x7fffffff;
#pragma unroll
                    for (int s = 32; s > 0; s >>= 1) {
                        const double od = __shfl_xor(d, s);
                        const double ob = __shfl_xor(b, s);
                        const int ol = __shfl_xor(li, s);
                        if (od < d || (od == d && ol < li)) {
                            d = od;
                            b = ob;
                            li = ol;
                        }
                    }
                    if (lane == src) {
                        if (r) {
                            best1 = b;
                            lab1 = li == 0x7fffffff ? -1 : li;
                        } else {
                            best0 = b;
                            lab0 = li == 0x7fffffff ? -1 : li;
                        }
                    }
                }
            }
        } else if (any_slow) {   // wave-uniform: the exact sweep (centres from global memory: uniform addresses, L2 resident)
            double bb0 = INFINITY, bb1 = INFINITY, thr0 = INFINITY, thr1 = INFINITY;
            int ll0 = -1, ll1 = -1;
            for (int j = 0; j < K; ++j) {
                raw_f32x4 y[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    double* yd = reinterpret_cast<double*>(&y[g]);
                    yd[0] = 2 * g < m ? Y[(size_t)j * m + 2 * g] : 0.0;
                    yd[1] = 2 * g + 1 < m ? Y[(size_t)j * m + 2 * g + 1] : 0.0;
                }
                small3_centre<double, M_EUCLIDEAN, NG>(y, x0, x1, j, P.m, bb0, bb1, thr0, thr1, ll0, ll1);
            }
            if (need0) {
                best0 = bb0;
                lab0 = ll0;
            }
            if (need1) {
                best1 = bb1;
                lab1 = ll1;
            }
        }
        if (screening) {
            ++tiles_done;
            tiles_slow += nslow > 12 ? 1 : 0;   // tiles that paid for a whole exact sweep
            if (tiles_done >= 8 && 2 * tiles_slow > tiles_done) screening = false;
        }
        double d0 = 1.7976931348623157e308, d1 = 1.7976931348623157e308;
        if (lab0 >= 0) d0 = sqrt(best0);
        if (lab1 >= 0) d1 = sqrt(best1);
        if (i0 < P.n) {
            P.labels[i0] = lab0 < 0 ? 0 : lab0;
            if (P.min_dist) P.min_dist[i0] = d0;
            inertia += d0;
        }
        if (i1 < P.n) {
            P.labels[i1] = lab1 < 0 ? 0 : lab1;
            if (P.min_dist) P.min_dist[i1] = d1;
            inertia += d1;
        }
    }
    red[tid] = inertia;
    __syncthreads();
    for (int s = DT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) P.partial[blockIdx.x] = red[0];
}

// true when the shape was launched on the screened kernel
static bool launch_screen_f64(int grid, const PairArgs& P)
{
    const char* env = getenv("MSM_ASSIGN_SCREEN");   // read per call: A/B switch of the tests
    const bool off = env && atoi(env) == 0;
    const int ng = (int)((P.m + 1) / 2);
    const size_t lds = (size_t)P.K * ng * 8 + (size_t)P.K * 8;
    if (off || P.K < 2 || P.K > 1024 || ng < 1 || ng > 8 || lds > 48 * 1024 || P.n < 16384) return false;
    switch (ng) {
#define MSM_SC(NG_) case NG_: hipLaunchKernelGGL((assign_screen_kernel<NG_>), dim3(grid), dim3(DT), lds, stream(), P); return true;
        MSM_SC(1) MSM_SC(2) MSM_SC(3) MSM_SC(4) MSM_SC(5) MSM_SC(6) MSM_SC(7) MSM_SC(8)
#undef MSM_SC
    }
    return false;
}

template <typename T, int M>
static bool launch_small3(int grid, const PairArgs& P)
{
    constexpr int GS = 16 / (int)sizeof(T);
    const int ng = (int)((P.m + GS - 1) / GS);
    switch (ng) {
#define MSM_S3(NG_) case NG_: hipLaunchKernelGGL((assign_small3_kernel<T, M, NG_>), dim3(grid), dim3(DT), 0, stream(), P); return true;
        MSM_S3(1) MSM_S3(2) MSM_S3(3) MSM_S3(4) MSM_S3(5) MSM_S3(6) MSM_S3(7) MSM_S3(8)
#undef MSM_S3
    }
    return false;
}

bool launch_small3_f32(int metric, int grid, const PairArgs& P)
{
    return metric == M_SQEUCLIDEAN ? launch_small3<float, M_SQEUCLIDEAN>(grid, P) : launch_small3<float, M_EUCLIDEAN>(grid, P);
}

bool launch_small3_f64(int metric, int grid, const PairArgs& P)
{
    if (metric == M_EUCLIDEAN && launch_screen_f64(grid, P)) return true;
    return metric == M_SQEUCLIDEAN ? launch_small3<double, M_SQEUCLIDEAN>(grid, P) : launch_small3<double, M_EUCLIDEAN>(grid, P);
}

}  // namespace msm
