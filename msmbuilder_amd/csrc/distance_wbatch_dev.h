// distance_wbatch_dev.h -- kwb_* kernels: SEVERAL centres per screened pass of wide rows (round 6; included by distance.hip
// after distance_wscreen_dev.h and distance_kcbatch_dev.h).
//
// The screened passes of distance_wscreen_dev.h cost m + 8 bytes per row -- but one pass per centre: KCenters(200) on
// 280,000 x 171 float32 is 196 launches of ~33 us for 50 MB each, launch- and latency-bound (profiles/r06_kcenters_wide.txt).
// The threshold lists of distance_kcbatch_dev.h (float64 rows of <= 16 features) carry over unchanged in their logic: a pass
// lists every row whose rounded-up distance exceeds theta; kwb_select_kernel, one workgroup, plays the algorithm on the
// list alone -- the listed row of largest distance (lowest row on ties) is the next centre (it beats every unlisted row
// strictly), the other listed rows get d = min(d, dist(row, centre)) in the pass's exact arithmetic, the largest of them is
// the centre after that if it still exceeds theta, ... up to `jmax` centres; the next pass applies them all, in order, to
// every row it streams.  What differs from the narrow case: the rows do not fit a thread's registers, so
//   * the selector reads a listed row from global memory for every centre it tries (rows of an L2-resident list: 2,048
//     rows x m) and keeps the centre's coordinates in LDS;
//   * the pass keeps the batch's centres in LDS -- float32 relative to the copy's origin for the screen, the rows themselves
//     for the exact re-evaluation -- accumulates JB squared distances per row while it streams the byte planes once
//     (4 features of a row: one word, unpacked once, 8 VALU operations per centre), marks the (row, centre) pairs the
//     screen cannot prove unchanged in a bit mask per row, and re-evaluates a marked row against its marked centres IN ORDER
//     from its own coordinates (staged through LDS as in the one-centre pass), exactly as the separate passes would have
//     met it: a pair proven "no update" against the distance the row had BEFORE the batch stays proven when an earlier
//     centre of the batch lowers that distance.
// Centre ids, labels_ and distances_ are those of the one-centre-per-pass loop, bit for bit (tests/test_gpu_kcenters_wide.py).
#pragma once
#include "common.h"
#include "distance_dev.h"
#include "distance_kcbatch_dev.h"
#include "distance_wscreen_dev.h"

namespace msm {

template <typename T>
__global__ __launch_bounds__(1024) void kwb_select_kernel(KwsArgs P, KcbState* S, int K, int jmax, int cap)
{
    extern __shared__ __attribute__((aligned(16))) char kwb_smem[];
    T* cs = reinterpret_cast<T*>(kwb_smem);   // [m] the centre being applied
    __shared__ double rv[1024];
    __shared__ long long ri[1024];
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x;
    const long long m = P.m;
    const int k0 = S->k_done;
    if (k0 >= K) {
        if (tid == 0) S->J = 0;
        return;
    }
    auto reduce = [&](double v, long long i, double& ov, long long& oi) {
        double wv;
        long long wi;
        kcb_wave_argmax(v, i, wv, wi);
        __syncthreads();   // the previous call's readers are done with rv / ri
        if ((tid & 63) == 0) {
            rv[tid >> 6] = wv;
            ri[tid >> 6] = wi;
        }
        __syncthreads();
        const int l = tid & 63;
        kcb_wave_argmax(l < 16 ? rv[l] : -1.0, l < 16 ? ri[l] : -1, ov, oi);
    };
    // the row the per-block partials of the last pass name (the one-centre-per-pass loop's choice)
    double vP;
    long long iP;
    {
        double v = -1.0;
        long long i = -1;
        for (int k = tid; k < P.nblk; k += 1024) {
            const KcPartial q = P.prev[k];
            if (q.i >= 0 && (i < 0 || kc_better(q.v, q.i, v, i))) {
                v = q.v;
                i = q.i;
            }
        }
        reduce(v, i, vP, iP);
    }
    const float theta = S->theta;
    const unsigned cnt = S->count;
    const bool usable = cnt > 0 && cnt <= (unsigned)cap && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    const unsigned target = (unsigned)(cap - cap / 4);   // (kcb: 1,536 of 2,048)
    long long ci[2] = {-1, -1};
    double cv[2] = {-1.0, -1.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned c = (unsigned)tid + 1024u * u;
        if (usable && c < cnt) {
            ci[u] = S->list[c];
            cv[u] = P.dist[ci[u]];
        }
    }
    int J = 0, fell = 0;
    double vlast = vP;
    for (;;) {
        double v = -1.0, vb;
        long long i = -1, ib;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (ci[u] >= 0 && (i < 0 || kc_better(cv[u], ci[u], v, i))) {
                v = cv[u];
                i = ci[u];
            }
        reduce(v, i, vb, ib);
        long long centre;
        if (J == 0) {
            if (usable && ib == iP) {
                centre = ib;
            } else {
                centre = iP;
                fell = 1;
            }
            vlast = vP;
        } else {
            if (!(ib >= 0 && vb > tau)) break;
            centre = ib;
            vlast = vb;
        }
        __syncthreads();   // (the previous centre's readers are done with cs)
        for (long long f = tid; f < m; f += 1024) cs[f] = X[centre * m + f];
        if (tid == 0) P.ids[k0 + J] = centre;
        __syncthreads();
        ++J;
        if (fell || k0 + J >= K || J >= jmax) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ci[u] < 0) continue;
            if (ci[u] == centre) {
                ci[u] = -1;
                continue;
            }
            const T* x = X + ci[u] * m;
            double a = 0.0, b = 0.0;
            for (long long f = 0; f < m; ++f) m_update<T, M_EUCLIDEAN>(a, b, x[f], cs[f]);
            const double d = m_final<M_EUCLIDEAN>(a, b, m);
            if (d < cv[u]) cv[u] = d;   // the pass's own update (kcenters.py:93)
        }
    }
    if (tid == 0) {
        // threshold of the next list (kcb_select_kernel's rule)
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (cnt > target) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (S->lev[q] <= target) l = q;
            th = theta * kcb_level(l);
        }
        if (th > vl) th = vl;
        S->theta = th;
        S->count = 0;
        for (int q = 0; q < KCB_NLEV; ++q) S->lev[q] = 0;
        S->J = J;
        S->k_done = k0 + J;
        S->rounds += 1;
        S->fallbacks += fell;
    }
}

// ---- the selector on SEVERAL workgroups -------------------------------------------------------------------------------------
// kwb_select_kernel moves every listed row through ONE compute unit for every centre it tries: 2,048 rows x 684 bytes =
// 1.4 MB per centre at 171 float32 features, ~100 us per round of a dozen centres and up to 400 us when the list is full
// (profiles/r06_kcenters_wide.txt) -- as long as the pass it prepares.  Here NB workgroups (8 .. 64: as many as it takes for
// a slice of the list to fit the LDS) each keep their slice's rows IN LDS for the whole round and replay the selection
// together: per centre a workgroup reduces its slice to one candidate, publishes it, meets the others at a grid barrier,
// reads all NB candidates and makes the same decision as everybody else, then updates its slice's distances from LDS (one
// thread per row, the reference's arithmetic).  One barrier per centre (candidate slots double-buffered by parity).
//   The barrier: an arrival counter in device memory, agent-scope release on arrival and acquire after the wait (the XCDs'
// L2s are not coherent with each other), a launch base left by the previous launch of the same stream (every workgroup
// passes the same number of barriers per launch: the decisions are functions of the published data).  The NB workgroups
// are co-resident by construction -- the launch is alone on the device's compute queue at that point of the stream and NB <=
// 64 of 256 CUs -- and the wait is BOUNDED: a workgroup that waits ~2 s flags the state, and the fit fails loudly.
struct KwbSync {
    unsigned bar;        // arrivals since the fit began
    unsigned bar_base;   // ... at the start of the current launch
    unsigned timed_out;
    unsigned pad;
    KcPartial slot[2][64];
};

__device__ __forceinline__ bool kwb_grid_barrier(KwbSync* Y, unsigned target)
{
    // (called by one thread per workgroup, between two __syncthreads)
    __hip_atomic_fetch_add(&Y->bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    for (long long spin = 0; spin < 2000000LL; ++spin) {   // (~1 us per look: ~2 s)
        if (__hip_atomic_load(&Y->bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(&Y->timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

template <typename T>
__global__ __launch_bounds__(DT) void kwb_select_multi_kernel(KwsArgs P, KcbState* S, KwbSync* Y, int K, int jmax, int cap)
{
    extern __shared__ __attribute__((aligned(16))) char kwb_smem[];
    __shared__ double rv[DT / 64];
    __shared__ long long ri[DT / 64];
    __shared__ double bc_v;
    __shared__ long long bc_i;
    __shared__ int bc_ok;
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x, wg = blockIdx.x, NB = gridDim.x;
    const long long m = P.m;
    const int pitch = (int)(m | 1);                       // odd: a thread per row walks its row without bank conflicts
    T* cs = reinterpret_cast<T*>(kwb_smem);               // [m] the centre being applied
    T* rows = cs + ((m + 3) & ~3LL);                      // [rows of the slice][pitch]
    const int k0 = S->k_done;
    if (k0 >= K) {
        if (wg == 0 && tid == 0) S->J = 0;
        return;
    }
    // block argmax of (value, row): largest value, lowest row on ties -- result in every thread
    auto reduce = [&](double v, long long i, double& ov, long long& oi) {
        double wv;
        long long wi;
        kcb_wave_argmax(v, i, wv, wi);
        __syncthreads();
        if ((tid & 63) == 0) {
            rv[tid >> 6] = wv;
            ri[tid >> 6] = wi;
        }
        __syncthreads();
        const int l = tid & 63;
        kcb_wave_argmax(l < DT / 64 ? rv[l] : -1.0, l < DT / 64 ? ri[l] : -1, ov, oi);
    };
    double vP;
    long long iP;
    {
        double v = -1.0;
        long long i = -1;
        for (int k = tid; k < P.nblk; k += DT) {
            const KcPartial q = P.prev[k];
            if (q.i >= 0 && (i < 0 || kc_better(q.v, q.i, v, i))) {
                v = q.v;
                i = q.i;
            }
        }
        reduce(v, i, vP, iP);
    }
    const float theta = S->theta;
    const unsigned cnt = S->count;
    const bool usable = cnt > 0 && cnt <= (unsigned)cap && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    const unsigned target = (unsigned)(cap - cap / 4);
    // this workgroup's slice of the list: rows [r0, r0 + nr), one per thread, staged in LDS (coalesced)
    const unsigned rpw = usable ? (cnt + NB - 1) / NB : 0u;
    const unsigned r0 = (unsigned)wg * rpw;
    const int nr = usable && r0 < cnt ? (int)(cnt - r0 < rpw ? cnt - r0 : rpw) : 0;
    long long ci = -1;
    double cv = -1.0;
    if (tid < nr) {
        ci = S->list[r0 + tid];
        cv = P.dist[ci];
    }
    for (int r = tid >> 6; r < nr; r += DT / 64) {        // a wave per row: coalesced
        const long long row = S->list[r0 + r];
        for (long long f = tid & 63; f < m; f += 64) rows[(size_t)r * pitch + f] = X[row * m + f];
    }
    unsigned nbar = 0;
    const unsigned base = Y->bar_base;
    int J = 0, fell = 0, par = 0;
    double vlast = vP;
    bool alive = true;
    for (;;) {
        double vb, v1;
        long long ib, i1;
        reduce(ci >= 0 ? cv : -1.0, ci, v1, i1);
        if (tid == 0) {
            KcPartial q;
            q.v = v1;
            q.i = i1;
            Y->slot[par][wg] = q;
            ++nbar;
            bc_ok = kwb_grid_barrier(Y, base + (unsigned)NB * nbar) ? 1 : 0;
        }
        __syncthreads();
        if (!bc_ok) {
            alive = false;
            break;
        }
        if (tid < 64) {
            double sv = -1.0;
            long long si = -1;
            if (tid < NB) {
                const KcPartial q = Y->slot[par][tid];
                sv = q.v;
                si = q.i;
            }
            double ov;
            long long oi;
            kcb_wave_argmax(sv, si, ov, oi);
            if (tid == 0) {
                bc_v = ov;
                bc_i = oi;
            }
        }
        __syncthreads();
        vb = bc_v;
        ib = bc_i;
        par ^= 1;
        long long centre;
        if (J == 0) {
            if (usable && ib == iP) {
                centre = ib;
            } else {
                centre = iP;
                fell = 1;
            }
            vlast = vP;
        } else {
            if (!(ib >= 0 && vb > tau)) break;
            centre = ib;
            vlast = vb;
        }
        for (long long f = tid; f < m; f += DT) cs[f] = X[centre * m + f];
        if (wg == 0 && tid == 0) P.ids[k0 + J] = centre;
        __syncthreads();
        ++J;
        if (fell || k0 + J >= K || J >= jmax) break;
        if (ci >= 0) {
            if (ci == centre) {
                ci = -1;
            } else {
                const T* x = rows + (size_t)tid * pitch;
                double a = 0.0, b = 0.0;
                for (long long f = 0; f < m; ++f) m_update<T, M_EUCLIDEAN>(a, b, x[f], cs[f]);
                const double d = m_final<M_EUCLIDEAN>(a, b, m);
                if (d < cv) cv = d;   // the pass's own update (kcenters.py:93)
            }
        }
    }
    if (wg == 0 && tid == 0) {
        // threshold of the next list (kcb_select_kernel's rule)
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (cnt > target) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (S->lev[q] <= target) l = q;
            th = theta * kcb_level(l);
        }
        if (th > vl) th = vl;
        S->theta = th;
        S->count = 0;
        for (int q = 0; q < KCB_NLEV; ++q) S->lev[q] = 0;
        S->J = alive ? J : 0;
        S->k_done = k0 + (alive ? J : 0);
        S->rounds += 1;
        S->fallbacks += fell;
        Y->bar_base = base + (unsigned)NB * nbar;
    }
}

// JS: the centres the body is compiled for (the batch's J rounded up to a multiple of 4: the kernel switches on it); the LDS
// holds the centres of a feature side by side ([4 nb4][JB] floats), so two centres' differences and squares are ONE packed
// operation each (v_pk_add_f32 / v_pk_fma_f32): 4 VALU operations per word of a row and centre.
// candidate rows re-evaluated together (staged in LDS): 64 rows of up to 175 floats; rows at a pitch of 16 x odd bytes, so
// the lanes' 16-byte reads of their own rows fall on different bank quads
constexpr int KWB_CB = 64;
constexpr int KWB_CFLOATS = 11264;   // 44 KiB

template <int KWS_R>
struct KwbShared {   // the pass kernel's static LDS (one object, shared by the bodies the kernel switches between)
    double rv[DT];
    long long ri[DT];
    __attribute__((aligned(16))) float cstage[KWB_CFLOATS];
    int cand[KWS_R * DT];
    unsigned short cmk[KWS_R * DT];
    unsigned slev[KCB_NLEV];
    int ncand;
};

template <typename T, int JB, int JS, int KWS_R, int KWS_U>
__device__ __forceinline__ void kwb_pass_body(const KwsArgs& P, KcbState* S, char* kwb_smem, KwbShared<KWS_R>& sh)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float* ycf = reinterpret_cast<float*>(kwb_smem);                                    // [4 nb4][JB] centres relative to c0, float32, zero padded
    T* yraw = reinterpret_cast<T*>(kwb_smem + (size_t)JB * 4 * P.nb4 * sizeof(float));    // [JB][ypitch] the centres themselves (rows 16-byte aligned)
    const int ypitch = (int)((P.m * sizeof(T) + 15) / 16 * 16 / sizeof(T));
    double* const rv = sh.rv;
    long long* const ri = sh.ri;
    int* const cand = sh.cand;
    unsigned short* const cmk = sh.cmk;
    int& ncand = sh.ncand;
    unsigned* const slev = sh.slev;
    float* const cstage = sh.cstage;
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x, lane = tid & 63;
    const long long n = P.n, m = P.m;
    const int J = S->J;
    const int kbase = S->k_done - J;
    const float theta = S->theta;
    const int cw = 4 * P.nb4;
    if (tid < KCB_NLEV) slev[tid] = 0;
    for (int e = tid; e < JS * cw; e += DT) {
        const int j = e / cw, f = e - j * cw;
        const long long row = P.ids[kbase + (j < J ? j : J - 1)];
        const T yv = f < m ? X[row * m + f] : (T)0;
        if (f < m) yraw[(size_t)j * ypitch + f] = yv;
        ycf[f * JB + j] = f < m ? (float)((double)yv - P.c0[f]) : 0.f;
    }
    __syncthreads();
    const double G = sqrt(__longlong_as_double((long long)P.gmax2[0]));
    const float Gf = (float)G * 1.0000002f;
    const float ea = (float)(m + 8) * 2.3841858e-07f;
    const float eq = 0.5001f * sqrtf((float)m);
    const float refl = (sizeof(T) == 4) ? (1.f - 2.3841858e-07f) : 1.f;
    const int cpitch = (int)(((((long long)m * sizeof(T) + 15) / 16) | 1) * 16 / sizeof(T));   // 16 x odd bytes
    const int cbatch = (int)((long long)KWB_CFLOATS * sizeof(float) / ((long long)cpitch * sizeof(T)));
    const int cb = cbatch < 1 ? 0 : (cbatch < KWB_CB ? cbatch : KWB_CB);
    T* cst = reinterpret_cast<T*>(cstage);

    float bestf = -1.f;
    long long besti = -1;
    unsigned nlev[KCB_NLEV];
#pragma unroll
    for (int l = 0; l < KCB_NLEV; ++l) nlev[l] = 0;
    auto consider = [&](long long i, float cf) {
        if (besti < 0 || cf > bestf) {
            bestf = cf;
            besti = i;
        } else if (cf == bestf) {
            const double di = __hip_atomic_load(P.dist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double db = __hip_atomic_load(P.dist + besti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (kc_better(di, i, db, besti)) besti = i;
        }
    };
    // the list of the next selection and the level counts that place its threshold: called by EVERY lane of the wave
    // (`has`: this lane holds a settled row i of rounded-up distance cf)
    auto settle = [&](bool has, long long i, float cf) {
        const bool lst = has && cf > theta;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(lst);
        if (bal) {
            const int leader = __builtin_ctzll(bal);
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&S->count, (unsigned)__builtin_popcountll(bal));
            base = __shfl(base, leader);
            if (lst) {
                const unsigned slot = base + (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (slot < (unsigned)KCB_CAP) S->list[slot] = i;
            }
        }
#pragma unroll
        for (int l = 1; l < KCB_NLEV; ++l) nlev[l] += (has && cf > theta * kcb_level(l)) ? 1u : 0u;
        if (has) consider(i, cf);
    };

    const long long nsuper = (n + (long long)KWS_R * DT - 1) / ((long long)KWS_R * DT);
    for (long long t = blockIdx.x; t < nsuper; t += gridDim.x) {
        const long long base = t * (KWS_R * DT);
        long long row[KWS_R];
        f32x2 acc[KWS_R][JS / 2];
        float sfr[KWS_R], cur[KWS_R];
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            const long long i = base + k * DT + tid;
            row[k] = i < n ? i : n - 1;
#pragma unroll
            for (int j = 0; j < JS / 2; ++j) acc[k][j] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            sfr[k] = __uint_as_float((unsigned)P.sf[row[k]] << 16);
            cur[k] = P.curf[row[k]];
        }
        for (int b0 = 0; b0 < P.nb4; b0 += KWS_U) {
            unsigned w[KWS_U][KWS_R];
#pragma unroll
            for (int u = 0; u < KWS_U; ++u) {
                const int b = b0 + u < P.nb4 ? b0 + u : P.nb4 - 1;
#pragma unroll
                for (int k = 0; k < KWS_R; ++k) w[u][k] = P.q[(size_t)b * n + row[k]];
            }
#pragma unroll
            for (int u = 0; u < KWS_U; ++u) {
                if (b0 + u < P.nb4) {   // uniform
                    float x0[KWS_R], x1[KWS_R], x2[KWS_R], x3[KWS_R];
#pragma unroll
                    for (int k = 0; k < KWS_R; ++k) {   // q sf: exact (8-bit integer x 8-bit significand)
                        const int wi = (int)w[u][k];
                        x0[k] = (float)((wi << 24) >> 24) * sfr[k];
                        x1[k] = (float)((wi << 16) >> 24) * sfr[k];
                        x2[k] = (float)((wi << 8) >> 24) * sfr[k];
                        x3[k] = (float)(wi >> 24) * sfr[k];
                    }
                    const float* yb = ycf + (size_t)4 * (b0 + u) * JB;
#pragma unroll
                    for (int j = 0; j < JS / 2; ++j) {
                        const f32x2 y0 = *reinterpret_cast<const f32x2*>(yb + 2 * j), y1 = *reinterpret_cast<const f32x2*>(yb + JB + 2 * j);
                        const f32x2 y2 = *reinterpret_cast<const f32x2*>(yb + 2 * JB + 2 * j), y3 = *reinterpret_cast<const f32x2*>(yb + 3 * JB + 2 * j);
#pragma unroll
                        for (int k = 0; k < KWS_R; ++k) {
                            const f32x2 t0 = f32x2{x0[k], x0[k]} - y0, t1 = f32x2{x1[k], x1[k]} - y1;
                            const f32x2 t2 = f32x2{x2[k], x2[k]} - y2, t3 = f32x2{x3[k], x3[k]} - y3;
                            f32x2 a = acc[k][j];
                            a = __builtin_elementwise_fma(t0, t0, a);
                            a = __builtin_elementwise_fma(t1, t1, a);
                            a = __builtin_elementwise_fma(t2, t2, a);
                            a = __builtin_elementwise_fma(t3, t3, a);
                            acc[k][j] = a;
                        }
                    }
                }
            }
        }
        if (tid == 0) ncand = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            const long long i = base + k * DT + tid;
            unsigned mk = 0;
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                const float dt = sqrtf((j & 1) ? acc[k][j >> 1].y : acc[k][j >> 1].x);
                const float E = eq * sfr[k] + (dt + Gf) * ea;
                const bool safe = (dt - E) * refl >= cur[k];   // NaN anywhere: false -> candidate
                if (j < J && !safe) mk |= 1u << j;
            }
            const bool in = i < n;
            if (in && mk) {
                const int slot = atomicAdd(&ncand, 1);
                cand[slot] = k * DT + tid;
                cmk[slot] = (unsigned short)mk;
            }
            settle(in && mk == 0, i, cur[k]);
        }
        __syncthreads();
        const int nc = ncand;
        const int step = cb ? cb : DT;
        for (int c0 = 0; c0 < nc; c0 += step) {
            const int cn = nc - c0 < step ? nc - c0 : step;
            const long long icp = base + cand[c0 + (tid < cn ? tid : 0)];
            const double dold = P.dist[icp];
            const float cfo = P.curf[icp];
            if (cb) {
                if (c0) __syncthreads();
                for (int e = tid; e < cn * (int)m; e += DT) {
                    const int c = e / (int)m, f = e - c * (int)m;
                    cst[c * cpitch + f] = X[(base + cand[c0 + c]) * m + f];
                }
                __syncthreads();
            }
            float cf = cfo;
            long long ic = icp;
            if (tid < cn) {
                ic = base + cand[c0 + tid];
                unsigned mk = cmk[c0 + tid];
                const T* x = cb ? cst + tid * cpitch : X + ic * m;
                double c = dold;
                int lab = -1;
                while (mk) {   // the batch's centres in order, as the separate passes would meet the row
                    const int j = __builtin_ctz(mk);
                    mk &= mk - 1;
                    const T* y = yraw + (size_t)j * ypitch;
                    double d;
                    if (cb) {
                        d = kws_exact_euclid<T>(x, y, m);
                    } else {   // rows too long to stage: straight from global memory (any alignment)
                        double a = 0.0, bb = 0.0;
                        for (long long f = 0; f < m; ++f) m_update<T, M_EUCLIDEAN>(a, bb, x[f], y[f]);
                        d = m_final<M_EUCLIDEAN>(a, bb, m);
                    }
                    if (d < c) {   // strict, kcenters.py:93
                        c = d;
                        lab = kbase + j;
                    }
                }
                if (lab >= 0) {
                    P.dist[ic] = c;
                    P.labels[ic] = lab;
                    cf = ksc_round_up(c);
                    P.curf[ic] = cf;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
            }
            settle(tid < cn, ic, cf);
        }
        __syncthreads();
    }
#pragma unroll
    for (int l = 1; l < KCB_NLEV; ++l)
        if (nlev[l]) atomicAdd(&slev[l], nlev[l]);
    rv[tid] = besti >= 0 ? __hip_atomic_load(P.dist + besti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.0;
    ri[tid] = besti;
    __syncthreads();
    if (tid >= 1 && tid < KCB_NLEV && slev[tid]) atomicAdd(&S->lev[tid], slev[tid]);
    for (int s = DT / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const long long oi = ri[tid + s];
            if (oi >= 0 && (ri[tid] < 0 || kc_better(rv[tid + s], oi, rv[tid], ri[tid]))) {
                rv[tid] = rv[tid + s];
                ri[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        KcPartial qn;
        qn.v = rv[0];
        qn.i = ri[0];
        P.next[blockIdx.x] = qn;
    }
}

template <typename T, int JB, int KWS_R, int KWS_U>
__global__ __launch_bounds__(DT) void kwb_pass_kernel(KwsArgs P, KcbState* S)
{
    extern __shared__ __attribute__((aligned(16))) char kwb_smem[];
    __shared__ KwbShared<KWS_R> sh;
    const int J = S->J;
    if (J == 0) return;   // uniform over the grid: all K centres are fixed
    if (J <= 4) kwb_pass_body<T, JB, 4, KWS_R, KWS_U>(P, S, kwb_smem, sh);
    else if (J <= 8 || JB == 8) kwb_pass_body<T, JB, 8, KWS_R, KWS_U>(P, S, kwb_smem, sh);
    else if (J <= 12) kwb_pass_body<T, JB, JB >= 12 ? 12 : 8, KWS_R, KWS_U>(P, S, kwb_smem, sh);
    else kwb_pass_body<T, JB, JB >= 16 ? 16 : 8, KWS_R, KWS_U>(P, S, kwb_smem, sh);
}

}  // namespace msm
