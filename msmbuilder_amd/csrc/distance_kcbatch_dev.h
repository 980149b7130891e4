// distance_kcbatch_dev.h -- kcb_* kernels: several centres per pass (threshold lists), single GPU and row-sharded
// (round 5: cut out of distance.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"
#include "distance_dev.h"

namespace msm {

// ---------------------------------------------------------------------------
// Several centres per pass (round 3; byte copy, single process).
//
// k-centers is sequential -- centre k+1 is the argmax of the distances AFTER centre k -- but the argmax can usually be
// read off a short list.  A pass appends every row whose updated, rounded-up distance exceeds a threshold theta to a list
// (`curf > theta` implies distance > theta; every row NOT listed has distance <= theta =: tau, and distances only
// shrink).  kcb_select_kernel, one workgroup, then plays the algorithm on the list alone: the listed row of largest
// distance (lowest row on ties) is the next centre -- it beats every unlisted row strictly; the remaining listed rows get
// d = min(d, dist(row, centre)) in the pass kernel's exact arithmetic; the largest of them is the centre after that IF it
// still exceeds tau, and so on, up to KCB_JMAX centres.  The next pass applies them all, in order, to every row it streams
// (a row's candidate centres by the screen, then the exact `d < distances_` of kcenters.py:93 centre after centre): the
// centres, labels_ and distances_ of the one-centre-per-pass loop, in a fraction of its passes (simulated on a 10-dimensional
// projection: 30 passes instead of 199 with lists of 16).  The per-block argmax partials are still written: the first
// centre of a batch must be the row they name (numpy's argmax under this file's NaN rules), otherwise -- and whenever the
// list is empty or overflowed -- the batch is that one row.  theta follows the data: a pass also counts the rows above five
// lower levels, and the selector takes the lowest level that held at most KCB_TARGET rows (counts at a fixed level can only
// fall from pass to pass, so the next list fits).
// ---------------------------------------------------------------------------
constexpr int KCB_JMAX = 32, KCB_CAP = 2048, KCB_NLEV = 6, KCB_TARGET = 1536;
// ---- wave argmax of (value, row): largest value, lowest row among equal values; rows < 0 do not take part ---------------
// The value goes through DPP row operations and readlanes (a 64-bit __shfl_xor is two ds_bpermute round trips per step:
// the selection kernels make ~35 block reductions between two passes and were 30-45 us, most of it shuffles).
template <int CTRL>
__device__ __forceinline__ double kcb_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double kcb_readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void kcb_wave_argmax(double v, long long i, double& ov, long long& oi)
{
    const double w = i >= 0 ? v : -1.0;     // distances are >= 0; a NaN loses every fmax
    double x = w;
    x = fmax(x, kcb_dpp_f64<0xB1>(x));      // quad_perm [1,0,3,2]
    x = fmax(x, kcb_dpp_f64<0x4E>(x));      // quad_perm [2,3,0,1]
    x = fmax(x, kcb_dpp_f64<0x141>(x));     // row_half_mirror
    x = fmax(x, kcb_dpp_f64<0x140>(x));     // row_mirror: every lane holds the maximum of its row of 16
    const double vm = fmax(fmax(kcb_readlane_f64(x, 0), kcb_readlane_f64(x, 16)), fmax(kcb_readlane_f64(x, 32), kcb_readlane_f64(x, 48)));
    unsigned long long mask = __builtin_amdgcn_ballot_w64(i >= 0 && w == vm);
    if (!mask) mask = __builtin_amdgcn_ballot_w64(i >= 0);   // only NaN values took part: the lowest row, like a scan that never sees `>`
    long long best = -1;
    while (mask) {   // one lane, except on exact ties
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const long long c = ((long long)__builtin_amdgcn_readlane((int)(i >> 32), l) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(i & 0xffffffffLL), l);
        if (best < 0 || c < best) best = c;
    }
    ov = vm;
    oi = best;
}


struct KcbState {
    int k_done;               // centres fixed so far: ids[0 .. k_done)
    int J;                    // centres the next pass applies: ids[k_done - J .. k_done)
    int rounds, fallbacks;
    float theta;              // listing threshold of the last pass = bound on every row it did not list
    unsigned count;           // rows it listed (more than KCB_CAP: list unusable)
    unsigned lev[KCB_NLEV];   // rows above theta * kcb_level(l) after it; lev[0] mirrors count
    double cen[KCB_JMAX][16];
    long long list[KCB_CAP];
};
// (levels below 0.91 were tried in round 4 -- twelve levels down to 0.58: the number of rounds did not move, 19 on the bench's
//  projection at 10M and at 1.25M rows: what ends a round is the list's capacity, not the threshold's rate of descent -- and
//  the extra level counters cost 15 % of a fit)
// Rounds the host queues before it looks at the progress counter again.  A synchronisation costs 35-50 us of idle GPU, an
// empty round (all K centres fixed: three early-returning launches) about 14: so the first group aims at the whole fit at
// a typical 12 centres per round, and the later ones at what is left at the rate seen so far, plus one.  The value depends
// on nothing but K and the counter, which every rank of a sharded fit holds identically.
static int kcb_group(int K, int done, int rounds_so_far, int done_at_start)
{
    const int left = K - done;
    if (left <= 0) return 0;
    int per = 12;
    if (rounds_so_far > 0) per = std::max(1, (done - done_at_start) / rounds_so_far);
    const int g = (left + per - 1) / per + (rounds_so_far > 0 ? 1 : 0);
    return std::min(std::max(g, 1), 24);
}

// the state before the first round: `k_done` centres fixed by the plain passes, no list yet.  (A launch instead of a copy
// from the host's stack and the synchronisation that keeps the stack alive: 20-30 us per fit.)
__global__ void kcb_init_kernel(KcbState* S, int k_done)
{
    if (threadIdx.x == 0) {
        S->k_done = k_done;
        S->J = S->rounds = S->fallbacks = 0;
        S->theta = INFINITY;
        S->count = 0;
    }
    if (threadIdx.x < KCB_NLEV) S->lev[threadIdx.x] = 0;
}
__device__ __forceinline__ float kcb_level(int l) { return l == 0 ? 1.f : l == 1 ? 0.985f : l == 2 ? 0.97f : l == 3 ? 0.955f : l == 4 ? 0.94f : 0.91f; }

template <int NP>
__global__ __launch_bounds__(1024) void kcb_select_kernel(KscArgs P, KcbState* S, int K)
{
    __shared__ double rv[1024];
    __shared__ long long ri[1024];
    __shared__ double cs[16];
    const int tid = threadIdx.x, m = (int)P.m;
    const int k0 = S->k_done;
    if (k0 >= K) {
        if (tid == 0) S->J = 0;
        return;
    }
    // block argmax (largest value, lowest row on ties; rows < 0 never win): inside a wave by kcb_wave_argmax, then every wave
    // reduces the 16 wave winners by itself -- two barriers per call (the selection loop makes up to 17 calls between passes)
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
        if (tid < P.nblk) {
            const KcPartial q = P.prev[tid];
            if (q.i >= 0) {
                v = q.v;
                i = q.i;
            }
        }
        reduce(v, i, vP, iP);
    }
    const float theta = S->theta;
    const unsigned cnt = S->count;
    const bool usable = cnt > 0 && cnt <= (unsigned)KCB_CAP && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    // this thread's (up to) two listed rows: index, current distance, coordinates
    long long ci[2] = {-1, -1};
    double cv[2] = {-1.0, -1.0}, cx[2][2 * NP];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned c = (unsigned)tid + 1024u * u;
        if (usable && c < cnt) {
            ci[u] = S->list[c];
            cv[u] = P.dist[ci[u]];
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) cx[u][f] = f < m ? P.X[ci[u] * P.m + f] : 0.0;
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
        if (fell) {
            if (tid < 16) cs[tid] = tid < m ? P.X[centre * P.m + tid] : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ci[u] == centre) {   // the thread that holds the row: no trip to global memory inside the loop
#pragma unroll
                    for (int f = 0; f < 16; ++f) cs[f] = f < 2 * NP ? cx[u][f] : 0.0;
                }
        }
        if (tid == 0) P.ids[k0 + J] = centre;
        __syncthreads();
        if (tid < 16) S->cen[J][tid] = cs[tid];
        ++J;
        if (fell || k0 + J >= K || J >= KCB_JMAX) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ci[u] < 0) continue;
            if (ci[u] == centre) {
                ci[u] = -1;
                continue;
            }
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, cx[u][f], cs[f]);
            const double d = m_final<M_EUCLIDEAN>(a, b, P.m);
            if (d < cv[u]) cv[u] = d;   // the pass's own update (kcenters.py:93)
        }
        __syncthreads();
    }
    if (tid == 0) {
        // threshold of the next list
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (cnt > (unsigned)KCB_TARGET) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (S->lev[q] <= (unsigned)KCB_TARGET) l = q;
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

// ---- row-sharded fit: the same rounds with ONE exchange per round -----------------------------------------------------
// A rank's round record (doubles): [0] rows it listed (more than KCB_CAPR: list unusable), [8] value and [9] GLOBAL row of its
// per-block-partials argmax (-1: none), [10..25] that row's coordinates, [32 + l] the count of level l, then from [KCB_HDR]
// on the listed rows as {distance, global row, coordinates[m]}.  The records are all-gathered and every rank runs the same selection on
// the same numbers: no rank learns anything another does not, so the batches -- and the number of rounds -- agree.
constexpr int KCB_CAPR = 1024, KCB_HDR = 48, KCB_LEV0 = 32;
__host__ __device__ constexpr size_t kcb_rec_doubles(long long m) { return (size_t)KCB_HDR + (size_t)KCB_CAPR * (size_t)(2 + m); }

// the records of the first round, made from the one-centre protocol's gathered candidates {value, global row, coordinates}
__global__ void kcb_boot_records_kernel(const double* __restrict__ cands, int world, long long m, double* __restrict__ recs)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= world) return;
    const double* c = cands + (size_t)r * (2 + m);
    double* o = recs + (size_t)r * kcb_rec_doubles(m);
    if (tid < KCB_HDR) {
        double v = 0.0;
        if (tid == 8) v = c[0];
        else if (tid == 9) v = c[1];
        else if (tid >= 10 && tid < 10 + 16) v = tid - 10 < m ? c[2 + tid - 10] : 0.0;
        o[tid] = v;
    }
}

// the shard's record of a round.  (A launch of its own: folding it into the pass kernel -- the last workgroup to arrive packs
// -- was tried in round 4 and cost 24 us per pass instead of the 7 + 4 us of this launch: the agent-scope release that
// every one of the pass's ~5,000 workgroups must then make before it counts itself in is an L2 write-back each.)
template <int NP>
__global__ __launch_bounds__(DT) void kcb_pack_kernel(KscArgs P, KcbState* S, double* __restrict__ rec)
{
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    const int tid = threadIdx.x, m = (int)P.m;
    if (S->J == 0) return;   // an empty round: nobody reads the record
    double bv = -1.0;
    long long bi = -1;
    for (int k = tid; k < P.nblk; k += DT) {
        const KcPartial q = P.next[k];
        if (q.i >= 0 && (bi < 0 || kc_better(q.v, q.i, bv, bi))) {
            bv = q.v;
            bi = q.i;
        }
    }
    rv[tid] = bv;
    ri[tid] = bi;
    __syncthreads();
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (tid < k) {
            const long long oi = ri[tid + k];
            if (oi >= 0 && (ri[tid] < 0 || kc_better(rv[tid + k], oi, rv[tid], ri[tid]))) {
                rv[tid] = rv[tid + k];
                ri[tid] = oi;
            }
        }
        __syncthreads();
    }
    const long long w = ri[0];
    const unsigned cnt = S->count;
    if (tid < KCB_HDR) {
        double v = 0.0;
        if (tid == 0) v = (double)cnt;
        else if (tid > KCB_LEV0 && tid < KCB_LEV0 + KCB_NLEV) v = (double)S->lev[tid - KCB_LEV0];
        else if (tid == 8) v = w >= 0 ? rv[0] : -1.0;
        else if (tid == 9) v = w >= 0 ? (double)(P.row_offset + w) : -1.0;
        else if (tid >= 10 && tid < 26) v = (w >= 0 && tid - 10 < m) ? P.X[w * P.m + (tid - 10)] : 0.0;
        rec[tid] = v;
    }
    const unsigned ne = cnt <= (unsigned)KCB_CAPR ? cnt : 0u;
    for (unsigned e = tid; e < ne; e += DT) {
        const long long p = S->list[e];
        double* o = rec + KCB_HDR + (size_t)e * (2 + m);
        o[0] = P.dist[p];
        o[1] = (double)(P.row_offset + p);
        for (int f = 0; f < m; ++f) o[2 + f] = P.X[p * P.m + f];
    }
}

template <int NP>
__global__ __launch_bounds__(DT) void kcenters_batch_pass_kernel(KscArgs P, KcbState* S)
{
    constexpr int R = 2;
    constexpr int NW = ksc_words(NP, 2), RW = NW + 1;
    constexpr int SB = 2 * NP;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) f32x2 ycf[KCB_JMAX][NP];
    __shared__ double yd[KCB_JMAX][2 * NP];
    __shared__ float epsb[KCB_JMAX];
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    __shared__ unsigned slev[KCB_NLEV];
    const int tid = threadIdx.x, m = (int)P.m;
    const int J = S->J;
    if (J == 0) return;
    const int kbase = S->k_done - J;
    const float theta = S->theta;
    const long long ntile = (P.n + (long long)R * DT - 1) / ((long long)R * DT);
    unsigned qn[R][RW];
    auto load_tile = [&](long long t) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long p0 = t * (R * DT) + k * DT + tid;
            const long long pc = p0 < P.n ? p0 : P.n - 1;
            const unsigned* xr = static_cast<const unsigned*>(P.xs) + pc * RW;
            if ((RW & 3) == 0) {
#pragma unroll
                for (int j = 0; j < RW / 4; ++j) {
                    const uint4 v = reinterpret_cast<const uint4*>(xr)[j];
                    qn[k][4 * j] = v.x;
                    qn[k][4 * j + 1] = v.y;
                    qn[k][4 * j + 2] = v.z;
                    qn[k][4 * j + 3] = v.w;
                }
            } else if ((RW & 1) == 0) {
#pragma unroll
                for (int j = 0; j < RW / 2; ++j) {
                    const uint2 v = reinterpret_cast<const uint2*>(xr)[j];
                    qn[k][2 * j] = v.x;
                    qn[k][2 * j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < RW; ++j) qn[k][j] = xr[j];
            }
        }
    };
    if ((long long)blockIdx.x < ntile) load_tile(blockIdx.x);
    // the batch's centres: exact coordinates, float32 coordinates relative to the copy's origin, and the part of eps that
    // belongs to the centre (see kcenters_screen_pass_kernel for the terms)
    if (tid < KCB_NLEV) slev[tid] = 0;
    for (int e = tid; e < KCB_JMAX * 2 * NP; e += DT) {
        const int j = e / (2 * NP), f = e - j * (2 * NP);
        const double y = j < J ? S->cen[j][f] : 0.0;
        yd[j][f] = y;
        reinterpret_cast<float*>(&ycf[j][0])[f] = (float)(y - (f < m ? P.c0[f] : 0.0));
    }
    if (tid < KCB_JMAX) {
        double c0n2 = 0.0, yn2 = 0.0, ycn2 = 0.0;
        for (int f = 0; f < 2 * NP; ++f) {
            const double y = tid < J ? S->cen[tid][f] : 0.0, c0f = f < m ? P.c0[f] : 0.0;
            c0n2 = fma(c0f, c0f, c0n2);
            yn2 = fma(y, y, yn2);
            ycn2 = fma(y - c0f, y - c0f, ycn2);
        }
        const double g2 = __longlong_as_double((long long)P.gmax2[0]), r2 = __longlong_as_double((long long)P.gmax2[1]);
        double eps0 = (sqrt(r2) + sqrt(yn2) + 2.0 * sqrt(c0n2)) * 0x1p-48 + 1e-37;
        if (!(g2 < 1e36) || !(r2 < 1e76) || !(ycn2 < 1e36)) eps0 = NAN;
        epsb[tid] = fmaf(0x1p-19f, (float)(sqrt(ycn2) * 1.000001), (float)(eps0 * 1.000001));
    }
    __syncthreads();
    constexpr float E32 = 0x1p-19f;
    constexpr float QSQ = NP == 1 ? 1.4143f : NP == 2 ? 2.f : NP == 3 ? 2.4495f : NP == 4 ? 2.8285f : NP == 5 ? 3.1623f
                        : NP == 6 ? 3.4642f : NP == 7 ? 3.7417f : 4.f;
    constexpr float QA = 0.51f * 1.02f * QSQ + E32 * 127.f * QSQ * 1.001f;
    float bf = -1.f;
    long long bi = -1;
    double bx = 0.0;
    bool bknown = false;
    unsigned nlev[KCB_NLEV];
#pragma unroll
    for (int l = 0; l < KCB_NLEV; ++l) nlev[l] = 0;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        float cf[R];
        unsigned cmask[R];
        long long pr[R];
        unsigned q[R][RW];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            pr[k] = t * (R * DT) + k * DT + tid;
#pragma unroll
            for (int j = 0; j < RW; ++j) q[k][j] = qn[k][j];
            cf[k] = __uint_as_float(q[k][NW]);
        }
        if (t + gridDim.x < ntile) load_tile(t + gridDim.x);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const float sf = __uint_as_float(((q[k][SB >> 2] >> (8 * (SB & 3))) & 0xffffu) << 16);
            f32x2 xt[NP];
#pragma unroll
            for (int g = 0; g < NP; ++g) {
                xt[g].x = (float)((int)(q[k][(2 * g) >> 2] << (24 - 8 * ((2 * g) & 3))) >> 24) * sf;       // q sf: exact
                xt[g].y = (float)((int)(q[k][(2 * g + 1) >> 2] << (24 - 8 * ((2 * g + 1) & 3))) >> 24) * sf;
            }
            // a row is left alone by centre j when  sqrt(a_j) - eps_j >= curf.  Compared as squares, without the square root:
            // a_j >= T^2 with T = (curf + eps_j)(1 + 2^-20) evaluated in float32 (three roundings of 2^-24 each, and one more
            // in the product T T, leave T^2 above the real (curf + eps_j)^2): the real-arithmetic inequality with room to
            // spare -- eps_j already allows for float32 roundings of the original form
            const float base = cf[k] + sf * QA;
            unsigned mk = 0;
            for (int j = 0; j < J; ++j) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int g = 0; g < NP; ++g) {
                    const f32x2 d = xt[g] - ycf[j][g];
                    acc = __builtin_elementwise_fma(d, d, acc);
                }
                const float T = (base + epsb[j]) * (1.f + 0x1p-20f);
                if (!(acc.x + acc.y >= T * T)) mk |= 1u << j;
            }
            cmask[k] = pr[k] < P.n ? mk : 0u;
        }
        if (cmask[0] | cmask[1]) {
            double x[R][2 * NP], cur[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const long long pc = pr[k] < P.n ? pr[k] : P.n - 1;
                const double* xp = P.X + pc * P.m;
                if (P.vecw == 16 && (m & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const raw_f32x4 v = *reinterpret_cast<const raw_f32x4*>(xp + 2 * j);
                        x[k][2 * j] = reinterpret_cast<const double*>(&v)[0];
                        x[k][2 * j + 1] = reinterpret_cast<const double*>(&v)[1];
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) x[k][f] = xp[f < m ? f : m - 1];
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f)
                        if (f >= m) x[k][f] = 0.0;
                }
                cur[k] = P.dist[pc];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                unsigned mk = cmask[k];
                int lab = -1;
                double c = cur[k];
                while (mk) {   // the batch's centres in order, as the separate passes would meet the row
                    const int j = __builtin_ctz(mk);
                    mk &= mk - 1;
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, x[k][f], yd[j][f]);
                    const double d = m_final<M_EUCLIDEAN>(a, b, P.m);
                    if (d < c) {   // strict, kcenters.py:93
                        c = d;
                        lab = kbase + j;
                    }
                }
                if (lab >= 0) {
                    P.dist[pr[k]] = c;
                    P.labels[pr[k]] = lab;
                    cf[k] = ksc_round_up(c);
                    static_cast<unsigned*>(P.xs)[pr[k] * RW + NW] = __float_as_uint(cf[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const long long p = pr[k];
            const bool in = p < P.n;
            // the list of the next selection, and the level counts that place its threshold
            const bool lst = in && cf[k] > theta;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(lst);
            if (bal) {
                const int lane = tid & 63, leader = __builtin_ctzll(bal);
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(&S->count, (unsigned)__builtin_popcountll(bal));
                base = __shfl(base, leader);
                if (lst) {
                    const unsigned slot = base + (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                    if (slot < (unsigned)KCB_CAP) S->list[slot] = p;
                }
            }
#pragma unroll
            for (int l = 1; l < KCB_NLEV; ++l) nlev[l] += (in && cf[k] > theta * kcb_level(l)) ? 1u : 0u;
            if (in) {
                if (cf[k] > bf || bi < 0) {
                    bf = cf[k];
                    bi = p;
                    bknown = false;
                } else if (cf[k] == bf) {
                    if (!bknown) {
                        bx = P.dist[bi];
                        bknown = true;
                    }
                    const double v = P.dist[p];
                    if (v > bx) {
                        bx = v;
                        bi = p;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int l = 1; l < KCB_NLEV; ++l)
        if (nlev[l]) atomicAdd(&slev[l], nlev[l]);
    double bvx = -1.0;
    if (bi >= 0) bvx = bknown ? bx : P.dist[bi];
    rv[tid] = bvx;
    ri[tid] = bi;
    __syncthreads();
    if (tid >= 1 && tid < KCB_NLEV && slev[tid]) atomicAdd(&S->lev[tid], slev[tid]);
    for (int k = DT / 2; k > 0; k >>= 1) {
        if (tid < k) {
            const long long oi = ri[tid + k];
            if (oi >= 0 && (ri[tid] < 0 || kc_better(rv[tid + k], oi, rv[tid], ri[tid]))) {
                rv[tid] = rv[tid + k];
                ri[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        KcPartial q;
        q.v = rv[0];
        q.i = ri[0];
        P.next[blockIdx.x] = q;
    }
}

// a rank without rows: nothing listed, no argmax
__global__ void kcb_empty_record_kernel(double* __restrict__ rec)
{
    if (threadIdx.x < KCB_HDR) rec[threadIdx.x] = (threadIdx.x == 8 || threadIdx.x == 9) ? -1.0 : 0.0;
}

template <int NP>
__global__ __launch_bounds__(1024) void kcb_select_sharded_kernel(const double* __restrict__ recs, int world, long long mm, KcbState* S, int K,
                                                                   double* __restrict__ cen_out, msm_idx_t* __restrict__ ids_out)
{
    __shared__ double rv[1024];
    __shared__ long long ri[1024];
    __shared__ double cs[16];
    const int tid = threadIdx.x, m = (int)mm;
    const size_t RD = kcb_rec_doubles(mm);
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
    // the row the one-centre protocol would take: best of the ranks' own argmax records (value, lowest GLOBAL row on ties)
    double vP;
    long long iP;
    int rP = -1;
    {
        double v = -1.0;
        long long i = -1;
        if (tid < world) {
            const double* h = recs + (size_t)tid * RD;
            if (h[9] >= 0.0) {
                v = h[8];
                i = (long long)h[9];
            }
        }
        reduce(v, i, vP, iP);
        for (int r = 0; r < world; ++r)
            if (iP >= 0 && (long long)recs[(size_t)r * RD + 9] == iP) rP = r;
    }
    // union of the ranks' lists, level counts summed
    unsigned total = 0, truecount = 0;
    bool fits = true;
    unsigned lev[KCB_NLEV];
#pragma unroll
    for (int q = 0; q < KCB_NLEV; ++q) lev[q] = 0;
    for (int r = 0; r < world; ++r) {
        const double* h = recs + (size_t)r * RD;
        const unsigned c = (unsigned)h[0];
        truecount += c;
        if (c > (unsigned)KCB_CAPR) fits = false;
        else total += c;
#pragma unroll
        for (int q = 1; q < KCB_NLEV; ++q) lev[q] += (unsigned)h[KCB_LEV0 + q];
    }
    const float theta = S->theta;
    const bool usable = fits && total > 0 && total <= (unsigned)KCB_CAP && theta > 0.f && theta < 3e38f;
    const double tau = (double)theta;
    long long ci[2] = {-1, -1};
    double cv[2] = {-1.0, -1.0}, cx[2][2 * NP];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        unsigned c = (unsigned)tid + 1024u * u;
        if (usable && c < total) {
            int r = 0;
            for (; r < world; ++r) {
                const unsigned cr = (unsigned)recs[(size_t)r * RD];
                if (c < cr) break;
                c -= cr;
            }
            const double* e = recs + (size_t)r * RD + KCB_HDR + (size_t)c * (2 + m);
            cv[u] = e[0];
            ci[u] = (long long)e[1];
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) cx[u][f] = f < m ? e[2 + f] : 0.0;
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
        if (fell) {
            if (tid < 16) cs[tid] = (rP >= 0 && tid < m) ? recs[(size_t)rP * RD + 10 + tid] : 0.0;
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ci[u] == centre) {
#pragma unroll
                    for (int f = 0; f < 16; ++f) cs[f] = f < 2 * NP ? cx[u][f] : 0.0;
                }
        }
        if (tid == 0) ids_out[k0 + J] = centre;
        __syncthreads();
        if (tid < 16) S->cen[J][tid] = cs[tid];
        if (tid < m) cen_out[(size_t)(k0 + J) * m + tid] = cs[tid];
        ++J;
        if (fell || k0 + J >= K || J >= KCB_JMAX) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (ci[u] < 0) continue;
            if (ci[u] == centre) {
                ci[u] = -1;
                continue;
            }
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int f = 0; f < 2 * NP; ++f) m_update<double, M_EUCLIDEAN>(a, b, cx[u][f], cs[f]);
            const double d = m_final<M_EUCLIDEAN>(a, b, mm);
            if (d < cv[u]) cv[u] = d;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float th;
        const float vl = ksc_round_up(vlast > 0.0 ? vlast : 0.0);
        if (!(theta > 0.f) || !(theta < 3e38f)) {
            th = 0.97f * vl;
        } else if (truecount > (unsigned)KCB_CAPR) {
            th = theta * 1.02f;
        } else {
            int l = 0;
            for (int q = 1; q < KCB_NLEV; ++q)
                if (lev[q] <= (unsigned)KCB_CAPR) l = q;
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

// c0 = coordinates of the first centre (ids[0]), for the copy's origin
// (sharded fit: `centre0` = the first centre's coordinates as selected from the exchanged records -- it may be another rank's row)
__global__ void ksc_origin_kernel(const double* __restrict__ X, const msm_idx_t* __restrict__ ids, long long m, double* __restrict__ c0,
                                  const double* __restrict__ centre0)
{
    if (threadIdx.x < 16) c0[threadIdx.x] = threadIdx.x < m ? (centre0 ? centre0[threadIdx.x] : X[ids[0] * m + threadIdx.x]) : 0.0;
}

}  // namespace msm
