// distance_wscreen_dev.h -- k-centers passes of WIDE rows (and of any float32 rows) screened on a feature-major byte copy.
// (round 6; included by distance.hip after distance_screen_dev.h)
//
// What was there: the screened passes of rounds 2-5 (distance_screen_dev.h, distance_kcbatch_dev.h) keep a row's byte copy in
// a thread's registers and exist for float64 rows of <= 16 features.  Everything else took one PLAIN pass per centre over the
// rows themselves: KCenters(200) on 280,000 x 171 float32 (SURVEY 8(d)'s C3 stress shape) 200 passes over 191 MB = 10.7 ms,
// KCenters(500) on 1M x 171 78 ms, and float64 rows of 17 features ran 5 x slower than rows of 16
// (profiles/r05_cluster_probe.txt).
//
// Same idea, a layout that scales with the row length: at the switch-over the rows are copied once, centred on the first
// centre c0, as signed bytes q_j with one bfloat16 scale per row (x~_j = q_j sf exactly, sf >= max_j |x_j - c0_j| / 127),
// FEATURE-major: word plane b holds features 4b .. 4b + 3 of every row, so a thread that owns a row reads one coalesced
// 4-byte word per four features -- m + 8 bytes per row and pass (the bytes, the scale, the rounded-up distance) instead of
// m sizeof(T) + 16.  A later pass evaluates d~ = || x~ - (y - c0) || in float32 and proves "no update" for a row when
//     (d~ - E) (1 - 2^-22) >= curf >= distances_,     E = 0.5001 sqrt(m) sf + (d~ + G) (m + 8) 2^-22
// where 0.5 sqrt(m) sf bounds || x~ - (x - c0) || (round to nearest of every feature), the second term bounds the float32
// arithmetic of the pass (m positive terms: relative (m + 2) 2^-24 on d~^2; the centre's coordinates rounded to float32:
// 2^-24 ||y - c0|| <= 2^-24 G, G = max_i ||x_i - c0||; the fp64 centring of x and y: 2^-52), and (1 - 2^-22) covers the reference's
// own float32 subtraction per feature (distance_kernels.h: relative 2^-24 per term; float64 rows: nothing to cover).  Every
// other row -- a few per cent -- is a CANDIDATE and is re-evaluated from its own coordinates with the reference's arithmetic
// (one fp64 accumulator, features in order, sqrt, strict <): labels_ / distances_ / the centre ids are bit-identical to the
// plain passes'.  Non-finite data or data beyond the float32 range make E NaN: every row is a candidate, the pass is exact.
// The next centre's argmax runs on curf (a monotone rounding of distances_: curf_i > curf_j proves distances_i >
// distances_j) and looks at the float64 values only on exact float32 ties, as in kcenters_screen_pass_kernel.
#pragma once
#include "common.h"
#include "distance_dev.h"
#include "distance_screen_dev.h"

namespace msm {

struct KwsArgs {
    const void* X;               // the rows (T), as the plain passes read them
    unsigned* q;                 // [nb4][n] byte planes: word b of row i = signed bytes of features 4b .. 4b + 3
    unsigned short* sf;          // [n] the row's scale as bfloat16 bits
    float* curf;                 // [n] distances_ rounded UP to float32
    unsigned long long* gmax2;   // [0]: bits of max_i ||x_i - c0||^2
    const double* c0;            // [m] the copy's origin (the first centre)
    long long n, m;
    int nb4, it, nblk;
    const KcPartial* prev;
    KcPartial* next;
    double* dist;
    msm_idx_t* labels;
    msm_idx_t* ids;
    unsigned long long* stats;   // [0] candidates re-evaluated, [1] rows updated (diagnostics)
};

template <typename T>
__global__ void kws_origin_kernel(const T* __restrict__ X, const msm_idx_t* __restrict__ ids, long long m, double* __restrict__ c0,
                                  unsigned long long* __restrict__ gmax2)
{
    const long long row = ids[0];
    for (long long f = threadIdx.x; f < m; f += blockDim.x) c0[f] = (double)X[row * m + f];
    if (threadIdx.x == 0) gmax2[0] = 0ull;
}

// rows -> byte planes + scale + rounded-up distance; G^2 = max ||x - c0||^2.  Once per fit, one streaming pass: a WAVE reads a
// row (lanes over the features: coalesced), reduces its largest |x_j - c0_j| and its squared norm, quantises its own features
// and leaves the row's words in LDS; the workgroup then writes every plane's words of its rows as one contiguous segment.
// (First version: a thread per row reading it twice, 4 bytes at a time at a 688-byte stride -- 2.6 ms per 1M x 171.)
constexpr int KWS_CROWS = 64;          // rows per workgroup (fewer when the rows are long: KWS_CWORDS of LDS)
constexpr int KWS_CWORDS = 64 * 65;    // LDS words of the transposition tile

template <typename T>
__global__ __launch_bounds__(DT) void kws_convert_kernel(KwsArgs P, int rows_per_block)
{
    extern __shared__ double c0s[];   // [m]
    __shared__ unsigned words[KWS_CWORDS];   // [rows_per_block][nb4 + 1]
    __shared__ double gred[DT / 64];
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = P.nb4 + 1;
    for (long long f = tid; f < P.m; f += DT) c0s[f] = P.c0[f];
    __syncthreads();
    double gmax = 0.0;
    for (long long r0 = (long long)blockIdx.x * rows_per_block; r0 < P.n; r0 += (long long)gridDim.x * rows_per_block) {
        const int nr = (int)(P.n - r0 < rows_per_block ? P.n - r0 : rows_per_block);
        // (TWO rows per wave and trip: a row is two dependent trips to memory -- one row at a time the kernel ran at 0.85 TB/s)
        for (int rr = wave; rr < nr; rr += 2 * (DT / 64)) {
            constexpr int NRW = 2;
            int rw[NRW];
            const T* xw[NRW];
            bool on[NRW];
#pragma unroll
            for (int u = 0; u < NRW; ++u) {
                rw[u] = rr + u * (DT / 64);
                on[u] = rw[u] < nr;
                xw[u] = X + (r0 + (on[u] ? rw[u] : rr)) * P.m;
            }
            // pass 1 over the rows (registers hold nothing: the second reading below hits the L1 / L2 lines of the first)
            float mx[NRW];
            double n2[NRW];
            bool bad[NRW];
#pragma unroll
            for (int u = 0; u < NRW; ++u) {
                mx[u] = 0.f;
                n2[u] = 0.0;
                bad[u] = false;
            }
            for (long long f = lane; f < P.m; f += 64) {
                double d[NRW];
#pragma unroll
                for (int u = 0; u < NRW; ++u) d[u] = (double)xw[u][f] - c0s[f];
#pragma unroll
                for (int u = 0; u < NRW; ++u) {
                    n2[u] = fma(d[u], d[u], n2[u]);
                    const float a = fabsf((float)d[u]);
                    mx[u] = a > mx[u] ? a : mx[u];
                    bad[u] |= !(d[u] == d[u]);
                }
            }
#pragma unroll
            for (int msk = 32; msk > 0; msk >>= 1) {
#pragma unroll
                for (int u = 0; u < NRW; ++u) {
                    const float om = __shfl_xor(mx[u], msk, 64);
                    mx[u] = om > mx[u] ? om : mx[u];
                    n2[u] += __shfl_xor(n2[u], msk, 64);
                    bad[u] |= (bool)__shfl_xor((int)bad[u], msk, 64);
                }
            }
            float inv[NRW];
            unsigned short sbw[NRW];
#pragma unroll
            for (int u = 0; u < NRW; ++u) {
                if (bad[u] || !(n2[u] == n2[u])) {
                    mx[u] = NAN;
                    n2[u] = NAN;
                }
                if (on[u]) gmax = (n2[u] > gmax || !(n2[u] == n2[u])) ? n2[u] : gmax;
                // the row's scale: the smallest bfloat16 >= mx / 127 (a zero row: any scale)
                float s = mx[u] / 127.f;
                if (!(s > 0.f)) s = (s == s) ? 1.f : s;   // zero row -> 1; NaN stays NaN
                s = s * 1.0000002f;                        // (the division rounded to nearest: stay on the safe side of mx / 127)
                sbw[u] = (unsigned short)ksc_bf16_up(s);
                const float sfv = __uint_as_float((unsigned)sbw[u] << 16);
                inv[u] = 1.f / sfv;
            }
            for (int b = lane; b < P.nb4; b += 64) {
                unsigned w[NRW];
#pragma unroll
                for (int u = 0; u < NRW; ++u) w[u] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long f = 4LL * b + e;
                    if (f < P.m) {
#pragma unroll
                        for (int u = 0; u < NRW; ++u) {
                            const float d = (float)((double)xw[u][f] - c0s[f]);
                            float q = rintf(d * inv[u]);
                            q = q > 127.f ? 127.f : (q < -127.f ? -127.f : q);   // (cannot bind for finite rows: sf >= mx / 127)
                            const int qv = (q == q) ? (int)q : 0;
                            w[u] |= ((unsigned)qv & 0xffu) << (8 * e);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NRW; ++u)
                    if (on[u]) words[rw[u] * pitch + b] = w[u];
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < NRW; ++u)
                    if (on[u]) {
                        P.sf[r0 + rw[u]] = sbw[u];
                        P.curf[r0 + rw[u]] = ksc_round_up(P.dist[r0 + rw[u]]);
                    }
            }
        }
        __syncthreads();
        // plane b of these rows: nr consecutive words
        for (int e = tid; e < P.nb4 * rows_per_block; e += DT) {
            const int b = e / rows_per_block, r = e - b * rows_per_block;
            if (r < nr) P.q[(size_t)b * P.n + r0 + r] = words[r * pitch + b];
        }
        __syncthreads();
    }
    // maximum of the squared norms (NaN poisons it on purpose) -> one atomic per block
    if (lane == 0) gred[wave] = gmax;
    __syncthreads();
    if (tid == 0) {
        double g = gred[0];
        for (int w = 1; w < DT / 64; ++w)
            if (gred[w] > g || !(gred[w] == gred[w])) g = gred[w];
        if (!(g == g)) g = INFINITY;   // (non-negative doubles order like their bit patterns; +inf is the largest)
        atomicMax(P.gmax2, (unsigned long long)__double_as_longlong(g));
    }
}

// Short rows (up to 192 bytes): a THREAD per row.  With a wave per row only m of 64 lanes work and every row is two dependent trips
// to memory: 2M x 17 float64 took 0.99 ms (0.33 TB/s), a fifth of the whole fit (profiles/r06_kcenters_wide.txt).  A thread
// walks its own row twice (the lanes of a wave cover 64 consecutive rows: whole cache lines, the second walk out of the
// L1 / L2) and writes its words of the planes coalesced -- no transposition through LDS.  Same scale, same bytes, same
// rounded-up distance as kws_convert_kernel (the maximum is order-independent; the squared norm only feeds the bound G).
template <typename T>
__global__ __launch_bounds__(DT) void kws_convert_rowthread_kernel(KwsArgs P)
{
    extern __shared__ double c0s[];   // [m]
    __shared__ double gred[DT / 64];
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (long long f = tid; f < P.m; f += DT) c0s[f] = P.c0[f];
    __syncthreads();
    double gmax = 0.0;
    for (long long i = (long long)blockIdx.x * DT + tid; i < P.n; i += (long long)gridDim.x * DT) {
        const T* x = X + i * P.m;
        float mx = 0.f;
        double n2 = 0.0;
        bool bad = false;
        for (long long f = 0; f < P.m; ++f) {
            const double d = (double)x[f] - c0s[f];
            n2 = fma(d, d, n2);
            const float a = fabsf((float)d);
            mx = a > mx ? a : mx;
            bad |= !(d == d);
        }
        if (bad || !(n2 == n2)) {
            mx = NAN;
            n2 = NAN;
        }
        gmax = (n2 > gmax || !(n2 == n2)) ? n2 : gmax;
        float s = mx / 127.f;
        if (!(s > 0.f)) s = (s == s) ? 1.f : s;   // zero row -> 1; NaN stays NaN
        s = s * 1.0000002f;
        const unsigned short sb = (unsigned short)ksc_bf16_up(s);
        const float sfv = __uint_as_float((unsigned)sb << 16);
        const float inv = 1.f / sfv;
        for (int b = 0; b < P.nb4; ++b) {
            unsigned w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long f = 4LL * b + e;
                int qv = 0;
                if (f < P.m) {
                    const float d = (float)((double)x[f] - c0s[f]);
                    float q = rintf(d * inv);
                    q = q > 127.f ? 127.f : (q < -127.f ? -127.f : q);
                    qv = (q == q) ? (int)q : 0;
                }
                w |= ((unsigned)qv & 0xffu) << (8 * e);
            }
            P.q[(size_t)b * P.n + i] = w;
        }
        P.sf[i] = sb;
        P.curf[i] = ksc_round_up(P.dist[i]);
    }
    // maximum of the squared norms (NaN poisons it on purpose) -> one atomic per block
    for (int msk = 32; msk > 0; msk >>= 1) {
        const double o = __shfl_xor(gmax, msk, 64);
        gmax = (o > gmax || !(o == o)) ? o : gmax;
    }
    if (lane == 0) gred[wave] = gmax;
    __syncthreads();
    if (tid == 0) {
        double g = gred[0];
        for (int w = 1; w < DT / 64; ++w)
            if (gred[w] > g || !(gred[w] == gred[w])) g = gred[w];
        if (!(g == g)) g = INFINITY;
        atomicMax(P.gmax2, (unsigned long long)__double_as_longlong(g));
    }
}

// KWS_R rows per thread and super-tile, KWS_U byte planes loaded together (template parameters): KWS_R x KWS_U = 32 words in
// flight per thread.  A pass is a chain of memory round trips per super-tile (planes, candidate rows, distances_): short rows
// take many rows per thread (few super-tiles per workgroup), long rows many planes per trip.
constexpr int KWS_RMAX = 8;
constexpr int KWS_CB = 32;      // candidates re-evaluated together (their rows staged in LDS)
constexpr int KWS_CFLOATS = 6144;   // LDS floats (or doubles / 2) of that staging area: KWS_CB rows of up to 191 floats, fewer rows when longer

// the reference's distance of a row to a centre -- one fp64 accumulator, features in order (m_update / m_final) -- with the
// operands fetched 16 bytes at a time, two fetches ahead of the chain (x and y 16-byte aligned, LDS or global).  Element by
// element (a chain that waits for an LDS round trip per feature) this was ~10 us per row and centre at 171 features.
template <typename T>
__device__ __forceinline__ double kws_exact_euclid(const T* x, const T* y, long long m)
{
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    double a = 0.0, b = 0.0;
    long long f = 0;
    for (; f + 2 * V <= m; f += 2 * V) {
        const vec_t x0 = *reinterpret_cast<const vec_t*>(x + f), x1 = *reinterpret_cast<const vec_t*>(x + f + V);
        const vec_t y0 = *reinterpret_cast<const vec_t*>(y + f), y1 = *reinterpret_cast<const vec_t*>(y + f + V);
#pragma unroll
        for (int e = 0; e < V; ++e) m_update<T, M_EUCLIDEAN>(a, b, x0[e], y0[e]);
#pragma unroll
        for (int e = 0; e < V; ++e) m_update<T, M_EUCLIDEAN>(a, b, x1[e], y1[e]);
    }
    for (; f < m; ++f) m_update<T, M_EUCLIDEAN>(a, b, x[f], y[f]);
    return m_final<M_EUCLIDEAN>(a, b, m);
}

template <typename T, int KWS_R, int KWS_U>
__global__ __launch_bounds__(DT) void kcenters_wscreen_pass_kernel(KwsArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char kws_smem[];
    float* ycf = reinterpret_cast<float*>(kws_smem);                 // [4 nb4] the centre relative to c0, float32, zero padded
    T* yraw = reinterpret_cast<T*>(kws_smem + (size_t)4 * P.nb4 * sizeof(float));   // [m] the centre itself (exact re-evaluation)
    __shared__ double rv[DT];
    __shared__ long long ri[DT];
    __shared__ int cand[KWS_R * DT];
    __shared__ int ncand;
    __shared__ __attribute__((aligned(16))) float cstage[KWS_CFLOATS];   // candidate rows, pitch m + 1 elements of T
    const T* X = static_cast<const T*>(P.X);
    const int tid = threadIdx.x;
    const long long n = P.n, m = P.m;

    // ---- prologue: centre of this pass = argmax of the previous pass's per-block partials (P.it >= 1 here) ----
    {
        double fv = -1.0;
        long long fi = 0x7fffffffffffffffLL;
        KcPartial qp[KC_MAXBLK / DT];
#pragma unroll
        for (int j = 0; j < KC_MAXBLK / DT; ++j) {
            const int k = tid + j * DT;
            qp[j] = P.prev[k < P.nblk ? k : P.nblk - 1];
        }
#pragma unroll
        for (int j = 0; j < KC_MAXBLK / DT; ++j) {
            const int k = tid + j * DT;
            if (k < P.nblk && qp[j].i >= 0 && kc_better(qp[j].v, qp[j].i, fv, fi)) {
                fv = qp[j].v;
                fi = qp[j].i;
            }
        }
        rv[tid] = fv;
        ri[tid] = fi;
        __syncthreads();
        for (int k = DT / 2; k > 0; k >>= 1) {
            if (tid < k && kc_better(rv[tid + k], ri[tid + k], rv[tid], ri[tid])) {
                rv[tid] = rv[tid + k];
                ri[tid] = ri[tid + k];
            }
            __syncthreads();
        }
    }
    const long long cidx = ri[0];
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) P.ids[P.it] = cidx;
    for (long long f = tid; f < 4LL * P.nb4; f += DT) {
        const T yv = f < m ? X[cidx * m + f] : (T)0;
        if (f < m) yraw[f] = yv;
        ycf[f] = f < m ? (float)((double)yv - P.c0[f]) : 0.f;
    }
    __syncthreads();
    const double G = sqrt(__longlong_as_double((long long)P.gmax2[0]));
    const float Gf = (float)G * 1.0000002f;                       // rounded up (inf / NaN stay what they are)
    const float ea = (float)(m + 8) * 2.3841858e-07f;              // (m + 8) 2^-22
    const float eq = 0.5001f * sqrtf((float)m);                    // >= 0.5 sqrt(m) with float32 roundings to spare
    const float refl = (sizeof(T) == 4) ? (1.f - 2.3841858e-07f) : 1.f;   // the reference's own float32 subtraction (float rows)
    // candidate staging: rows of pitch m + 1 (odd or not, lanes of a wave then hit different banks), as many as fit
    const int cpitch = (int)(((((long long)m * sizeof(T) + 15) / 16) | 1) * 16 / sizeof(T));   // 16 x odd bytes: the lanes' 16-byte reads of their rows fall on different bank quads
    const int cbatch = (int)((long long)KWS_CFLOATS * sizeof(float) / ((long long)cpitch * sizeof(T)));
    const int cb = cbatch < 1 ? 0 : (cbatch < KWS_CB ? cbatch : KWS_CB);   // 0: rows too long to stage -> straight from global memory
    T* cst = reinterpret_cast<T*>(cstage);

    float bestf = -1.f;        // argmax of the next pass: best row by curf ...
    long long besti = -1;
    unsigned long long n_cand = 0, n_upd = 0;
    // argmax bookkeeping of one row: largest curf; an exact float32 tie is decided by the float64 values (largest, lowest row)
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
    // SUPER-tiles of KWS_R x DT rows: a thread owns KWS_R rows DT apart, their words of KWS_U planes are in flight together
    // (a pass is a latency-bound stream: with one row and a barrier pair per 256 rows it ran at 0.7 TB/s), and the candidates
    // of the whole super-tile are re-evaluated in one go
    const long long nsuper = (n + (long long)KWS_R * DT - 1) / ((long long)KWS_R * DT);
    for (long long t = blockIdx.x; t < nsuper; t += gridDim.x) {
        const long long base = t * (KWS_R * DT);
        long long row[KWS_R];
        float acc[KWS_R], sfr[KWS_R], cur[KWS_R];
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            const long long i = base + k * DT + tid;
            row[k] = i < n ? i : n - 1;
            acc[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            sfr[k] = __uint_as_float((unsigned)P.sf[row[k]] << 16);
            cur[k] = P.curf[row[k]];
        }
        // d~^2 over the byte planes: one coalesced word per row and four features, the centre's floats broadcast from LDS
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
                    const float4 y4 = *reinterpret_cast<const float4*>(ycf + 4 * (b0 + u));
#pragma unroll
                    for (int k = 0; k < KWS_R; ++k) {
                        const int wi = (int)w[u][k];
                        const float q0 = (float)((wi << 24) >> 24), q1 = (float)((wi << 16) >> 24), q2 = (float)((wi << 8) >> 24), q3 = (float)(wi >> 24);
                        const float t0 = fmaf(q0, sfr[k], -y4.x), t1 = fmaf(q1, sfr[k], -y4.y), t2 = fmaf(q2, sfr[k], -y4.z), t3 = fmaf(q3, sfr[k], -y4.w);
                        acc[k] = fmaf(t0, t0, acc[k]);
                        acc[k] = fmaf(t1, t1, acc[k]);
                        acc[k] = fmaf(t2, t2, acc[k]);
                        acc[k] = fmaf(t3, t3, acc[k]);
                    }
                }
            }
        }
        // screen: the rows that provably stay go straight to the argmax bookkeeping, the others into the block's list
        if (tid == 0) ncand = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KWS_R; ++k) {
            const long long i = base + k * DT + tid;
            if (i < n) {
                const float dt = sqrtf(acc[k]);
                const float E = eq * sfr[k] + (dt + Gf) * ea;
                const bool safe = (dt - E) * refl >= cur[k];   // NaN anywhere: false -> candidate
                if (safe) consider(i, cur[k]);
                else cand[atomicAdd(&ncand, 1)] = k * DT + tid;
            }
        }
        __syncthreads();
        const int nc = ncand;
        // candidates: the reference's arithmetic on the row itself.  Their rows are staged in LDS by the whole workgroup
        // (coalesced), then lane c walks row c front to back with ONE fp64 accumulator -- the sum's order is the reference's,
        // and no lane waits on global memory inside its chain of m steps
        for (int c0 = 0; c0 < nc; c0 += (cb ? cb : DT)) {
            const int cn = cb ? (nc - c0 < cb ? nc - c0 : cb) : (nc - c0 < DT ? nc - c0 : DT);
            // (this lane's candidate: its distances_ / curf travel with the staging loads, not behind the chain)
            const long long icp = base + cand[c0 + (tid < cn ? tid : 0)];
            const double dold = P.dist[icp];
            const float cfo = P.curf[icp];
            if (cb) {
                if (c0) __syncthreads();   // (the previous batch's readers are done)
                for (int e = tid; e < cn * (int)m; e += DT) {
                    const int c = e / (int)m, f = e - c * (int)m;
                    cst[c * cpitch + f] = X[(base + cand[c0 + c]) * m + f];
                }
                __syncthreads();
            }
            if (tid < cn) {
                const long long ic = base + cand[c0 + tid];
                const T* x = cb ? cst + tid * cpitch : X + ic * m;
                double d;
                if (cb) {
                    d = kws_exact_euclid<T>(x, yraw, m);
                } else {   // rows too long to stage: straight from global memory (any alignment)
                    double a = 0.0, bb = 0.0;
                    for (long long f = 0; f < m; ++f) m_update<T, M_EUCLIDEAN>(a, bb, x[f], yraw[f]);
                    d = m_final<M_EUCLIDEAN>(a, bb, m);
                }
                ++n_cand;
                float cf = cfo;
                if (d < dold) {   // strict, kcenters.py:93
                    P.dist[ic] = d;
                    P.labels[ic] = P.it;
                    cf = ksc_round_up(d);
                    P.curf[ic] = cf;
                    ++n_upd;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
                consider(ic, cf);   // (a tie re-reads distances_ of both rows: this lane's own store, or settled rows)
            }
        }
        __syncthreads();   // the list (and the staging area) are free for the next super-tile
    }
    // block reduction on the float64 values of the threads' winners
    rv[tid] = besti >= 0 ? __hip_atomic_load(P.dist + besti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.0;
    ri[tid] = besti;
    __syncthreads();
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
    if (P.stats && (n_cand | n_upd)) {
        atomicAdd(P.stats, n_cand);
        atomicAdd(P.stats + 1, n_upd);
    }
}

}  // namespace msm
