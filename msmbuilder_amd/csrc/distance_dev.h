// distance_dev.h -- device-side pieces shared by the libdistance translation units (distance.hip, distance_small.hip):
// the per-metric update/finalise arithmetic (separately rounded: both units are built with -ffp-contract=off), the row
// loaders and the argument block of the pairwise kernels.
#pragma once
#include "common.h"

#include <cmath>

namespace msm {

constexpr int DT = 256;  // threads = rows per tile
constexpr int CJ = 8;    // centres per register tile
constexpr int KC_MAXBLK = 1024;

template <typename T> struct FeatChunk;
template <> struct FeatChunk<float> { static constexpr int FC = 32; };
template <> struct FeatChunk<double> { static constexpr int FC = 16; };

// ---- per-element update / finalisation: distance_kernels.h:41-243 ---------
template <typename T, int M>
__device__ __forceinline__ void m_update(double& a, double& b, const T u, const T v)
{
    if (M == M_EUCLIDEAN || M == M_SQEUCLIDEAN) {
        const T df = u - v;
        const double d = (double)df;
        // float32 rows: d has a 24-bit significand, so d * d is EXACT in float64 (48 bits, exponent within range) and the
        // fused multiply-add rounds once exactly like the reference's multiply-then-add -- one instruction fewer per element
        if (sizeof(T) == 4)
            a = __builtin_fma(d, d, a);
        else
            a = a + d * d;
    } else if (M == M_CITYBLOCK) {
        const T df = u - v;
        a = a + fabs((double)df);
    } else if (M == M_CHEBYSHEV) {
        const T df = u - v;
        const double d = fabs((double)df);
        if (d > a) a = d;
    } else if (M == M_CANBERRA) {
        const T df = u - v;
        const double snum = fabs((double)df);
        // the reference's translation unit (Cython C++: Python.h first) resolves fabs(float) to the FLOAT overload, so
        // for float rows the denominator is a float add, widened afterwards (the compiled reference agrees bit for bit)
        double sdenom;
        if constexpr (sizeof(T) == 4)
            sdenom = (double)(__builtin_fabsf((float)u) + __builtin_fabsf((float)v));
        else
            sdenom = fabs((double)u) + fabs((double)v);
        if (sdenom > 0.0) a = a + snum / sdenom;
    } else if (M == M_BRAYCURTIS) {
        const T df = u - v;
        const T sf = u + v;
        a = a + fabs((double)df);
        b = b + fabs((double)sf);
    } else if (M == M_HAMMING) {
        a = a + (double)(u != v);
    } else if (M == M_JACCARD) {
        const int nz = (u != (T)0) | (v != (T)0);
        a = a + (double)((u != v) & nz);
        b = b + (double)nz;
    }
}

// One 16-byte fragment of a row against the same fragment of a centre.  float32 euclidean family: the four differences
// in two packed subtracts (v_pk_add_f32), then convert + exact fused multiply-add each (2.5 VALU instructions per element
// instead of 4); everything else: element by element.
template <typename T, int M>
__device__ __forceinline__ void m_update_frag(double& a, double& b, const raw_f32x4& xq, const raw_f32x4& yq)
{
    if constexpr (sizeof(T) == 4 && (M == M_EUCLIDEAN || M == M_SQEUCLIDEAN)) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // written as instructions: left to the compiler the pair subtracts are scalarised again when both operands come
        // straight from 16-byte LDS reads.  One wait state between a packed write and its first reader (s_nop).
        f32x2 d0, d1;
        const f32x2 x01 = {xq.x, xq.y}, x23 = {xq.z, xq.w}, y01 = {yq.x, yq.y}, y23 = {yq.z, yq.w};
        asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0"
            : "=&v"(d0), "=&v"(d1)
            : "v"(x01), "v"(y01), "v"(x23), "v"(y23));
        a = __builtin_fma((double)d0.x, (double)d0.x, a);
        a = __builtin_fma((double)d0.y, (double)d0.y, a);
        a = __builtin_fma((double)d1.x, (double)d1.x, a);
        a = __builtin_fma((double)d1.y, (double)d1.y, a);
    } else {
        constexpr int E = 16 / (int)sizeof(T);
        const T* xe = reinterpret_cast<const T*>(&xq);
        const T* ye = reinterpret_cast<const T*>(&yq);
#pragma unroll
        for (int e = 0; e < E; ++e) m_update<T, M>(a, b, xe[e], ye[e]);
    }
}

template <int M>
__device__ __forceinline__ double m_final(double a, double b, long long n)
{
    if (M == M_EUCLIDEAN) return sqrt(a);
    if (M == M_BRAYCURTIS || M == M_JACCARD) return a / b;
    if (M == M_HAMMING) return a / (double)n;
    return a;
}

template <typename T>
__device__ __forceinline__ void stage_rows(T* Xs, const T* __restrict__ X,
                                           const msm_idx_t* __restrict__ X_indices, long long row0,
                                           long long n, long long m, int f0, int fw, int tid)
{
    constexpr int FC = FeatChunk<T>::FC;  // power of two: lane -> (row, feature) needs no division
    constexpr int RPP = DT / FC;          // rows covered per pass of the workgroup
    const int ff = tid & (FC - 1);
    const int rr0 = tid / FC;
    if (ff >= fw) {
        // nothing to load for this lane in a partial last chunk; the tile columns >= fw are never read
        return;
    }
#pragma unroll 4
    for (int rr = rr0; rr < DT; rr += RPP) {
        const long long i = row0 + rr;
        T v = (T)0;
        if (i < n) {
            const long long r = X_indices ? X_indices[i] : i;
            v = X[r * m + f0 + ff];
        }
        Xs[rr * (FC + 1) + ff] = v;
    }
}

// Small-m fast path (m <= FC, e.g. clustering in tICA space): each lane keeps its whole row
// in registers, fetched with the widest aligned vector loads the row size allows; a wave's 64
// rows are contiguous in memory, so HBM still sees a linear stream.  No LDS round trip.
template <typename T>
__device__ __forceinline__ void load_row_regs(T (&x)[FeatChunk<T>::FC], const T* __restrict__ p, int m, int vecw)
{
    constexpr int FC = FeatChunk<T>::FC;
#pragma unroll
    for (int f = 0; f < FC; ++f) x[f] = (T)0;
    if (vecw == 16) {
        constexpr int E = 16 / sizeof(T);
#pragma unroll
        for (int v = 0; v < FC / E; ++v)
            if (v * E < m) {
                const float4 q = *reinterpret_cast<const float4*>(p + v * E);
                const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
                for (int e = 0; e < E; ++e) x[v * E + e] = qe[e];
            }
    } else if (vecw == 8) {
        constexpr int E = 8 / sizeof(T);
#pragma unroll
        for (int v = 0; v < FC / E; ++v)
            if (v * E < m) {
                const float2 q = *reinterpret_cast<const float2*>(p + v * E);
                const T* qe = reinterpret_cast<const T*>(&q);
#pragma unroll
                for (int e = 0; e < E; ++e) x[v * E + e] = qe[e];
            }
    } else {
#pragma unroll
        for (int f = 0; f < FC; ++f)
            if (f < m) x[f] = p[f];
    }
}

struct PairArgs {
    const void* X;
    const msm_idx_t* X_indices;
    const void* Y;        // device [K, m]
    long long n, K, m;
    msm_idx_t* labels;    // assign
    double* min_dist;     // assign (nullable)
    double* partial;      // assign: per-block inertia partials
    double* out;          // cdist / dist
    int vecw;             // fast path: vector width in bytes of the per-lane row loads (0 = LDS path)
};

// distance_small.hip: compile-time-width exact assign (euclidean family); false when the shape has no instantiation
bool launch_small3_f32(int metric, int grid, const PairArgs& P);
bool launch_small3_f64(int metric, int grid, const PairArgs& P);

}  // namespace msm
