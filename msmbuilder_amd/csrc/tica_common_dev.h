// tica_common_dev.h -- constants, chunk / argument structs and block-id helpers shared by the tICA kernels
// (round 5: cut out of tica.hip by kernel family, unchanged; included by it in this order)
#pragma once
#include "common.h"

namespace msm {

constexpr int TM = 128;     // output tile is TM x TM features
constexpr int NT = 256;     // threads per workgroup: 4 waves as 2x2, 64x64 outputs per wave
constexpr int BK32 = 32;    // frames per K-step, fp32 kernel
constexpr int BK64 = 16;    // frames per K-step, fp64 kernel
constexpr int KCMAX = 4096; // max frames per chunk (load-balance granule)
constexpr int KFLUSH = 8192; // max frames accumulated in fp32 registers before an fp64 merge
constexpr int NCB = 1024;   // column-sum partial slots (4 blocks per CU)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

struct TicaChunk {
    const void* base;  // row 0 of the trajectory
    long long row0;    // first row of this chunk inside the trajectory
    long long len;     // trajectory length
    int n;             // rows in this chunk
    int pad;
    long long last;    // last ADDRESSABLE row of the trajectory's storage (len - 1, or the end of the
                       // slice a rank holds when one long trajectory is split over ranks)
    long long g0;      // bf16 image path: first 8-pair group of this chunk in the packed image
};

struct TicaArgs {
    const TicaChunk* chunks;  // device table, or nullptr -> `single` split arithmetically by kc
    TicaChunk single;
    long long nchunks;
    long long ld;
    int kc;
    int F, lag, T, ntiles, S;
    double* slabs;    // [S*ntiles][TM*TM] fp64, owned per workgroup
    double* colpart;  // [NCB][2][F] fp64 partial column sums (temporary buffer)
    int* flag;        // sticky non-finite flag
    unsigned* cosync; // [S] per-cohort arrival counters (zeroed per launch): keeps a cohort's workgroups within one chunk of each other
    long long* dbg;   // profiling only: [shader clock start, end, 100 MHz wall start, end] of workgroup 0
    const float* shift; // [F] per-column reference row r (or nullptr): the fp32 / bf16 kernels accumulate (x - r), see "mean shift"
    int kflush;         // sum/difference kernel: frames accumulated in fp32 registers before the fp64 slab merge
    const float* zrow;  // [F] zeros: where the dummy loads of a non-staging half-step read when the column sums are folded
    double* colA;       // sum/difference kernel with folded column sums: [S (+ 1)][F] fp64 sums of the LEFT frames, one row per cohort
    long long n_main;   // sum/difference kernel, REM: chunks [0, n_main) belong to the whole cohorts, the rest to the remainder cohort
};

__device__ __forceinline__ TicaChunk get_chunk(const TicaArgs& P, long long c)
{
    if (P.chunks) return P.chunks[c];
    TicaChunk ch = P.single;
    ch.row0 = c * (long long)P.kc;
    long long rem = ch.len - ch.row0;
    ch.n = (int)(rem < P.kc ? rem : P.kc);
    return ch;
}

// persistent block id -> (cohort, tile); blocks land on XCD (blockIdx % 8), so remap to
// make consecutive p (= one cohort's tiles) share an XCD's L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_linear_id()
{
    const int G = gridDim.x, b = blockIdx.x;
    const int q = G / 8, r = G % 8, xcd = b % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
}

__device__ __forceinline__ void decode_tile(int tile, int T, int& I, int& J, int& isG)
{
    if (tile < T * T) {
        isG = 0;
        I = tile / T;
        J = tile % T;
    } else {
        isG = 1;
        int u = tile - T * T;
        I = 0;
        while (u >= T - I) {
            u -= T - I;
            ++I;
        }
        J = I + u;
    }
}

}  // namespace msm
