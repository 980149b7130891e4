// counts.hip -- post-clustering transition counts: the label-pair histogram at lag tau of
// msmbuilder.msm._transition_counts (/root/reference/msmbuilder/msm/core.py:487-596) over the
// integer labels the clustering kernels leave in HBM.
//
//   counts[i][j] = #{ (s, t) : map(y_s[t]) = i, map(y_s[t + tau]) = j }       (core.py:567-589)
// with pairs dropped when either label has no mapping (NaN / None upstream, here: a negative code or
// a label outside the remap table).  Integer work, HBM-bound: 8 B per frame read (twice: once as
// the "from" and once as the "to" stream, tau rows apart), results exact (int64).
//
// MSM trajectories are metastable -- the great majority of pairs are (i, i) for a slowly changing
// i -- so naive global atomics serialise on a handful of addresses.  Each workgroup therefore
// stages a tile of 4096 consecutive pair positions into LDS as int32 state codes (coalesced loads,
// remap applied on the way in), every thread run-length encodes its 16 consecutive positions and
// issues ONE 64-bit atomic per run.
#include "common.h"

#include <algorithm>
#include <vector>

namespace msm {

constexpr int CNT = 256;      // threads
constexpr int CSEG = 16;      // consecutive pair positions per thread
constexpr int CPITCH = 17;    // LDS pitch of a thread's segment (odd: conflict-free)
constexpr int CTILE = CNT * CSEG;

struct CountChunk {
    const msm_idx_t* y;  // the sequence
    long long t0;        // first pair position of this tile
    long long npos;      // positions in this tile (<= CTILE)
};

struct CountArgs {
    const CountChunk* chunks;
    long long nchunks;
    long long lag;
    long long lo;            // label value of remap[0] / bin 0
    const int32_t* remap;    // [n_bins] -> state or -1; nullptr: state = label - lo
    long long n_bins;
    long long n_states;
    unsigned long long* out; // MODE 0: [n_states^2]; MODE 1: [n_bins]
    msm_idx_t* range;        // MODE 2: [2] = {min, max} (atomics)
};

__device__ __forceinline__ int label_code(const CountArgs& P, msm_idx_t v)
{
    const long long b = v - P.lo;
    if (b < 0 || b >= P.n_bins) return -1;
    return P.remap ? P.remap[b] : (int)b;
}

// MODE 0: pair counts, MODE 1: label histogram (class discovery), MODE 2: min / max of the labels
template <int MODE>
__global__ __launch_bounds__(CNT) void counts_kernel(CountArgs P)
{
    __shared__ int from_s[CNT * CPITCH];
    __shared__ int to_s[MODE == 0 ? CNT * CPITCH : 1];
    __shared__ long long rmin[CNT], rmax[CNT];
    const int tid = threadIdx.x;
    long long vmin = 0x7fffffffffffffffLL, vmax = -0x7fffffffffffffffLL - 1;
    for (long long c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
        const CountChunk ch = P.chunks[c];
        const global_ptr<msm_idx_t> y = as_global<msm_idx_t>(ch.y);
        if (MODE == 2) {
            for (long long k = tid; k < ch.npos; k += CNT) {
                const msm_idx_t v = y[ch.t0 + k];
                vmin = v < vmin ? v : vmin;
                vmax = v > vmax ? v : vmax;
            }
            continue;
        }
        __syncthreads();  // previous tile's scans are done
#pragma unroll
        for (int j = 0; j < CSEG; ++j) {
            const int k = tid + j * CNT;  // coalesced: consecutive lanes, consecutive labels
            int cf = -1, ct = -1;
            if (k < ch.npos) {
                cf = label_code(P, y[ch.t0 + k]);
                if (MODE == 0) ct = label_code(P, y[ch.t0 + k + P.lag]);
            }
            const int slot = (k / CSEG) * CPITCH + (k % CSEG);
            from_s[slot] = cf;
            if (MODE == 0) to_s[slot] = ct;
        }
        __syncthreads();
        // run-length encode this thread's 16 consecutive positions
        long long cur = -1;
        unsigned long long run = 0;
#pragma unroll
        for (int j = 0; j < CSEG; ++j) {
            const int cf = from_s[tid * CPITCH + j];
            long long key;
            if (MODE == 0) {
                const int ct = to_s[tid * CPITCH + j];
                key = (cf >= 0 && ct >= 0) ? (long long)cf * P.n_states + ct : -1;
            } else {
                key = cf;
            }
            if (key != cur) {
                if (cur >= 0) atomicAdd(P.out + cur, run);
                cur = key;
                run = 0;
            }
            ++run;
        }
        if (cur >= 0) atomicAdd(P.out + cur, run);
    }
    if (MODE == 2) {
        rmin[tid] = vmin;
        rmax[tid] = vmax;
        __syncthreads();
        for (int s = CNT / 2; s > 0; s >>= 1) {
            if (tid < s) {
                rmin[tid] = rmin[tid + s] < rmin[tid] ? rmin[tid + s] : rmin[tid];
                rmax[tid] = rmax[tid + s] > rmax[tid] ? rmax[tid + s] : rmax[tid];
            }
            __syncthreads();
        }
        if (tid == 0) {
            atomicMin(reinterpret_cast<long long*>(P.range), rmin[0]);
            atomicMax(reinterpret_cast<long long*>(P.range) + 1, rmax[0]);
        }
    }
}

// stage host sequences (or take device ones) and build the tile table for `lag`
static int build_chunks(const msm_idx_t* const* ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                        long long lag, std::vector<CountChunk>& tab, DevBuf& stage, long long* total_pos)
{
    int rc;
    std::vector<const msm_idx_t*> dptr((size_t)n_seq);
    if (on_device) {
        for (msm_idx_t s = 0; s < n_seq; ++s) dptr[(size_t)s] = ptrs[s];
    } else {
        size_t bytes = 0;
        for (msm_idx_t s = 0; s < n_seq; ++s) bytes += ((size_t)n_rows[s] * sizeof(msm_idx_t) + 255) & ~(size_t)255;
        if ((rc = stage.reserve(bytes ? bytes : 256))) return rc;
        size_t off = 0;
        for (msm_idx_t s = 0; s < n_seq; ++s) {
            msm_idx_t* d = reinterpret_cast<msm_idx_t*>(stage.as<char>() + off);
            if (n_rows[s] > 0)
                MSM_HIP_CHECK(hipMemcpyAsync(d, ptrs[s], (size_t)n_rows[s] * sizeof(msm_idx_t), hipMemcpyHostToDevice, stream()));
            dptr[(size_t)s] = d;
            off += ((size_t)n_rows[s] * sizeof(msm_idx_t) + 255) & ~(size_t)255;
        }
    }
    long long tot = 0;
    for (msm_idx_t s = 0; s < n_seq; ++s) {
        const long long npos = n_rows[s] - lag;  // pair positions t in [0, len - lag)
        for (long long t0 = 0; t0 < npos; t0 += CTILE) {
            CountChunk ch;
            ch.y = dptr[(size_t)s];
            ch.t0 = t0;
            ch.npos = std::min<long long>(CTILE, npos - t0);
            tab.push_back(ch);
        }
        if (npos > 0) tot += npos;
    }
    *total_pos = tot;
    return MSM_OK;
}

static int check_seqs(const msm_idx_t* const* ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, const char* who)
{
    if (n_seq < 0 || (n_seq > 0 && (!ptrs || !n_rows))) return fail(MSM_ERR_INVALID, "%s: bad sequence table", who);
    for (msm_idx_t s = 0; s < n_seq; ++s)
        if (n_rows[s] < 0 || (n_rows[s] > 0 && !ptrs[s])) return fail(MSM_ERR_INVALID, "%s: bad sequence %lld", who, (long long)s);
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    return MSM_OK;
}

template <int MODE>
static int run_counts(CountArgs& P, const std::vector<CountChunk>& tab)
{
    DevBuf& dTab = pool(PS_IDS);
    int rc;
    if ((rc = dTab.reserve(tab.size() * sizeof(CountChunk) + 16))) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(dTab.p, tab.data(), tab.size() * sizeof(CountChunk), hipMemcpyHostToDevice, stream()));
    P.chunks = dTab.as<CountChunk>();
    P.nchunks = (long long)tab.size();
    const unsigned grid = (unsigned)std::min<size_t>(tab.size(), (size_t)8 * num_cus());
    hipLaunchKernelGGL(counts_kernel<MODE>, dim3(grid), dim3(CNT), 0, stream(), P);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_label_range(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                    msm_idx_t* lo, msm_idx_t* hi, msm_idx_t* n_total)
{
    int rc = check_seqs(y_ptrs, n_rows, n_seq, "msm_label_range");
    if (rc) return rc;
    if (!lo || !hi) return fail(MSM_ERR_INVALID, "msm_label_range: null output");
    std::vector<CountChunk> tab;
    long long tot = 0;
    if ((rc = build_chunks(y_ptrs, n_rows, n_seq, on_device, 0, tab, pool(PS_X), &tot))) return rc;
    if (n_total) *n_total = tot;
    *lo = 0;
    *hi = -1;
    if (tab.empty()) return MSM_OK;
    DevBuf& dR = pool(PS_SUM);
    if ((rc = dR.reserve(2 * sizeof(msm_idx_t)))) return rc;
    const msm_idx_t init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
    MSM_HIP_CHECK(hipMemcpyAsync(dR.p, init, sizeof(init), hipMemcpyHostToDevice, stream()));
    CountArgs P;
    memset(&P, 0, sizeof(P));
    P.range = dR.as<msm_idx_t>();
    if ((rc = run_counts<2>(P, tab))) return rc;
    msm_idx_t out[2];
    MSM_HIP_CHECK(hipMemcpyAsync(out, dR.p, sizeof(out), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    *lo = out[0];
    *hi = out[1];
    return MSM_OK;
}

int msm_label_histogram(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                        msm_idx_t lo, msm_idx_t n_bins, int64_t* hist)
{
    int rc = check_seqs(y_ptrs, n_rows, n_seq, "msm_label_histogram");
    if (rc) return rc;
    if (!hist || n_bins < 0 || n_bins > ((msm_idx_t)1 << 31) - 1) return fail(MSM_ERR_INVALID, "msm_label_histogram: bad bins");
    if (n_bins == 0) return MSM_OK;
    std::vector<CountChunk> tab;
    long long tot = 0;
    if ((rc = build_chunks(y_ptrs, n_rows, n_seq, on_device, 0, tab, pool(PS_X), &tot))) return rc;
    DevBuf& dOut = pool(PS_OUT);
    if ((rc = dOut.reserve((size_t)n_bins * sizeof(int64_t)))) return rc;
    MSM_HIP_CHECK(hipMemsetAsync(dOut.p, 0, (size_t)n_bins * sizeof(int64_t), stream()));
    if (!tab.empty()) {
        CountArgs P;
        memset(&P, 0, sizeof(P));
        P.lo = lo;
        P.n_bins = n_bins;
        P.out = dOut.as<unsigned long long>();
        if ((rc = run_counts<1>(P, tab))) return rc;
    }
    MSM_HIP_CHECK(hipMemcpyAsync(hist, dOut.p, (size_t)n_bins * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_transition_counts(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                          msm_idx_t lag_time, msm_idx_t lo, const int32_t* remap, msm_idx_t n_bins,
                          msm_idx_t n_states, int64_t* counts)
{
    int rc = check_seqs(y_ptrs, n_rows, n_seq, "msm_transition_counts");
    if (rc) return rc;
    if (lag_time < 1) return fail(MSM_ERR_INVALID, "msm_transition_counts: lag_time must be >= 1");
    if (n_states < 0 || n_states > ((msm_idx_t)1 << 20) || n_bins < 0 || n_bins > ((msm_idx_t)1 << 31) - 1 || (n_states > 0 && !counts))
        return fail(MSM_ERR_INVALID, "msm_transition_counts: bad table size");
    if (!remap && n_bins != n_states) return fail(MSM_ERR_INVALID, "msm_transition_counts: n_bins must equal n_states without a remap table");
    if (n_states == 0) return MSM_OK;
    std::vector<CountChunk> tab;
    long long tot = 0;
    if ((rc = build_chunks(y_ptrs, n_rows, n_seq, on_device, lag_time, tab, pool(PS_X), &tot))) return rc;
    DevBuf &dOut = pool(PS_OUT), &dMap = pool(PS_LAB);
    const size_t nn = (size_t)n_states * (size_t)n_states;
    if ((rc = dOut.reserve(nn * sizeof(int64_t)))) return rc;
    MSM_HIP_CHECK(hipMemsetAsync(dOut.p, 0, nn * sizeof(int64_t), stream()));
    if (!tab.empty()) {
        CountArgs P;
        memset(&P, 0, sizeof(P));
        P.lag = lag_time;
        P.lo = lo;
        P.n_bins = n_bins;
        P.n_states = n_states;
        P.out = dOut.as<unsigned long long>();
        if (remap) {
            if ((rc = dMap.reserve((size_t)n_bins * sizeof(int32_t) + 16))) return rc;
            for (msm_idx_t b = 0; b < n_bins; ++b)
                if (remap[b] >= n_states) return fail(MSM_ERR_INVALID, "msm_transition_counts: remap[%lld] = %d >= n_states", (long long)b, remap[b]);
            MSM_HIP_CHECK(hipMemcpyAsync(dMap.p, remap, (size_t)n_bins * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
            P.remap = dMap.as<int32_t>();
        }
        if ((rc = run_counts<0>(P, tab))) return rc;
    }
    MSM_HIP_CHECK(hipMemcpyAsync(counts, dOut.p, nn * sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // extern "C"
