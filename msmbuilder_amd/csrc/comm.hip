// comm.hip -- the library's own communicator: RCCL over xGMI, one process per GPU.
//
// The exchange steps of the hot path -- the all-reduce of the packed tICA accumulators, the per-centre all-gather of the
// k-centers candidate records, the per-step all-reduce of the MiniBatchKMeans centroid sums -- are issued by the library
// itself on its own stream, between its own kernels, on device buffers: nothing is staged through the host and no
// Python runs inside a centre / step loop.  The reference has no collective at all (single process, single thread), so
// there is no call pattern to follow; these are sized for point-to-point xGMI: ONE 4 MB all-reduce per tICA fit, and
// latency-bound 100-byte records per k-centers centre.
//
// Two transports behind the same two primitives (comm_allreduce_f64, comm_allgather):
//   * RCCL (`msm_comm_init_rccl`): librccl is dlopen'ed (the copy PyTorch loaded when there is one); the 128-byte
//     unique id is created on rank 0 (`msm_comm_unique_id`) and handed to the other ranks by whatever bootstrap the
//     host program has (msmbuilder_amd/parallel.py broadcasts it through torch.distributed).
//   * host callback (`msm_comm_init_host`): the library stages the buffer through pinned host memory and calls a
//     function of the host program (gloo in the CPU-only / single-GPU test runs, where RCCL cannot form a communicator
//     because several ranks share one device).  Same code path above the primitive, so the multi-rank logic of the
//     centre and step loops is exercised by world_size-2 tests on one GPU.
#include "common.h"

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <atomic>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace msm {

namespace {

typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_destroy)(nccl_comm);
typedef int (*fn_abort)(nccl_comm);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
constexpr int NCCL_CHAR = 0, NCCL_F64 = 8, NCCL_SUM = 0;

struct Rccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_abort abort = nullptr;      // ncclCommAbort (optional: older builds): frees a communicator whose collectives will never complete
    fn_errstr errstr = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_allgather allgather = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        // a copy already in the process (PyTorch's, under whichever name it was loaded) before a fresh one: two RCCL
        // instances in one process would each bring their own bootstrap threads and IPC state
        for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
        for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            r.error = "librccl.so not found";
            return;
        }
        r.get_uid = (fn_get_uid)dlsym(r.lib, "ncclGetUniqueId");
        r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
        r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
        r.abort = (fn_abort)dlsym(r.lib, "ncclCommAbort");
        r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
        r.allreduce = (fn_allreduce)dlsym(r.lib, "ncclAllReduce");
        r.allgather = (fn_allgather)dlsym(r.lib, "ncclAllGather");
        if (!r.get_uid || !r.init_rank || !r.destroy || !r.allreduce || !r.allgather) r.error = "RCCL entry points not found in librccl";
    });
    return r;
}

struct Comm {
    int kind = 0;  // 0 none, 1 RCCL, 2 host callback
    int rank = 0, world = 1;
    nccl_comm comm = nullptr;
    msm_host_collective_fn cb = nullptr;
    void* pinned = nullptr;  // host-callback staging
    size_t pinned_cap = 0;
    bool wedged = false;     // a collective of this communicator never completed and could not be aborted: never wait for its stream
};
Comm g_comm;

int pinned_reserve(size_t bytes)
{
    if (bytes <= g_comm.pinned_cap) return MSM_OK;
    if (g_comm.pinned) (void)hipHostFree(g_comm.pinned);
    g_comm.pinned = nullptr;
    g_comm.pinned_cap = 0;
    MSM_HIP_CHECK(hipHostMalloc(&g_comm.pinned, bytes, hipHostMallocDefault));
    g_comm.pinned_cap = bytes;
    return MSM_OK;
}

const char* nccl_err(int st)
{
    Rccl& r = rccl();
    return r.errstr ? r.errstr(st) : "?";
}

}  // namespace

bool comm_active() { return g_comm.kind != 0; }  // a communicator is installed (a world of one still runs the collectives)
int comm_rank() { return g_comm.kind ? g_comm.rank : 0; }
int comm_world() { return g_comm.kind ? g_comm.world : 1; }

// in place, device buffer, ordered on stream(); returns without synchronising (RCCL) or synchronised (host callback)
int comm_allreduce_f64(double* dbuf, size_t n)
{
    if (g_comm.kind == 0 || n == 0) return MSM_OK;
    if (g_comm.kind == 1) {
        const int st = rccl().allreduce(dbuf, dbuf, n, NCCL_F64, NCCL_SUM, g_comm.comm, stream());
        if (st != 0) return fail(MSM_ERR_HIP, "ncclAllReduce failed: %s", nccl_err(st));
        return MSM_OK;
    }
    int rc = pinned_reserve(n * sizeof(double));
    if (rc) return rc;
    MSM_HIP_CHECK(hipMemcpyAsync(g_comm.pinned, dbuf, n * sizeof(double), hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (g_comm.cb(0, g_comm.pinned, g_comm.pinned, (int64_t)(n * sizeof(double))) != 0)
        return fail(MSM_ERR_STATE, "host collective callback failed (all-reduce)");
    MSM_HIP_CHECK(hipMemcpyAsync(dbuf, g_comm.pinned, n * sizeof(double), hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

// drecv[r * bytes ...] = rank r's dsend[0 .. bytes); device buffers, ordered on stream()
int comm_allgather(const void* dsend, void* drecv, size_t bytes)
{
    if (bytes == 0) return MSM_OK;
    if (g_comm.kind == 0) {
        if (dsend != drecv) MSM_HIP_CHECK(hipMemcpyAsync(drecv, dsend, bytes, hipMemcpyDeviceToDevice, stream()));
        return MSM_OK;
    }
    if (g_comm.kind == 1) {
        const int st = rccl().allgather(dsend, drecv, bytes, NCCL_CHAR, g_comm.comm, stream());
        if (st != 0) return fail(MSM_ERR_HIP, "ncclAllGather failed: %s", nccl_err(st));
        return MSM_OK;
    }
    const size_t W = (size_t)g_comm.world;
    int rc = pinned_reserve((W + 1) * bytes);
    if (rc) return rc;
    char* hs = static_cast<char*>(g_comm.pinned);
    char* hr = hs + bytes;
    MSM_HIP_CHECK(hipMemcpyAsync(hs, dsend, bytes, hipMemcpyDeviceToHost, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    if (g_comm.cb(1, hs, hr, (int64_t)bytes) != 0) return fail(MSM_ERR_STATE, "host collective callback failed (all-gather)");
    MSM_HIP_CHECK(hipMemcpyAsync(drecv, hr, W * bytes, hipMemcpyHostToDevice, stream()));
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

extern "C" {

int msm_comm_unique_id(char* id128)
{
    if (!id128) return fail(MSM_ERR_INVALID, "msm_comm_unique_id: null pointer");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(MSM_ERR_STATE, "RCCL unavailable: %s", r.error.c_str());
    nccl_uid id;
    const int st = r.get_uid(&id);
    if (st != 0) return fail(MSM_ERR_HIP, "ncclGetUniqueId failed: %s", nccl_err(st));
    memcpy(id128, id.internal, 128);
    return MSM_OK;
}

/* 1 when this process could join an RCCL communicator (librccl loadable with every entry point, a device visible).
 * Ranks agree on this BEFORE any of them enters ncclCommInitRank, which blocks until all ranks have joined. */
int msm_comm_rccl_available(void)
{
    if (msm_device_count() <= 0) return 0;
    return rccl().error.empty() ? 1 : 0;
}

int msm_comm_init_rccl(const char* id128, int rank, int world)
{
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(MSM_ERR_INVALID, "msm_comm_init_rccl: bad argument");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(MSM_ERR_STATE, "RCCL unavailable: %s", r.error.c_str());
    msm_comm_destroy();
    nccl_uid id;
    memcpy(id.internal, id128, 128);
    // ncclCommInitRank blocks until EVERY rank has joined: a rank that died, took another branch or cannot reach the
    // others would hang the job for good.  It runs on a helper thread (on the caller's device); the caller gives up after
    // MSM_COMM_TIMEOUT_S seconds (180) with an error that names the rank -- the Python side then agrees on the host
    // transport with the ranks that are still there (the helper stays parked inside RCCL: there is no cancelling it).
    int dev = 0;
    MSM_HIP_CHECK(hipGetDevice(&dev));
    typedef std::pair<int, nccl_comm> InitResult;
    struct Join {
        std::promise<InitResult> prom;
        std::atomic<int> state{0};   // 0 waiting, 1 the caller gave up (a communicator that still arrives belongs to the helper), 2 delivered
    };
    auto join = std::make_shared<Join>();
    std::future<InitResult> fut = join->prom.get_future();
    const fn_init_rank init = r.init_rank;
    const fn_abort abort_fn = r.abort;
    const fn_destroy destroy_fn = r.destroy;
    std::thread([join, init, abort_fn, destroy_fn, dev, world, id, rank] {
        (void)hipSetDevice(dev);
        nccl_comm c = nullptr;
        const int st = init(&c, world, id, rank);
        int expect = 0;
        if (join->state.compare_exchange_strong(expect, 2)) {
            join->prom.set_value(InitResult(st, c));
        } else if (st == 0 && c) {
            // the caller timed out and has moved on to the host transport: nobody will ever use or free this communicator
            if (abort_fn) (void)abort_fn(c);
            else (void)destroy_fn(c);
        }
    }).detach();
    const char* te = getenv("MSM_COMM_TIMEOUT_S");
    const int tmo = te && atoi(te) > 0 ? atoi(te) : 180;
    if (fut.wait_for(std::chrono::seconds(tmo)) != std::future_status::ready) {
        int expect = 0;
        if (join->state.compare_exchange_strong(expect, 1))
            return fail(MSM_ERR_STATE, "ncclCommInitRank timed out after %d s on rank %d of %d (device %d): not every rank joined", tmo, rank,
                        world, dev);
        // (the helper delivered between the wait and the exchange: take the result)
    }
    const InitResult res = fut.get();
    const int st = res.first;
    nccl_comm c = res.second;
    if (st != 0) return fail(MSM_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, nccl_err(st));
    g_comm.kind = 1;
    g_comm.rank = rank;
    g_comm.world = world;
    g_comm.comm = c;
    return MSM_OK;
}

int msm_comm_init_host(msm_host_collective_fn fn, int rank, int world)
{
    if (!fn || world < 1 || rank < 0 || rank >= world) return fail(MSM_ERR_INVALID, "msm_comm_init_host: bad argument");
    msm_comm_destroy();
    g_comm.kind = 2;
    g_comm.rank = rank;
    g_comm.world = world;
    g_comm.cb = fn;
    return MSM_OK;
}

int msm_comm_destroy(void)
{
    if (g_comm.kind == 1 && g_comm.comm) {
        if (g_comm.wedged) {
            // its stream will never drain: no synchronisation, abort where the library has it, otherwise leaked on purpose
            if (rccl().abort) (void)rccl().abort(g_comm.comm);
        } else {
            (void)hipStreamSynchronize(stream());
            (void)rccl().destroy(g_comm.comm);
        }
    }
    if (g_comm.pinned) (void)hipHostFree(g_comm.pinned);
    g_comm = Comm();
    return MSM_OK;
}

int msm_comm_info(int* rank, int* world, int* kind)
{
    if (rank) *rank = comm_rank();
    if (world) *world = comm_world();
    if (kind) *kind = g_comm.kind;
    return MSM_OK;
}

/* One all-reduce and one all-gather of a few doubles through the installed communicator, checked against what they must
 * return, with a time limit: 0 = the transport works, MSM_ERR_STATE = wrong numbers or no answer within `timeout_s`
 * (message names the rank).  Called by every rank right after the communicator is built, BEFORE any fit depends on it. */
int msm_comm_selftest(int timeout_s)
{
    if (g_comm.kind == 0) return MSM_OK;
    const int W = g_comm.world, me = g_comm.rank, tmo = timeout_s > 0 ? timeout_s : 60;
    // d[0..1]: all-reduced {rank + 1, 1}; d[2]: this rank's tag; d[3 .. 3 + W): the gathered tags
    double* d = nullptr;
    MSM_HIP_CHECK(hipMalloc((void**)&d, (size_t)(3 + W) * sizeof(double)));
    const double mine[3] = {(double)(me + 1), 1.0, 1000.0 + me};
    MSM_HIP_CHECK(hipMemcpyAsync(d, mine, sizeof(mine), hipMemcpyHostToDevice, stream()));
    int rc = comm_allreduce_f64(d, 2);
    if (!rc) rc = comm_allgather(d + 2, d + 3, sizeof(double));
    if (!rc) {
        hipEvent_t ev = nullptr;
        MSM_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        MSM_HIP_CHECK(hipEventRecord(ev, stream()));
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipEventQuery(ev)) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(tmo)) break;
            usleep(200);
        }
        if (q == hipErrorNotReady) {
            // A collective that will never complete sits on the library's stream, and everything queued behind it -- the
            // host transport the caller falls back to included -- would wait for it.  ncclCommAbort makes the communicator's
            // kernels exit; the stream is then given a bounded time to drain.  The buffer and the event are leaked on purpose
            // when it does not (the caller is told that the stream is unusable).
            bool drained = false;
            if (g_comm.kind == 1 && g_comm.comm && rccl().abort) {
                (void)rccl().abort(g_comm.comm);
                g_comm.comm = nullptr;
                g_comm.kind = 0;
                const auto t1 = std::chrono::steady_clock::now();
                while ((q = hipEventQuery(ev)) == hipErrorNotReady && std::chrono::steady_clock::now() - t1 < std::chrono::seconds(20)) usleep(1000);
                drained = q != hipErrorNotReady;
            }
            if (drained) {
                (void)hipEventDestroy(ev);
                (void)hipFree(d);
            } else {
                g_comm.wedged = true;
            }
            return fail(MSM_ERR_STATE, "rank %d of %d: the communicator's first collectives did not complete within %d s (%s)", me, W, tmo,
                        drained ? "communicator aborted, the stream drained" : "the library's stream is still blocked: give the library a new stream");
        }
        (void)hipEventDestroy(ev);
        if (q != hipSuccess) rc = fail(MSM_ERR_HIP, "rank %d of %d: collective self-test: %s", me, W, hipGetErrorString(q));
    }
    if (!rc) {
        std::vector<double> host((size_t)(3 + W), 0.0);
        MSM_HIP_CHECK(hipMemcpy(host.data(), d, (size_t)(3 + W) * sizeof(double), hipMemcpyDeviceToHost));
        bool ok = host[0] == 0.5 * W * (W + 1) && host[1] == (double)W;
        for (int r = 0; r < W && ok; ++r) ok = host[3 + r] == 1000.0 + r;
        if (!ok)
            rc = fail(MSM_ERR_STATE, "rank %d of %d: the communicator's first collectives returned wrong numbers (sum %g, count %g)", me, W,
                      host[0], host[1]);
    }
    (void)hipFree(d);
    return rc;
}

/* test / bootstrap helpers on caller-owned DEVICE buffers */
int msm_comm_allreduce_f64(double* dbuf, msm_idx_t n)
{
    if (!dbuf || n < 0) return fail(MSM_ERR_INVALID, "msm_comm_allreduce_f64: bad argument");
    int rc = comm_allreduce_f64(dbuf, (size_t)n);
    if (rc) return rc;
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

int msm_comm_allgather(const void* dsend, void* drecv, msm_idx_t bytes)
{
    if (!dsend || !drecv || bytes < 0) return fail(MSM_ERR_INVALID, "msm_comm_allgather: bad argument");
    int rc = comm_allgather(dsend, drecv, (size_t)bytes);
    if (rc) return rc;
    MSM_HIP_CHECK(hipStreamSynchronize(stream()));
    return MSM_OK;
}

/* Latency of the library's collectives as the k-centers and tICA loops see them: `reps` collectives of `bytes` per rank
 * queued back to back on the library stream between two events (3 untimed ones first), microseconds per call.
 * kind 0: in-place all-reduce of bytes / 8 doubles; kind 1: all-gather of `bytes` per rank.  Every rank calls it. */
int msm_comm_measure(int kind, msm_idx_t bytes, int reps, float* us_per_call)
{
    if (!us_per_call || bytes < 8 || reps < 1 || (kind != 0 && kind != 1)) return fail(MSM_ERR_INVALID, "msm_comm_measure: bad argument");
    if (!comm_active()) return fail(MSM_ERR_STATE, "msm_comm_measure: no communicator");
    const int W = g_comm.world;
    char* d = nullptr;
    MSM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), (size_t)bytes * (kind == 1 ? (size_t)W + 1 : 1)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MSM_OK;
    if (hipMemsetAsync(d, 0, (size_t)bytes * (kind == 1 ? (size_t)W + 1 : 1), stream()) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess)
        rc = fail(MSM_ERR_HIP, "msm_comm_measure: setup failed");
    auto once = [&]() -> int {
        return kind == 0 ? comm_allreduce_f64(reinterpret_cast<double*>(d), (size_t)bytes / 8) : comm_allgather(d, d + bytes, (size_t)bytes);
    };
    for (int i = 0; i < 3 && !rc; ++i) rc = once();
    if (!rc && hipEventRecord(e0, stream()) != hipSuccess) rc = fail(MSM_ERR_HIP, "msm_comm_measure: event");
    for (int i = 0; i < reps && !rc; ++i) rc = once();
    if (!rc && (hipEventRecord(e1, stream()) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = fail(MSM_ERR_HIP, "msm_comm_measure: event");
    float ms = 0.f;
    if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(MSM_ERR_HIP, "msm_comm_measure: event");
    *us_per_call = 1e3f * ms / (float)reps;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(stream());
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
