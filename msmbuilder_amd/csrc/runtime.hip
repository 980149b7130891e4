// runtime.hip -- device selection, stream, error state, memory helpers of libmsmhip.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace msm {

static thread_local char g_err[512] = "";
static hipStream_t g_stream = nullptr;
static int g_num_cus = 0;

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

hipStream_t stream() { return g_stream; }

// ---------------------------------------------------------------------------------------------------
// h2d_bulk: pageable host memory -> HBM through pinned buffers filled by a small thread pool
// ---------------------------------------------------------------------------------------------------
namespace {

struct CopyPool {
    // Workers SPIN for a short while after a job before they sleep on the condition variable: a streamed upload hands them
    // a 20-32 MB slice every ~0.4 ms, and a futex wake of sleeping threads on a many-core host costs a sizeable part of
    // that.  The submitting thread spins too (a slice takes ~0.3 ms: 64-70 GB/s with 4-8 threads, scripts/micro/h2d_pipe.hip).
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    const char* src = nullptr;
    char* dst = nullptr;
    size_t len = 0;
    std::atomic<unsigned long> gen{0};
    std::atomic<int> pending{0};
    std::atomic<bool> quit{false};
    static constexpr long long SPIN_US = 2000;

    explicit CopyPool(int n)
    {
        for (int i = 0; i < n; ++i) th.emplace_back([this, i] { run(i); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> l(m);
            quit.store(true);
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void run(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            bool got = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; ++it) {
                if (gen.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_relaxed)) {
                    got = true;
                    break;
                }
                __builtin_ia32_pause();
                if ((it & 255u) == 255u &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > SPIN_US)
                    break;
            }
            if (!got) {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return quit.load() || gen.load(std::memory_order_acquire) != seen; });
            }
            if (quit.load()) return;
            seen = gen.load(std::memory_order_acquire);   // (the job's fields were written before this generation was published)
            const char* s = src;
            char* d = dst;
            const size_t n = len;
            const size_t T = th.size();
            const size_t per = (((n + T - 1) / T) + 4095) & ~(size_t)4095;
            const size_t o = (size_t)id * per;
            if (o < n) memcpy(d + o, s + o, std::min(per, n - o));
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
    void copy(char* d, const char* s, size_t n)  // blocking, all workers (one submitter at a time: the library's host thread)
    {
        src = s;
        dst = d;
        len = n;
        pending.store((int)th.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> l(m);   // (a worker between its predicate check and its sleep must not miss the wake)
            gen.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }
};

struct BulkH2D {
    static constexpr int NBUF = 4;
    static constexpr size_t BUF = (size_t)32 << 20;
    CopyPool pool;
    char* pin[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    bool used[NBUF] = {false, false, false, false};
    int next = 0;  // ring position, kept across calls: a call's first memcpy overlaps the previous call's last DMA
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipStream_t cs = nullptr;
    bool ok = false;

    BulkH2D() : pool(pool_threads())
    {
        hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        for (int i = 0; i < NBUF && e == hipSuccess; ++i) {
            e = hipHostMalloc((void**)&pin[i], BUF, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_in, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_out, hipEventDisableTiming);
        ok = (e == hipSuccess);
        if (!ok) (void)hipGetLastError();
    }
    static int pool_threads()
    {
        const unsigned hw = std::thread::hardware_concurrency();
        // (measured, round 5: 4 to 24 staging threads all gave the same rate on the bench's 20 MB trajectories, 32 fewer:
        //  profiles/r05_h2d_threads.txt)
        return (int)std::max(2u, std::min(8u, hw / 2));
    }
};

BulkH2D* bulk_instance()
{
    static BulkH2D* inst = new BulkH2D();  // process lifetime (the pool's threads only ever memcpy)
    return inst;
}

}  // namespace

int h2d_bulk(void* dst_device, const void* src_host, size_t bytes, bool after_compute)
{
    const bool small = bytes < ((size_t)1 << 20);   // (1 MB: below it the runtime's own staging of a pageable copy costs less than a slice hand-off)
    BulkH2D* B = (small && after_compute) ? nullptr : bulk_instance();   // (the pinned ring is built at the first copy that needs it)
    if (!B || !B->ok) {
        MSM_HIP_CHECK(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, g_stream));
        return MSM_OK;
    }
    if (small) {   // a small copy that must not queue behind the compute stream's kernels
        MSM_HIP_CHECK(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, B->cs));
        MSM_HIP_CHECK(hipEventRecord(B->ev_out, B->cs));
        MSM_HIP_CHECK(hipStreamWaitEvent(g_stream, B->ev_out, 0));
        return MSM_OK;
    }
    if (after_compute) {   // the destination may still be read by work queued on the caller's stream
        MSM_HIP_CHECK(hipEventRecord(B->ev_in, g_stream));
        MSM_HIP_CHECK(hipStreamWaitEvent(B->cs, B->ev_in, 0));
    }
    const char* s = static_cast<const char*>(src_host);
    char* d = static_cast<char*>(dst_device);
    for (size_t off = 0; off < bytes; off += BulkH2D::BUF) {
        const size_t len = std::min(BulkH2D::BUF, bytes - off);
        const int slot = B->next;
        B->next = (B->next + 1) % BulkH2D::NBUF;
        if (B->used[slot]) MSM_HIP_CHECK(hipEventSynchronize(B->done[slot]));  // its previous DMA has drained
        B->pool.copy(B->pin[slot], s + off, len);
        MSM_HIP_CHECK(hipMemcpyAsync(d + off, B->pin[slot], len, hipMemcpyHostToDevice, B->cs));
        MSM_HIP_CHECK(hipEventRecord(B->done[slot], B->cs));
        B->used[slot] = true;
    }
    MSM_HIP_CHECK(hipEventRecord(B->ev_out, B->cs));
    MSM_HIP_CHECK(hipStreamWaitEvent(g_stream, B->ev_out, 0));
    return MSM_OK;
}

int d2h_bulk(void* dst_host, const void* src_device, size_t bytes)
{
    BulkH2D* B = nullptr;
    if (bytes >= ((size_t)8 << 20)) {
        BulkH2D* inst = bulk_instance();
        if (inst->ok) B = inst;
    }
    if (!B) {
        MSM_HIP_CHECK(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, g_stream));
        MSM_HIP_CHECK(hipStreamSynchronize(g_stream));
        return MSM_OK;
    }
    MSM_HIP_CHECK(hipEventRecord(B->ev_in, g_stream));
    MSM_HIP_CHECK(hipStreamWaitEvent(B->cs, B->ev_in, 0));
    const char* s = static_cast<const char*>(src_device);
    char* d = static_cast<char*>(dst_host);
    // DMA of slice i+1 into the next pinned buffer while the workers copy slice i out of its buffer
    int prev_slot = -1;
    size_t prev_off = 0, prev_len = 0;
    for (size_t off = 0; off < bytes; off += BulkH2D::BUF) {
        const size_t len = std::min(BulkH2D::BUF, bytes - off);
        const int slot = B->next;
        B->next = (B->next + 1) % BulkH2D::NBUF;
        if (B->used[slot]) MSM_HIP_CHECK(hipEventSynchronize(B->done[slot]));
        MSM_HIP_CHECK(hipMemcpyAsync(B->pin[slot], s + off, len, hipMemcpyDeviceToHost, B->cs));
        MSM_HIP_CHECK(hipEventRecord(B->done[slot], B->cs));
        B->used[slot] = true;
        if (prev_slot >= 0) {
            MSM_HIP_CHECK(hipEventSynchronize(B->done[prev_slot]));
            B->pool.copy(d + prev_off, B->pin[prev_slot], prev_len);
        }
        prev_slot = slot;
        prev_off = off;
        prev_len = len;
    }
    MSM_HIP_CHECK(hipEventSynchronize(B->done[prev_slot]));
    B->pool.copy(d + prev_off, B->pin[prev_slot], prev_len);
    return MSM_OK;
}

int num_cus()
{
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return MSM_OK;
    release();
    size_t want = bytes + (bytes >> 3) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(MSM_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return MSM_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

DevBuf& pool(int slot)
{
    static DevBuf* slots = new DevBuf[PS_COUNT];  // intentionally leaked (outlives HIP teardown)
    return slots[slot];
}

// The bf16 image path's ring (tica.hip): ONE buffer per process, created with the first bf16-mode handle, never per fit --
// round 3 reserved a whole-input image per handle inside the timed fit (an 8 - 16 GB hipMalloc = 0.25 - 0.5 s).
ImgRing* img_ring()
{
    static ImgRing* P = nullptr;   // intentionally leaked
    if (P) return P;
    ImgRing* q = new ImgRing();
    const char* e = getenv("MSM_TICA_IMG_RING_MB");
    size_t mb = e ? (size_t)atoll(e) : 2048;   // (round 6: two halves of 1 GB, the carried pack fills one while the other is multiplied)
    if (mb < 64) mb = 64;
    if (mb > 65536) mb = 65536;
    q->bytes = mb << 20;
    const hipError_t err = hipMalloc((void**)&q->p, q->bytes);
    if (err != hipSuccess) {
        set_error("bf16 image ring: hipMalloc(%zu MB) failed: %s", mb, hipGetErrorString(err));
        delete q;
        return nullptr;
    }
    P = q;
    return P;
}

int metric_id(const char* name)
{
    if (!name) return -1;
    static const char* names[M_COUNT] = {"euclidean", "sqeuclidean", "cityblock", "chebyshev",
                                         "canberra",  "braycurtis",  "hamming",   "jaccard"};
    for (int i = 0; i < M_COUNT; ++i)
        if (strcmp(name, names[i]) == 0) return i;
    return -1;
}

__global__ void gather_rows_kernel(const char* __restrict__ X, size_t row_bytes,
                                   const msm_idx_t* __restrict__ rows, char* __restrict__ out)
{
    const msm_idx_t r = rows[blockIdx.x];
    const char* src = X + (size_t)r * row_bytes;
    char* dst = out + (size_t)blockIdx.x * row_bytes;
    for (size_t b = threadIdx.x; b < row_bytes; b += blockDim.x) dst[b] = src[b];
}

}  // namespace msm

using namespace msm;

extern "C" {

const char* msm_last_error(void) { return g_err; }

const char* msm_version(void) { return "msmhip 0.1 (gfx950)"; }

int msm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

int msm_init(int device)
{
    int n = msm_device_count();
    if (n == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(MSM_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    MSM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MSM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MSM_ERR_NODEVICE, "device %d is %s; libmsmhip is built for gfx950 only", device,
                    prop.gcnArchName);
    g_num_cus = prop.multiProcessorCount;
    return MSM_OK;
}

int msm_set_stream(void* hip_stream)
{
    g_stream = static_cast<hipStream_t>(hip_stream);
    return MSM_OK;
}

int msm_synchronize(void)
{
    MSM_HIP_CHECK(hipStreamSynchronize(g_stream));
    return MSM_OK;
}

int msm_device_info(char* name, int name_len, int* n_cu, int64_t* hbm_bytes)
{
    int dev = 0;
    MSM_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MSM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return MSM_OK;
}

int msm_malloc(void** dptr, size_t bytes)
{
    if (!dptr) return fail(MSM_ERR_INVALID, "msm_malloc: null output pointer");
    MSM_HIP_CHECK(hipMalloc(dptr, bytes ? bytes : 1));
    return MSM_OK;
}

int msm_free(void* dptr)
{
    if (dptr) MSM_HIP_CHECK(hipFree(dptr));
    return MSM_OK;
}

int msm_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return MSM_OK;   // (an empty tensor has a null data pointer: a zero-length copy is not an error)
    if (!dst || !src) return fail(MSM_ERR_INVALID, "msm_memcpy_h2d: null pointer");
    const int rc = h2d_bulk(dst, src, bytes);   // (large copies through the pinned ring: 50+ GB/s against ~10 from pageable memory)
    if (rc) return rc;
    MSM_HIP_CHECK(hipStreamSynchronize(g_stream));
    return MSM_OK;
}

int msm_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return MSM_OK;
    if (!dst || !src) return fail(MSM_ERR_INVALID, "msm_memcpy_d2h: null pointer");
    return d2h_bulk(dst, src, bytes);   // blocking
}

/* n host buffers (src[i], nbytes[i]) back to back into ONE device buffer, staged through the pinned ring without a
 * synchronisation between them; returns when the data is on the device. */
int msm_upload_list(void* dst, const void* const* src, const msm_idx_t* nbytes, msm_idx_t n)
{
    if (!dst || (n > 0 && (!src || !nbytes)) || n < 0) return fail(MSM_ERR_INVALID, "msm_upload_list: bad argument");
    char* d = static_cast<char*>(dst);
    // the FIRST copy that moves bytes orders the copy stream behind the caller's stream (the destination may be a block the
    // caching allocator recycled from a tensor whose kernels are still queued); the later ones follow it on the copy stream.
    // (tied to i == 0, an empty first buffer left every copy unordered.)
    bool first_done = false;
    for (msm_idx_t i = 0; i < n; ++i) {
        if (nbytes[i] < 0 || (nbytes[i] > 0 && !src[i])) return fail(MSM_ERR_INVALID, "msm_upload_list: bad buffer %lld", (long long)i);
        if (nbytes[i] > 0) {
            const int rc = h2d_bulk(d, src[i], (size_t)nbytes[i], !first_done);
            if (rc) return rc;
            first_done = true;
        }
        d += nbytes[i];
    }
    MSM_HIP_CHECK(hipStreamSynchronize(g_stream));
    return MSM_OK;
}

int msm_memcpy_d2d(void* dst, const void* src, size_t bytes)
{
    MSM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_stream));
    return MSM_OK;
}

int msm_gather_rows(const void* X, int elem_size, msm_idx_t n_features, const msm_idx_t* rows,
                    msm_idx_t n_rows, void* out, int on_device)
{
    if (!X || !rows || !out || elem_size <= 0 || n_features <= 0 || n_rows < 0)
        return fail(MSM_ERR_INVALID, "msm_gather_rows: bad argument");
    const size_t row_bytes = (size_t)elem_size * n_features;
    if (n_rows == 0) return MSM_OK;
    if (!on_device) {
        for (msm_idx_t i = 0; i < n_rows; ++i)
            memcpy((char*)out + i * row_bytes, (const char*)X + rows[i] * row_bytes, row_bytes);
        return MSM_OK;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, g_stream,
                       (const char*)X, row_bytes, rows, (char*)out);
    MSM_HIP_CHECK(hipGetLastError());
    return MSM_OK;
}

int msm_event_create(void** ev)
{
    hipEvent_t e;
    MSM_HIP_CHECK(hipEventCreate(&e));
    *ev = e;
    return MSM_OK;
}

int msm_event_record(void* ev)
{
    MSM_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev), g_stream));
    return MSM_OK;
}

int msm_event_elapsed_ms(void* start, void* stop, float* ms)
{
    MSM_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    MSM_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return MSM_OK;
}

int msm_event_destroy(void* ev)
{
    MSM_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return MSM_OK;
}

}  // extern "C"
