// npyio.hip -- the on-disk side of the hot path: MSMBuilder's "dir-npy" datasets are
// directories of %08d.npy files, one 2-D array per trajectory
// (/root/reference/msmbuilder/dataset.py:290-331, NumpyDirDataset.get = np.load).  This is a native
// reader that takes such a file straight into HBM: header parse, pread() of the payload into
// pinned host buffers on a worker thread, hipMemcpyAsync on a dedicated copy stream -- so the disk
// read of file k+1, the PCIe transfer of file k and the covariance kernel on file k-1 overlap.
// Host-side code only (no kernels); the bytes that land in HBM are exactly the file's payload.
#include "common.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace msm {

struct NpyInfo {
    int dtype_bytes = 0;
    char kind = 0;  // 'f', 'i', 'u', 'b'
    int fortran = 0;
    int ndim = 0;
    long long shape[4] = {0, 0, 0, 0};
    long long data_offset = 0;
    long long payload_bytes = 0;
};

// .npy format 1.0 / 2.0 / 3.0: magic, version, little-endian header length, python dict literal
static int parse_npy_header(const char* path, int fd, NpyInfo* out)
{
    unsigned char pre[12];
    if (pread(fd, pre, 10, 0) != 10 || memcmp(pre, "\x93NUMPY", 6) != 0)
        return fail(MSM_ERR_INVALID, "%s: not a .npy file", path);
    const int major = pre[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = (size_t)pre[8] | ((size_t)pre[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (pread(fd, pre + 10, 2, 10) != 2) return fail(MSM_ERR_INVALID, "%s: truncated header", path);
        hlen = (size_t)pre[8] | ((size_t)pre[9] << 8) | ((size_t)pre[10] << 16) | ((size_t)pre[11] << 24);
        hoff = 12;
    } else {
        return fail(MSM_ERR_INVALID, "%s: unsupported .npy version %d", path, major);
    }
    if (hlen > (1u << 20)) return fail(MSM_ERR_INVALID, "%s: implausible header length", path);
    std::string h(hlen, '\0');
    if ((size_t)pread(fd, &h[0], hlen, (off_t)hoff) != hlen) return fail(MSM_ERR_INVALID, "%s: truncated header", path);
    auto value_after = [&](const char* key) -> size_t {
        size_t p = h.find(key);
        if (p == std::string::npos) return p;
        p = h.find(':', p);
        if (p == std::string::npos) return p;
        ++p;
        while (p < h.size() && h[p] == ' ') ++p;
        return p;
    };
    size_t p = value_after("'descr'");
    if (p == std::string::npos || (h[p] != '\'' && h[p] != '"')) return fail(MSM_ERR_INVALID, "%s: no simple dtype descr", path);
    const size_t q = h.find(h[p], p + 1);
    const std::string descr = h.substr(p + 1, q - p - 1);
    if (descr.size() < 3 || (descr[0] != '<' && descr[0] != '|' && descr[0] != '='))
        return fail(MSM_ERR_INVALID, "%s: dtype '%s' is not little-endian native", path, descr.c_str());
    out->kind = descr[1];
    out->dtype_bytes = atoi(descr.c_str() + 2);
    if (!(out->kind == 'f' || out->kind == 'i' || out->kind == 'u' || out->kind == 'b') || out->dtype_bytes < 1 || out->dtype_bytes > 8)
        return fail(MSM_ERR_INVALID, "%s: unsupported dtype '%s'", path, descr.c_str());
    p = value_after("'fortran_order'");
    if (p == std::string::npos) return fail(MSM_ERR_INVALID, "%s: no fortran_order", path);
    out->fortran = h.compare(p, 4, "True") == 0;
    p = value_after("'shape'");
    if (p == std::string::npos || h[p] != '(') return fail(MSM_ERR_INVALID, "%s: no shape", path);
    ++p;
    out->ndim = 0;
    long long count = 1;
    while (p < h.size() && h[p] != ')') {
        if (h[p] >= '0' && h[p] <= '9') {
            if (out->ndim >= 4) return fail(MSM_ERR_INVALID, "%s: more than 4 dimensions", path);
            long long v = 0;
            while (p < h.size() && h[p] >= '0' && h[p] <= '9') v = v * 10 + (h[p++] - '0');
            out->shape[out->ndim++] = v;
            count *= v;
        } else {
            ++p;
        }
    }
    out->data_offset = (long long)(hoff + hlen);
    out->payload_bytes = count * out->dtype_bytes;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < out->data_offset + out->payload_bytes)
        return fail(MSM_ERR_INVALID, "%s: file shorter than its header promises", path);
    return MSM_OK;
}

}  // namespace msm

using namespace msm;

// n_buffers reader threads, each with ONE pinned buffer and its own stream.  A submitted file is cut
// into buffer-sized pieces that go to a shared queue; a reader takes a piece, pread()s it (page cache /
// disk -> pinned memory, the CPU-bound part, which is why there are several readers), copies it to
// its place in HBM and synchronises its own stream before taking the next piece.  A job is complete
// when its last piece has landed; wait(j) returns when every job <= j is complete.
struct msm_npy_loader {
    struct Piece {
        long long job;
        int fd;
        long long file_off, n;
        char* dst;
        hipEvent_t fence;  // recorded on the consumer's stream at submit time: the copy must not start before it
    };
    struct JobState {
        int fd = -1;
        long long remaining = 0;
        hipEvent_t fence = nullptr;
    };
    size_t buf_bytes = 0;
    std::vector<char*> bufs;  // pinned, one per reader
    std::vector<hipStream_t> streams;
    std::vector<std::thread> readers;
    int device = 0;
    std::mutex mu;
    std::condition_variable cv_piece, cv_done;
    std::deque<Piece> queue;
    std::deque<std::pair<long long, JobState>> jobs;  // incomplete jobs in submission order
    long long next_id = 1;
    std::string error;
    bool stop = false;

    JobState* find(long long id)
    {
        for (auto& j : jobs)
            if (j.first == id) return &j.second;
        return nullptr;
    }

    void run(int r)
    {
        (void)hipSetDevice(device);
        for (;;) {
            Piece pc;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_piece.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;  // stop requested and nothing left
                pc = queue.front();
                queue.pop_front();
            }
            std::string err;
            long long got = 0;
            while (got < pc.n && err.empty()) {
                const ssize_t k = pread(pc.fd, bufs[r] + got, (size_t)(pc.n - got), (off_t)(pc.file_off + got));
                if (k <= 0) err = "short read in .npy payload";
                else got += k;
            }
            // The destination usually comes from a stream-ordered caching allocator (torch): the block may have been
            // handed back by a tensor that kernels queued on the consumer's stream are still reading.  The reader's own
            // stream therefore waits for the fence recorded on that stream when the job was submitted.
            if (err.empty() && pc.n > 0 && pc.fence && hipStreamWaitEvent(streams[r], pc.fence, 0) != hipSuccess)
                err = "hipStreamWaitEvent on the submit fence failed";
            if (err.empty() && pc.n > 0 &&
                (hipMemcpyAsync(pc.dst, bufs[r], (size_t)pc.n, hipMemcpyHostToDevice, streams[r]) != hipSuccess ||
                 hipStreamSynchronize(streams[r]) != hipSuccess))
                err = "host-to-device copy of a .npy piece failed";
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!err.empty() && error.empty()) error = err;
                JobState* js = find(pc.job);
                if (js && --js->remaining == 0) {
                    close(js->fd);
                    js->fd = -1;
                    if (js->fence) (void)hipEventDestroy(js->fence);
                    js->fence = nullptr;
                    while (!jobs.empty() && jobs.front().second.remaining == 0) jobs.pop_front();
                }
            }
            cv_done.notify_all();
        }
    }
};

extern "C" {

int msm_npy_info(const char* path, int* dtype_bytes, int* kind, int* fortran_order, int* ndim, msm_idx_t* shape4,
                 msm_idx_t* data_offset)
{
    if (!path) return fail(MSM_ERR_INVALID, "msm_npy_info: null path");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(MSM_ERR_INVALID, "%s: cannot open", path);
    NpyInfo info;
    const int rc = parse_npy_header(path, fd, &info);
    close(fd);
    if (rc) return rc;
    if (dtype_bytes) *dtype_bytes = info.dtype_bytes;
    if (kind) *kind = info.kind;
    if (fortran_order) *fortran_order = info.fortran;
    if (ndim) *ndim = info.ndim;
    if (shape4)
        for (int i = 0; i < 4; ++i) shape4[i] = info.shape[i];
    if (data_offset) *data_offset = info.data_offset;
    return MSM_OK;
}

int msm_npy_loader_create(msm_npy_loader_t** out, int n_buffers, size_t buffer_bytes)
{
    if (!out || n_buffers < 1 || n_buffers > 64 || buffer_bytes < 4096)
        return fail(MSM_ERR_INVALID, "msm_npy_loader_create: need 1..64 buffers of >= 4 KiB");
    if (msm_device_count() == 0) return fail(MSM_ERR_NODEVICE, "no HIP device visible");
    msm_npy_loader* h = new msm_npy_loader();
    h->buf_bytes = buffer_bytes;
    (void)hipGetDevice(&h->device);
    hipError_t e = hipSuccess;
    for (int i = 0; i < n_buffers && e == hipSuccess; ++i) {
        char* p = nullptr;
        e = hipHostMalloc((void**)&p, buffer_bytes, hipHostMallocDefault);
        if (e == hipSuccess) {
            h->bufs.push_back(p);
            hipStream_t st;
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            if (e == hipSuccess) h->streams.push_back(st);
        }
    }
    if (e != hipSuccess) {
        for (char* p : h->bufs) (void)hipHostFree(p);
        for (hipStream_t st : h->streams) (void)hipStreamDestroy(st);
        delete h;
        return fail(MSM_ERR_HIP, "msm_npy_loader_create: %s", hipGetErrorString(e));
    }
    for (int r = 0; r < n_buffers; ++r) h->readers.emplace_back([h, r] { h->run(r); });
    *out = h;
    return MSM_OK;
}

int msm_npy_loader_submit(msm_npy_loader_t* h, const char* path, void* dptr, msm_idx_t nbytes, msm_idx_t* job_id)
{
    if (!h || !path || !dptr || nbytes < 0 || !job_id) return fail(MSM_ERR_INVALID, "msm_npy_loader_submit: bad argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(MSM_ERR_INVALID, "%s: cannot open", path);
    NpyInfo info;
    int rc = parse_npy_header(path, fd, &info);
    if (rc == MSM_OK && info.payload_bytes != nbytes)
        rc = fail(MSM_ERR_INVALID, "%s: payload is %lld bytes, caller expects %lld", path, info.payload_bytes, (long long)nbytes);
    if (rc) {
        close(fd);
        return rc;
    }
    // fence: everything queued so far on the library stream (= the consumer's stream, e.g. torch's current stream)
    hipEvent_t fence = nullptr;
    if (hipEventCreateWithFlags(&fence, hipEventDisableTiming) != hipSuccess || hipEventRecord(fence, stream()) != hipSuccess) {
        if (fence) (void)hipEventDestroy(fence);
        close(fd);
        return fail(MSM_ERR_HIP, "msm_npy_loader_submit: could not record the submit fence");
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        const long long id = h->next_id++;
        msm_npy_loader::JobState js;
        js.fd = fd;
        js.fence = fence;
        js.remaining = std::max<long long>(1, ceil_div(nbytes, (long long)h->buf_bytes));
        h->jobs.emplace_back(id, js);
        long long off = 0;
        do {
            msm_npy_loader::Piece pc;
            pc.job = id;
            pc.fd = fd;
            pc.file_off = info.data_offset + off;
            pc.n = std::min<long long>((long long)h->buf_bytes, nbytes - off);
            pc.dst = static_cast<char*>(dptr) + off;
            pc.fence = fence;
            h->queue.push_back(pc);
            off += pc.n;
        } while (off < nbytes);
        *job_id = id;
    }
    h->cv_piece.notify_all();
    return MSM_OK;
}

int msm_npy_loader_wait(msm_npy_loader_t* h, msm_idx_t job_id)
{
    if (!h) return fail(MSM_ERR_STATE, "null loader");
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv_done.wait(lk, [&] { return h->jobs.empty() || h->jobs.front().first > job_id; });
    if (!h->error.empty()) {
        const std::string e = h->error;
        h->error.clear();
        return fail(MSM_ERR_INVALID, "%s", e.c_str());
    }
    return MSM_OK;
}

int msm_npy_loader_destroy(msm_npy_loader_t* h)
{
    if (!h) return MSM_OK;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->stop = true;
    }
    h->cv_piece.notify_all();
    for (auto& t : h->readers)
        if (t.joinable()) t.join();  // readers drain the queue first: no copy is left in flight
    for (auto& j : h->jobs) {
        if (j.second.fd >= 0) close(j.second.fd);
        if (j.second.fence) (void)hipEventDestroy(j.second.fence);
    }
    for (char* p : h->bufs) (void)hipHostFree(p);
    for (hipStream_t st : h->streams) (void)hipStreamDestroy(st);
    delete h;
    return MSM_OK;
}

}  // extern "C"
