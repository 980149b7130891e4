"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` collectives.

The hot path shards by FRAMES: every rank owns whole trajectories (tICA: lagged pairs
never cross a trajectory boundary, reference tica.py:417, so there is no halo) or a
contiguous block of rows (clustering).  The only exchange steps are

* tICA: ONE all-reduce(sum) of the packed fp64 accumulators
  ``[C (F^2) | G (F^2) | s0 (F) | stau (F) | n_obs | n_seq]`` after the local passes
  (F=512: 4.2 MB, F=2048: 67 MB) -- ``allreduce_tica``;
* MiniBatchKMeans: one all-reduce(sum) per step of ``[K*F sums | K counts | inertia]``;
* KCenters: ``(max distance, global row)`` per centre (tiny, latency-bound).

With the ``nccl`` backend these are RCCL collectives over xGMI on device buffers (the
tICA buffer is exported device-to-device, never staged through the host); with ``gloo``
(CPU-only test runs of the host logic) they go through host tensors.
"""
import os
import warnings

import numpy as np

_enabled = None


def _dist():
    import torch.distributed as dist
    return dist


def active():
    """True when torch.distributed is initialised with more than one rank (and sharded
    operation has not been disabled with MSMBUILDER_AMD_PARALLEL=0)."""
    if os.environ.get("MSMBUILDER_AMD_PARALLEL", "1") == "0":
        return False
    try:
        dist = _dist()
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return _dist().get_rank() if active() else 0


def world_size():
    return _dist().get_world_size() if active() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if present.
    Returns (rank, world_size, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("MSMBUILDER_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=r, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=r, world_size=world)
    return r, world, local


# ---------------------------------------------------------------------------------------------------
# The library's own communicator (csrc/comm.hip): the collectives of the hot path -- tICA's all-reduce, the
# per-centre all-gather of k-centers, MiniBatchKMeans' per-step all-reduce -- are issued by libmsmhip itself on
# its stream between its kernels.  torch.distributed is only the bootstrap (rendezvous + one broadcast of RCCL's
# 128-byte unique id).  With the gloo backend (CPU-only test runs, several ranks sharing one GPU) RCCL cannot
# form a communicator, so the library is given a host-side transport instead: same library code above it.
# ---------------------------------------------------------------------------------------------------
_lib_comm_kind = None      # None | "rccl" | "host"
_comm_failures = []        # ranks whose RCCL join / self-test failed when the communicator was last built (then: host transport)
_host_cb_keepalive = None
_fresh_stream = None         # the stream this thread was moved to after an un-abortable RCCL collective blocked the old one
_lib_comm_key = None       # (backend, world, rank, default group identity) the communicator belongs to
_atexit_registered = False


def _host_collective(op, send, recv, nbytes):
    """msm_host_collective_fn: op 0 = all-reduce(sum) of nbytes/8 doubles in place, op 1 = all-gather of nbytes."""
    import ctypes as C
    try:
        import torch
        dist = _dist()
        nccl = _backend_is_nccl()
        if op == 0:
            a = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(nbytes // 8,))
            t = torch.from_numpy(a)
            if nccl:
                d = t.cuda()
                dist.all_reduce(d, op=dist.ReduceOp.SUM)
                t.copy_(d.cpu())
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            w = dist.get_world_size()
            a = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            o = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(w, nbytes))
            t = torch.from_numpy(a)
            outs = [torch.from_numpy(o[r]) for r in range(w)]
            if nccl:
                douts = [x.cuda() for x in outs]
                dist.all_gather(douts, t.cuda())
                for x, d in zip(outs, douts):
                    x.copy_(d.cpu())
            else:
                dist.all_gather(outs, t)
        return 0
    except Exception:   # never let an exception cross the C boundary
        import traceback
        traceback.print_exc()
        return 1


def _world_key():
    """Identity of the torch.distributed world a library communicator was built for."""
    dist = _dist()
    return (dist.get_backend(), dist.get_world_size(), dist.get_rank(), id(dist.distributed_c10d._get_default_group()))


def library_comm():
    """Make sure libmsmhip has a communicator for the current torch.distributed world (collective call: every
    rank reaches it from the same SPMD entry point).  Returns "rccl", "host" or None (single process).
    A communicator built for an earlier process group (destroyed and re-created with another size / rank / backend)
    is torn down first."""
    global _lib_comm_kind, _lib_comm_key, _host_cb_keepalive
    if not active():
        if _lib_comm_kind is not None:
            library_comm_shutdown()
        return None
    key = _world_key()
    if _lib_comm_kind is not None:
        if key == _lib_comm_key:
            return _lib_comm_kind
        library_comm_shutdown()
    import ctypes as C
    import torch
    from . import _lib
    dist = _dist()
    L = _lib.lib()
    r, w = dist.get_rank(), dist.get_world_size()
    _comm_failures[:] = []
    want = os.environ.get("MSMBUILDER_AMD_COMM", "auto")     # auto | rccl | host
    kind = None
    if want != "host" and _backend_is_nccl():
        _lib.ensure_device()
        # 1) every rank says whether it could join (librccl loadable, device visible) BEFORE anyone enters
        #    ncclCommInitRank, which blocks until all ranks have joined: one rank bailing out early would hang the rest
        usable = torch.tensor([float(L.msm_comm_rccl_available())], device="cuda")
        dist.all_reduce(usable, op=dist.ReduceOp.MIN)
        if usable.item() > 0:
            uid = (C.c_char * 128)()
            ok = 1
            if r == 0:
                ok = 1 if L.msm_comm_unique_id(uid) == 0 else 0
            t = torch.tensor(list(bytes(uid)) + [ok], dtype=torch.uint8, device="cuda")
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().numpy().tolist())
            if raw[128]:
                # 2) join (time-limited inside the library), then 3) one checked all-reduce + all-gather through the new
                #    communicator before any fit depends on it.  Every rank reports; the ranks that failed are named in
                #    `comm_failures` (bench.py prints them) and everybody falls back to the host transport together.
                rc = L.msm_comm_init_rccl(raw[:128], r, w)
                err = _lib.last_error() if rc else ""
                if rc == 0:
                    rc = L.msm_comm_selftest(int(os.environ.get("MSM_COMM_TIMEOUT_S", "180")))
                    err = _lib.last_error() if rc else ""
                if rc and "new stream" in err:
                    # a collective that could not be aborted still blocks the stream the library (= torch's current stream)
                    # was on: everything queued behind it -- the flag exchange right below, the host transport -- would wait
                    # for ever, and with this rank every other rank inside that all_reduce.  Move this thread, and with it
                    # the library, to a fresh stream BEFORE touching torch again (this used to happen after the
                    # flag exchange, i.e. too late).
                    global _fresh_stream
                    _fresh_stream = torch.cuda.Stream()
                    torch.cuda.set_stream(_fresh_stream)
                    _lib.set_stream(_fresh_stream.cuda_stream)
                flags = torch.zeros(w, device="cuda")
                flags[r] = 1.0 if rc == 0 else 0.0
                dist.all_reduce(flags, op=dist.ReduceOp.SUM)
                bad = [i for i, v in enumerate(flags.cpu().tolist()) if v < 0.5]
                if not bad:
                    kind = "rccl"
                else:
                    _comm_failures[:] = [{"rank": i, "error": err if i == r else None} for i in bad]
                    if rc:
                        warnings.warn("libmsmhip RCCL communicator, rank %d of %d: %s" % (r, w, err))
                    L.msm_comm_destroy()
        if kind is None and want == "rccl":
            raise RuntimeError("libmsmhip could not create its RCCL communicator (failed ranks: %s): %s"
                               % ([f["rank"] for f in _comm_failures], _lib.last_error()))
    if kind is None:
        cbt = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
        _host_cb_keepalive = cbt(_host_collective)
        _lib.check(L.msm_comm_init_host(C.cast(_host_cb_keepalive, C.c_void_p), r, w))
        _lib.check(L.msm_comm_selftest(int(os.environ.get("MSM_COMM_TIMEOUT_S", "180"))))
        kind = "host"
    _lib_comm_kind, _lib_comm_key = kind, key
    global _atexit_registered
    if not _atexit_registered:
        import atexit
        atexit.register(library_comm_shutdown)      # before interpreter teardown destroys the process group
        _atexit_registered = True
    return kind


def library_comm_shutdown():
    global _lib_comm_kind, _lib_comm_key, _host_cb_keepalive
    if _lib_comm_kind is not None:
        from . import _lib
        _lib.lib().msm_comm_destroy()
    _lib_comm_kind = None
    _lib_comm_key = None
    _host_cb_keepalive = None


def shard_sequences(sequences, rank_=None, world=None):
    """Greedy longest-first assignment of WHOLE trajectories to ranks (balanced frame
    counts, zero halo).  Returns the indices this rank owns."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    lengths = [len(s) for s in sequences]
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    owner = [0] * len(lengths)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += lengths[i]
    return [i for i in range(len(lengths)) if owner[i] == rank_]


def split_frames(lengths, lag_time, rank_=None, world=None):
    """Balanced split of the FRAME axis over ranks, cutting inside trajectories where needed.

    The usable trajectories (length > lag_time) are laid end to end and the axis is cut into
    ``world`` equal spans; a rank owns, as LEFT frames of the lagged pairs, the rows of every
    trajectory that fall into its span.  Returns this rank's ``[(sequence index, own_begin,
    own_end), ...]``.  To accumulate a piece a rank needs rows ``[own_begin, min(own_end +
    lag_time, length))`` of that trajectory (the right halo of ``lag_time`` rows is read, not
    owned).  Trajectories shorter than the lag are skipped by every rank, like tica.py:410-412.
    """
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    usable = [(i, int(n)) for i, n in enumerate(lengths) if int(n) > lag_time]
    total = sum(n for _, n in usable)
    lo = (total * rank_) // world
    hi = (total * (rank_ + 1)) // world
    out, pos = [], 0
    for i, n in usable:
        b, e = max(lo, pos), min(hi, pos + n)
        if e > b:
            out.append((i, b - pos, e - pos))
        pos += n
    return out


def _backend_is_nccl(group=None):
    return _dist().get_backend(group) == "nccl"


def allreduce_array(a, group=None, op="sum"):
    """Reduce (sum or max) a host float64 numpy array over all ranks; returns the reduced copy."""
    if not active():
        return a
    import torch
    dist = _dist()
    t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
    if _backend_is_nccl(group):
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def all_gather_rows(out2d, row, group=None):
    """out2d[r] = rank r's `row` (device tensors; RCCL all-gather with the nccl backend)."""
    dist = _dist()
    try:
        dist.all_gather_into_tensor(out2d.view(-1), row, group=group)
    except (RuntimeError, NotImplementedError):   # backends without the flat variant (gloo)
        dist.all_gather([out2d[r] for r in range(out2d.shape[0])], row, group=group)
    return out2d


def allreduce_tica(model, group=None):
    """All-reduce(sum) a fitted local tICA's accumulators in place: ``msm_tica_allreduce`` -- the packed fp64
    buffer is exported, reduced over the library communicator (RCCL over xGMI) and re-imported device to device."""
    if not active():
        return model
    from . import _lib
    L = _lib.lib()
    if not model._initialized:
        raise RuntimeError("allreduce() before any data was seen on this rank")
    model._ensure_handle()
    library_comm()
    import torch
    if torch.cuda.is_available():
        _lib.set_stream(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.msm_tica_allreduce(model._handle))
    import ctypes as C
    nobs, nseq = C.c_int64(0), C.c_int64(0)
    _lib.check(L.msm_tica_counts(model._handle, C.byref(nobs), C.byref(nseq)))
    model.n_observations_ = int(nobs.value)
    model.n_sequences_ = int(nseq.value)
    model._host_stale = True
    model._is_dirty = True
    model._mu_raw = None   # a mean cached by a device solve of the LOCAL accumulators is not the reduced model's
    return model


class RowShard:
    """Bookkeeping for a row-sharded array: this rank owns global rows [offset, offset+n_local)."""

    def __init__(self, n_local):
        self.n_local = int(n_local)
        if active():
            counts = allreduce_array(np.eye(world_size())[rank()] * n_local)
            self.counts = counts.astype(np.int64)
        else:
            self.counts = np.array([n_local], dtype=np.int64)
        self.offset = int(self.counts[:rank()].sum())
        self.n_total = int(self.counts.sum())

    def local(self, global_idx):
        """(positions in global_idx that this rank owns, their shard-local row numbers)"""
        global_idx = np.asarray(global_idx, dtype=np.int64)
        mine = (global_idx >= self.offset) & (global_idx < self.offset + self.n_local)
        pos = np.nonzero(mine)[0]
        return pos, global_idx[pos] - self.offset

    def gather_rows(self, fetch_local, global_idx, n_features, dtype=np.float32):
        """Rows global_idx of the sharded array on every rank: each rank fills in the rows it
        owns (``fetch_local(local_rows) -> [len, F]``), one all-reduce(sum) completes them."""
        out = np.zeros((len(global_idx), n_features), dtype=np.float64)
        pos, loc = self.local(global_idx)
        if len(pos):
            out[pos] = fetch_local(loc)
        if active():
            out = allreduce_array(out.ravel()).reshape(out.shape)
        return out.astype(dtype)
