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
    """All-reduce(sum) a fitted local tICA's accumulators in place."""
    if not active():
        return model
    import ctypes as C
    import torch
    from . import _lib
    dist = _dist()
    L = _lib.lib()
    if not model._initialized:
        raise RuntimeError("allreduce() before any data was seen on this rank")
    model._ensure_handle()
    n = int(L.msm_tica_packed_size(model._handle))
    if _backend_is_nccl(group):
        buf = torch.empty(n, dtype=torch.float64, device="cuda")
        _lib.set_stream(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.msm_tica_export_packed(model._handle, C.c_void_p(buf.data_ptr()), 1))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        torch.cuda.current_stream().synchronize()
        _lib.check(L.msm_tica_import_packed(model._handle, C.c_void_p(buf.data_ptr()), 1))
        counts = buf[-2:].cpu().numpy()
    else:
        host = np.empty(n, dtype=np.float64)
        _lib.check(L.msm_tica_export_packed(model._handle, C.c_void_p(host.ctypes.data), 0))
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        _lib.check(L.msm_tica_import_packed(model._handle, C.c_void_p(host.ctypes.data), 0))
        counts = host[-2:]
    model.n_observations_ = int(round(counts[0]))
    model.n_sequences_ = int(round(counts[1]))
    model._host_stale = True
    model._is_dirty = True
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
