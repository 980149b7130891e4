"""ctypes binding of libmsmhip.so (include/msmhip.h).

The product path has no CPU fallback: if the shared library is missing, cannot
be loaded, or no gfx950 device is visible when a compute entry point is called,
an exception is raised -- nothing is silently routed elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsmhip.so")

MSM_OK = 0
MSM_ERR_INVALID = -1
MSM_ERR_METRIC = -2
MSM_ERR_HIP = -3
MSM_ERR_NODEVICE = -4
MSM_ERR_NONFINITE = -5
MSM_ERR_STATE = -6

TICA_F32 = 0
TICA_F64 = 1
TICA_BF16 = 2
TICA_BF16X2 = 3

_i64 = C.c_int64
_p = C.c_void_p
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)

_lib = None
_initialized_device = None


class MsmHipError(RuntimeError):
    pass


class NoDeviceError(MsmHipError):
    pass


def _declare(lib):
    def f(name, restype, *argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    f("msm_last_error", C.c_char_p)
    f("msm_version", C.c_char_p)
    f("msm_device_count", C.c_int)
    f("msm_init", C.c_int, C.c_int)
    f("msm_set_stream", C.c_int, _p)
    f("msm_synchronize", C.c_int)
    f("msm_device_info", C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), _i64p)
    f("msm_malloc", C.c_int, C.POINTER(_p), C.c_size_t)
    f("msm_free", C.c_int, _p)
    f("msm_memcpy_h2d", C.c_int, _p, _p, C.c_size_t)
    f("msm_memcpy_d2h", C.c_int, _p, _p, C.c_size_t)
    f("msm_memcpy_d2d", C.c_int, _p, _p, C.c_size_t)
    f("msm_upload_list", C.c_int, _p, C.POINTER(_p), _i64p, _i64)
    f("msm_gather_rows", C.c_int, _p, C.c_int, _i64, _p, _i64, _p, C.c_int)
    f("msm_event_create", C.c_int, C.POINTER(_p))
    f("msm_event_record", C.c_int, _p)
    f("msm_event_elapsed_ms", C.c_int, _p, _p, C.POINTER(C.c_float))
    f("msm_event_destroy", C.c_int, _p)

    f("msm_tica_create", C.c_int, C.POINTER(_p), _i64, _i64, C.c_int)
    f("msm_tica_destroy", C.c_int, _p)
    f("msm_tica_reset", C.c_int, _p)
    f("msm_tica_accumulate", C.c_int, _p, _p, C.c_int, _i64, _i64, C.c_int, C.c_int, C.POINTER(C.c_int))
    f("msm_tica_accumulate_batch", C.c_int, _p, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, C.c_int,
      C.c_int, _i64p)
    f("msm_tica_accumulate_segments", C.c_int, _p, C.POINTER(_p), _i64p, _i64p, _i64, C.c_int, _i64, C.c_int,
      C.c_int, _i64p)
    f("msm_tica_nonfinite", C.c_int, _p, C.POINTER(C.c_int))
    f("msm_tica_lagged_symmetrised", C.c_int, _p, C.POINTER(C.c_int))
    f("msm_tica_last_kernel_ms", C.c_int, _p, C.POINTER(C.c_float))
    f("msm_tica_debug_clocks", C.c_int, _p, _i64p)
    f("msm_tica_debug_profile", C.c_int, _p, _i64p)
    f("msm_tica_export", C.c_int, _p, _p, _p, _p, _p, _i64p, _i64p)
    f("msm_tica_import", C.c_int, _p, _p, _p, _p, _p, _i64, _i64)
    f("msm_tica_packed_size", _i64, _p)
    f("msm_tica_export_packed", C.c_int, _p, _p, C.c_int)
    f("msm_tica_import_packed", C.c_int, _p, _p, C.c_int)
    f("msm_tica_project", C.c_int, _p, C.c_int, _i64, _i64, _i64, _p, _p, _i64, _p, C.c_int, C.c_int)
    f("msm_tica_project_batch", C.c_int, C.POINTER(_p), C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _p, _p, _i64, C.c_int)
    f("msm_tica_project_host_list", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _p, _p, _i64, _p, C.c_int)
    f("msm_tica_last_folded", C.c_int, _p, C.POINTER(C.c_int))
    f("msm_tica_last_img_fused", C.c_int, _p, C.POINTER(C.c_int))
    f("msm_tica_last_img_carried", C.c_int, _p, C.POINTER(C.c_int))
    f("msm_tica_allreduce", C.c_int, _p)
    f("msm_tica_counts", C.c_int, _p, _i64p, _i64p)
    f("msm_comm_rccl_available", C.c_int)
    f("msm_comm_unique_id", C.c_int, _p)
    f("msm_comm_init_rccl", C.c_int, _p, C.c_int, C.c_int)
    f("msm_comm_init_host", C.c_int, _p, C.c_int, C.c_int)
    f("msm_comm_destroy", C.c_int)
    f("msm_comm_selftest", C.c_int, C.c_int)
    f("msm_comm_info", C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
    f("msm_comm_allreduce_f64", C.c_int, _p, _i64)
    f("msm_comm_allgather", C.c_int, _p, _p, _i64)
    f("msm_comm_measure", C.c_int, C.c_int, _i64, C.c_int, C.POINTER(C.c_float))
    f("msm_mbk_zero_packed", C.c_int, _p)
    f("msm_mbk_allreduce", C.c_int, _p, _f64p, _p)
    for sfx in ("f32", "f64"):
        f("msm_kcenters_fit_sharded_" + sfx, C.c_int, _p, _i64, _i64, _i64, C.c_char_p, _i64, _i64, _p, _p, _p, _p, _f64p)
    f("msm_kcenters_last_stats", C.c_int, _i64p)
    f("msm_kcenters_last_wide_stats", C.c_int, _i64p)
    f("msm_tica_export_sums", C.c_int, _p, _p, _p)
    f("msm_tica_reduce", C.c_int, _p, C.c_double, _i64, _p, _p, _p, _p)
    f("msm_tica_backsolve", C.c_int, _p, _p, _i64, _p)
    f("msm_tica_solve_topk", C.c_int, _p, C.c_double, _i64, _p, _i64, _p, _p, _p, _p, _p, C.POINTER(C.c_int))
    f("msm_potrf", C.c_int, _p, _i64, C.POINTER(C.c_int), C.c_int)
    f("msm_tica_solve_device", C.c_int, _p, C.c_double, _i64, _p, _i64, _p, _p, _p, _p)

    f("msm_label_range", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64p, _i64p, _i64p)
    f("msm_label_histogram", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _i64, _p)
    f("msm_transition_counts", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _i64, _p, _i64, _i64, _p)
    f("msm_npy_info", C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
      _i64p, _i64p)
    f("msm_npy_loader_create", C.c_int, C.POINTER(_p), C.c_int, C.c_size_t)
    f("msm_npy_loader_submit", C.c_int, _p, C.c_char_p, _p, _i64, _i64p)
    f("msm_npy_loader_wait", C.c_int, _p, _i64)
    f("msm_npy_loader_destroy", C.c_int, _p)
    f("msm_sygv_top", C.c_int, _p, _p, _i64, _i64, _p, _p, C.c_int)
    f("msm_colstats", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _i64, C.c_int, _p, C.POINTER(C.c_int))
    f("msm_col_digit_hist", C.c_int, C.POINTER(_p), _i64p, _i64, C.c_int, _i64, _i64, _p, C.c_int, C.c_int, C.c_int, _p)
    f("msm_scale_apply", C.c_int, _p, C.c_int, _i64, _i64, _i64, _p, _p, C.c_int, _p, _i64, C.c_int)

    for sfx in ("f32", "f64"):
        f("msm_dist_" + sfx, C.c_int, _p, _p, C.c_char_p, _i64, _i64, _p, _i64, _p, C.c_int)
        f("msm_pdist_" + sfx, C.c_int, _p, C.c_char_p, _i64, _i64, _p, _i64, _p, C.c_int)
        f("msm_sumdist_" + sfx, C.c_int, _p, C.c_char_p, _i64, _i64, _p, _i64, _f64p, C.c_int)
        f("msm_cdist_" + sfx, C.c_int, _p, _p, C.c_char_p, _i64, _i64, _i64, _p, C.c_int)
        f("msm_assign_nearest_" + sfx, C.c_int, _p, _p, C.c_char_p, _p, _i64, _i64, _i64, _i64, _p, _p,
          _f64p, C.c_int)
        f("msm_kcenters_fit_" + sfx, C.c_int, _p, _i64, _i64, _i64, C.c_char_p, _i64, _p, _p, _p, _f64p,
          C.c_int)
        f("msm_kcenters_fit2_" + sfx, C.c_int, _p, _i64, _i64, _i64, C.c_char_p, _i64, _p, _p, _p, _f64p,
          C.c_int, _p)
    for sfx in ("f32", "f64"):
        f("msm_kcenters_pass_" + sfx, C.c_int, _p, _i64, _i64, _p, _i64, C.c_char_p, _p, _p, _f64p, _i64p, _p, C.c_int)
        f("msm_kcenters_pass_dev_" + sfx, C.c_int, _p, _i64, _i64, _p, _i64, C.c_char_p, _p, _p, _i64, _p)
        f("msm_kcenters_select_" + sfx, C.c_int, _p, _i64, _i64, _p, _p, _p, _i64)
    for sfx in ("f32", "f64"):
        f("msm_kmeans_label_" + sfx, C.c_int, _p, _i64, _i64, _p, _i64, _p, _f64p, C.c_int)
        f("msm_kmeans_plusplus_" + sfx, C.c_int, _p, _i64, _i64, _i64, _i64, _p, C.c_int, _p, _p, C.c_int)
        f("msm_mbk_step_" + sfx, C.c_int, _p, _i64, _i64, _p, _i64, _p, _p, _i64, _f64p, _p, _p, C.c_int, C.c_int)
    f("msm_mbk_create", C.c_int, C.POINTER(_p), _i64, _i64)
    f("msm_mbk_create_f64", C.c_int, C.POINTER(_p), _i64, _i64)
    f("msm_mbk_is_f64", C.c_int, _p)
    f("msm_mbk_destroy", C.c_int, _p)
    f("msm_mbk_set", C.c_int, _p, _p, _p)
    f("msm_mbk_set_counts", C.c_int, _p, _p)
    f("msm_mbk_get", C.c_int, _p, _p, _p)
    f("msm_mbk_step", C.c_int, _p, _p, _i64, _p, _i64, _f64p, _p, C.c_int, C.c_int)
    f("msm_mbk_run_begin", C.c_int, _p, _p, _i64, _p, _i64, _i64, _i64, C.c_double, _i64, _p)
    f("msm_mbk_run_end", C.c_int, _p, _p, C.POINTER(C.c_int64), C.POINTER(C.c_int), _p, _p)
    f("msm_mbk_run", C.c_int, _p, _p, _i64, _p, _i64, _i64, _i64, C.c_double, _i64, _p, C.POINTER(C.c_int64),
      C.POINTER(C.c_int), _p, _p)
    f("msm_mbk_run_sharded", C.c_int, _p, _p, _i64, _p, _p, _i64, _i64, _i64, C.c_double, _i64, _p, C.POINTER(C.c_int64),
      C.POINTER(C.c_int), _p, _p)
    f("msm_mbk_packed_size", _i64, _p)
    f("msm_mbk_export_packed", C.c_int, _p, _p, C.c_int)
    f("msm_mbk_apply_packed", C.c_int, _p, _p, _p, C.c_int)
    f("msm_mbk_reassign", C.c_int, _p, _p, _i64, _p, _p, _i64, C.c_double, C.c_int)
    f("msm_mbk_label", C.c_int, _p, _p, _i64, _p, _f64p, C.c_int)


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsmHipError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C msmbuilder_amd/csrc` (hipcc --offload-arch=gfx950). "
                "msmbuilder_amd has no CPU fallback." % LIB_PATH)
        try:
            # torch (if installed) must bring in ITS libamdhip64 first so both share one HIP runtime
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def last_error() -> str:
    return lib().msm_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc == MSM_OK:
        return
    msg = last_error()
    if rc == MSM_ERR_NONFINITE:
        raise ValueError(msg)
    if rc == MSM_ERR_METRIC:
        raise ValueError(msg)
    if rc == MSM_ERR_INVALID:
        raise ValueError(msg)
    if rc == MSM_ERR_NODEVICE:
        raise NoDeviceError(msg + " (msmbuilder_amd needs an MI355X/gfx950 GPU; there is no CPU fallback)")
    raise MsmHipError("libmsmhip error %d: %s" % (rc, msg))


def device_count() -> int:
    return lib().msm_device_count()


def ensure_device(device=None):
    """Select the GPU (LOCAL_RANK-aware) once per process and bind torch's current stream."""
    global _initialized_device
    L = lib()
    if _initialized_device is None or (device is not None and device != _initialized_device):
        if device is None:
            device = int(os.environ.get("MSMBUILDER_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = L.msm_device_count()
            if n > 0:
                device %= n
        check(L.msm_init(int(device)))
        _initialized_device = int(device)
    return _initialized_device


def set_stream(stream_handle):
    check(lib().msm_set_stream(C.c_void_p(stream_handle or 0)))


def synchronize():
    check(lib().msm_synchronize())


# --------------------------------------------------------------------------
# array plumbing: numpy (host) or torch CUDA tensors (device)
# --------------------------------------------------------------------------
def is_device_array(x) -> bool:
    return hasattr(x, "data_ptr") and bool(getattr(x, "is_cuda", False))


def adjacent_rows(sequences):
    """Row counts (int64 array) of a list of 2-D torch CUDA tensors that are contiguous views of ONE tensor lying back to
    back in its storage, first to last -- or None.  A few list comprehensions: a thousand trajectories are checked in
    well under a millisecond (a Python loop with six checks per trajectory is 1 us each)."""
    head = sequences[0]
    base = getattr(head, "_base", None)
    if base is None or head.dim() != 2 or not head.is_contiguous():
        return None
    if not all(getattr(s, "_base", None) is base for s in sequences):   # views of ONE tensor: one storage, one dtype, one device
        return None
    import operator
    F = head.shape[1]
    shapes = [tuple(s.shape) for s in sequences]
    if any(len(sh) != 2 or sh[1] != F for sh in shapes) or len(set(map(operator.attrgetter("dtype"), sequences))) != 1:
        return None
    rows = np.array([sh[0] for sh in shapes], dtype=np.int64)
    ptrs = np.array(list(map(type(head).data_ptr, sequences)), dtype=np.int64)   # (unbound methods through map: half the
    want = ptrs[0] + np.concatenate(([0], np.cumsum(rows[:-1]))) * (F * head.element_size())   #  cost of a comprehension)
    live = rows > 0   # an empty trajectory occupies nothing (and its pointer need not follow its neighbours)
    if not np.array_equal(ptrs[live], want[live]) or not all(map(type(head).is_contiguous, sequences)):
        return None
    return rows


def adjacent_view(sequences):
    """ONE [total, F] array over a list of 2-D trajectories that lie back to back in one allocation -- what
    ``X.view(n, T, F).unbind(0)``, ``np.split`` or slices of a joined array give --, or None.  Lets the per-frame
    operations (`transform`, `predict`) run as one launch over all trajectories instead of one small launch each,
    without copying anything."""
    if len(sequences) < 2:
        return None
    head = sequences[0]
    if hasattr(head, "shape") and len(head.shape) == 2 and head.shape[0] == 0:
        return None   # (keep it simple: the first trajectory anchors the view)
    try:
        if is_device_array(head):
            import torch
            rows = adjacent_rows(sequences)
            if rows is None:
                return None
            return torch.as_strided(head, (int(rows.sum()), head.shape[1]), (head.shape[1], 1))
        if isinstance(head, np.ndarray):
            if head.ndim != 2 or not head.flags.c_contiguous or head.base is None:
                return None
            root = head.base
            while getattr(root, "base", None) is not None:
                root = root.base
            F, total, nxt = head.shape[1], 0, head.ctypes.data
            for s in sequences:
                if not isinstance(s, np.ndarray) or s.ndim != 2 or s.shape[1] != F or s.dtype != head.dtype:
                    return None
                if s.shape[0] == 0:
                    continue
                if not s.flags.c_contiguous or s.ctypes.data != nxt:
                    return None
                r = s.base
                while getattr(r, "base", None) is not None:
                    r = r.base
                if r is not root:
                    return None
                nxt += s.nbytes
                total += s.shape[0]
            return np.lib.stride_tricks.as_strided(head, shape=(total, F), strides=(F * head.itemsize, head.itemsize), writeable=False)
    except Exception:
        return None
    return None


def cut_rows(joined, lengths):
    """Per-trajectory views of an array / tensor indexed like the joined frames (one C++ call for a thousand pieces)."""
    lengths = [int(n) for n in lengths]
    if hasattr(joined, "split") and not isinstance(joined, np.ndarray):
        return list(joined.split(lengths)) if lengths else []
    out, start = [], 0
    for n in lengths:
        out.append(joined[start:start + n])
        start += n
    return out


class Arr:
    """(pointer, shape, dtype, on_device) view of a numpy array or torch CUDA tensor."""
    __slots__ = ("ptr", "shape", "dtype", "on_device", "keep")

    def __init__(self, x, dtype=None):
        if is_device_array(x):
            import torch
            if dtype is not None:
                want = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                        np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32}[np.dtype(dtype)]
                if x.dtype != want:
                    x = x.to(want)
            x = x.contiguous()
            ensure_device(x.device.index)
            set_stream(torch.cuda.current_stream(x.device).cuda_stream)
            self.ptr = x.data_ptr()
            self.shape = tuple(x.shape)
            self.dtype = np.dtype(str(x.dtype).replace("torch.", ""))
            self.on_device = 1
        else:
            x = np.ascontiguousarray(x, dtype=dtype)
            ensure_device()
            self.ptr = x.ctypes.data
            self.shape = x.shape
            self.dtype = x.dtype
            self.on_device = 0
        self.keep = x

    @property
    def vp(self):
        return C.c_void_p(self.ptr)


def empty_like_placement(ref: Arr, shape, dtype):
    """Allocate an output next to `ref`: torch CUDA tensor for device inputs, numpy otherwise."""
    if ref.on_device:
        import torch
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.int64): torch.int64,
               np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}[np.dtype(dtype)]
        return torch.empty(shape, dtype=tdt, device=ref.keep.device)
    return np.zeros(shape, dtype=dtype)
