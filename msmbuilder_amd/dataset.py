"""dir-npy datasets feeding the hot path.

``NumpyDirDataset`` mirrors the reference container of the same name
(/root/reference/msmbuilder/dataset.py:290-331 and the ``_BaseDataset`` protocol, :100-277): a
directory of ``%08d.npy`` files, one 2-D array per trajectory, with ``keys() / get(i, mmap) /
set(i, x) / __getitem__ / __setitem__ / __len__ / items() / fit_with / transform_with /
fit_transform_with`` -- host arrays through numpy exactly as upstream.

``device_sequences()`` is the MI355X side: a re-iterable view whose iteration streams every
trajectory straight into HBM through the native loader of csrc/npyio.hip (worker-thread pread
into pinned buffers + hipMemcpyAsync on a private stream, ``prefetch`` files in flight), so
``tICA().fit(ds.device_sequences())`` overlaps disk, PCIe and the covariance kernel.
"""
import ctypes as C
import os
import re
from os.path import exists, join

import numpy as np

from . import _lib
from ._lib import check

__all__ = ['NumpyDirDataset', 'dataset']


def _keynat(string):
    """natural sort key (dataset.py:451-463)"""
    r = []
    for c in string:
        if c.isdigit():
            if r and isinstance(r[-1], int):
                r[-1] = r[-1] * 10 + int(c)
            else:
                r.append(int(c))
        else:
            r.append(9 + ord(c))
    return r


def dataset(path, mode='r', fmt=None, verbose=False, **kwargs):
    """Open a dir-npy dataset (the only on-disk format in scope here; dataset.py:30-97)."""
    if fmt not in (None, 'dir-npy'):
        raise NotImplementedError("msmbuilder_amd.dataset only implements fmt='dir-npy', got %r" % (fmt,))
    return NumpyDirDataset(path, mode=mode, verbose=verbose)


class NumpyDirDataset(object):
    _ITEM_FORMAT = '%08d.npy'
    _ITEM_RE = re.compile(r'(\d{8}).npy')
    _PROVENANCE_FILE = 'PROVENANCE.txt'

    def __init__(self, path, mode='r', verbose=False):
        self.path = path
        self.mode = mode
        self.verbose = verbose
        if mode not in ('r', 'w', 'a'):
            raise ValueError('mode must be one of "r", "w", "a"')
        if mode in 'wa':
            if mode == 'w' and exists(path):
                raise ValueError('File exists: %s' % path)
            try:
                os.makedirs(path)
            except OSError:
                pass
            self._write_provenance()

    # ------------------------------------------------------------------ container protocol
    def get(self, i, mmap=False):
        filename = join(self.path, self._ITEM_FORMAT % i)
        if self.verbose:
            print('[NumpydirDataset] loading %s' % filename)
        try:
            return np.load(filename, 'r' if mmap else None)
        except IOError as e:
            raise IndexError(e)

    def set(self, i, x):
        if self.mode not in 'wa':
            raise IOError('Dataset not opened for writing')
        filename = join(self.path, self._ITEM_FORMAT % i)
        if self.verbose:
            print('[NumpydirDataset] saving %s' % filename)
        if hasattr(x, "is_cuda"):
            x = x.cpu().numpy()
        return np.save(filename, x)

    def keys(self):
        for fn in sorted(os.listdir(os.path.expanduser(self.path)), key=_keynat):
            match = self._ITEM_RE.match(fn)
            if match:
                yield int(match.group(1))

    def items(self):
        for key in self.keys():
            yield (key, self.get(key))

    def __iter__(self):
        for key in self.keys():
            yield self.get(key)

    def __len__(self):
        return sum(1 for _ in self.keys())

    def __getitem__(self, i):
        return self.get(i)

    def __setitem__(self, i, x):
        return self.set(i, x)

    def close(self):
        pass

    def flush(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc_info):
        self.close()

    @property
    def provenance(self):
        try:
            with open(join(self.path, self._PROVENANCE_FILE), 'r') as f:
                return f.read()
        except IOError:
            return 'No available provenance'

    def _write_provenance(self, previous=None, comments=''):
        from . import __version__
        with open(join(self.path, self._PROVENANCE_FILE), 'w') as f:
            f.write('MSMBuilder Dataset:\n  msmbuilder_amd:\t%s\n  Path:\t\t%s\n  Comments:\t\t%s\n'
                    % (__version__, self.path, comments))
            if previous:
                f.write('\n== Derived from ==\n%s\n' % previous)

    def create_derived(self, out_path, comments='', fmt=None):
        out = dataset(out_path, mode='w', verbose=self.verbose, fmt=fmt)
        out._write_provenance(previous=self.provenance, comments=comments)
        return out

    # -------------------------------------------------------------- estimator hooks (dataset.py:158-233)
    def fit_with(self, estimator):
        estimator.fit(self)
        return estimator

    def transform_with(self, estimator, out_ds, fmt=None):
        if isinstance(out_ds, str):
            out_ds = self.create_derived(out_ds, fmt=fmt)
        elif getattr(out_ds, "mode", "w") not in 'wa':
            raise ValueError('out_ds must be opened for writing')
        for key in self.keys():
            out_ds[key] = estimator.partial_transform(self.get(key))
        return out_ds

    def fit_transform_with(self, estimator, out_ds, fmt=None):
        self.fit_with(estimator)
        return self.transform_with(estimator, out_ds, fmt=fmt)

    # ------------------------------------------------------------------------- device streaming
    def device_sequences(self, prefetch=4, buffer_bytes=8 << 20, readers=4, device=None):
        """Re-iterable view yielding each trajectory as a ``torch`` CUDA tensor loaded by the
        native pipelined reader (float32 / float64 / int32 / int64 C-ordered files): ``readers``
        threads with one pinned ``buffer_bytes`` buffer each, ``prefetch`` files in flight."""
        return _DeviceView(self, prefetch, buffer_bytes, device, readers)


_TORCH_DTYPES = {('f', 4): 'float32', ('f', 8): 'float64', ('i', 4): 'int32', ('i', 8): 'int64',
                 ('u', 1): 'uint8', ('b', 1): 'bool', ('i', 2): 'int16', ('i', 1): 'int8'}


class _DeviceView(object):
    def __init__(self, ds, prefetch, buffer_bytes, device, readers=4):
        self.ds = ds
        self.readers = max(1, int(readers))
        self.prefetch = max(1, int(prefetch))
        self.buffer_bytes = int(buffer_bytes)
        self.device = device
        self._keys = list(ds.keys())

    def __len__(self):
        return len(self._keys)

    def _info(self, key):
        path = join(self.ds.path, self.ds._ITEM_FORMAT % key).encode()
        nb, kind, fo, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        shape, off = (C.c_int64 * 4)(), C.c_int64()
        check(_lib.lib().msm_npy_info(path, C.byref(nb), C.byref(kind), C.byref(fo), C.byref(nd), shape, C.byref(off)))
        if fo.value and nd.value > 1:
            raise ValueError("%s is Fortran-ordered; the device loader needs C-ordered arrays" % path.decode())
        dt = _TORCH_DTYPES.get((chr(kind.value), nb.value))
        if dt is None:
            raise TypeError("%s: dtype %s%d has no device loader" % (path.decode(), chr(kind.value), nb.value))
        return path, dt, tuple(shape[i] for i in range(nd.value))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self._stream(self._keys[i]))
        for x in self._stream([self._keys[i]]):      # list semantics: negative indices count from the end, IndexError beyond
            return x

    def __iter__(self):
        return self._stream(self._keys)

    def _stream(self, keys):
        import torch
        dev = torch.device("cuda", _lib.ensure_device(self.device))
        L = _lib.lib()
        h = C.c_void_p()
        check(L.msm_npy_loader_create(C.byref(h), self.readers, self.buffer_bytes))
        try:
            pending = []           # (job id, tensor)
            it = iter(keys)

            def submit():
                key = next(it, None)
                if key is None:
                    return False
                path, dt, shape = self._info(key)
                t = torch.empty(shape, dtype=getattr(torch, dt), device=dev)
                # `t` may re-use a block that kernels queued on torch's stream are still reading (a trajectory the consumer
                # has already dropped): submit() records a fence on the library stream, which must be that stream
                _lib.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                job = C.c_int64(0)
                check(L.msm_npy_loader_submit(h, path, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), C.byref(job)))
                pending.append((job.value, t))
                return True

            for _ in range(self.prefetch):
                if not submit():
                    break
            while pending:
                job, t = pending.pop(0)
                check(L.msm_npy_loader_wait(h, job))
                submit()
                yield t
        finally:
            L.msm_npy_loader_destroy(h)
