"""dir-npy trajectory stores feeding the hot path.

On-disk contract (the reference's "dir-npy" format, /root/reference/msmbuilder/dataset.py:290-331): a
directory holding one ``%08d.npy`` file per trajectory -- the integer in the name is the key,
keys iterate in ascending order -- plus a free-text ``PROVENANCE.txt``.  Estimator hooks
(dataset.py:158-237): ``fit_with(est)``, ``transform_with(est, out)``, ``fit_transform_with(est, out)``.

This container is built around the native reader of csrc/npyio.hip rather than around ``np.load``:

* every file is described once by ``msm_npy_info`` (dtype, shape, payload offset); host reads are
  ``np.memmap`` / ``np.fromfile`` views of that payload, so ``get(i, mmap=True)`` costs no copy and
  no header re-parse in Python;
* ``device_sequences()`` streams the same payloads straight into HBM (reader threads pread into pinned
  buffers, hipMemcpyAsync on private streams, ``prefetch`` files in flight), so
  ``tICA().fit(ds.device_sequences())`` overlaps disk, PCIe and the covariance kernel;
* writes go to a temporary name and are renamed into place, so a reader never sees half a trajectory.
"""
import ctypes as C
import os
import re

import numpy as np

from . import _lib
from ._lib import check

__all__ = ['NumpyDirDataset', 'dataset']

_NAME = re.compile(r'\d{8}\.npy')
_NOTES = 'PROVENANCE.txt'
_NP_KINDS = {'f': 'f', 'i': 'i', 'u': 'u', 'b': 'b'}
_TORCH_NAMES = {'f4': 'float32', 'f8': 'float64', 'i4': 'int32', 'i8': 'int64', 'u1': 'uint8', 'b1': 'bool',
                'i2': 'int16', 'i1': 'int8', 'f2': 'float16'}


class _Payload(object):
    """What msm_npy_info says about one file."""
    __slots__ = ('path', 'code', 'shape', 'offset', 'fortran')

    def __init__(self, path):
        nb, kind, fo, nd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        shape, off = (C.c_int64 * 4)(), C.c_int64()
        check(_lib.lib().msm_npy_info(os.fsencode(path), C.byref(nb), C.byref(kind), C.byref(fo), C.byref(nd),
                                      shape, C.byref(off)))
        self.path = path
        self.code = '%s%d' % (_NP_KINDS[chr(kind.value)], nb.value)
        self.shape = tuple(int(shape[d]) for d in range(nd.value))
        self.offset = int(off.value)
        self.fortran = bool(fo.value)

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * int(self.code[1:])

    def host_array(self, mapped):
        dt = np.dtype('<' + self.code if self.code[1:] != '1' else '|' + self.code)
        order = 'F' if self.fortran else 'C'
        if mapped:
            if 0 in self.shape:
                return np.empty(self.shape, dtype=dt, order=order)
            return np.memmap(self.path, dtype=dt, mode='r', offset=self.offset, shape=self.shape, order=order)
        flat = np.fromfile(self.path, dtype=dt, count=int(np.prod(self.shape, dtype=np.int64)), offset=self.offset)
        return flat.reshape(self.shape, order=order)


def dataset(path, mode='r', fmt=None, verbose=False, **kwargs):
    """Open a trajectory store.  Only ``fmt='dir-npy'`` (or None) exists here: the reference's HDF5 and
    mdtraj containers (dataset.py:30-97) are file formats outside the hot path."""
    if fmt not in (None, 'dir-npy'):
        raise NotImplementedError("msmbuilder_amd.dataset only implements fmt='dir-npy', got %r" % (fmt,))
    return NumpyDirDataset(path, mode=mode, verbose=verbose)


class NumpyDirDataset(object):
    def __init__(self, path, mode='r', verbose=False):
        if mode not in ('r', 'w', 'a'):
            raise ValueError('mode must be one of "r", "w", "a"')
        self.path, self.mode, self.verbose = path, mode, verbose
        if self.writable:
            if mode == 'w' and os.path.exists(path):
                raise ValueError('File exists: %s' % path)
            os.makedirs(path, exist_ok=True)
            self._note()

    writable = property(lambda self: self.mode != 'r')

    def _file(self, key):
        return os.path.join(self.path, '%08d.npy' % key)

    def _say(self, what, name):
        if self.verbose:
            print('[dir-npy] %s %s' % (what, name))

    # ------------------------------------------------------------------ keyed access
    def keys(self):
        with os.scandir(os.path.expanduser(self.path)) as entries:
            found = sorted(int(e.name[:8]) for e in entries if _NAME.fullmatch(e.name))
        return iter(found)

    def get(self, i, mmap=False):
        name = self._file(i)
        if not os.path.isfile(name):
            raise IndexError('%s: no trajectory %r in this dataset' % (name, i))
        self._say('reading', name)
        try:
            return _Payload(name).host_array(mmap)
        except (ValueError, OSError, _lib.MsmHipError):
            # structured / object / big-endian payloads are not device material; numpy still reads them -- and a plain
            # host read of a file (no compute) must not depend on libmsmhip.so being loadable on this machine
            return np.load(name, mmap_mode='r' if mmap else None, allow_pickle=False)

    def set(self, i, x):
        if not self.writable:
            raise IOError('Dataset not opened for writing')
        if hasattr(x, 'detach'):                       # torch tensor, possibly resident in HBM
            x = x.detach().cpu().numpy()
        name = self._file(i)
        self._say('writing', name)
        # a private temporary file next to the target, created with mode 0666 so that the KERNEL applies the umask (np.save,
        # what the reference uses at dataset.py:323, gets the same mode; reading the umask with os.umask(0) would open a window
        # in which another thread's files are created world-writable), then an atomic rename
        tmp = None
        for attempt in range(100):
            cand = os.path.join(self.path, '.%s.%d.%d.part' % (os.path.basename(name), os.getpid(), attempt))
            try:
                fd = os.open(cand, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o666)
            except FileExistsError:
                continue
            tmp = cand
            break
        if tmp is None:
            raise IOError('could not create a temporary file in %s' % self.path)
        try:
            with os.fdopen(fd, 'wb') as f:
                np.lib.format.write_array(f, np.asanyarray(x), allow_pickle=False)
            os.replace(tmp, name)
        except BaseException:
            if os.path.exists(tmp):
                os.unlink(tmp)
            raise

    __getitem__ = get
    __setitem__ = set

    def items(self):
        return ((k, self.get(k)) for k in self.keys())

    def __iter__(self):
        return (self.get(k) for k in self.keys())

    def __len__(self):
        return len(list(self.keys()))

    def close(self):
        pass

    flush = close

    def __enter__(self):
        return self

    def __exit__(self, *exc_info):
        self.close()

    # ------------------------------------------------------------------ provenance chain
    @property
    def provenance(self):
        try:
            with open(os.path.join(self.path, _NOTES)) as f:
                return f.read()
        except IOError:
            return 'No available provenance'

    def _note(self, parent=None, comments=''):
        from . import __version__
        lines = ['MSMBuilder Dataset:', '  msmbuilder_amd:\t%s' % __version__, '  Path:\t\t%s' % self.path,
                 '  Comments:\t\t%s' % comments]
        if parent:
            lines += ['', '== Derived from ==', parent]
        with open(os.path.join(self.path, _NOTES), 'w') as f:
            f.write('\n'.join(lines) + '\n')

    def create_derived(self, out_path, comments='', fmt=None):
        child = dataset(out_path, mode='w', verbose=self.verbose, fmt=fmt)
        child._note(parent=self.provenance, comments=comments)
        return child

    # ------------------------------------------------------------------ estimator hooks
    def fit_with(self, estimator):
        estimator.fit(self)
        return estimator

    def transform_with(self, estimator, out_ds, fmt=None):
        sink = self.create_derived(out_ds, fmt=fmt) if isinstance(out_ds, str) else out_ds
        if getattr(sink, 'mode', 'w') == 'r':
            raise ValueError('out_ds must be opened for writing')
        for k in self.keys():
            sink[k] = estimator.partial_transform(self.get(k))
        return sink

    def fit_transform_with(self, estimator, out_ds, fmt=None):
        return self.transform_with(self.fit_with(estimator), out_ds, fmt=fmt)

    # ------------------------------------------------------------------ device streaming
    def device_sequences(self, prefetch=4, buffer_bytes=8 << 20, readers=4, device=None):
        """Re-iterable view yielding each trajectory as a ``torch`` CUDA tensor loaded by the
        native pipelined reader (float32 / float64 / int32 / int64 C-ordered files): ``readers``
        threads with one pinned ``buffer_bytes`` buffer each, ``prefetch`` files in flight."""
        return DeviceSequences(self, prefetch, buffer_bytes, device, readers)


class DeviceSequences(object):
    """The trajectories of a dir-npy store as device tensors, loaded on demand (see module docstring)."""

    def __init__(self, ds, prefetch, buffer_bytes, device, readers=4):
        self.ds = ds
        self.readers = max(1, int(readers))
        self.prefetch = max(1, int(prefetch))
        self.buffer_bytes = int(buffer_bytes)
        self.device = device
        self._keys = list(ds.keys())

    def __len__(self):
        return len(self._keys)

    def _describe(self, key):
        p = _Payload(self.ds._file(key))
        if p.fortran and len(p.shape) > 1:
            raise ValueError("%s is Fortran-ordered; the device loader needs C-ordered arrays" % p.path)
        if p.code not in _TORCH_NAMES:
            raise TypeError("%s: dtype %s has no device loader" % (p.path, p.code))
        return p

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
                p = self._describe(key)
                t = torch.empty(p.shape, dtype=getattr(torch, _TORCH_NAMES[p.code]), device=dev)
                # `t` may re-use a block that kernels queued on torch's stream are still reading (a trajectory the consumer
                # has already dropped): submit() records a fence on the library stream, which must be that stream
                _lib.set_stream(torch.cuda.current_stream(dev).cuda_stream)
                job = C.c_int64(0)
                check(L.msm_npy_loader_submit(h, os.fsencode(p.path), C.c_void_p(t.data_ptr()), p.nbytes, C.byref(job)))
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
