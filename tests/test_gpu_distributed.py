"""Two ranks sharing the one GPU of the box (gloo collectives through host tensors; on a real
multi-GPU node the same code runs RCCL): sharded tICA / KCenters / MiniBatchKMeans must equal the
single-process result."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, warnings
import numpy as np
sys.path.insert(0, {root!r})
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
import torch
import torch.distributed as dist
from msmbuilder_amd import tICA, KCenters, MiniBatchKMeans, parallel, _lib
rank, world, local = parallel.init_from_env(backend="gloo")
_lib.ensure_device(0)
rs = np.random.RandomState(11)
seqs = [(rs.randn(int(n), 12) + 0.5).astype(np.float32) for n in rs.randint(30, 900, size=17)]
warnings.simplefilter("ignore")

# ---- tICA: whole trajectories per rank + one all-reduce
mine = parallel.shard_sequences(seqs)
m = tICA(n_components=3, lag_time=5).fit([seqs[i] for i in mine]).allreduce()
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
ref = tICA(n_components=3, lag_time=5).fit(seqs)
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert m.n_observations_ == ref.n_observations_ and m.n_sequences_ == ref.n_sequences_
np.testing.assert_allclose(m.eigenvalues_, ref.eigenvalues_, rtol=1e-11)
np.testing.assert_allclose(m.means_, ref.means_, rtol=1e-12)

# partial_fit -> read eigenvalues_ (the device solve caches the LOCAL mean) -> allreduce -> means_ / covariance_ must be
# those of the reduced model (ADVICE r3: the cached mean used to survive the all-reduce)
m2 = tICA(n_components=3, lag_time=5)
for i in mine:
    m2.partial_fit(seqs[i])
_ = m2.eigenvalues_
m2.allreduce()
np.testing.assert_allclose(m2.means_, ref.means_, rtol=1e-12)
np.testing.assert_allclose(m2.covariance_, ref.covariance_, rtol=1e-10, atol=1e-13)
np.testing.assert_allclose(m2.offset_correlation_, ref.offset_correlation_, rtol=1e-10, atol=1e-13)
np.testing.assert_allclose(m2.eigenvalues_, ref.eigenvalues_, rtol=1e-11)

# ---- tICA: SPMD fit of ONE long trajectory cut between the ranks (lag-row right halo)
long = [(rs.randn(5000, 12) * 0.7 + 0.3).astype(np.float32), seqs[0], seqs[1][:4]]
ms = tICA(n_components=3, lag_time=7).fit_sharded(long)
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
rl = tICA(n_components=3, lag_time=7).fit(long)
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert ms.n_observations_ == rl.n_observations_ and ms.n_sequences_ == rl.n_sequences_ == 2
np.testing.assert_allclose(ms.eigenvalues_, rl.eigenvalues_, rtol=1e-11)
np.testing.assert_allclose(ms.offset_correlation_, rl.offset_correlation_, rtol=1e-10, atol=1e-13)
np.testing.assert_allclose(ms.covariance_, rl.covariance_, rtol=1e-10, atol=1e-13)

assert parallel._lib_comm_kind == "host"      # the library ran the exchange itself (host transport under gloo)

# ---- tICA, default f32 mode on UN-CENTRED features (|mean|/std = 50): every rank has its own shift row r; the raw
# moments are restored before the all-reduce, so the sum is the unsplit fit's
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f32"
unc = [(rs.randn(int(n), 12) + 50.0 * np.sign(rs.randn(12))).astype(np.float32) for n in (700, 900, 650, 1200)]
mu_ = tICA(n_components=3, lag_time=5).fit([unc[i] for i in parallel.shard_sequences(unc)]).allreduce()
os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
ru = tICA(n_components=3, lag_time=5).fit(unc)
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
np.testing.assert_allclose(mu_.eigenvalues_, ru.eigenvalues_, rtol=1e-5)
np.testing.assert_allclose(mu_.covariance_, ru.covariance_, rtol=0, atol=1e-5 * np.abs(ru.covariance_).max())

# ---- KCenters: consecutive row blocks per rank
X = np.concatenate(seqs).astype(np.float64)
X[40:44] = X[3]                                   # duplicates: argmax ties across the shard boundary
cut = 2000
block = X[:cut] if rank == 0 else X[cut:]
kc = KCenters(n_clusters=25, random_state=4).fit([block])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
kref = KCenters(n_clusters=25, random_state=4).fit([X])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert kc.cluster_ids_ == kref.cluster_ids_, (kc.cluster_ids_, kref.cluster_ids_)
lo, hi = (0, cut) if rank == 0 else (cut, len(X))
assert np.array_equal(kc.labels_[0].cpu().numpy(), kref.labels_[0][lo:hi])
assert np.array_equal(kc.distances_[0].cpu().numpy(), kref.distances_[0][lo:hi])
assert abs(kc.inertia_ - kref.inertia_) <= 1e-12 * kref.inertia_
assert np.array_equal(kc.cluster_centers_, kref.cluster_centers_)
assert np.array_equal(kc.predict([block])[0], kref.predict([X])[0][lo:hi])

# SCREENED passes in the sharded loop (float64 rows): centres far from the origin, duplicates across the boundary, a small and
# an EMPTY shard.  Default: several centres per exchange (threshold lists, the same selection on every rank: fewer passes
# than centres, the same number on both ranks); MSM_KC_BATCH=0: one centre per exchange, where screening is a rank's own
# decision (>= 65536 rows) under an identical exchange pattern.
import ctypes as C
zs = np.cumsum(rs.randn(200_000, 10) * 0.05, axis=0) + rs.randn(200_000, 10) * 0.3 + 7.0
zs[150_000:150_004] = zs[17]
for cutz, batch in ((120_000, "1"), (160_000, "1"), (200_000, "1"), (120_000, "0"), (160_000, "0")):
    os.environ["MSM_KC_BATCH"] = batch
    blk = zs[:cutz] if rank == 0 else zs[cutz:]
    kz = KCenters(n_clusters=40, random_state=2).fit([blk])
    st = (C.c_int64 * 5)()
    _lib.check(_lib.lib().msm_kcenters_last_stats(st))
    assert st[0] == len(blk), list(st)
    if batch == "1":
        assert st[1] == 2 and 1 <= st[2] < 38, list(st)
        both = torch.tensor([int(st[2])]); dist.all_reduce(both, op=dist.ReduceOp.MAX)
        assert int(both[0]) == st[2]                                   # every rank ran the same rounds
    else:
        assert st[1] + st[2] == 40 and (st[2] == 36) == (len(blk) >= 65536), list(st)
    os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
    kzr = KCenters(n_clusters=40, random_state=2).fit([zs])
    os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
    assert kz.cluster_ids_ == kzr.cluster_ids_, (kz.cluster_ids_, kzr.cluster_ids_)
    lz, hz = (0, cutz) if rank == 0 else (cutz, len(zs))
    assert np.array_equal(kz.labels_[0].cpu().numpy(), kzr.labels_[0][lz:hz])
    assert np.array_equal(kz.distances_[0].cpu().numpy(), kzr.distances_[0][lz:hz])
    assert np.array_equal(kz.cluster_centers_, kzr.cluster_centers_)
    assert abs(kz.inertia_ - kzr.inertia_) <= 1e-12 * kzr.inertia_
os.environ.pop("MSM_KC_BATCH")

# an EMPTY shard on one rank, float32 rows longer than one chunk (wide-row kernel)
Xw = rs.randn(3000, 40).astype(np.float32)
blockw = Xw if rank == 0 else Xw[:0]
kcw = KCenters(n_clusters=9, random_state=1).fit([blockw])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
kwref = KCenters(n_clusters=9, random_state=1).fit([Xw])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert kcw.cluster_ids_ == kwref.cluster_ids_ and np.array_equal(kcw.cluster_centers_, kwref.cluster_centers_)
if rank == 0:
    assert np.array_equal(kcw.labels_[0].cpu().numpy(), kwref.labels_[0])
    assert np.array_equal(kcw.distances_[0].cpu().numpy(), kwref.distances_[0])
assert abs(kcw.inertia_ - kwref.inertia_) <= 1e-12 * kwref.inertia_

# ---- MiniBatchKMeans: global batches, per-rank partial sums, one all-reduce per step
Xf = X.astype(np.float32)
init = Xf[rs.choice(len(Xf), 6, replace=False)].copy()
kw = dict(n_clusters=6, init=init, n_init=1, batch_size=200, max_iter=3, random_state=9, reassignment_ratio=0.0)
blockf = Xf[:cut] if rank == 0 else Xf[cut:]
mb = MiniBatchKMeans(**kw).fit([blockf])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
mref = MiniBatchKMeans(**kw).fit([Xf])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert mb.n_steps_ == mref.n_steps_
np.testing.assert_allclose(mb.cluster_centers_, mref.cluster_centers_, rtol=1e-5, atol=1e-6)
np.testing.assert_allclose(mb.inertia_, mref.inertia_, rtol=1e-5)
assert (mb.labels_[0] != mref.labels_[0][lo:hi]).mean() < 5e-3
# device-resident shards: the queued runs (msm_mbk_run_sharded: one library all-reduce per step, no Python inside a run)
# against the step-by-step sharded path and the single-process fit; enough steps for several runs and an early stop
kw2 = dict(n_clusters=6, init=init, n_init=1, batch_size=200, max_iter=40, random_state=9, reassignment_ratio=0.01,
           max_no_improvement=5)
dblock = torch.from_numpy(blockf).cuda()
mb_run = MiniBatchKMeans(**kw2).fit([dblock])
os.environ["MSMBUILDER_AMD_MBK_RUNS"] = "0"
mb_step = MiniBatchKMeans(**kw2).fit([dblock])
os.environ["MSMBUILDER_AMD_MBK_RUNS"] = "1"
os.environ["MSMBUILDER_AMD_PARALLEL"] = "0"
mref2 = MiniBatchKMeans(**kw2).fit([torch.from_numpy(Xf).cuda()])
os.environ["MSMBUILDER_AMD_PARALLEL"] = "1"
assert mb_run.n_steps_ == mb_step.n_steps_ == mref2.n_steps_, (mb_run.n_steps_, mb_step.n_steps_, mref2.n_steps_)
assert mb_run.n_steps_ < (40 * len(Xf)) // 200                      # the criterion fired inside a queued run
np.testing.assert_array_equal(mb_run.cluster_centers_, mb_step.cluster_centers_)     # same kernels, same reductions
np.testing.assert_allclose(mb_run.cluster_centers_, mref2.cluster_centers_, rtol=1e-4, atol=1e-5)
np.testing.assert_allclose(mb_run.inertia_, mref2.inertia_, rtol=1e-4)
# random_state=None: rank 0's draw seeds every rank (ADVICE r1) -- the ranks must agree on every centre
mn = MiniBatchKMeans(n_clusters=5, batch_size=150, max_iter=2, n_init=1, tol=1e-4).fit([blockf])
spread = parallel.allreduce_array(mn.cluster_centers_.astype(np.float64).ravel(), op="max") + \
    parallel.allreduce_array(-mn.cluster_centers_.astype(np.float64).ravel(), op="max")
assert np.all(spread == 0.0), spread.max()
dist.barrier()
parallel.library_comm_shutdown()
dist.destroy_process_group()
print("rank", rank, "ok")
'''

_RCCL1 = r'''
import os, sys, warnings, ctypes as C
import numpy as np
sys.path.insert(0, {root!r})
import torch
from msmbuilder_amd import tICA, KCenters, _lib
from msmbuilder_amd._lib import Arr
warnings.simplefilter("ignore")
_lib.ensure_device(0)
L = _lib.lib()
uid = (C.c_char * 128)()
_lib.check(L.msm_comm_unique_id(uid))
_lib.check(L.msm_comm_init_rccl(bytes(uid), 0, 1))        # a world of one: RCCL itself runs every collective
r, w, k = C.c_int(), C.c_int(), C.c_int()
L.msm_comm_info(C.byref(r), C.byref(w), C.byref(k))
assert (r.value, w.value, k.value) == (0, 1, 1)
t = torch.arange(1000, dtype=torch.float64, device="cuda")
_lib.check(L.msm_comm_allreduce_f64(C.c_void_p(t.data_ptr()), 1000))
assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
o = torch.zeros(1000, dtype=torch.float64, device="cuda")
_lib.check(L.msm_comm_allgather(C.c_void_p(t.data_ptr()), C.c_void_p(o.data_ptr()), 8000))
assert torch.equal(o, t)
# tICA all-reduce through RCCL (device to device)
rs = np.random.RandomState(2)
seqs = [(rs.randn(800, 40) + 20.0).astype(np.float32) for _ in range(3)]
m = tICA(n_components=3, lag_time=4).fit(seqs)
e0 = m.eigenvalues_.copy()
_lib.check(L.msm_tica_allreduce(m._handle))
m._is_dirty = True; m._host_stale = True
np.testing.assert_allclose(m.eigenvalues_, e0, rtol=1e-12)
# sharded k-centers loop with RCCL all-gathers == the single-process fit
X = torch.from_numpy(rs.randn(5000, 10)).cuda()
ref = KCenters(n_clusters=30, random_state=3).fit([X])
ax = Arr(X)
lab = torch.empty(5000, dtype=torch.int64, device="cuda"); dist_ = torch.empty(5000, dtype=torch.float64, device="cuda")
ids = np.zeros(30, dtype=np.int64); cen = np.zeros((30, 10)); inertia = C.c_double()
_lib.check(L.msm_kcenters_fit_sharded_f64(ax.vp, 5000, 10, 30, b"euclidean", ref.cluster_ids_[0], 0, C.c_void_p(lab.data_ptr()),
                                          C.c_void_p(dist_.data_ptr()), ids.ctypes.data, cen.ctypes.data, C.byref(inertia)))
assert list(ids) == ref.cluster_ids_
assert torch.equal(lab, ref.labels_[0]) and torch.equal(dist_, ref.distances_[0])
assert np.array_equal(cen, ref.cluster_centers_.cpu().numpy() if hasattr(ref.cluster_centers_, "cpu") else ref.cluster_centers_)
assert abs(inertia.value - ref.inertia_) <= 1e-12 * ref.inertia_
# ... and with enough rows for the screened passes
X2 = torch.from_numpy(np.cumsum(rs.randn(100_000, 10) * 0.05, axis=0) + 3.0).cuda()
ref2 = KCenters(n_clusters=30, random_state=3).fit([X2])
ax2 = Arr(X2)
lab2 = torch.empty(100_000, dtype=torch.int64, device="cuda"); dist2 = torch.empty(100_000, dtype=torch.float64, device="cuda")
_lib.check(L.msm_kcenters_fit_sharded_f64(ax2.vp, 100_000, 10, 30, b"euclidean", ref2.cluster_ids_[0], 0, C.c_void_p(lab2.data_ptr()),
                                          C.c_void_p(dist2.data_ptr()), ids.ctypes.data, cen.ctypes.data, C.byref(inertia)))
st = (C.c_int64 * 5)()
_lib.check(L.msm_kcenters_last_stats(st))
assert st[0] == 100_000 and st[1] == 2 and 1 <= st[2] <= 28, list(st)   # several centres per exchange: at most one round per centre
assert list(ids) == ref2.cluster_ids_
assert torch.equal(lab2, ref2.labels_[0]) and torch.equal(dist2, ref2.distances_[0])
L.msm_comm_destroy()
print("rccl world-of-one ok")
'''


def test_two_ranks_match_single_process(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
        assert "ok" in o


def test_library_rccl_world_of_one(gpu, tmp_path):
    """The RCCL transport of csrc/comm.hip on the one GPU a test box has: a communicator of a single rank still
    goes through ncclCommInitRank / ncclAllReduce / ncclAllGather on the library stream (dlopen, prototypes, stream
    ordering), under the tICA all-reduce and the sharded k-centers loop."""
    script = tmp_path / "rccl1.py"
    script.write_text(_RCCL1.format(root=ROOT))
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "rccl world-of-one ok" in p.stdout, p.stdout[-3000:]


def test_bench_self_launches_its_ranks(gpu):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command form) starts its own two
    ranks under torch.distributed.run and prints ONE JSON line; on a one-GPU box the ranks share the device and the
    library's collectives run on the host transport (`comm` = "host", `rccl_ranks` = 0), on a multi-GPU node over RCCL."""
    import json
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames", "400000", "--no-extras", "--no-mbk", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["total_frames"] == 400000
    assert out["config"]["frames_per_gpu"] == 200000
    if torch.cuda.device_count() >= 2:
        assert out["comm"] == "rccl" and out["rccl_ranks"] == 2
    else:
        assert out["comm"] == "host" and out["rccl_ranks"] == 0
    assert out["value"] > 0 and set(out["phases_ms"]) >= {"fit", "allreduce", "solve", "transform", "kcenters_fit", "kcenters_predict"}
    # (several centres per exchange: at most one round per centre)
    assert out["clustering"]["kcenters_plain_passes"] == 2 and 1 <= out["clustering"]["kcenters_screened_passes"] <= 198


def test_bench_world_of_eight_on_the_box(gpu):
    """The driver's 8-GPU command form, `python bench.py --gpus 8`, with EIGHT ranks before the driver's node is the first
    to try: on a one-GPU box the ranks share the device over the host transport, so the sharded code -- the dealing of the
    trajectories, the packed all-reduce, the k-centers round records of eight shards, the barrier / max-over-ranks timing,
    the single JSON line -- meets world = 8 here (VERDICT r3 #2d).  On an 8-GPU node the same command runs over RCCL."""
    import json
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                        "--frames", "800000", "--no-extras", "--no-mbk", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["total_frames"] == 800000
    assert out["config"]["frames_per_gpu"] == 100000 and out["comm_ok"] and out["comm_failed_ranks"] == []
    if torch.cuda.device_count() >= 8:
        assert out["comm"] == "rccl" and out["rccl_ranks"] == 8
    else:
        assert out["comm"] == "host" and out["rccl_ranks"] == 0
    assert out["value"] > 0 and set(out["phases_ms"]) >= {"fit", "allreduce", "solve", "transform", "kcenters_fit", "kcenters_predict"}
    cm = out["comm_measured_us"]
    assert cm["transport"] == out["comm"] and all(v is not None and v > 0 for k_, v in cm.items() if k_.endswith("_us"))
    ev = out["top_eigenvalues"] if "top_eigenvalues" in out else None
    assert ev is None or (0.0 < ev[0] < 1.0)
