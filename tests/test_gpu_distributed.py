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
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
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
