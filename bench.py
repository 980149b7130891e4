#!/usr/bin/env python
"""bench.py -- frames/sec of the MSMBuilder hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the rank's HBM-resident synthetic shard
(BASELINE.json configs[3]: 10,000,000 x 512 fp32 as 1,000 trajectories x 10,000 frames,
tICA lag 100):

    tICA.fit        MFMA covariance accumulation of every frame  (dominant kernel)
    allreduce       one RCCL all-reduce of the packed fp64 accumulators (N > 1)
    solve           finalise + generalized eigensolve (host scipy, F x F)
    transform       fused projection to n_components = 10 (fp64 out, stays in HBM)
    KCenters.fit    K = 200 fused k-centers passes on the projected frames (exact arithmetic)
    KCenters.predict  assign_nearest of every frame (exact arithmetic, bit-identical labels)

Scaling is weak: every rank holds its own 10M x 512 shard (20.5 GB of the 288 GB HBM), so
`value` = N x frames-per-rank / max-over-ranks step time.  Data is generated on the
device before timing (AR(1) slow modes mixed into 512 features).
Prints ONE JSON line on rank 0.  Besides the contract's keys it carries `roofline` (dominant kernel: algorithmic and
executed TFLOP/s against the fp32 MFMA peak, L2-fabric traffic), `clustering` (the HBM-bound half on its own),
`cpu_baseline` (the oracle on the host cores, bounded sample; N = 1 only) and `minibatchkmeans` (configs[3]'s
clusterer, MiniBatchKMeans(k=1000).fit on the same projection, run once OUTSIDE the timed steps; N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import warnings

# torchrun exports OMP_NUM_THREADS=1 to every rank unless the user set it ("please further tune the variable"): the
# host eigensolve of a step would then run on one BLAS thread at N > 1 but on four at N = 1.  OpenBLAS sizes its
# buffers from this variable when numpy / scipy are first imported, so it has to be raised HERE, before that import.
if "LOCAL_RANK" in os.environ and os.environ.get("OMP_NUM_THREADS") == "1":
    os.environ["OMP_NUM_THREADS"] = "8"

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def synth(torch, n_seq, n_frames, F, seed, device, n_slow=16):
    """AR(1) slow modes -> F features, generated on the device.  Returns [n_seq*n_frames, F] f32."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gm = torch.Generator(device="cpu")
    gm.manual_seed(4321)
    M = (torch.randn(n_slow, F, generator=gm) / np.sqrt(n_slow)).to(device)
    b = (torch.rand(F, generator=gm) * 2 - 1).to(device)
    ts = torch.logspace(np.log10(20.0), np.log10(5000.0), n_slow)
    a = torch.exp(-1.0 / ts).to(device)
    sig = torch.sqrt(1 - a * a)
    X = torch.empty((n_seq * n_frames, F), dtype=torch.float32, device=device)
    Xv = X.view(n_seq, n_frames, F)
    z = torch.randn(n_seq, n_slow, generator=g, device=device)
    Z = torch.empty(n_seq, n_frames, n_slow, device=device)
    eps_block = 500
    for t0 in range(0, n_frames, eps_block):
        e = torch.randn(n_seq, min(eps_block, n_frames - t0), n_slow, generator=g, device=device)
        for t in range(e.shape[1]):
            z = a * z + sig * e[:, t]
            Z[:, t0 + t] = z
    sc = max(1, (1 << 28) // (n_frames * F))  # ~1 GiB of fp32 per chunk
    for s0 in range(0, n_seq, sc):
        s1 = min(n_seq, s0 + sc)
        blk = Z[s0:s1] @ M
        blk += 0.5 * torch.randn(blk.shape, generator=g, device=device)
        blk += b
        Xv[s0:s1] = blk
    del Z
    return X


def cpu_baseline(X_host_list, lag, k_comp, k_clusters, budget_s=12.0):
    """The CPU checker timed on the host cores on a bounded sample of the same workload:
    oracle tICA (the reference's op sequence: f64 up-cast + 3 dgemm, tica.py:402-422),
    eigensolve, projection, then the C restatement of KCenters.fit + assign_nearest."""
    from oracle.tica_oracle import TicaOracle
    from oracle.libdistance_oracle import Oracle
    o = TicaOracle(n_components=k_comp, lag_time=lag)
    t0 = time.perf_counter()
    used = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for X in X_host_list:
            o.partial_fit(X)
            used.append(X)
            if time.perf_counter() - t0 > budget_s:
                break
    t_fit = time.perf_counter() - t0
    t1 = time.perf_counter()
    Y = np.concatenate(o.transform(used))
    t_proj = time.perf_counter() - t1
    lo = Oracle()
    t2 = time.perf_counter()
    ids, labels, dist = lo.kcenters_fit(Y, k_clusters, "euclidean", 0)
    lab, _ = lo.assign_nearest(Y, np.ascontiguousarray(Y[ids]), "euclidean")
    t_clu = time.perf_counter() - t2
    n = sum(len(x) for x in used)
    total = t_fit + t_proj + t_clu
    return dict(value=n / total, unit="frames/s", cores=os.cpu_count(), kind="port",
                sample="%d trajectories x %d frames x %d f32 of the same synthetic data (oracle tICA via numpy BLAS "
                       "on all host threads %.2fs + projection %.2fs + single-thread C KCenters K=%d fit+assign %.2fs)"
                       % (len(used), len(used[0]), used[0].shape[1], t_fit, t_proj, k_clusters, t_clu),
                tica_fit_frames_per_s=n / t_fit)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=10_000_000, help="frames per GPU")
    ap.add_argument("--features", type=int, default=512)
    ap.add_argument("--traj-len", type=int, default=10_000)
    ap.add_argument("--lag", type=int, default=100)
    ap.add_argument("--components", type=int, default=10)
    ap.add_argument("--clusters", type=int, default=200)
    ap.add_argument("--mode", default=os.environ.get("MSMBUILDER_AMD_TICA_MODE", "f32"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mbk", action="store_true", help="skip the untimed MiniBatchKMeans(k=1000) leg")
    args = ap.parse_args()
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = args.mode

    import torch
    import torch.distributed as dist
    from msmbuilder_amd import tICA, KCenters, _lib, parallel

    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (got WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    local = local % max(1, torch.cuda.device_count())  # identity on a real N-GPU node; lets 2 ranks share 1 GPU in smoke tests
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.ensure_device(local)
    _lib.set_stream(torch.cuda.current_stream().cuda_stream)

    F, T = args.features, args.traj_len
    n_seq = max(1, args.frames // T)
    frames = n_seq * T
    X = synth(torch, n_seq, T, F, 1234 + rank, dev)
    seqs = list(X.view(n_seq, T, F).unbind(0))
    torch.cuda.synchronize()

    times = {}

    def step(record):
        t = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tica = tICA(n_components=args.components, lag_time=args.lag)
            tica.fit(seqs)
            if record is not None:
                ms = C.c_float(0.0)
                _lib.check(_lib.lib().msm_tica_last_kernel_ms(tica._handle, C.byref(ms)))
                record.setdefault("mfma_ms", []).append(ms.value)
                record["sym"] = tica._lagged_symmetrised
                torch.cuda.synchronize()
                record.setdefault("fit", []).append(time.perf_counter() - t)
            if world > 1:
                tica.allreduce()
            ev = tica.eigenvalues_          # finalise + eigensolve (host)
            t1 = time.perf_counter()
            Y = tica.transform([X])[0]      # [frames, k] float64, device resident
            if record is not None:
                torch.cuda.synchronize()
                record.setdefault("transform", []).append(time.perf_counter() - t1)
            t2 = time.perf_counter()
            kc = KCenters(n_clusters=args.clusters, random_state=0).fit([Y])
            if record is not None:
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                record.setdefault("kcenters_fit", []).append(t3 - t2)
            labels = kc.predict([Y])[0]
            if record is not None:
                torch.cuda.synchronize()
                record.setdefault("kcenters_predict", []).append(time.perf_counter() - t3)
                record.setdefault("cluster", []).append(time.perf_counter() - t2)
        return ev, labels, kc, Y

    for _ in range(args.warmup):
        step(None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev, labels, kc, Y = step(times)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = float(parallel.allreduce_array(np.array([elapsed]), op="max")[0])

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * frames / (elapsed / args.steps)
        mfma_ms = float(np.mean(times["mfma_ms"]))
        sym = bool(times.pop("sym", False))
        flops = 4.0 * F * F * frames                     # algorithmic: 2 dense F x F rank-1 updates per frame
        # MFMA flops the kernel really issues per frame, in 128 x 128 tiles: G upper tiles + all C tiles, or -- fp32
        # sum/difference kernel -- the H and the D block of the upper tiles only
        nt = (F + 127) // 128
        tiles = nt * (nt + 1) if sym else nt * nt + nt * (nt + 1) // 2
        executed = 2.0 * 128 * 128 * tiles * frames
        kernel = "tica_sym_f32_kernel" if sym else "tica_mfma_%s_kernel" % args.mode
        achieved = flops / (mfma_ms * 1e-3) / 1e12
        peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f64": PEAK_F64_MFMA_TFLOPS}.get(args.mode, 2500.0)  # bf16 dense MFMA
        traffic = None   # PMC counters need their own rocprofv3 pass: read the committed measurement
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
            if tj["workload"].startswith("%dx%d " % (frames, F)):
                traffic = tj["bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "frames/sec tICA fit + KCenters assign, 10M x 512 feats",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.mode, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3] shape per GPU: %d x %d fp32 as %d trajectories x %d, "
                                   "tICA(n_components=%d, lag_time=%d) fit+solve+transform -> KCenters(k=%d) fit+predict"
                                   % (frames, F, n_seq, T, args.components, args.lag, args.clusters),
                       "frames_per_gpu": frames, "n_features": F, "lag_time": args.lag,
                       "n_components": args.components, "n_clusters": args.clusters,
                       "parallelism": "frames sharded x%d, 1 all-reduce" % world},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "executed": executed / (mfma_ms * 1e-3) / 1e12, "executed_frac": executed / (mfma_ms * 1e-3) / 1e12 / peak,
                         "frac_note": "achieved = SURVEY 8d's algorithmic 4 F^2 flop/frame / kernel time; executed = the MFMA flops "
                                      "actually issued (%d tile products of 128x128 per frame). The symmetric kernel needs fewer "
                                      "flops than 4 F^2, so achieved/peak can exceed 1; executed_frac is the pipe utilisation" % tiles,
                         "traffic_note": "bytes/launch at the L2 fabric side (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC pass, "
                                         "profiles/traffic.json); includes Infinity-Cache hits; algorithmic bytes/launch = %d" % (frames * F * 4),
                         "kernel_ms": mfma_ms, "algorithmic_flop_per_frame": 4 * F * F,
                         "tica_accumulate_frames_per_s": frames / (mfma_ms * 1e-3)},
            "phases_ms": {k: 1e3 * float(np.mean(v)) for k, v in times.items() if k not in ("mfma_ms", "sym")},
            "top_eigenvalues": [float(x) for x in ev[:3]],
        }
        # the clustering half of the metric on its own: HBM-bound exact-arithmetic scans of the [frames, k] float64 projection
        fit_s, pred_s = float(np.mean(times["kcenters_fit"])), float(np.mean(times["kcenters_predict"]))
        pass_bytes = frames * (args.components * 8 + 16)          # read X row + distances_, update distances_/labels_
        out["clustering"] = {"kcenters_fit_frames_per_s": world * frames / fit_s, "assign_frames_per_s": world * frames / pred_s,
                             "kcenters_pass_TBps": args.clusters * pass_bytes / fit_s / 1e12, "hbm_peak_TBps": 8.0}
        if world == 1 and not args.no_mbk:
            # BASELINE configs[3] names MiniBatchKMeans(k=1000) as the clusterer of this shape: the same projection through
            # it, once, OUTSIDE the timed steps (`value` stays the metric's tICA + KCenters pipeline)
            from msmbuilder_amd import MiniBatchKMeans
            Y32 = Y.float().contiguous()
            torch.cuda.synchronize()
            tm = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mb = MiniBatchKMeans(n_clusters=1000, random_state=0).fit([Y32])
            torch.cuda.synchronize()
            tm = time.perf_counter() - tm
            out["minibatchkmeans"] = {"n_clusters": 1000, "fit_ms": 1e3 * tm, "n_steps": int(mb.n_steps_),
                                      "fit_frames_per_s": frames / tm, "inertia_per_frame": float(mb.inertia_) / frames,
                                      "note": "MiniBatchKMeans(n_clusters=1000).fit on the [frames, %d] projection (fp32), "
                                              "k-means++ seeding + mini-batch steps + labels_ of every frame" % args.components}
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            sample = [s.cpu().numpy() for s in seqs[:64]]
            out["cpu_baseline"] = cpu_baseline(sample, args.lag, args.components, args.clusters)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
