#!/usr/bin/env python
"""bench.py -- frames/sec of the MSMBuilder hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (any N: with N > 1 and no WORLD_SIZE in the environment the
                                                            script re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the HBM-resident synthetic data set of BASELINE.json configs[3]
(10,000,000 x 512 fp32 as 1,000 trajectories x 10,000 frames, tICA lag 100):

    fit               tICA.fit: column sums + MFMA covariance accumulation of every frame   (dominant kernel)
    allreduce         one RCCL all-reduce of the packed fp64 accumulators                    (N > 1)
    solve             finalise + generalized eigensolve, top n_components = 10
    transform         fused projection (fp64 out, stays in HBM)
    kcenters_fit      K = 200 fused k-centers passes on the projected frames (exact arithmetic)
    kcenters_predict  assign_nearest of every frame (exact arithmetic, bit-identical labels)

Scaling.  N = 1: the whole problem on one GPU.  N > 1 defaults to STRONG scaling -- the same 10M x 512 problem, its
trajectories dealt out over the ranks (`frames // N` per GPU), which is what BASELINE's metric and configs[3] describe;
`--scaling weak` gives every rank its own 10M x 512 shard instead.  `value` = total frames / max-over-ranks step time.
Data is generated on the device before timing (AR(1) slow modes mixed into 512 features).

Prints ONE JSON line on rank 0.  Besides the contract's keys:
  roofline      dominant kernel (the MFMA accumulation): `achieved` = the MFMA flop it EXECUTES / its HIP-event duration,
                `frac` = achieved / dense fp32-MFMA peak (pipe utilisation); `algorithmic` = SURVEY 8(d)'s 4 F^2 flop per
                frame / the same duration (can exceed the peak: the sum/difference kernel needs 0.625 of those flops)
  phases_ms     per-phase wall times (host clock around synchronised phases), summing to the step
  self_check    the HIP path and the float64 oracle fitted on the same 64 trajectories (eigenvalues, rtol 1e-5), and the
                full run's leading eigenvalues against that sample (loose)
  cpu_baseline  the oracle on the host cores (bounded sample): all-threads and 1-thread BLAS tICA, single-thread and
                row-parallel exact KCenters, scikit-learn's own MiniBatchKMeans(k=1000)
  f64, config2, config1_width[_f64], config3_width, config3_stress, kcenters_wide_1m, config4_label_wide, config5_width,
  config5_per_gpu, minibatchkmeans, strong_scaling_model   untimed secondary legs (N = 1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import warnings

# torchrun exports OMP_NUM_THREADS=1 to every rank unless the user set it ("please further tune the variable"): the
# host part of the eigensolve would then run on one BLAS thread at N > 1 but on four at N = 1.  OpenBLAS sizes its
# buffers from this variable when numpy / scipy are first imported, so it has to be raised HERE, before that import.
if "LOCAL_RANK" in os.environ and os.environ.get("OMP_NUM_THREADS") == "1":
    os.environ["OMP_NUM_THREADS"] = "8"

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense peaks
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0, "bf16x2": 2500.0}
PEAK_HBM_GBS = 8000.0


def synth(torch, n_seq, n_frames, F, seed, device, n_slow=16, mean_scale=1.0):
    """AR(1) slow modes -> F features, generated on the device.  Returns [n_seq*n_frames, F] f32."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gm = torch.Generator(device="cpu")
    gm.manual_seed(4321)
    M = (torch.randn(n_slow, F, generator=gm) / np.sqrt(n_slow)).to(device)
    b = ((torch.rand(F, generator=gm) * 2 - 1) * mean_scale).to(device)
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


def synth_bf16(torch, n_seq, n_frames, F, seed, device, block=125):
    """The same recipe stored as bfloat16 (BASELINE configs[4]: half the bytes), generated block-wise so that the fp32
    intermediate never exceeds `block` trajectories.  Returns [n_seq * n_frames, F] bf16."""
    X = torch.empty((n_seq * n_frames, F), dtype=torch.bfloat16, device=device)
    for i, s0 in enumerate(range(0, n_seq, block)):
        s1 = min(n_seq, s0 + block)
        X[s0 * n_frames:s1 * n_frames] = synth(torch, s1 - s0, n_frames, F, seed + 1000 * i, device).to(torch.bfloat16)
    return X


def executed_flop_per_frame(F, sym):
    """MFMA flop the accumulation kernel issues per frame, in 128 x 128 tile products: all T^2 lagged tiles + the
    T(T+1)/2 upper Gram tiles, or -- fp32 sum/difference kernel -- the H and the D block of the upper tiles only."""
    nt = (F + 127) // 128
    tiles = nt * (nt + 1) if sym else nt * nt + nt * (nt + 1) // 2
    return 2.0 * 128 * 128 * tiles, tiles


def executed_flop_per_frame_symw(F):
    """Whole-matrix sum/difference kernel (F <= 256, csrc/tica_symw_dev.h): 16 x 16 MFMA blocks of the upper triangle of H and
    of D, the columns in groups of 16 / 32 / 64 (the variant table of tica.hip)."""
    il, ng = ((1, 1) if F <= 16 else (2, 1) if F <= 32 else (4, 1) if F <= 64 else (2, 3) if F <= 96 else (4, 2) if F <= 128
              else (2, 5) if F <= 160 else (4, 3) if F <= 192 else (4, 4))
    nblk = ng * (il * (il + 1) // 2) + (ng * (ng - 1) // 2) * il * il
    return 2.0 * 16 * 16 * 2 * nblk, nblk


def kernel_ms_of(tica, _lib):
    ms = C.c_float(0.0)
    _lib.check(_lib.lib().msm_tica_last_kernel_ms(tica._handle, C.byref(ms)))
    return float(ms.value)


def executed_flop_per_frame_bf16(F, x2):
    """bf16 image kernel: H and D blocks of the upper 256 x 256 tiles (bf16x2: four bf16 products per block)."""
    nt2 = (F + 255) // 256
    return 2.0 * 256 * 256 * nt2 * (nt2 + 1) * (4 if x2 else 1)


def cpu_baseline(X_host_list, lag, k_comp, k_clusters, budget_s=8.0):
    """The CPU checkers timed on the host cores on a bounded sample of the same workload (SURVEY 8(d) i-iii):
    (i) oracle tICA = the reference's op sequence (f64 up-cast + 3 dgemm, tica.py:402-422) through numpy's BLAS, with the
    thread pool as numpy found it ("all") and limited to one thread (threadpoolctl; both pools are reported as
    threadpoolctl.threadpool_info() saw them INSIDE the timed region); (ii) KCenters.fit + assign_nearest through the
    reference's own libdistance (oracle/_ref, `kinds.clustering` = "reference": the loop of kcenters.py:79-102 around
    libdistance.dist, single-threaded like the reference) when that prebuilt file travelled with the snapshot, else
    through the C restatement ("port"); assign_nearest also row-parallel over the cores; (iii) scikit-learn's
    MiniBatchKMeans(k=1000) itself."""
    from oracle.tica_oracle import TicaOracle
    from oracle.libdistance_oracle import Oracle, Ref
    from concurrent.futures import ThreadPoolExecutor
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
    except Exception:  # reported, not required
        threadpool_limits = threadpool_info = None
    ncpu = os.cpu_count() or 1

    def pools():
        if threadpool_info is None:
            return None
        return [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads", "threading_layer", "version")}
                for p in threadpool_info()]

    def fit_for(budget, limit):
        """partial_fit trajectories until `budget` seconds have passed; (oracle, used trajectories, seconds, pools)."""
        o = TicaOracle(n_components=k_comp, lag_time=lag)
        used, seen = [], None
        ctx = threadpool_limits(limit, "blas") if (limit and threadpool_limits) else None
        if ctx is not None:
            ctx.__enter__()
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                wo = TicaOracle(n_components=k_comp, lag_time=lag)       # BLAS warm-up (thread start-up: ~1 s on first use)
                for X in X_host_list[:3]:
                    wo.partial_fit(X)
                seen = pools()
                t0 = time.perf_counter()
                for X in X_host_list:
                    o.partial_fit(X)
                    used.append(X)
                    if time.perf_counter() - t0 > budget:
                        break
                el = time.perf_counter() - t0
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        return o, used, el, seen

    o, used, t_fit, pools_all = fit_for(budget_s, None)
    _o1, used1, t_fit1, pools_one = fit_for(3.0, 1)
    _o8, used8, t_fit8, _p8 = fit_for(3.0, 8)
    n = sum(len(x) for x in used)
    n1 = sum(len(x) for x in used1)
    n8 = sum(len(x) for x in used8)
    # `value` uses the FASTEST of the three thread settings for the tICA part (many-thread BLAS on 10,000-row dgemms is
    # not always the fastest on a 256-core host; the baseline should not be handicapped by a bad default)
    rate_best = max(n / t_fit, n1 / t_fit1, n8 / t_fit8)
    t_fit_best = n / rate_best
    t1 = time.perf_counter()
    Y = np.concatenate(o.transform(used))
    t_proj = time.perf_counter() - t1

    have_ref = Ref.available()
    lo = Ref() if have_ref else Oracle()
    t2 = time.perf_counter()
    if have_ref:
        # kcenters.py:79-102 around the reference's libdistance.dist
        dist_ = np.full(len(Y), np.inf)
        labels = np.zeros(len(Y), dtype=int)
        ids, c = [], 0
        for i in range(k_clusters):
            d = lo.dist(Y, Y[c], "euclidean")
            mask = d < dist_
            dist_[mask] = d[mask]
            labels[mask] = i
            ids.append(c)
            c = int(np.argmax(dist_))
    else:
        ids, labels, dist_ = lo.kcenters_fit(Y, k_clusters, "euclidean", 0)
    centers = np.ascontiguousarray(Y[ids])
    t2b = time.perf_counter()
    lab = lo.assign_nearest(Y, centers, "euclidean")[0]
    t_clu = time.perf_counter() - t2
    t_assign1 = time.perf_counter() - t2b
    # row-parallel variant of the same scalar routine (ctypes releases the GIL)
    nth = min(ncpu, 64)
    bounds = np.linspace(0, len(Y), nth + 1).astype(np.int64)
    t3 = time.perf_counter()
    with ThreadPoolExecutor(nth) as ex:
        parts = list(ex.map(lambda i: lo.assign_nearest(np.ascontiguousarray(Y[bounds[i]:bounds[i + 1]]), centers, "euclidean")[0],
                            range(nth)))
    t_assign_par = time.perf_counter() - t3
    assert np.array_equal(np.concatenate(parts), lab)
    total = t_fit_best + t_proj + t_clu
    blas_threads = None
    if pools_all:
        blas_threads = max([p["num_threads"] or 1 for p in pools_all if p["user_api"] == "blas"] or [1])
    out = dict(value=n / total, unit="frames/s", cores=ncpu, kind="port",
               kinds={"tica": "port (oracle/tica_oracle.py: the Python reference cannot travel to the GPU box)",
                      "clustering": "reference (oracle/_ref: the reference's libdistance headers compiled)" if have_ref
                      else "port (oracle/libdistance_oracle.c)"},
               blas_threads=blas_threads, threadpool_info=pools_all, threadpool_info_limited=pools_one,
               sample="%d trajectories x %d frames x %d f32 of the same synthetic data: oracle tICA (numpy BLAS, default pool of %s threads: "
                      "%.2fs; `value` uses the fastest of default / 8 / 1 threads) + projection %.2fs + single-thread KCenters K=%d fit+assign %.2fs"
                      % (len(used), len(used[0]), used[0].shape[1], blas_threads, t_fit, t_proj, k_clusters, t_clu),
               tica_fit_frames_per_s=n / t_fit, tica_fit_frames=n,
               tica_fit_1thread_frames_per_s=(n1 / t_fit1) if t_fit1 else None, tica_fit_1thread_frames=n1,
               tica_fit_8thread_frames_per_s=(n8 / t_fit8) if t_fit8 else None,
               tica_fit_best_frames_per_s=rate_best,
               assign_1thread_frames_per_s=len(Y) / t_assign1,
               assign_row_parallel_frames_per_s=len(Y) / t_assign_par, assign_row_parallel_threads=nth)
    # (iii) scikit-learn's MiniBatchKMeans on the projected sample
    try:
        from sklearn.cluster import MiniBatchKMeans as SkMBK
        Y64 = np.ascontiguousarray(Y, dtype=np.float64)   # the reference pipeline's type: tICA.transform emits float64
        t4 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = SkMBK(n_clusters=1000, random_state=0, n_init=1).fit(Y64)
        t_sk = time.perf_counter() - t4
        out["sklearn_minibatchkmeans"] = dict(n_clusters=1000, dtype="f64", frames=len(Y64), fit_s=t_sk, n_steps=int(sk.n_steps_),
                                              fit_frames_per_s=len(Y64) / t_sk)
    except Exception as e:  # baseline only
        out["sklearn_minibatchkmeans"] = dict(error=str(e)[:100])
    return out, o, used


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run on a free
    local port and hand its exit code back.  When the node shows fewer devices than ranks (a 1-GPU test box) the ranks
    share devices and torch.distributed falls back to gloo (RCCL cannot put two ranks on one device); the JSON line says
    so (`rccl_ranks`, `comm`)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        from msmbuilder_amd import _lib
        ndev = _lib.device_count()
    except Exception:
        ndev = 0
    if 0 < ndev < n:
        env.setdefault("MSMBUILDER_AMD_DIST_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # stdout of the job carries ONE JSON line (rank 0's); anything else a backend prints there (gloo's connection banner)
    # is passed on through stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(rc if rc else (0 if line is not None else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=10_000_000, help="frames of the whole problem (strong) / per GPU (weak)")
    ap.add_argument("--features", type=int, default=512)
    ap.add_argument("--traj-len", type=int, default=10_000)
    ap.add_argument("--lag", type=int, default=100)
    ap.add_argument("--components", type=int, default=10)
    ap.add_argument("--clusters", type=int, default=200)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--mode", default=os.environ.get("MSMBUILDER_AMD_TICA_MODE", "f32"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mbk", action="store_true", help="skip the untimed MiniBatchKMeans(k=1000) leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed f64 / config2 / config5 / model legs")
    args = ap.parse_args()
    os.environ["MSMBUILDER_AMD_TICA_MODE"] = args.mode
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

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
    n_seq_total = max(1, args.frames // T)
    if args.scaling == "strong":   # the same problem dealt out over the ranks, whole trajectories
        n_seq = n_seq_total // world + (1 if rank < n_seq_total % world else 0)
        total_frames = n_seq_total * T
    else:
        n_seq = n_seq_total
        total_frames = n_seq_total * T * world
    frames = n_seq * T
    X = synth(torch, n_seq, T, F, 1234 + rank, dev)
    seqs = list(X.view(n_seq, T, F).unbind(0))
    torch.cuda.synchronize()

    def step(record, seqs, X):
        def mark(name, t_prev):
            if record is None:
                return t_prev
            torch.cuda.synchronize()
            now = time.perf_counter()
            record.setdefault(name, []).append(now - t_prev)
            return now
        t = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tica = tICA(n_components=args.components, lag_time=args.lag)
            tica.fit(seqs)
            if record is not None:
                record.setdefault("mfma_ms", []).append(kernel_ms_of(tica, _lib))
                record["sym"] = tica._lagged_symmetrised
                fl = C.c_int(0)
                _lib.check(_lib.lib().msm_tica_last_folded(tica._handle, C.byref(fl)))
                record["folded"] = bool(fl.value)
            t = mark("fit", t)
            if world > 1:
                tica.allreduce()
                t = mark("allreduce", t)
            ev = tica.eigenvalues_          # finalise + eigensolve
            comps = tica.components_
            t = mark("solve", t)
            Y = tica.transform([X])[0]      # [frames, k] float64, device resident
            t = mark("transform", t)
            kc = KCenters(n_clusters=args.clusters, random_state=0).fit([Y])
            t = mark("kcenters_fit", t)
            labels = kc.predict([Y])[0]
            t = mark("kcenters_predict", t)
        return ev, labels, kc, Y, tica

    def timed(record, seqs, X, warmup, steps):
        res = None
        for _ in range(warmup):
            step(None, seqs, X)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step(record, seqs, X)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            el = float(parallel.allreduce_array(np.array([el]), op="max")[0])
        return el, res

    times = {}
    elapsed, (ev, labels, kc, Y, tica) = timed(times, seqs, X, args.warmup, args.steps)
    # which transport the library's own collectives ran on: msm_comm_info -> (rank, world, kind 0 none / 1 RCCL / 2 host)
    ci = [C.c_int(0), C.c_int(1), C.c_int(0)]
    _lib.lib().msm_comm_info(C.byref(ci[0]), C.byref(ci[1]), C.byref(ci[2]))
    comm_kind = {0: "none", 1: "rccl", 2: "host"}[ci[2].value]
    rccl_ranks = ci[1].value if ci[2].value == 1 else (1 if world == 1 else 0)
    # On a node with a device per rank the library's collectives must have run over RCCL with all N ranks; anything else
    # (a fallback to the host transport) is reported in the line and turns the exit code non-zero AFTER the line is printed
    comm_ok = True
    if world > 1:
        comm_ok = ci[1].value == world and (torch.cuda.device_count() < world or (comm_kind == "rccl" and rccl_ranks == world))
    comm_failed_ranks = [f["rank"] for f in parallel._comm_failures]
    if world > 1 and not comm_ok and rank == 0:
        sys.stderr.write("bench.py: the library's collectives did not run over RCCL with all %d ranks (transport %s, %d ranks%s)\n"
                         % (world, comm_kind, ci[1].value, ", RCCL join / self-test failed on ranks %s" % comm_failed_ranks
                            if comm_failed_ranks else ""))
    kst = (C.c_int64 * 5)()
    _lib.check(_lib.lib().msm_kcenters_last_stats(kst))

    # N > 1: the latencies of the library's collectives as its loops see them (queued back to back on the library stream),
    # MEASURED on this run's communicator -- RCCL over xGMI on a node with a GPU per rank -- for the sizes the step uses: the
    # tICA all-reduce (2 F^2 + 2 F doubles), the k-centers candidate all-gather and the k-centers round record
    comm_measured = None
    if world > 1 and ci[2].value != 0:
        comm_measured = {"transport": comm_kind}
        rec_bytes = 8 * (48 + 1024 * (2 + args.components))
        for name, kind, nbytes in (("allreduce_tica_%dKB" % ((2 * F * F + 2 * F) * 8 // 1024), 0, (2 * F * F + 2 * F) * 8),
                                   ("allgather_per_centre_%dB" % (8 * (2 + args.components)), 1, 8 * (2 + args.components)),
                                   ("allgather_per_round_record_%dKB" % (rec_bytes // 1024), 1, rec_bytes)):
            us = C.c_float(0.0)
            rc_m = _lib.lib().msm_comm_measure(kind, nbytes, 50, C.byref(us))   # every rank takes part
            comm_measured[name + "_us"] = float(us.value) if rc_m == 0 else None

    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_extras:
        # the other reading of "N GPUs", untimed secondary leg: every rank its own full-size shard (weak scaling)
        del labels, kc, Y, tica, seqs, X
        torch.cuda.empty_cache()
        Xw = synth(torch, n_seq_total, T, F, 1234 + rank, dev)
        seqsw = list(Xw.view(n_seq_total, T, F).unbind(0))
        rec_w = {}
        el_w, _r = timed(rec_w, seqsw, Xw, 1, 2)
        del _r
        rec_w.pop("sym", None)
        rec_w.pop("folded", None)
        weak = {"frames_per_gpu": n_seq_total * T, "ms_per_step": 1e3 * el_w / 2,
                "value": world * n_seq_total * T / (el_w / 2), "unit": "frames/s",
                "phases_ms": {k: 1e3 * float(np.mean(v)) for k, v in rec_w.items() if k != "mfma_ms"}}
        del Xw, seqsw
        labels = kc = Y = tica = seqs = X = None

    exit_code = 0 if comm_ok else 1
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_frames / (elapsed / args.steps)
        mfma_ms = float(np.mean(times["mfma_ms"]))
        sym = bool(times.pop("sym", False))
        folded = bool(times.pop("folded", False))
        alg_flop = 4.0 * F * F                      # two dense F x F rank-1 updates per frame
        exe_flop, tiles = executed_flop_per_frame(F, sym)
        kernel = "tica_sym_f32_kernel" if sym else "tica_mfma_%s_kernel" % args.mode
        peak = PEAK_TFLOPS.get(args.mode, 157.3)
        executed = exe_flop * frames / (mfma_ms * 1e-3) / 1e12
        algorithmic = alg_flop * frames / (mfma_ms * 1e-3) / 1e12
        # PMC counters need their own rocprofv3 passes (scripts/pmc.sh), so `traffic` cannot be measured inside this run.
        # It is reported only when the committed profile was taken on THIS kernel source (sha256 of csrc/tica.hip
        # recorded next to the counters) and this workload; otherwise null, with the stale profile named beside it.
        traffic, traffic_source = None, None
        try:
            import hashlib
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
            # (the kernel's source: csrc/tica_sym_dev.h since round 5's split of tica.hip by kernel family, + the staging helpers it uses)
            sha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "msmbuilder_amd", "csrc", f), "rb").read()
                                          for f in ("tica_common_dev.h", "tica_cg_dev.h", "tica_sym_dev.h"))).hexdigest()[:16]
            if tj["workload"].startswith("%dx%d " % (frames, F)) and tj.get("tica_hip_sha16") == sha:
                traffic = tj["bytes_per_launch"]
                traffic_source = "rocprofv3 PMC passes of this kernel source (tica_{common,cg,sym}_dev.h sha16 %s), separate run: %s" % (sha, tj["source"])
            else:
                traffic_source = "null: profiles/traffic.json (%s) was taken on another kernel source or workload" % tj.get("source", "?")
        except Exception:
            pass
        out = {
            "metric": "frames/sec tICA fit + KCenters assign, 10M x 512 feats",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None, "dtype": args.mode, "data": "synthetic",
            "rccl_ranks": rccl_ranks, "comm": comm_kind, "comm_ok": comm_ok, "comm_failed_ranks": comm_failed_ranks,
            "config": {"workload": "BASELINE configs[3]: %d x %d fp32 as %d trajectories x %d (%d frames on each of %d GPU%s), "
                                   "tICA(n_components=%d, lag_time=%d) fit+solve+transform -> KCenters(k=%d) fit+predict"
                                   % (total_frames, F, total_frames // T, T, frames, world, "s" if world > 1 else "",
                                      args.components, args.lag, args.clusters),
                       "total_frames": total_frames, "frames_per_gpu": frames, "n_features": F, "lag_time": args.lag,
                       "n_components": args.components, "n_clusters": args.clusters,
                       "parallelism": "whole trajectories dealt over %d rank%s, 1 all-reduce (tICA) + 1 all-gather per ROUND of centres (KCenters: %d exchanges in this fit)"
                                      % (world, "s" if world > 1 else "", int(kst[1] + kst[2]))},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": executed, "peak": peak, "unit": "TFLOP/s",
                         "frac": executed / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic": algorithmic, "algorithmic_frac": algorithmic / peak,
                         "frac_note": "achieved = MFMA flop the kernel executes (%d tile products of 128x128 per frame = %.0f flop) / "
                                      "HIP-event kernel time; algorithmic = SURVEY 8d's 4 F^2 = %.0f flop per frame / the same time (the "
                                      "sum/difference kernel needs fewer flops than 4 F^2, so algorithmic_frac can exceed 1 and is never a "
                                      "utilisation).  The timed steps fit the same device tensors as the warm-up: the library re-uses the "
                                      "device chunk table it built for that pointer table (about -0.3 ms of `fit` per step, 0.5 %%)"
                                      % (tiles, exe_flop, alg_flop),
                         "traffic_note": "bytes/launch at the L2 fabric side (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes); "
                                         "includes Infinity-Cache hits; algorithmic bytes/launch = %d" % (frames * F * 4),
                         "kernel_ms": mfma_ms, "frames_per_launch": frames,
                         "column_sums_folded": folded,
                         "column_sums_note": "true: the kernel's staging lanes also sum the left frames in fp64 (msm_tica_last_folded) and "
                                             "no column-sum pass reads X ahead of it; MSM_TICA_FOLD=0 restores that pass (kernel 49.2 ms + pass 3.6 ms "
                                             "against 51.1 ms at 10M x 512)",
                         "tica_accumulate_frames_per_s": frames / (mfma_ms * 1e-3)},
            "phases_ms": {k: 1e3 * float(np.mean(v)) for k, v in times.items() if k not in ("mfma_ms", "sym", "folded")},
            "top_eigenvalues": [float(x) for x in ev[:3]],
        }
        if weak is not None:
            out["weak_scaling"] = weak
        if comm_measured is not None:
            out["comm_measured_us"] = comm_measured
        fit_s, pred_s = float(np.mean(times["kcenters_fit"])), float(np.mean(times["kcenters_predict"]))
        # bytes the K passes of one fit READ on this rank (msm_kcenters_last_stats): plain passes stream the float64 row +
        # distances_ + labels_, screened passes the bfloat16 copy + the rounded-up distance; the one-off conversion reads
        # the rows once more and writes the copy.  Exact re-evaluations of screen candidates and the label / distance
        # updates are not counted, so this is a lower bound on the traffic and stays under the HBM peak by construction.
        kc_rows, kc_plain, kc_scr, kc_pb, kc_sb = [int(v) for v in kst]
        kc_bytes = kc_rows * (kc_plain * kc_pb + kc_scr * kc_sb + ((kc_pb - 8 + kc_sb) if kc_scr else 0))
        out["clustering"] = {"kcenters_fit_frames_per_s": total_frames / fit_s, "assign_frames_per_s": total_frames / pred_s,
                             "kcenters_plain_passes": kc_plain, "kcenters_screened_passes": kc_scr,
                             "kcenters_plain_pass_bytes_per_row": kc_pb, "kcenters_screened_pass_bytes_per_row": kc_sb,
                             "kcenters_streamed_bytes_per_fit": kc_bytes,
                             "kcenters_pass_TBps_per_gpu": kc_bytes / fit_s / 1e12, "hbm_peak_TBps": 8.0}

        extras = world == 1 and not args.no_extras
        if world == 1 and not args.no_mbk:
            # BASELINE configs[3] names MiniBatchKMeans(k=1000) as the clusterer of this shape: the same projection through
            # it, once, OUTSIDE the timed steps (`value` stays the metric's tICA + KCenters pipeline)
            from msmbuilder_amd import MiniBatchKMeans
            # Round 6: on the float64 projection itself -- the type `transform` emits and the reference pipeline feeds
            # scikit-learn, which then computes in float64 (msmbuilder/cluster/__init__.py:67-69, tica.py:329-352): labelling on
            # the fp64 matrix pipe (kmeans_label_f64_kernel), float64 centres.  The fp32 timing of the same fit (what rounds 1-5
            # reported, after narrowing Y) is kept beside it as `fit_ms_f32`.
            Y64 = Y if Y.dtype == torch.float64 else Y.double()
            Y32 = Y.float().contiguous()

            def mbk_fit(rows, **kw):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    m = MiniBatchKMeans(n_clusters=1000, random_state=0, **kw).fit([rows])
                torch.cuda.synchronize()
                return m, time.perf_counter() - t0
            mbk_fit(Y64[:200_000])   # (first use of the float64 kernels: module load outside the timings)
            mb, tm = mbk_fit(Y64)
            _, tm32 = mbk_fit(Y32)
            assert mb.cluster_centers_.dtype == np.float64
            out["minibatchkmeans"] = {"n_clusters": 1000, "dtype": "f64", "fit_ms": 1e3 * tm, "fit_ms_f32": 1e3 * tm32, "n_steps": int(mb.n_steps_),
                                      "fit_frames_per_s": frames / tm, "inertia_per_frame": float(mb.inertia_) / frames,
                                      "note": "MiniBatchKMeans(n_clusters=1000).fit on the [frames, %d] float64 projection as `transform` "
                                              "emits it (float64 arithmetic like scikit-learn's on such input), "
                                              "k-means++ seeding + mini-batch steps + labels_ of every frame; fit_ms_f32: the same "
                                              "fit on the projection narrowed to float32" % args.components}
            # SURVEY 8(d)'s "large-batch" variant: 65,536 rows per step (64 steps' worth of rows per launch group)
            mbl, tm = mbk_fit(Y64, batch_size=65536)
            _, tm32 = mbk_fit(Y32, batch_size=65536)
            out["minibatchkmeans_batch65536"] = {"n_clusters": 1000, "dtype": "f64", "batch_size": 65536, "fit_ms": 1e3 * tm, "fit_ms_f32": 1e3 * tm32,
                                                 "n_steps": int(mbl.n_steps_), "fit_frames_per_s": frames / tm,
                                                 "rows_through_steps_per_s": int(mbl.n_steps_) * 65536 / tm,
                                                 "inertia_per_frame": float(mbl.inertia_) / frames}
            del Y32, Y64, mb, mbl
        del labels, kc, Y

        if extras:
            # PCIe-inclusive rate (never `value`): tICA.fit on HOST numpy trajectories of the same data, staged through the
            # library's pinned ring (runtime.hip h2d_bulk) while the kernels run
            nh = min(n_seq, 200)
            host_seqs = [t.cpu().numpy() for t in seqs[:nh]]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tICA(n_components=args.components, lag_time=args.lag).fit(host_seqs[:20])      # warm the staging buffers
                torch.cuda.synchronize()
                th = time.perf_counter()
                mh = tICA(n_components=args.components, lag_time=args.lag).fit(host_seqs)
                torch.cuda.synchronize()
                th = time.perf_counter() - th
            out["h2d_inclusive"] = {"what": "tICA.fit on %d host (numpy, pageable) trajectories x %d x %d f32: PCIe staging + column "
                                            "sums + MFMA accumulation" % (nh, T, F),
                                    "h2d_inclusive_frames_per_s": nh * T / th, "GBps": nh * T * F * 4 / th / 1e9,
                                    "pcie_gen5_x16_GBps": 63.0}
            del host_seqs, mh

        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            sample = [s.cpu().numpy() for s in seqs[:64]]
            base, oracle, used = cpu_baseline(sample, args.lag, args.components, args.clusters)
            out["cpu_baseline"] = base
            # self-check: the HIP path against the float64 oracle on the SAME trajectories, at the stated tolerance
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m_s = tICA(n_components=args.components, lag_time=args.lag).fit(seqs[:len(used)])
                e_hip, e_ref = np.asarray(m_s.eigenvalues_), np.asarray(oracle.eigenvalues_)
            rel = float(np.abs(e_hip / e_ref - 1).max())
            full_vs_sample = float(np.abs(np.asarray(ev[:3]) / e_ref[:3] - 1).max())
            tol = {"f32": 1e-5, "f64": 1e-9, "bf16x2": 1e-5, "bf16": 5e-3}.get(args.mode, 1e-5)
            ok = bool(rel <= tol and full_vs_sample <= 0.25)
            out["self_check"] = {"sample_trajectories": len(used), "hip_vs_oracle_eigenvalue_max_rel_err": rel, "rtol": tol,
                                 "full_run_top3_vs_sample_max_rel_diff": full_vs_sample, "loose_bound": 0.25, "ok": ok}
            if not ok:
                exit_code = 1
            del m_s, sample, used

        if extras:
            ev32 = np.asarray(ev, dtype=np.float64)
            # --- f64 mode: the reference's own arithmetic (fp64 MFMA on widened inputs) on the same data
            os.environ["MSMBUILDER_AMD_TICA_MODE"] = "f64"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(2):
                    m64 = tICA(n_components=args.components, lag_time=args.lag).fit(seqs)
                ms64 = kernel_ms_of(m64, _lib)
                ev64 = np.asarray(m64.eigenvalues_)
            e64, _t = executed_flop_per_frame(F, False)
            out["f64"] = {"kernel": "tica_mfma_f64_kernel", "kernel_ms": ms64,
                          "executed_TFLOPs": e64 * frames / ms64 / 1e9, "frac_of_fp64_mfma_peak": e64 * frames / ms64 / 1e9 / PEAK_TFLOPS["f64"],
                          "algorithmic_TFLOPs": alg_flop * frames / ms64 / 1e9, "frames_per_s": frames / ms64 * 1e3,
                          "eigenvalues_max_rel_diff_vs_%s_run" % args.mode: float(np.abs(ev32 / ev64 - 1).max()),
                          "top_eigenvalues": [float(x) for x in ev64[:3]]}
            del m64
            os.environ["MSMBUILDER_AMD_TICA_MODE"] = args.mode
            # --- strong-scaling model: what ONE rank of an N-GPU run of this problem executes (1/N of the trajectories), for
            # N = 2, 4, 8, measured on this GPU without the collectives.  NO hardware curve exists for this repo: the driver's
            # SCALE run is the measurement, this is a model (measured phases + ASSUMED collective latencies, labelled so).
            # ASSUMED RCCL latencies over xGMI -- NOT measured (this leg runs on one GPU).  The sharded k-centers loop makes one
            # small all-gather per centre for its first plain passes and then ONE all-gather of a 99 KB round record per ROUND
            # of several centres (msm_kcenters_last_stats of the sharded fit: plain passes + rounds = the number of its exchanges)
            comm_us = {"allreduce_4MB": 150.0, "allgather_per_centre": 25.0, "allgather_per_round_record": 55.0}
            ladder = {}
            for nshare in (2, 4, 8):
                nN = max(1, n_seq // nshare)
                XN = X[: nN * T]
                seqsN = seqs[:nN]
                recN = {}
                # the ROW-SHARDED k-centers loop (msm_kcenters_fit_sharded_*, a world of one: its all-gathers are copies), i.e.
                # the code an N-rank run executes, not the single-GPU fit
                KCenters._force_sharded = True
                try:
                    step(None, seqsN, XN)
                    step(None, seqsN, XN)
                    for _ in range(7):   # (the phases' MEDIANS go into the model: a share's sub-millisecond phases move with any hiccup)
                        step(recN, seqsN, XN)
                finally:
                    KCenters._force_sharded = False
                recN.pop("sym", None)
                recN.pop("folded", None)
                kstN = (C.c_int64 * 5)()
                _lib.check(_lib.lib().msm_kcenters_last_stats(kstN))
                phN = {k: 1e3 * float(np.median(v)) for k, v in recN.items() if k != "mfma_ms"}
                stepN = sum(phN.values())
                kc_exchanges = {"one_centre": int(kstN[1]), "rounds": int(kstN[2])}
                comm_ms = (comm_us["allreduce_4MB"] + kc_exchanges["one_centre"] * comm_us["allgather_per_centre"]
                           + kc_exchanges["rounds"] * comm_us["allgather_per_round_record"]) / 1e3
                ladder[nshare] = {"trajectories": nN, "phases_ms": phN, "mfma_ms": float(np.median(recN["mfma_ms"])),
                                  "measured_step_ms": stepN, "kcenters_exchanges": kc_exchanges, "assumed_comm_ms": comm_ms,
                                  "modelled_step_ms": stepN + comm_ms, "modelled_speedup": ms_per_step / (stepN + comm_ms),
                                  "serial_fraction_of_n1_step": (phN.get("solve", 0.0) + comm_ms) / ms_per_step}
                del XN, seqsN
            l8 = ladder[8]
            out["strong_scaling_model"] = {
                "what": "one rank's share at N = 2, 4, 8 (1/N of the %d trajectories) run alone on this GPU through the sharded code "
                        "path (msm_kcenters_fit_sharded, screened passes): per-phase ms measured (medians of 7 steps), collectives "
                        "MODELLED with assumed latencies" % n_seq,
                "hardware_curve": "none: no run of this repository with N > 1 RCCL ranks on N GPUs exists; the driver's SCALE run is the measurement",
                "assumed_comm_us": comm_us, "assumed_comm_us_note": "UNMEASURED assumptions (no multi-GPU node in this run)",
                "ladder": {str(k): v for k, v in ladder.items()},
                # (the N = 8 row under the keys earlier rounds printed)
                "phases_ms": l8["phases_ms"], "mfma_ms": l8["mfma_ms"], "measured_step_ms": l8["measured_step_ms"],
                "kcenters_exchanges": l8["kcenters_exchanges"], "modelled_step_ms": l8["modelled_step_ms"],
                "modelled_speedup_at_8": l8["modelled_speedup"], "modelled_speedup_at_4": ladder[4]["modelled_speedup"],
                "modelled_speedup_at_2": ladder[2]["modelled_speedup"],
                "serial_fraction_of_n1_step": l8["serial_fraction_of_n1_step"]}
        del X, seqs
        if extras:
            torch.cuda.empty_cache()
            # --- BASELINE configs[1]: 1M x 128, lag 100, ONE trajectory (single-tile C/G kernel)
            X2 = synth(torch, 1, 1_000_000, 128, 99, dev)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(3):
                    m2 = tICA(n_components=args.components, lag_time=100).fit([X2])
                ms2 = kernel_ms_of(m2, _lib)
                t2 = time.perf_counter()
                m2 = tICA(n_components=args.components, lag_time=100).fit([X2])
                e2 = m2.eigenvalues_
                torch.cuda.synchronize()
                t2 = time.perf_counter() - t2
            sym2 = bool(m2._lagged_symmetrised)
            ex2 = executed_flop_per_frame_symw(128)[0] if sym2 else executed_flop_per_frame(128, False)[0]
            out["config2"] = {"workload": "1,000,000 x 128 fp32, one trajectory, lag 100",
                              "kernel": "tica_symw_f32_kernel" if sym2 else "tica_mfma_f32_kernel",
                              "kernel_ms": ms2, "executed_TFLOPs": ex2 * 1e6 / ms2 / 1e9, "frac": ex2 * 1e6 / ms2 / 1e9 / PEAK_TFLOPS["f32"],
                              "algorithmic_TFLOPs": 4.0 * 128 * 128 * 1e6 / ms2 / 1e9,
                              "accumulate_frames_per_s": 1e6 / ms2 * 1e3, "fit_plus_solve_ms": 1e3 * t2,
                              "top_eigenvalues": [float(x) for x in e2[:3]]}
            del X2, m2
            # --- BASELINE configs[2]'s and configs[0]'s WIDTHS through the accumulation (round 6): 2M x 171 (contact
            # features of a 21-residue peptide, featurizer.py:1149-1179) as 200 x 10,000, lag 100, and 8M x 4 (dihedral sin / cos,
            # featurizer.py:572-657) as 800 x 10,000, lag 10 -- executed = the MFMA blocks the kernel issues, algorithmic = 4 F^2
            for tag, Fw, nw, lagw in (("config3_width", 171, 200, 100), ("config1_width", 4, 800, 10)):
                Xw = synth(torch, nw, 10_000, Fw, 7 + Fw, dev)
                seqs_w = list(Xw.view(nw, 10_000, Fw).unbind(0))
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    kms, fts = [], []
                    for it in range(5):
                        torch.cuda.synchronize()
                        tw = time.perf_counter()
                        mw = tICA(n_components=min(args.components, Fw), lag_time=lagw).fit(seqs_w)
                        torch.cuda.synchronize()
                        fts.append(time.perf_counter() - tw)
                        kms.append(kernel_ms_of(mw, _lib))
                symw_on = bool(mw._lagged_symmetrised)
                exw = executed_flop_per_frame_symw(Fw)[0] if (symw_on and Fw <= 256) else executed_flop_per_frame(Fw, symw_on)[0]
                kmw, nfw = min(kms[2:]), nw * 10_000
                out[tag] = {"workload": "%d x %d fp32 as %d x 10,000, lag %d" % (nfw, Fw, nw, lagw),
                            "kernel": "tica_symw_f32_kernel" if (symw_on and Fw <= 256) else "128-wide tile kernels",
                            "kernel_ms": kmw, "fit_ms": 1e3 * min(fts[2:]), "accumulate_frames_per_s": nfw / kmw * 1e3,
                            "executed_TFLOPs": exw * nfw / kmw / 1e9, "frac_of_fp32_mfma_peak": exw * nfw / kmw / 1e9 / PEAK_TFLOPS["f32"],
                            "algorithmic_TFLOPs": 4.0 * Fw * Fw * nfw / kmw / 1e9,
                            "rows_TBps": nfw * Fw * 4 / kmw / 1e9, "hbm_peak_TBps": 8.0}
                del Xw, seqs_w, mw
            # --- float64 rows of configs[0]'s width (round 6: tica_symw_f64_kernel; scikit-learn scalers and tICA.transform emit
            # float64, the reference computes in float64): 8M x 4 float64 as 800 x 10,000, lag 10
            X64 = synth(torch, 800, 10_000, 4, 11, dev).double()
            seqs64 = list(X64.view(800, 10_000, 4).unbind(0))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kms, fts = [], []
                for it in range(5):
                    torch.cuda.synchronize()
                    tw = time.perf_counter()
                    m64 = tICA(n_components=4, lag_time=10).fit(seqs64)
                    torch.cuda.synchronize()
                    fts.append(time.perf_counter() - tw)
                    kms.append(kernel_ms_of(m64, _lib))
            out["config1_width_f64"] = {"workload": "8000000 x 4 float64 as 800 x 10,000, lag 10",
                                        "kernel": "tica_symw_f64_kernel" if bool(m64._lagged_symmetrised) else "tica_mfma_f64_kernel",
                                        "kernel_ms": min(kms[2:]), "fit_ms": 1e3 * min(fts[2:]),
                                        "rows_TBps": 8_000_000 * 4 * 8 / min(kms[2:]) / 1e9, "hbm_peak_TBps": 8.0,
                                        "note": "rounds 1-5: 13.4 ms (every float64 row took the 128-wide fp64 tile kernel)"}
            del X64, seqs64, m64
            # --- wide k-centers at scale (round 6: several centres per screened pass, distance_wbatch_dev.h): KCenters(500) on
            # 1,000,000 x 171 float32 (50 blobs), fit only
            gW = torch.Generator(device=dev).manual_seed(5)
            cW = torch.randn(50, 171, generator=gW, device=dev) * 3
            XW = (cW[torch.randint(0, 50, (1_000_000,), generator=gW, device=dev)] + torch.randn(1_000_000, 171, generator=gW, device=dev)).contiguous()
            KCenters(n_clusters=500, random_state=0).fit([XW])
            torch.cuda.synchronize()
            tW = time.perf_counter()
            kcW = KCenters(n_clusters=500, random_state=0).fit([XW])
            torch.cuda.synchronize()
            tW = time.perf_counter() - tW
            stW = (C.c_int64 * 5)()
            _lib.check(_lib.lib().msm_kcenters_last_stats(stW))
            out["kcenters_wide_1m"] = {"workload": "1,000,000 x 171 fp32 (50 blobs), KCenters(k=500, euclidean) fit",
                                       "kcenters_fit_ms": 1e3 * tW, "plain_passes": int(stW[1]), "screened_passes": int(stW[2]),
                                       "screened_pass_bytes_per_row": int(stW[4]),
                                       "note": "rounds 1-5: 500 plain passes, 78.5 ms; one centre per screened pass: 28.9 ms"}
            del XW, kcW, cW
            # --- SURVEY 8(d)'s C3 stress variant: KCenters(200) and assign_nearest on RAW contact-like features, 280,000 x 171
            # float32 as 28 trajectories x 10,000, no tICA in front (odd row length: the scalar-staged exact kernels)
            gC = torch.Generator(device=dev).manual_seed(171)
            XC = (torch.linspace(0.4, 2.5, 171, device=dev) + 0.2 * torch.randn(280_000, 171, generator=gC, device=dev)).abs().float().contiguous()
            seqsC = list(XC.view(28, 10_000, 171).unbind(0))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kcC = KCenters(n_clusters=200, random_state=0).fit(seqsC)
                torch.cuda.synchronize()
                tC = time.perf_counter()
                kcC = KCenters(n_clusters=200, random_state=0).fit(seqsC)
                torch.cuda.synchronize()
                tC = time.perf_counter() - tC
                kcC.predict(seqsC)
                torch.cuda.synchronize()
                tP = time.perf_counter()
                labC = kcC.predict(seqsC)
                torch.cuda.synchronize()
                tP = time.perf_counter() - tP
            out["config3_stress"] = {"workload": "280,000 x 171 fp32 (28 x 10,000), KCenters(k=200, euclidean) fit + predict, no tICA",
                                     "kcenters_fit_ms": 1e3 * tC, "fit_frames_per_s": 280_000 / tC,
                                     "fit_pass_TBps_if_every_pass_read_X": 200 * 280_000 * 171 * 4 / tC / 1e12,
                                     "assign_nearest_ms": 1e3 * tP, "assign_frames_per_s": 280_000 / tP,
                                     "assign_pair_elements_per_s": 280_000 * 200 * 171 / tP,
                                     "inertia": float(kcC.inertia_)}
            del XC, seqsC, kcC, labC
            # --- BASELINE configs[3]'s WIDE clusterer: the final labelling pass of MiniBatchKMeans(k=1000) on one rank's share of
            # the 8-GPU run, 1,250,000 x 512 fp32 ("final labelling pass dominates at large N"):
            # kmeans_label_v4_kernel + kmeans_inertia_kernel through msm_kmeans_label_f32, HIP events around the call
            from msmbuilder_amd.cluster.minibatchkmeans import label_inertia
            gL = torch.Generator(device=dev).manual_seed(1000)
            CL = torch.randn(1000, 512, generator=gL, device=dev) * 2
            XL = CL[torch.randint(0, 1000, (1_250_000,), generator=gL, device=dev)] + torch.randn(1_250_000, 512, generator=gL, device=dev)
            CLh = CL.cpu().numpy()
            label_inertia(XL, CLh)
            label_inertia(XL, CLh)
            tl = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                labL, inL = label_inertia(XL, CLh)
                e1.record()
                torch.cuda.synchronize()
                tl.append(e0.elapsed_time(e1))
            msL = float(min(tl))
            exL = 2.0 * 1_250_000 * 1024 * 512        # executed: 8 centre tiles of 128 x 512 features per row
            trafficL = None
            try:
                trafficL = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kmeans_label_v4_kernel"]
            except Exception:
                pass
            out["config4_label_wide"] = {
                "workload": "1,250,000 x 512 fp32 (one rank's share of BASELINE configs[3] over 8 GPUs), K = 1000: labels + inertia "
                            "(kmeans_label_v4_kernel: one workgroup per row block x 4 centre tiles, grouped per XCD; kmeans_inertia_kernel "
                            "merges the splits' candidates; the 2 MB of centres uploaded inside the call)",
                "label_plus_inertia_ms": msL, "rows_per_s": 1_250_000 / msL * 1e3,
                "executed_TFLOPs_whole_call": exL / msL / 1e9, "frac_of_fp32_mfma_peak_whole_call": exL / msL / 1e9 / PEAK_TFLOPS["f32"],
                "algorithmic_bytes": 1_250_000 * 512 * 4,
                "traffic": trafficL, "inertia_per_row": inL / 1_250_000,
                "note": "the MFMA kernel alone (rocprofv3 kernel trace, profiles/r05_label_wide.txt, last session): 10.07 ms = 0.85 of the "
                        "fp32 MFMA peak, 11.5 GB fetched + written per pass = 4.5 x the rows (all 8 tiles per workgroup: 21.8 GB, 9.9 ms; "
                        "2 tiles: 5.4 GB, 10.6 ms); kmeans_inertia_kernel 0.58 ms"}
            del XL, CL, labL
            torch.cuda.empty_cache()
            # --- BASELINE configs[4] width: F = 2048, fp32 vs bf16x2 vs bf16 MFMA
            n5 = 100
            X5 = synth(torch, n5, T, 2048, 7, dev)
            seqs5 = list(X5.view(n5, T, 2048).unbind(0))
            c5 = {"workload": "%d x 2048 fp32 as %d trajectories x %d, lag %d" % (n5 * T, n5, T, args.lag), "modes": {}}
            ref5 = None
            for mode in ("f32", "bf16x2", "bf16"):
                os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    for _ in range(2):
                        m5 = tICA(n_components=args.components, lag_time=args.lag).fit(seqs5)
                    ms5 = kernel_ms_of(m5, _lib)     # bf16 modes: the whole pack + multiply pipeline (overlapped)
                    sym5 = m5._lagged_symmetrised
                    e5 = np.asarray(m5.eigenvalues_)
                if ref5 is None:
                    ref5 = e5
                if mode == "f32":
                    ex5, _t = executed_flop_per_frame(2048, sym5)
                else:
                    ex5 = executed_flop_per_frame_bf16(2048, mode == "bf16x2")
                c5["modes"][mode] = {"accumulate_ms": ms5,
                                     "frames_per_s": n5 * T / ms5 * 1e3,
                                     "executed_TFLOPs": ex5 * n5 * T / ms5 / 1e9,
                                     "frac_of_its_mfma_peak": ex5 * n5 * T / ms5 / 1e9 / PEAK_TFLOPS[mode],
                                     "algorithmic_TFLOPs": 4.0 * 2048 * 2048 * n5 * T / ms5 / 1e9,
                                     "eigenvalues_max_rel_diff_vs_f32": float(np.abs(e5 / ref5 - 1).max())}
                del m5
            os.environ["MSMBUILDER_AMD_TICA_MODE"] = args.mode
            out["config5_width"] = c5
            del X5, seqs5
            torch.cuda.empty_cache()
            # --- BASELINE configs[4] at ONE GPU's share of the 8-GPU run: 6,250,000 x 2048, bfloat16-STORED (25.6 GB),
            # through the bf16 image path (fit = column sums + image pre-pass + MFMA kernel; wall time of the whole fit)
            from msmbuilder_amd.decomposition import tica as tica_mod
            tica_mod.release_parked_handles()
            n5g = 625
            X5g = synth_bf16(torch, n5g, T, 2048, 11, dev)
            seqs5g = list(X5g.view(n5g, T, 2048).unbind(0))
            c5g = {"workload": "%d x 2048 bfloat16-stored (%.1f GB) as %d trajectories x %d, lag %d: one rank's share of 50M x 2048 over 8 GPUs"
                               % (n5g * T, n5g * T * 2048 * 2 / 1e9, n5g, T, args.lag), "modes": {}}
            ref5g = None
            for mode in ("bf16x2", "bf16"):
                os.environ["MSMBUILDER_AMD_TICA_MODE"] = mode
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    m5 = tICA(n_components=args.components, lag_time=args.lag).fit(seqs5g)      # warm-up
                    del m5
                    torch.cuda.synchronize()
                    t5 = time.perf_counter()
                    m5 = tICA(n_components=args.components, lag_time=args.lag).fit(seqs5g)
                    torch.cuda.synchronize()
                    t5 = time.perf_counter() - t5
                    ms5 = kernel_ms_of(m5, _lib)
                    e5 = np.asarray(m5.eigenvalues_)
                ref5g = e5 if ref5g is None else ref5g
                ex5 = executed_flop_per_frame_bf16(2048, mode == "bf16x2")
                c5g["modes"][mode] = {"fit_wall_ms": 1e3 * t5, "fit_frames_per_s": n5g * T / t5,
                                      "accumulate_ms": ms5,
                                      "executed_TFLOPs": ex5 * n5g * T / ms5 / 1e9,
                                      "frac_of_bf16_mfma_peak": ex5 * n5g * T / ms5 / 1e9 / PEAK_TFLOPS["bf16"],
                                      "algorithmic_TFLOPs_whole_fit": 4.0 * 2048 * 2048 * n5g * T / t5 / 1e12,
                                      "hbm_GBps_if_read_once": n5g * T * 2048 * 2 / t5 / 1e9,
                                      "eigenvalues_max_rel_diff_vs_bf16x2": float(np.abs(e5 / ref5g - 1).max())}
                del m5
                tica_mod.release_parked_handles()
                torch.cuda.empty_cache()
            os.environ["MSMBUILDER_AMD_TICA_MODE"] = args.mode
            out["config5_per_gpu"] = c5g
            del X5g, seqs5g
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
