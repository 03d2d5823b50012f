#!/usr/bin/env python
"""Benchmark of the importance-sampling density evaluation path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no launcher environment (no RANK), this process starts N fresh rank processes itself — before it
touches the GPU — and relays rank 0's JSON line and the exit code.  Under a launcher it is one rank.

A step = one pass of the hot path over one batch of synthetic query points already resident in HBM:
H-forward (proposal centres) -> fused sampler + G forward + scoring + log-sum-exp kernel -> finalize
(-> one RCCL all-gather of the results when N > 1, on the same stream).  Default workload = BASELINE.json configs[1]:
the eight_octagon_gmm 100x100 evaluation grid on [-5,5]^2 (/root/reference/evaluate.py:237), dx = dy = 2,
S = 10000 (evaluate.py:225), sd_y = 0.1, scale = 0.5, log=False, G = [2, 512x10, 2], H = [2, 256x10, 2],
Glorot-uniform weights from RandomState(1234).  With N GPUs the grid is tiled N times
and sharded contiguously (weak scaling: 10000 query points per GPU).  --workload selects BASELINE.md §4's other rows
(config 1: indep-gmm; config 2's second grid: involute; config 3: arem, casp, S = 40000; config 4: hepmass, bank, and
--g-units W for its synthetic width sweep; config 5: YearPredictionMSD 20/90, strong 262144 points / weak 131072 per GPU).
With N > 1 the line carries a "ranks" object (every rank's kernel and wall time per step, the all-gather alone, the
slowest rank), and --balance-ranks re-cuts the ranks' blocks by the rates measured on a warm-up step.

No torch: device buffers, the collective (RCCL) and the timing fences all go through librt_is.so's C ABI.

Prints ONE JSON line on rank 0.  The cpu_baseline leg (rank 0, N = 1 only) times the CPU restatements — ports, not
TensorFlow — on bounded samples of the same workload: oracle/is_oracle.c (C + OpenMP) and oracle/is_ref.py (numpy/scipy,
the reference's per-point structure, main_density_est.py:324-335).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X: 256 CU x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)

# name -> (dx, dy, sd_y, scale, default S, log, query points per GPU (weak) or in total (strong), scaling, description)
# (query points None: a 2-D evaluation grid, --grid points a side, over the rectangle in GRID_RANGES)
WORKLOADS = {
    "grid": (2, 2, 0.1, 0.5, 10000, False, None, "weak",
             "eight_octagon_gmm 2D evaluation grid %(g)dx%(g)d on [-5,5]^2 per GPU (evaluate.py shape)"),
    "involute": (2, 2, 0.4, 0.5, 10000, False, None, "weak",
                 "involute 2D evaluation grid %(g)dx%(g)d on [-6,5]x[-5,5] per GPU (evaluate.py:153-154, 238-239; BASELINE config 2)"),
    "indep-gmm": (2, 2, 0.05, 0.5, 40000, True, None, "weak",
                  "indep_gmm 2D evaluation grid %(g)dx%(g)d on [-1.5,1.5]^2 per GPU, S = 40000 (BASELINE config 1)"),
    "hepmass": (8, 21, 0.1, 0.1, 40000, True, 100000, "weak",
                "UCI HEPMASS 8/21, y ~ N(0,I) RandomState(42), 100000 query points per GPU (evaluate.py:163-168; BASELINE config 4)"),
    "bank": (8, 17, 0.1, 0.1, 40000, True, 100000, "weak",
             "UCI BANK 8/17, y ~ N(0,I) RandomState(42), 100000 query points per GPU (evaluate.py:163-168; BASELINE config 4)"),
    "arem": (3, 6, 0.1, 0.1, 40000, True, 125000, "weak",
             "UCI AReM 3/6, y ~ N(0,I) RandomState(42), 1M query points over 8 GPUs = 125000 per GPU (BASELINE config 3)"),
    "casp": (5, 9, 0.1, 0.1, 40000, True, 125000, "weak",
             "UCI CASP 5/9, y ~ N(0,I) RandomState(42), 125000 query points per GPU (BASELINE config 3)"),
    "ypmsd-weak": (20, 90, 0.5, 0.1, 40000, True, 131072, "weak",
                   "UCI YearPredictionMSD 20/90, y ~ N(0,I) RandomState(42), 131072 query points per GPU (BASELINE config 5)"),
    "ypmsd-strong": (20, 90, 0.5, 0.1, 40000, True, 262144, "strong",
                     "UCI YearPredictionMSD 20/90, y ~ N(0,I) RandomState(42), 262144 query points in total (BASELINE config 5)"),
}
GRID_RANGES = {"grid": (-5.0, 5.0, -5.0, 5.0), "involute": (-6.0, 5.0, -5.0, 5.0), "indep-gmm": (-1.5, 1.5, -1.5, 1.5)}


def build_nets(dx, dy, g_units=512):
    """G = [dx, g_units x 10, dy] (the reference's is always 512 wide, main_density_est.py:447; other widths are BASELINE
    config 4's synthetic LDS-capacity sweep), H = [dy, 256 x 10, dx] (:448)."""
    from roundtrip_amd import Generator
    rng = np.random.RandomState(1234)
    g = Generator(dx, dy, "g_net", nb_layers=10, nb_units=g_units, rng=rng)
    h = Generator(dy, dx, "h_net", nb_layers=10, nb_units=256, rng=rng)
    return g, h


def grid_points(n=100, lo=-5.0, hi=5.0, lo2=None, hi2=None):
    # create_2d_grid_data, /root/reference/util.py:568-573
    g1 = np.linspace(lo, hi, n)
    g2 = np.linspace(lo if lo2 is None else lo2, hi if hi2 is None else hi2, n)
    v1, v2 = np.meshgrid(g1, g2)
    return np.vstack((v1.ravel(), v2.ravel())).T


def cpu_baseline(g, h, y, S, sd, scale, log, n_points, n_points_numpy, eng):
    """Times the CPU restatements on bounded samples of the workload (same draws the GPU uses)."""
    from oracle import c_oracle, is_ref
    from roundtrip_amd import _lib
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    ys = np.ascontiguousarray(y[:: max(1, len(y) // n_points)][:n_points])
    xhat = eng.mlp_forward(_lib.RT_MLP_H, ys.astype(np.float32))
    xs = eng.draw_samples(xhat, S, scale, df=1.0, seed=0).astype(np.float64)
    og = c_oracle.OracleMlp(g.weights, g.biases, g.alpha)
    threads = max(1, min(threads, len(ys)))      # the C restatement runs one query point per thread
    t0 = time.perf_counter()
    out = c_oracle.is_estimate_given_samples(og, ys, xhat, xs, sd, scale, 1.0, log, c_oracle.ORDER_NATURAL,
                                             nthreads=threads)
    dt = time.perf_counter() - t0
    gpu = eng.is_estimate_given_samples(ys, xhat, xs, sd, scale, 1.0, log=log)
    ok = bool(np.allclose(gpu, out, rtol=1e-4, atol=1e-300))
    res = {"value": len(ys) * S / dt, "unit": "sample-evals/s", "cores": threads, "kind": "port",
           "sample": "%d query points x S=%d through oracle/is_oracle.c (C + OpenMP restatement, not TensorFlow); "
                     "matches GPU on this sample: %s" % (len(ys), S, ok),
           "seconds": dt}
    if n_points_numpy > 0:
        # the reference's own structure (main_density_est.py:324-335): one query point after another, G in 1024-row
        # sub-batches (predict_y), scipy.stats.t.logpdf, Python max(); numpy/scipy arithmetic, Python-level single thread
        # (the reference's pool is GIL-bound), BLAS threads as the box sets them
        try:
            from threadpoolctl import threadpool_info
            blas = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
        except Exception:
            blas = None
        n2 = min(n_points_numpy, len(ys))
        gn, hn = is_ref.DenseNet(g.weights, g.biases, g.alpha), is_ref.DenseNet(h.weights, h.biases, h.alpha)
        t0 = time.perf_counter()
        out2 = is_ref.estimate_py_with_IS(gn, hn, ys[:n2], sd_y=sd, scale=scale, sample_size=S, bs=1024, df=1, log=log,
                                          x_samples=xs[:n2])
        dt2 = time.perf_counter() - t0
        ok2 = bool(np.allclose(gpu[:n2], out2, rtol=1e-4, atol=1e-300))
        res["numpy_restatement"] = {
            "value": n2 * S / dt2, "unit": "sample-evals/s", "cores": blas, "kind": "port", "seconds": dt2,
            "sample": "%d query points x S=%d through oracle/is_ref.py (numpy/scipy statement-by-statement restatement of "
                      "main_density_est.py:266-348: per-point loop, 1024-row G sub-batches, scipy t.logpdf; one Python "
                      "thread, %s BLAS threads; not TensorFlow); matches GPU on this sample: %s" % (n2, S, blas, ok2)}
    return res


def recorded_traffic(key):
    """HBM-side bytes per launch of the fused kernel for THIS workload from the committed PMC passes
    (profiles/r*_pmc_v*.json, written by scripts/pmc.sh + scripts/pmc_profile.py).  The fallback of live_traffic()."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_v*.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        k = d.get("bench_key", {})
        if (k.get("workload", "grid"), k.get("grid"), k.get("sample_size"), k.get("dx"), k.get("dy"), k.get("g_units", 512)) == key \
                and "traffic_bytes_per_launch" in d:
            best = (d["traffic_bytes_per_launch"], "PMC passes recorded in profiles/%s" % os.path.basename(f))
    return best


def _pmc_pass(counter, child_args, timeout_s=150):
    """One rocprofv3 --pmc pass over a child run of this script (one launch of the same workload); returns the counter's
    value for the last dispatch of the fused kernel, summed over its rows, or None."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__)] + \
        child_args + ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-live-traffic"]
    try:
        proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            if proc.wait(timeout=timeout_s) != 0:
                return None
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)      # the exact process group started above
            proc.wait()
            return None
        acc = {}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "is_fused_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        d = int(row["Dispatch_Id"])
                        acc[d] = acc.get(d, 0.0) + float(row["Counter_Value"])
        return acc[max(acc)] if acc else None
    except (OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic(child_args):
    """HBM-side bytes of ONE launch of the fused kernel on this workload, measured now: FETCH_SIZE and WRITE_SIZE from two
    separate rocprofv3 --pmc passes over child runs of this script, corrected as MI355X_MICROARCH.md (HBM section)
    prescribes: both are in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B/lane loads — the weight
    stream — at 64 B, so it is doubled.  None when rocprofv3 is missing, a pass fails, or this process is itself being
    profiled (a profiler inside a profiled run)."""
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    fetch = _pmc_pass("FETCH_SIZE", child_args)
    if fetch is None:
        return None
    write = _pmc_pass("WRITE_SIZE", child_args)
    if write is None:
        return None
    return (2.0 * fetch * 1024.0 + write * 1024.0, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate child "
            "runs of one launch each), KiB units, FETCH doubled (gfx950 wide loads)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """--gpus N > 1 without a launcher: N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, like torch.distributed.run sets them), started before this process has made any GPU call.  This parent
    only relays: the ranks inherit stdout/stderr (rank 0 prints the JSON line), and the exit code is the first failure's.
    A rank that fails takes the others down (by their exact PIDs)."""
    port = _free_port()
    comm_port = _free_port()            # roundtrip_amd.dist's own rendezvous (RCCL's id): a port of its own, not MASTER_PORT + k
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RT_COMM_RUN_ID="bench-%d" % os.getpid(),
                   RT_COMM_PORT=str(comm_port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    import signal

    def stop_ranks(signum, frame):          # the parent is being stopped (a harness's time limit): so are its ranks
        for p in procs:
            if p.poll() is None:
                p.terminate()
        sys.exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, stop_ranks)
    rc = 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    for o in sorted(live):
                        procs[o].terminate()
            if live:
                time.sleep(0.05)
    finally:
        for p in procs:                     # nothing of ours is left behind, whatever ended the loop
            if p.poll() is None:
                p.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="grid", choices=sorted(WORKLOADS),
                    help="grid = BASELINE config 2 (the default, what the metric is quoted on at N = 1); the others are "
                         "BASELINE.md §4's other rows at full size (involute: config 2's second grid; indep-gmm: 1; arem, "
                         "casp: 3; hepmass, bank: 4; ypmsd-*: 5)")
    ap.add_argument("--g-units", type=int, default=512,
                    help="hidden width of G (10 layers): BASELINE config 4's synthetic width sweep {64,128,256,512,1024}; the "
                         "reference's G is always 512 wide")
    ap.add_argument("--balance-ranks", action="store_true",
                    help="N > 1: after the warm-up steps, re-cut the ranks' blocks of query points in proportion to the rate "
                         "each rank measured on them (same results: a point's estimate depends on its global index only)")
    ap.add_argument("--sample-size", type=int, default=None, help="S (default: the workload's)")
    ap.add_argument("--grid", type=int, default=100, help="grid workload: grid side; query points per GPU = grid^2")
    ap.add_argument("--points", type=int, default=None,
                    help="other workloads: query points per GPU (weak) or in total (strong) instead of the BASELINE size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the PMC passes recorded under profiles/ instead of two live rocprofv3 passes")
    ap.add_argument("--cpu-points", type=int, default=160,
                    help="query points of the cpu_baseline sample (160 x S=10000 is about 13 s on 16 host cores)")
    ap.add_argument("--cpu-points-numpy", type=int, default=64,
                    help="query points of the numpy-restatement leg of cpu_baseline (64 x S=10000 is about 5 s on 16 BLAS threads)")
    # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (never used by the driver):
    ap.add_argument("--backend", default="rccl", choices=["rccl", "nccl", "gloo"],
                    help="rccl (= nccl): the gather runs on RCCL through librt_is.so; gloo: through torch.distributed on "
                         "host tensors (rehearsal only: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    args = ap.parse_args()

    if args.gpus > 1:
        # this pool's host driver only supports dmabuf IPC: without it RCCL's peer mappings fail (hipIpcGetMemHandle).
        # Read when the HSA runtime starts, i.e. before the first HIP call of this process and of the ranks it starts.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))          # nothing above has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    # one step = ONE launch of the fused kernel (the library would otherwise split a process's first long call into a
    # short share-measuring block and the rest, DESIGN.md §4 "XCD balance"): the launches rocprofv3 --stats averages for
    # this command are then all alike, and the timed steps run on measured shares after the first warm-up step either way
    os.environ.setdefault("RT_IS_FIRST_BLOCK", "0")
    from roundtrip_amd import _lib, dist as rdist
    dev_index = 0 if args.share_gpu else local_rank
    use_gloo = args.backend == "gloo"
    tdist = None
    if world > 1 and use_gloo:
        import torch
        import torch.distributed as tdist
        tdist.init_process_group("gloo")

    dx, dy, sd, scale, S_default, log, n_points, scaling, desc = WORKLOADS[args.workload]
    S = args.sample_size or S_default
    g, h = build_nets(dx, dy, args.g_units)
    eng = _lib.Engine(dev_index)
    eng.set_mlp(_lib.RT_MLP_G, g.weights, g.biases, g.alpha)
    eng.set_mlp(_lib.RT_MLP_H, h.weights, h.biases, h.alpha)
    if world > 1 and not use_gloo:
        rdist.init(eng)                            # RCCL communicator on the engine's device and stream

    is_grid = n_points is None
    if is_grid:
        per_gpu = args.grid * args.grid
        n_total = per_gpu * world
        tile = grid_points(args.grid, *GRID_RANGES[args.workload])
        desc = desc % {"g": args.grid}
    else:
        n_points = args.points or n_points
        n_total = n_points * world if scaling == "weak" else n_points

    def rows(lo, hi):
        """Query points [lo, hi) of the job's one global list (the same whatever the rank count or the cut)."""
        if is_grid:       # the grid, tiled once per GPU
            idx = np.arange(lo, hi) % per_gpu
            return np.ascontiguousarray(tile[idx])
        # rows [lo, hi) of the one global stream RandomState(42).standard_normal((n_total, dy)), float32 values
        rs = np.random.RandomState(42)
        skip = lo * dy
        while skip > 0:
            k = min(skip, 1 << 24)
            rs.standard_normal(k)
            skip -= k
        return rs.standard_normal((hi - lo, dy)).astype(np.float32).astype(np.float64)

    # device buffers of this rank's block (re-made when --balance-ranks re-cuts the blocks after the warm-up)
    bufs = {}

    def place(bounds):
        """Uploads this rank's block under `bounds` (one (lo, hi) per rank): y (float64 + H's float32 feed), xhat, out;
        `per` = the largest block, which every rank contributes to the all-gather (padded)."""
        for k in ("y", "y32", "xhat", "out", "gathered"):
            if bufs.get(k):
                eng.dev_free(bufs[k])
        lo, hi = bounds[rank]
        per = max(1, max(b[1] - b[0] for b in bounds))
        y_mine = rows(lo, hi)
        bufs.update(lo=lo, hi=hi, n=hi - lo, per=per, bounds=bounds, y_host=y_mine,
                    y=eng.to_device(y_mine), y32=eng.to_device(y_mine.astype(np.float32)),
                    xhat=eng.dev_alloc(max(1, hi - lo) * dx * 4), out=eng.to_device(np.zeros(per, dtype=np.float64)),
                    gathered=eng.dev_alloc(per * world * 8) if world > 1 and not use_gloo else None)

    place([rdist.shard_bounds(n_total, world, r) for r in range(world)])
    gathered_host = None

    def step(gather=True):
        nonlocal gathered_host
        b = bufs
        eng.mlp_forward_dev(_lib.RT_MLP_H, b["y32"], b["n"], b["xhat"])
        eng.is_estimate_dev(b["y"], b["xhat"], b["n"], S, sd, scale, b["out"], df=1.0, seed=0, log=log, y_index_base=b["lo"])
        if world > 1 and gather:
            if use_gloo:
                loc = torch.from_numpy(eng.from_device(b["out"], (b["per"],), np.float64))       # waits for the stream
                gathered_host = torch.empty(b["per"] * world, dtype=torch.float64)
                tdist.all_gather_into_tensor(gathered_host, loc)
            else:
                eng.comm_all_gather_dev(b["out"], b["per"], b["gathered"])                   # same stream, no host hop

    def gather_doubles(vals):
        """(world, len(vals)) float64: every rank's `vals` (the bookkeeping of the N > 1 line, not the data path)."""
        v = np.ascontiguousarray(vals, dtype=np.float64)
        if world == 1:
            return v[None, :]
        if use_gloo:
            out = torch.empty(world * v.size, dtype=torch.float64)
            tdist.all_gather_into_tensor(out, torch.from_numpy(v))
            return out.numpy().reshape(world, v.size)
        return eng.comm_all_gather(v)

    def barrier():
        if world > 1:
            tdist.barrier() if use_gloo else eng.comm_barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        if use_gloo:
            t = torch.tensor([x], dtype=torch.float64)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            return float(t.item())
        return float(eng.comm_all_reduce_max([x])[0])

    # long workloads (the full-size configs run for minutes): a line on stderr every minute, so that a harness watching
    # for output does not take the run for hung; stdout stays the one JSON line
    import threading
    done = threading.Event()
    t_start = time.perf_counter()

    def heartbeat():
        while not done.wait(60.0):
            print("[bench] rank %d: %.0f s, still running" % (rank, time.perf_counter() - t_start), file=sys.stderr, flush=True)
    if rank == 0:
        threading.Thread(target=heartbeat, daemon=True).start()

    # not a warm-up step: one tiny H forward, so that loading the library's code object (all kernels are in one; by the
    # metric's definition library load is outside the measured call) never lands in a timed step, even under --warmup 0.
    # No extra launch of the fused kernel: the rocprofv3 --stats average of this command is over equal launches.
    if bufs["n"] > 0:
        eng.mlp_forward_dev(_lib.RT_MLP_H, bufs["y32"], min(bufs["n"], 64), bufs["xhat"])
        eng.sync()
    for _ in range(args.warmup):
        step()
    eng.sync()          # everything this process puts on the GPU runs on the engine's stream
    rank_weights = None
    if world > 1 and args.balance_ranks:
        # one more untimed step, measured: every rank's rate on ITS block (query points per ms of its fused kernel); the
        # blocks are then re-cut in proportion.  Ranks that would change by less than 0.2 % keep the equal cut.
        barrier()
        eng.reset_counters()
        step(gather=False)
        eng.sync()
        rate = bufs["n"] / max(1e-9, eng.counters()["kernel_ms"])
        rates = gather_doubles([rate])[:, 0]
        if rates.min() > 0 and rates.max() / rates.min() > 1.002:
            rank_weights = [float(v) for v in rates]
            place([rdist.shard_bounds(n_total, world, r, rank_weights) for r in range(world)])
            step(gather=False)                                # buffers touched once before the timed region
            eng.sync()
    barrier()
    eng.reset_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    local_elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(local_elapsed)
    ctr = eng.counters()
    attribution = None
    if world > 1:
        # what a scaling loss would be made of: every rank's kernel time and wall time per step, and the all-gather alone
        # (5 stand-alone gathers of the same payload after the timed region, between barriers)
        t_g = []
        for _ in range(5):
            barrier()
            tg0 = time.perf_counter()
            if use_gloo:
                loc = torch.zeros(bufs["per"], dtype=torch.float64)
                tdist.all_gather_into_tensor(torch.empty(bufs["per"] * world, dtype=torch.float64), loc)
            else:
                eng.comm_all_gather_dev(bufs["out"], bufs["per"], bufs["gathered"])
                eng.sync()
            t_g.append((time.perf_counter() - tg0) * 1e3)
        mine = [ctr["kernel_ms"] / args.steps, local_elapsed / args.steps * 1e3, float(np.median(t_g)), float(bufs["n"])]
        allr = gather_doubles(mine)
        k_ms = allr[:, 0]
        attribution = {
            "per_rank_kernel_ms": [round(float(v), 3) for v in k_ms],
            "per_rank_step_ms": [round(float(v), 3) for v in allr[:, 1]],
            "per_rank_query_points": [int(v) for v in allr[:, 3]],
            "slowest_rank": int(np.argmax(k_ms)),
            "gather_ms": round(float(allr[:, 2].max()), 3),
            "efficiency_vs_slowest": float(k_ms.mean() / k_ms.max()) if k_ms.max() > 0 else None,
            "non_kernel_ms": round(float(elapsed / args.steps * 1e3 - k_ms.max()), 3),
            "rank_weights": rank_weights,
            "note": "kernel = the fused IS kernel's launches (HIP events on each rank's stream); step = the rank's own wall "
                    "time per step; gather_ms = the all-gather alone (median of 5, max over ranks); efficiency_vs_slowest = "
                    "mean / max of the ranks' kernel times (1 = balanced); non_kernel_ms = job ms_per_step - slowest kernel",
        }

    if rank == 0:
        evals = float(n_total) * S * args.steps
        per, bounds = bufs["per"], bufs["bounds"]
        if world == 1:
            res = eng.from_device(bufs["out"], (per,), np.float64)[:n_total]
        else:
            flat = gathered_host.numpy() if use_gloo else eng.from_device(bufs["gathered"], (per * world,), np.float64)
            res = np.concatenate([flat[r * per: r * per + (b[1] - b[0])] for r, b in enumerate(bounds)])
        achieved = ctr["flops"] / (ctr["kernel_ms"] * 1e-3) / 1e12 if ctr["kernel_ms"] > 0 else 0.0
        traffic = None
        child_args = ["--workload", args.workload, "--grid", str(args.grid), "--sample-size", str(S), "--g-units",
                      str(args.g_units)] + (["--points", str(args.points)] if args.points else [])
        if world == 1 and not args.no_live_traffic:
            traffic = live_traffic(child_args)
        if traffic is None:
            traffic = recorded_traffic((args.workload, args.grid if is_grid else None, S, dx, dy, args.g_units))
        line = {
            "metric": "IS sample-evals/sec (N_y x S through G)",
            "value": evals / elapsed,
            "unit": "sample-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, dx=%d dy=%d, S=%d, sd_y=%g scale=%g df=1 log=%s, G=[%d,%dx10,%d] H=[%d,256x10,%d] "
                                   "Glorot RandomState(1234)" % (desc, dx, dy, S, sd, scale, log, dx, args.g_units, dy, dy, dx),
                       "query_points_total": int(n_total), "sample_size": S, "parallelism": "y-shard x%d" % world,
                       "collective": None if world == 1 else ("gloo (rehearsal)" if use_gloo else "rccl all-gather")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic[0] if traffic else None, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": eng.is_kernel_name(1.0),
                         "kernel_ms_avg": ctr["kernel_ms"] / max(1, ctr["launches"]),
                         "launches": ctr["launches"],
                         "flops_per_sample_eval": ctr["flops"] / max(1, ctr["sample_evals"])},
            "result_check": {"finite": bool(np.all(np.isfinite(res))), "mean": float(res.mean())},
        }
        if attribution is not None:
            line["ranks"] = attribution
            line["roofline"]["note"] = "rank 0's kernel; every rank's kernel time is in ranks.per_rank_kernel_ms"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(g, h, bufs["y_host"], S, sd, scale, log, args.cpu_points,
                                                args.cpu_points_numpy, eng)
        print(json.dumps(line), flush=True)
    done.set()
    barrier()
    if world > 1 and not use_gloo:
        rdist.shutdown()
    if tdist is not None:
        tdist.destroy_process_group()
    for k in ("y", "y32", "xhat", "out", "gathered"):
        eng.dev_free(bufs[k])
    eng.close()


if __name__ == "__main__":
    main()
