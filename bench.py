#!/usr/bin/env python
"""Benchmark of the importance-sampling density evaluation path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic query points already resident in HBM:
H-forward (proposal centres) -> fused sampler + G forward + scoring + log-sum-exp kernel -> finalize
(-> one RCCL all-gather of the results when N > 1).  Workload = BASELINE.json configs[1]: the
eight_octagon_gmm 100x100 evaluation grid on [-5,5]^2 (/root/reference/evaluate.py:237), dx = dy = 2,
S = 10000 (evaluate.py:225), sd_y = 0.1, scale = 0.5, log=False, G = [2, 512x10, 2], H = [2, 256x10, 2],
Glorot-uniform weights from RandomState(1234).  With N GPUs the grid is tiled N times
and sharded contiguously (weak scaling: 10000 query points per GPU).

Prints ONE JSON line on rank 0.  The cpu_baseline leg (rank 0, N = 1 only) times the C restatement
oracle/is_oracle.c — a port, not TensorFlow — on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X: 256 CU x 256 flop/clk x 2.4 GHz (MI355X_MICROARCH.md)


def build_nets(dx, dy):
    from roundtrip_amd import Generator
    rng = np.random.RandomState(1234)
    g = Generator(dx, dy, "g_net", nb_layers=10, nb_units=512, rng=rng)
    h = Generator(dy, dx, "h_net", nb_layers=10, nb_units=256, rng=rng)
    return g, h


def grid_points(n=100, lo=-5.0, hi=5.0):
    # create_2d_grid_data, /root/reference/util.py:568-573
    g1 = np.linspace(lo, hi, n)
    g2 = np.linspace(lo, hi, n)
    v1, v2 = np.meshgrid(g1, g2)
    return np.vstack((v1.ravel(), v2.ravel())).T


def cpu_baseline(g, h, y, S, sd, scale, n_points, eng):
    """Times the CPU restatement on n_points query points x S samples (same draws the GPU uses)."""
    from oracle import c_oracle
    from roundtrip_amd import _lib
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    ys = np.ascontiguousarray(y[:: max(1, len(y) // n_points)][:n_points])
    xhat = eng.mlp_forward(_lib.RT_MLP_H, ys.astype(np.float32))
    xs = eng.draw_samples(xhat, S, scale, df=1.0, seed=0).astype(np.float64)
    og = c_oracle.OracleMlp(g.weights, g.biases, g.alpha)
    t0 = time.perf_counter()
    out = c_oracle.is_estimate_given_samples(og, ys, xhat, xs, sd, scale, 1.0, False, c_oracle.ORDER_NATURAL,
                                             nthreads=threads)
    dt = time.perf_counter() - t0
    gpu = eng.is_estimate_given_samples(ys, xhat, xs, sd, scale, 1.0, log=False)
    ok = bool(np.allclose(gpu, out, rtol=1e-4, atol=1e-300))
    return {"value": len(ys) * S / dt, "unit": "sample-evals/s", "cores": threads, "kind": "port",
            "sample": "%d grid points x S=%d through oracle/is_oracle.c (C + OpenMP restatement, not TensorFlow); "
                      "matches GPU on this sample: %s" % (len(ys), S, ok),
            "seconds": dt}


def recorded_traffic(grid, S, dx, dy):
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
        if (k.get("grid"), k.get("sample_size"), k.get("dx"), k.get("dy")) == (grid, S, dx, dy) \
                and "traffic_bytes_per_launch" in d:
            best = (d["traffic_bytes_per_launch"], "PMC passes recorded in profiles/%s" % os.path.basename(f))
    return best


def _pmc_pass(counter, grid, S, timeout_s=150):
    """One rocprofv3 --pmc pass over a child run of this script (one launch of the same workload); returns the counter's
    value for the last dispatch of the fused kernel, summed over its rows, or None."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
           "--grid", str(grid), "--sample-size", str(S), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-live-traffic"]
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


def live_traffic(grid, S):
    """HBM-side bytes of ONE launch of the fused kernel on this workload, measured now: FETCH_SIZE and WRITE_SIZE from two
    separate rocprofv3 --pmc passes over child runs of this script, corrected as MI355X_MICROARCH.md (HBM section)
    prescribes: both are in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B/lane loads — the weight
    stream — at 64 B, so it is doubled.  None when rocprofv3 is missing, a pass fails, or this process is itself being
    profiled (a profiler inside a profiled run)."""
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    fetch = _pmc_pass("FETCH_SIZE", grid, S)
    if fetch is None:
        return None
    write = _pmc_pass("WRITE_SIZE", grid, S)
    if write is None:
        return None
    return (2.0 * fetch * 1024.0 + write * 1024.0, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate child "
            "runs of one launch each), KiB units, FETCH doubled (gfx950 wide loads)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sample-size", type=int, default=10000)
    ap.add_argument("--grid", type=int, default=100, help="grid side; query points per GPU = grid^2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the PMC passes recorded under profiles/ instead of two live rocprofv3 passes")
    ap.add_argument("--cpu-points", type=int, default=160,
                    help="query points of the cpu_baseline sample (160 x S=10000 is about 13 s on 16 host cores)")
    # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (never used by the driver):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo gathers through host tensors (rehearsal only)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs the torch.distributed.run launcher (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch
    dev_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    host_gather = args.backend == "gloo"
    if world > 1:
        import torch.distributed as dist
        if host_gather:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from roundtrip_amd import _lib, dist as rdist
    dx = dy = 2
    S, sd, scale = args.sample_size, 0.1, 0.5
    g, h = build_nets(dx, dy)
    eng = _lib.Engine(dev_index)
    eng.set_mlp(_lib.RT_MLP_G, g.weights, g.biases, g.alpha)
    eng.set_mlp(_lib.RT_MLP_H, h.weights, h.biases, h.alpha)

    per_gpu = args.grid * args.grid
    y_all = np.tile(grid_points(args.grid), (world, 1))          # global batch, weak scaling
    n_total = y_all.shape[0]
    lo, hi = rdist.shard_bounds(n_total, world, rank)
    y_dev = torch.from_numpy(np.ascontiguousarray(y_all[lo:hi])).to(dev)          # float64, resident
    y32_dev = y_dev.to(torch.float32)                                             # H's float32 feed
    xhat_dev = torch.empty((hi - lo, dx), dtype=torch.float32, device=dev)
    out_dev = torch.zeros(per_gpu, dtype=torch.float64, device=dev)
    gathered = (torch.empty(per_gpu * world, dtype=torch.float64, device="cpu" if host_gather else dev)
                if world > 1 else None)
    torch.cuda.synchronize()

    def step():
        eng.mlp_forward_dev(_lib.RT_MLP_H, y32_dev.data_ptr(), hi - lo, xhat_dev.data_ptr())
        eng.is_estimate_dev(y_dev.data_ptr(), xhat_dev.data_ptr(), hi - lo, S, sd, scale, out_dev.data_ptr(),
                            df=1.0, seed=0, log=False, y_index_base=lo)
        if world > 1:
            eng.sync()   # results are produced on the engine's stream; the gather runs on torch's
            dist.all_gather_into_tensor(gathered, out_dev.cpu() if host_gather else out_dev)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    eng.reset_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_gather else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctr = eng.counters()

    if rank == 0:
        evals = float(n_total) * S * args.steps
        res = (gathered if world > 1 else out_dev).cpu().numpy()
        achieved = ctr["flops"] / (ctr["kernel_ms"] * 1e-3) / 1e12 if ctr["kernel_ms"] > 0 else 0.0
        traffic = None
        if world == 1 and not args.no_live_traffic:
            traffic = live_traffic(args.grid, S)
        if traffic is None:
            traffic = recorded_traffic(args.grid, S, dx, dy)
        line = {
            "metric": "IS sample-evals/sec (N_y x S through G)",
            "value": evals / elapsed,
            "unit": "sample-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "eight_octagon_gmm 2D evaluation grid %dx%d on [-5,5]^2 per GPU (evaluate.py shape), "
                                   "dx=2 dy=2, S=%d, sd_y=0.1 scale=0.5 df=1 log=False, G=[2,512x10,2] H=[2,256x10,2] "
                                   "Glorot RandomState(1234)" % (args.grid, args.grid, S),
                       "query_points_total": int(n_total), "sample_size": S, "parallelism": "y-shard x%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": traffic[0] if traffic else None, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": "is_fused_kernel<512, 2, false, false>",
                         "kernel_ms_avg": ctr["kernel_ms"] / max(1, ctr["launches"]),
                         "launches": ctr["launches"],
                         "flops_per_sample_eval": ctr["flops"] / max(1, ctr["sample_evals"])},
            "result_check": {"finite": bool(np.all(np.isfinite(res))), "mean_density": float(res.mean())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(g, h, y_all, S, sd, scale, args.cpu_points, eng)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
