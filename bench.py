#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload convert|estep|dtw|traj]

Default workload = BASELINE.json configs[1]: GMMMap fvconvert, D=40, M=64, T=10^6 synthetic frames per GPU
(weak scaling: every rank converts its own shard of T frames; frames are independent, so there is no
data-path collective).  A "step" is one pass of the kernel over the rank's resident (D,T) matrix.  Inputs are in
HBM before the timed region.  For N>1 the driver launches this file through torch.distributed.run (one process
per GPU, RCCL); the timed region is bracketed by barrier + synchronize and the max over ranks is reported.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel vs the FP64 MFMA roof, timed with HIP events
on the launch stream) and `cpu_baseline` (the single-threaded C oracle -- a port of the reference's per-frame
loop, src/common.jl:17-19 + src/gmmmap.jl:101-118 -- on a bounded sample of the same frames).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FP64 roof of MI355X: 78.6 TFLOP/s (vector = matrix; AMD datasheet).  MI355X_MICROARCH.md lists no FP64 row;
# tools/microbench_f64.hip measured 78.3 TFLOP/s for back-to-back v_mfma_f64_16x16x4_f64 and showed that VALU
# FP64 shares that pipe (profiles/r01_microbench_f64.txt).
FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0


def julia_model(w, mu, sig):
    """numpy [m][d] / [m][col][row] buffers -> Julia-shaped (Dj,M) / (Dj,Dj,M) Fortran arrays (same bytes)."""
    return w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))


def convert_flops_per_frame(D, M):
    """Algorithmic FP64 flop per converted frame: M(3D^2 + 6D) + 25M."""
    return M * (3 * D * D + 6 * D) + 25 * M


def estep_flops_per_frame(Dj, M):
    return 8 * Dj * M + 25 * M


def dist_setup(n_gpus):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return world, rank, local


def barrier_sync(world):
    import torch
    import torch.distributed as dist

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    import torch.distributed as dist

    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step_fn, steps, warmup, world):
    """W untimed warmups, then exactly K steps between barrier+sync; also per-step HIP-event durations."""
    import torch

    for _ in range(warmup):
        step_fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier_sync(world)
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step_fn()
        b.record()
    barrier_sync(world)
    t1 = time.perf_counter()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    return max_over_ranks(t1 - t0, world), kernel_ms


# ------------------------------------------------------------------------------------------- convert
def bench_convert(args, world, rank):
    import torch

    import voiceconversion_jl_amd as vc
    from oracle import np_oracle as npo

    D, M, T = 40, 64, args.frames
    w, mu, sig = npo.synth_model(1002, 2 * D, M)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    X = npo.sample_frames(1002 + rank, w, mu, sig, T, 0, D)          # (T,D) == Julia (D,T) image
    Xd = torch.from_numpy(X).cuda()
    Yd = torch.empty_like(Xd)

    def step():
        vc.fvconvert(g, Xd.t(), out=Yd.t())

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)
    frames_per_s = world * T * args.steps / wall
    flops = convert_flops_per_frame(D, M) * T
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    out = {
        "metric": "converted frames/sec (D=40, M=64) at 1/2/4/8 MI355X vs CPU ref",
        "value": frames_per_s,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "GMMMap fvconvert (BASELINE configs[1])", "D": D, "M": M, "frames_per_gpu": T,
                   "sharding": f"frames x{world}, no collective"},
        "roofline": {"bound": "mfma", "kernel": "gmmmap_mfma_kernel<40,...>", "achieved": achieved,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                     "traffic": None, "flop_per_frame": convert_flops_per_frame(D, M), "kernel_ms": kernel_ms,
                     "hbm_GBps_algorithmic": 2 * D * 8 * T / (kernel_ms * 1e-3) / 1e9},
    }
    if rank == 0:
        from oracle import c_oracle as co

        ref = co.GMMMap(w, mu, sig)
        n0 = 512
        t0 = time.perf_counter()
        Yref0 = ref.fvconvert(X[:n0])
        dt0 = time.perf_counter() - t0
        n = int(min(T, max(n0, args.cpu_seconds / (dt0 / n0))))
        t0 = time.perf_counter()
        Yref = ref.fvconvert(X[:n])
        dt = time.perf_counter() - t0
        Y = Yd[:n].cpu().numpy()
        err = float(np.max(np.linalg.norm(Y - Yref, axis=1) / np.linalg.norm(Yref, axis=1)))
        out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {n} of the {T} frames, C oracle (reference-structured per-frame loop), "
                                         f"{dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        out["parity_max_rel_err_vs_oracle"] = err
        out["speedup_vs_cpu_baseline"] = frames_per_s / (n / dt)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="convert", choices=["convert"])
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (BASELINE: 10^6)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU time budget of the cpu_baseline sample")
    args = ap.parse_args()

    world, rank, _ = dist_setup(args.gpus)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
    out = {"convert": bench_convert}[args.workload](args, world, rank)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
