#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|convert|estep|estep_full|em_full|dtw|traj|trajgv]

The default (`all`) prints the headline line of configs[1] and, under `"workloads"`, the same measurement (ms_per_step,
kernel_ms, roofline, parity against the oracle, cpu_baseline) for each of BASELINE configs[1..4]: convert, the diagonal
E-step (with its all-reduce timed separately), DTW and the trajectory conversion -- K timed steps each.

Headline workload = BASELINE.json configs[1]: GMMMap fvconvert, D=40, M=64, T=10^6 synthetic frames per GPU
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


PMC_DIR = "r06_pmc"
SOURCE_FILES = ("*.hip", "*.hpp", "*.inc", "*.cpp", "Makefile")


def source_hash():
    """sha256 over the library's sources (csrc/): what a committed PMC file is stamped with (tools/pmc_traffic.sh) and
    what bench.py compares before quoting it -- traffic collected from another build of the kernels is refused."""
    import glob
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "voiceconversion.jl_amd", "csrc")
    for f in sorted(sum((glob.glob(os.path.join(csrc, pat)) for pat in SOURCE_FILES), [])):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# What FETCH_SIZE reports per byte actually fetched, by load type -- MI355X_MICROARCH.md (HBM section) gives the 16-byte
# streaming case (half); tools/microbench_fetch.hip + tools/fetch_calibration.py replay the patterns of this library's
# kernels on buffers of known size (profiles/r03_pmc/fetch_calibration.json): EVERY vector-memory load pattern tried (16
# and 8 bytes per lane, streaming or 16 rows 320 bytes apart) reports exactly half, scalar loads (s_load) and every store
# pattern report the bytes.  So FETCH is doubled for kernels that read through vector loads; the fused DTW kernel mixes
# both (template rows by vector loads, sequence columns by scalar loads) and gets the share measured for it once.
VECTOR_FETCH_FACTOR = 2.0
# fused DTW kernel: of its FETCH_SIZE, the part that is vector loads (a probe build that points every lane's template loads at
# one row leaves 227 of the 487 MB at D = 40 with three segments; 222 of 310 MB with whole-length jobs): 0.53 -> x 1.53
DTW_FUSED_FETCH_FACTOR = 1.53


def traffic_from_table(tr, kernel_prefix, fetch_factor=VECTOR_FETCH_FACTOR):
    """FETCH + WRITE bytes per bench step of every kernel whose name contains `kernel_prefix`, from a table made by
    tools/pmc_traffic.py.  KB counters x 1024, FETCH times `fetch_factor` (a number, or {substring of the kernel name:
    factor} with "" as the default)."""
    total = None
    for k, v in tr.items():
        if k.startswith("_") or kernel_prefix not in k:
            continue
        ff = fetch_factor
        if isinstance(fetch_factor, dict):
            ff = next((x for key, x in fetch_factor.items() if key and key in k), fetch_factor.get("", VECTOR_FETCH_FACTOR))
        f = v.get("FETCH_SIZE_KB_per_step", 0.0) * 1024.0
        w = v.get("WRITE_SIZE_KB_per_step", 0.0) * 1024.0
        total = (total or 0.0) + ff * f + w
    return total


def table_problems(tr, kernel_prefixes):
    """Why a PMC table must not be quoted: problems its collector recorded, or a kernel of interest whose launch count is not
    a whole number of launches per bench step (round 4: ~60 clock-warm launches were divided by 3 steps)."""
    rec = (tr.get("_meta") or {}).get("problems") or {}
    if isinstance(rec, dict):        # per kernel: only the kernels of interest count (model uploads, set-up calls do not)
        why = ["%s: %s" % (k[:60], v) for k, v in rec.items() if any(p in k for p in kernel_prefixes)]
    else:
        why = list(rec)
    for k, v in tr.items():
        if k.startswith("_") or not any(p in k for p in kernel_prefixes):
            continue
        lps = v.get("launches_per_step")
        if lps is not None and abs(lps - round(lps)) > 1e-9:
            why.append("%s: %.3f launches per step" % (k[:60], lps))
    return why


def pmc_traffic(fname, kernel_prefixes, fetch_factor=VECTOR_FETCH_FACTOR, live=None):
    """`roofline.traffic` and where it came from.  `live`: a table measured by THIS run (measure_traffic_live); otherwise
    the committed file profiles/<PMC_DIR>/<fname> -- used only when its `_meta.source_hash` equals the hash of the sources
    the loaded library was built from; a stale or inconsistent table yields traffic = None and says so."""
    if isinstance(kernel_prefixes, str):
        kernel_prefixes = (kernel_prefixes,)
    src = source_hash()
    if live is not None:
        tr, origin = live, {"source": "measured in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes)", "source_hash": src}
    else:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", PMC_DIR, fname)))
        except (OSError, ValueError):
            return None, {"source": f"profiles/{PMC_DIR}/{fname} missing"}
        meta = tr.get("_meta", {})
        if meta.get("source_hash") != src:
            return None, {"source": f"profiles/{PMC_DIR}/{fname} REFUSED: collected at source hash {meta.get('source_hash')}, "
                                    f"the library sources are now {src}"}
        origin = {"source": f"profiles/{PMC_DIR}/{fname}", "source_hash": src, "collected": meta.get("collected")}
    why = table_problems(tr, kernel_prefixes)
    if why:
        return None, {"source": origin["source"] + " REFUSED: " + "; ".join(why[:3])}
    vals = [traffic_from_table(tr, k, fetch_factor) for k in kernel_prefixes]
    vals = [v for v in vals if v is not None]
    return (sum(vals) if vals else None), origin


def attach_traffic(out, fname, kernel_prefixes, fetch_factor=VECTOR_FETCH_FACTOR, standard=True, live=None, algorithmic_bytes=None):
    """Fill roofline.traffic (+ traffic_source, traffic_GBps, traffic_x_algorithmic): HBM bytes per step, rocprofv3 FETCH_SIZE +
    WRITE_SIZE in separate passes, FETCH x `fetch_factor` (units and the gfx950 correction: profiles/README.md).  `standard`:
    the run has the sizes the PMC passes were collected at.  A figure that implies more than the HBM peak over the kernel's
    own time is not evidence and is refused."""
    roof = out["roofline"]
    if not standard and live is None:
        roof["traffic"], roof["traffic_source"] = None, {"source": "non-standard size: no PMC pass"}
        return
    roof["traffic"], roof["traffic_source"] = pmc_traffic(fname, kernel_prefixes, fetch_factor, live)
    roof["traffic_raw"], _ = pmc_traffic(fname, kernel_prefixes, 1.0, live)
    if roof["traffic"] is not None and roof.get("kernel_ms"):
        gbps = roof["traffic"] / (roof["kernel_ms"] * 1e-3) / 1e9
        if gbps > HBM_PEAK_GBS:
            roof["traffic_source"] = {"source": roof["traffic_source"]["source"] + " REFUSED: %.0f GB/s over the kernel time exceeds "
                                                "the %.0f GB/s HBM peak" % (gbps, HBM_PEAK_GBS)}
            roof["traffic"] = roof["traffic_raw"] = None
            return
        roof["traffic_GBps"] = gbps
        roof["hbm_frac"] = gbps / HBM_PEAK_GBS            # the other roof: measured HBM bytes over the kernel's time / 8 TB/s
        if algorithmic_bytes:
            roof["algorithmic_bytes"] = algorithmic_bytes
            roof["traffic_x_algorithmic"] = roof["traffic"] / algorithmic_bytes


def measure_traffic_live(workload, extra=(), timeout=360):
    """Run tools/pmc_traffic.py for `workload` as CHILD processes (two rocprofv3 --pmc passes of this same bench command,
    kernel-trace only).  Must be called before this process has touched the GPU.  Returns the per-kernel table or None."""
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3") or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return None
    out = tempfile.mkdtemp(prefix="vcmi_pmc_")
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), workload, "--out", out] + list(extra),
                           capture_output=True, text=True, timeout=timeout, cwd=out, env=dict(os.environ, TMPDIR=out))
        if p.returncode != 0:
            return None
        return json.load(open(os.path.join(out, "traffic.json")))
    except (OSError, ValueError, subprocess.SubprocessError):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(argv, n_gpus):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves, as the driver
    would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`), as a CHILD
    process, and exit with its code.  This runs before anything in this process has touched torch.cuda or HIP (a process
    that has initialised the GPU must never be replaced or forked into ranks); the launcher itself never imports
    torch."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def host_facts():
    """What the host-pointer path's speed depends on besides PCIe: sockets, huge pages, copy threads."""
    import glob

    def rd(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None

    return {"cores": os.cpu_count(), "numa_nodes": len(glob.glob("/sys/devices/system/node/node[0-9]*")),
            "transparent_hugepage": rd("/sys/kernel/mm/transparent_hugepage/enabled"),
            "copy_threads": os.environ.get("VCMI_HOST_THREADS", "default (min(16, cores/4))"),
            "placement": os.environ.get("VCMI_HOST_NUMA", "default (copy workers and pinned slots on the calling thread's socket)")}


BACKEND = {"name": None}
LIVE_PMC = {}            # workload -> per-kernel traffic table measured by this run (measure_traffic_live)


def dist_setup(n_gpus):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        # never report a line whose n_gpus is not the number of ranks that actually ran
        print(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world}: refusing to run", file=sys.stderr)
        sys.exit(2)
    # VCMI_BENCH_DEVICE / VCMI_BENCH_BACKEND exist only to smoke-test the multi-rank code path on a 1-GPU box
    # (all ranks on one device, gloo) or on a CPU box (`--workload selftest`); the real launch is one rank per GPU
    # over RCCL ("nccl").  VCMI_BENCH_FORCE_PG=1 creates the process group even for one rank, so that the RCCL
    # communicator and its all-reduce are exercised on a 1-GPU box.
    backend = os.environ.get("VCMI_BENCH_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available()
    dev = int(os.environ.get("VCMI_BENCH_DEVICE", local))
    if have_gpu:
        if backend == "nccl" and dev >= torch.cuda.device_count():
            print(f"bench.py: rank {rank} wants device {dev} but only {torch.cuda.device_count()} visible", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(dev)
        from voiceconversion_jl_amd import _lib

        _lib.set_device(dev)
    if world > 1 or os.environ.get("VCMI_BENCH_FORCE_PG") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != n_gpus:
            print(f"bench.py: process group has {dist.get_world_size()} ranks, expected {n_gpus}", file=sys.stderr)
            sys.exit(2)
        BACKEND["name"] = backend
    return world, rank, local


def _dev():
    import torch

    return "cuda" if torch.cuda.is_available() else "cpu"


def barrier_sync(world):
    import torch
    import torch.distributed as dist

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    import torch.distributed as dist

    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(x, world):
    """Every rank's value of x (rank order), on every rank."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return [x]
    t = torch.zeros(world, dtype=torch.float64, device=_dev())
    t[dist.get_rank()] = x
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


PER_RANK = {}


CLOCK_WARM_MS = 100.0     # --clock-warm-ms
CLOCK_WARM = {}           # the steady-clock pass of the last timed_steps() (reported BESIDE the protocol figure)
_MID = {"ev": None}       # the event split_mark() records inside the step being timed


def split_mark():
    """Called by a step between its kernels and its collective: records the step's middle event when one is armed (the
    E-step prices its kernels and its all-reduce from the SAME timed steps)."""
    ev = _MID["ev"]
    if ev is not None:
        ev.record()


def _timed_pass(step_fn, steps, warmup, world, split):
    import torch

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True) if split else None) for _ in range(steps)]
    for _ in range(warmup):
        step_fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for a, b, m in evs:
        a.record()
        _MID["ev"] = m
        step_fn()
        b.record()
    _MID["ev"] = None
    barrier_sync(world)
    t1 = time.perf_counter()
    per_step = [a.elapsed_time(b) for a, b, _ in evs]
    first = [a.elapsed_time(m) for a, _, m in evs] if split else None
    return t1 - t0, per_step, first


def timed_steps(step_fn, steps, warmup, world, steady=True, split=False):
    """The contract's protocol to the letter: W untimed warmups, then exactly K steps between barrier + synchronize, with
    per-step HIP events on the launch stream.  Returns (max-over-ranks wall seconds of the K steps, mean event ms per step) --
    `value`, `ms_per_step` and `roofline.kernel_ms` of every line come from THIS pass.  With `split` the step calls
    split_mark() once and PER_RANK["first_part_ms"] holds the mean duration up to the mark.

    The Python garbage collector is off inside the timed region (as `timeit` does): with torch imported a full collection
    pauses the interpreter for ~40 ms (tools/dtw_host_probe.py), which is the whole timed region of a 20 x 1.7 ms workload.

    Steady-clock pass (`--clock-warm-ms`, default 100; 0 = off; `steady=False` = skip): the MI355X raises its shader clock
    over the first ~30-40 ms of back-to-back kernels (profiles/r04_convert_kernel_calls.csv: 1.85 ms at the start of a burst,
    1.52 ms from the 20th launch on), so W = 5 warmups of a 1.7 ms step end inside the ramp.  AFTER the protocol pass the same
    K steps are therefore timed once more behind that many ms of untimed steps; the result goes into CLOCK_WARM
    (`steady_ms_per_step`, `steady_kernel_ms`) and is reported beside the protocol figure, never as `value`."""
    import gc

    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        wall, per_step, first = _timed_pass(step_fn, steps, warmup, world, split)
        wall = max_over_ranks(wall, world)
        kernel_ms = float(np.mean(per_step))
        PER_RANK["wall_s"] = gather_over_ranks(wall, world)
        PER_RANK["kernel_ms"] = gather_over_ranks(kernel_ms, world)
        if split:
            PER_RANK["first_part_ms"] = float(np.mean(first))
        CLOCK_WARM.clear()
        if steady and CLOCK_WARM_MS > 0:
            # the same number of untimed steps on every rank (a step may hold a collective): from the slowest rank's step time
            one = max_over_ranks(kernel_ms * 1e-3, world)
            n_warm = int(min(max(CLOCK_WARM_MS * 1e-3 / max(one, 1e-6), 1.0), 4000.0))
            swall, sper, _ = _timed_pass(step_fn, steps, n_warm, world, False)
            CLOCK_WARM.update({"steady_ms_per_step": round(max_over_ranks(swall, world) / max(steps, 1) * 1e3, 4),
                               "steady_kernel_ms": round(float(np.mean(sper)), 4), "untimed_steps": n_warm})
    finally:
        if gc_was_on:
            gc.enable()
    return wall, kernel_ms


# ------------------------------------------------------------------------------------------- convert
MFMA_FLOP = 2048.0        # one v_mfma_f64_16x16x4_f64: 16 x 16 x 4 multiply-adds

CONVERT_VARIANTS = {
    # BASELINE configs[1] / SURVEY 8(d): the prescribed synthetic model (eigenvalues of the covariances log-uniform in [1e-5, 1]):
    # peaked -- one mixture owns every frame
    "synthetic": {"label": "GMMMap fvconvert (BASELINE configs[1])", "M": 64, "lam_lo": 1e-5, "seed": 1002},
    # the same generator with eigenvalues in [1e-1, 1]: overlapping mixtures of similar shape
    "broad": {"label": "GMMMap fvconvert, synthetic model with covariance eigenvalues in [1e-1, 1] (not a BASELINE config)", "M": 64,
              "lam_lo": 1e-1, "seed": 1002},
    # the reference's own trained model (test/models/clb_to_slt_gmm32_order40_diff.jld -> tests/golden/model_*.npz): M = 32
    "fixture": {"label": "GMMMap fvconvert, the reference's trained model clb_to_slt_gmm32_order40_diff (M = 32; not a BASELINE config)",
                "M": 32, "seed": 1002},
    # ... and its other one (test/models/clb_and_slt_gmm32_order40.jld, the joint model test/vc.jl:40-51 converts with); `--workload
    # convert_joint` only, not part of the default run
    "joint": {"label": "GMMMap fvconvert, the reference's trained model clb_and_slt_gmm32_order40 (M = 32; not a BASELINE config)",
              "M": 32, "seed": 1002},
}


def convert_model(variant):
    import synthdata as npo

    v = CONVERT_VARIANTS[variant]
    if variant in ("fixture", "joint"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "model_clb_to_slt_gmm32_order40_diff.npz" if variant == "fixture" else
                                 "model_clb_and_slt_gmm32_order40.npz"))
        return z["weights"], z["means"], z["covars"]
    return npo.synth_model(v["seed"], 80, v["M"], lam_lo=v["lam_lo"])


def pcie_roof(nbytes=320_000_000):
    """What the link gives the library's own staging path: its pinned slots, its upload / download streams, the chunks
    staged_pipeline moves -- without host memcpy or kernels (vcmi_debug_pcie_probe).  GB/s."""
    import ctypes as C

    from voiceconversion_jl_amd import _lib

    out = (C.c_double * 3)()
    fn = _lib.lib.vcmi_debug_pcie_probe
    fn.argtypes, fn.restype = [C.c_size_t, C.POINTER(C.c_double)], C.c_int
    _lib.check(fn(nbytes, out))
    return {"h2d_GBps": out[0], "d2h_GBps": out[1], "duplex_GBps_per_direction": out[2],
            "duplex_ms_for_2x%dMB" % (nbytes // 1_000_000): nbytes / out[2] / 1e6,
            "how": "vcmi_debug_pcie_probe: hipMemcpyAsync of the pipeline's chunks between the ring's pinned slots and HBM on its "
                   "upload / download streams, alone and both at once, best of 4"}


def bench_convert(args, world, rank, variant="synthetic"):
    import torch

    import voiceconversion_jl_amd as vc
    import synthdata as npo

    headline = variant == "synthetic"
    D, T = 40, args.frames
    w, mu, sig = convert_model(variant)
    M = len(w)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    if args.prune is not None:
        g.set_prune(args.prune)
    X = npo.sample_frames(1002 + rank, w, mu, sig, T, 0, D)          # (T,D) == Julia (D,T) image
    Xd = torch.from_numpy(X).cuda()
    Yd = torch.empty_like(Xd)

    def step():
        vc.fvconvert(g, Xd.t(), out=Yd.t())

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)

    clock_warm = dict(CLOCK_WARM)
    frames_per_s = world * T * args.steps / wall
    flops = convert_flops_per_frame(D, M) * T
    tiles = -(-T // 16)
    # What the matrix pipe was actually given: the kernel counts its own v_mfma_f64_16x16x4 instructions (wave-uniform scalar
    # adds; vcmi_gmmmap_convert_plan) in one extra, untimed launch on the same frames; profiles/r04_clock/ holds the same
    # number from the SQ_INSTS_MFMA counter.
    # (profiling runs, --cpu-seconds 0, launch nothing but the warm-up and the timed steps: the PMC passes count there)
    _, shape, active_frac, undecided_frac = g.convert_plan()
    issued_mfma, nreg = None, None
    if not args.profile_run:
        g.prune_stats(True)
        step()
        torch.cuda.synchronize()
        issued_mfma = g.convert_plan()[0]
        nreg = g.prune_stats(False)
    per_pair_dense = 42 if D == 40 else None               # MFMA steps per (16-frame tile, mixture) with nothing skipped, D = 40
    kname = "gmmmap_screen_kernel" if shape == 3 else "gmmmap_mfma_kernel"     # the dominant kernel of this model's loop shape
    alg_tflops = flops / (kernel_ms * 1e-3) / 1e12
    iss_tflops = (issued_mfma * MFMA_FLOP / (kernel_ms * 1e-3) / 1e12) if issued_mfma else None
    achieved = min(alg_tflops, iss_tflops) if iss_tflops else alg_tflops
    out = {
        "metric": "converted frames/sec (D=40, M=64) at 1/2/4/8 MI355X vs CPU ref" if headline else
                  "converted frames/sec (D=40, M=%d), %s model" % (M, variant),
        "value": frames_per_s,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup, "steady": clock_warm,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if variant not in ("fixture", "joint") else "synthetic frames drawn from the reference's trained model",
        "config": {"workload": CONVERT_VARIANTS[variant]["label"], "D": D, "M": M, "frames_per_gpu": T,
                   "sharding": f"frames x{world}, no collective"},
        "roofline": {"bound": "mfma", "kernel": kname + ("<40,2,4>" if shape == 3 else "<40,2,4,0,2,%d>" % shape), "achieved": achieved,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                     "traffic": None, "kernel_ms": kernel_ms,
                     "flop_per_frame": convert_flops_per_frame(D, M),
                     "algorithmic_frac": alg_tflops / FP64_PEAK_TFLOPS,
                     "issued_mfma_frac": (iss_tflops / FP64_PEAK_TFLOPS) if iss_tflops else None,
                     "mfma_issued_per_launch": issued_mfma,
                     "work_skipped": (1.0 - issued_mfma / float(per_pair_dense * tiles * M)) if (per_pair_dense and issued_mfma) else None,
                     "regressions_evaluated_frac": (nreg / float(tiles * M)) if nreg is not None else None,
                     "loop_shape": {0: "dense", 1: "broad", 2: "peaked", 3: "screened"}.get(shape, str(shape)),
                     "model_active_frac": active_frac, "model_undecided_frac": undecided_frac,
                     "hbm_GBps_algorithmic": 2 * D * 8 * T / (kernel_ms * 1e-3) / 1e9},
    }
    if shape == 3:
        # The screened shape proves 63 of the 64 mixtures irrelevant for (nearly) every frame and never evaluates them: the
        # SURVEY 8(d) flop count is not performed (fp64_formulation_frac > 1 says exactly that), and the FP64 roof does not bind a
        # pass that reads x and writes y once.  Priced like the E-step's hard-assignment path: against HBM, with the ALGORITHMIC
        # bytes (2 D 8 per frame: x in, y out) over the step's kernel time.  The FP64 credit of this configuration is `dense`.
        gbs = 2.0 * D * 8 * T / (kernel_ms * 1e-3) / 1e9
        out["roofline"].update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "algorithmic_bytes_per_frame": 2 * D * 8,
                                "fp64_formulation_frac": out["roofline"].pop("algorithmic_frac"),
                                "fp64_formulation_note": "flops of the every-mixture formulation over this step's time; the screen rules "
                                                         "them out without performing them (see `dense` for the FP64 figure)"})
    # HBM traffic of the kernel: PMC passes of this same command, run as child processes before the timed run (LIVE_PMC)
    # or, failing that, the committed passes if they were collected from the same library sources
    if headline:
        attach_traffic(out, "convert_traffic.json", kname, standard=(T == 1_000_000), live=LIVE_PMC.get("convert"),
                       algorithmic_bytes=2.0 * D * 8 * T)
        if out["roofline"].get("traffic") is not None:
            # `traffic` is the dominant kernel's, as the roofline object is defined; the three grouping kernels in front of it read x
            # once more (the nearest-mean keys) and write the permutation
            tot, _ = pmc_traffic("convert_traffic.json", (kname, "gmmmap_group_"), live=LIVE_PMC.get("convert"))
            out["roofline"]["traffic_whole_step"] = tot
        live = LIVE_PMC.get("convert") or {}
        for k, v in live.items():
            if kname in k and "SQ_INSTS_MFMA_per_launch" in v:
                out["roofline"]["SQ_INSTS_MFMA_per_launch"] = v["SQ_INSTS_MFMA_per_launch"]
    # The same K steps with nothing skipped (vcmi_gmmmap_set_prune(inf): every mixture's whitening and regression for every
    # frame, the dense loop the flop count of SURVEY 8(d) assumes), and the pruned output against it.
    if (args.prune is None or args.prune < 1e300) and args.cpu_seconds > 0 and not args.profile_run:
        g.set_prune(float("inf"))
        wall_d, kernel_ms_d = timed_steps(step, args.steps, args.warmup, world)
        ach_d = flops / (kernel_ms_d * 1e-3) / 1e12
        out["roofline"]["dense"] = {"kernel_ms": kernel_ms_d, "frac": ach_d / FP64_PEAK_TFLOPS, "achieved": ach_d,
                                    "value": world * T * args.steps / wall_d, "unit": "frames/s",
                                    "ms_per_step": wall_d / args.steps * 1e3, "steady": dict(CLOCK_WARM),
                                    "note": "vcmi_gmmmap_set_prune(inf): gmmmap_mfma_kernel<40,2,4,0,2,0>, every MFMA step of every "
                                            "mixture for every frame; frac = algorithmic flop / time / peak"}
        if rank == 0:
            Yd_dense = Yd[:4096].clone()
        g.set_prune(46.0 if args.prune is None else args.prune)
        step()
        if rank == 0:
            a, b = Yd[:4096], Yd_dense
            out["roofline"]["max_rel_diff_vs_dense_4096_frames"] = float((torch.linalg.norm(a - b, dim=1) / torch.linalg.norm(b, dim=1)).max())
        PER_RANK["wall_s"] = gather_over_ranks(wall, world)
        PER_RANK["kernel_ms"] = gather_over_ranks(kernel_ms, world)
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
    # (profiling runs pass --cpu-seconds 0: nothing but the warm-up and the timed launches may reach the kernel trace and
    # the PMC passes, so the host-pointer measurement and the strong CPU baseline below are skipped there)
    if rank == 0 and args.cpu_seconds > 0 and headline:
        # the host-pointer (PCIe-inclusive) rate of the same call, measured -- never the reported `value`
        Xh = np.asfortranarray(X.T)
        vc.fvconvert(g, Xh)                                  # warm the library's staging buffers
        keep = []                                            # the results stay alive while the calls are timed: releasing a
        t0 = time.perf_counter()                             # 320 MB array is the caller's cost (munmap: 11-12 ms on these
        for _ in range(3):                                   # boxes, the kernel clears pages when they are freed), timed apart
            keep.append(vc.fvconvert(g, Xh))
        dth = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        while keep:
            keep.pop()
        dfree = (time.perf_counter() - t0) / 3
        Yh = np.empty_like(Xh, order="F")
        vc.fvconvert(g, Xh, out=Yh)
        t0 = time.perf_counter()
        for _ in range(3):
            vc.fvconvert(g, Xh, out=Yh)
        dtr = (time.perf_counter() - t0) / 3
        # a caller that keeps its arrays pins them once (vcmi_host_register): DMA straight from x into y, no staging copy
        try:
            t0 = time.perf_counter()
            vc.pin(Xh), vc.pin(Yh)
            dpin = time.perf_counter() - t0
            Yh[:] = 0.0
            vc.fvconvert(g, Xh, out=Yh)
            reg_equal = bool(np.array_equal(Yh.T, Yd.cpu().numpy()))
            t0 = time.perf_counter()
            for _ in range(3):
                vc.fvconvert(g, Xh, out=Yh)
            dtp = (time.perf_counter() - t0) / 3
            vc.unpin(Yh), vc.unpin(Xh)
            registered = {"value": T / dtp, "ms_per_call": dtp * 1e3, "pinning_both_arrays_once_ms": dpin * 1e3,
                          "parity_vs_device_path": reg_equal}
        except Exception as e:  # noqa: BLE001  (informative; never fail the bench on it)
            registered = {"error": repr(e)}
        try:
            pcie = pcie_roof(Xh.nbytes)
            floor_ms = pcie["duplex_ms_for_2x%dMB" % (Xh.nbytes // 1_000_000)]
        except Exception as e:  # noqa: BLE001
            pcie, floor_ms = {"error": repr(e)}, None
        out["host_inclusive"] = {"value": T / dth, "unit": "frames/s", "ms_per_call": dth * 1e3,
                                 "reused_output": {"value": T / dtr, "ms_per_call": dtr * 1e3,
                                                   "frac_of_pcie": (floor_ms / (dtr * 1e3)) if floor_ms else None},
                                 "frac_of_pcie": (floor_ms / (dth * 1e3)) if floor_ms else None,
                                 "registered": dict(registered, frac_of_pcie=(floor_ms / registered["ms_per_call"])
                                                    if (floor_ms and "ms_per_call" in registered) else None),
                                 "pcie": pcie,
                                 "releasing_one_result_ms": dfree * 1e3,
                                 "note": "vcmi_gmmmap_convert on pageable host arrays (what a Julia ccall passes): chunked "
                                         "pinned staging, H2D / kernel / D2H of consecutive chunks overlapped; `value` "
                                         "allocates a fresh output per call like `similar(X)` (first-touch page faults "
                                         "included; releasing the previous result is timed apart: it is the caller's munmap), "
                                         "`reused_output` writes into an existing array; frac_of_pcie = the time the link needs "
                                         "for the same 2 x 320 MB from PINNED memory, both directions at once, measured in this "
                                         "run (`pcie`), over the call's time",
                                 "parity_vs_device_path": bool(np.array_equal(Yh.T, Yd.cpu().numpy())),
                                 "host": host_facts()}
        # the honest strong CPU baseline as specified -- the same math restructured as blocked FP64 GEMMs
        # (oracle/vc_oracle_gemm.c: whitening and regression of 32-frame blocks per mixture, register-blocked FMA micro-kernel,
        # every mixture for every frame) with OpenMP over all host cores; the per-frame loop under OpenMP beside it
        try:
            ref.fvconvert_gemm(X[:4096])                         # thread pool up, ISA clone resolved
            ns = min(T, 200_000)
            t0 = time.perf_counter()
            Ys, nthr = ref.fvconvert_gemm(X[:ns])
            dts = time.perf_counter() - t0
            if dts < 1.0 and ns < T:                             # a big host: take enough frames for a second of work
                ns = int(min(T, ns * min(10.0, 1.5 / max(dts, 1e-3))))
                t0 = time.perf_counter()
                Ys, nthr = ref.fvconvert_gemm(X[:ns])
                dts = time.perf_counter() - t0
            errs = float(np.max(np.linalg.norm(Yd[:ns].cpu().numpy() - Ys, axis=1) / np.linalg.norm(Ys, axis=1)))
            erro = float(np.max(np.linalg.norm(Ys[:n] - Yref, axis=1) / np.linalg.norm(Yref, axis=1)))
            nm = int(min(T, 8 * n))
            t0 = time.perf_counter()
            _, nthr_pf = ref.fvconvert_mt(X[:nm])
            dtm = time.perf_counter() - t0
            out["cpu_baseline_strong"] = {"value": ns / dts, "unit": "frames/s", "cores": nthr, "kind": "port",
                                          "sample": f"first {ns} frames, GEMM-structured C (32-frame blocks, FMA micro-kernel, every "
                                                    f"mixture evaluated) with OpenMP on {nthr} threads, {dts:.2f} s",
                                          "max_rel_err_vs_gpu": errs, "max_rel_err_vs_per_frame_oracle": erro,
                                          "gflops_per_thread": ns / dts * M * 3 * D * D / nthr / 1e9,
                                          "per_frame_loop_openmp": {"value": nm / dtm, "cores": nthr_pf}}
        except Exception as e:  # noqa: BLE001  (baseline is informative; never fail the bench on it)
            out["cpu_baseline_strong"] = {"error": repr(e)}
    return out


# --------------------------------------------------------------------------------------------- E-step
def bench_estep(args, world, rank, variant="synthetic"):
    """BASELINE configs[2]: diagonal E-step, Dj=80, M=128, N=10^7 frames over 8 GPUs -> 1.25e6 frames per GPU
    (weak scaling), followed by ONE all-reduce of the packed statistics over RCCL.
    variant "fixture": the same call on frames that SHARE their mixtures -- drawn from the reference's trained joint model
    (test/models/clb_and_slt_gmm32_order40.jld, full covariances), E-step of the diagonal model with its means and variances
    (M = 32): what a train_gmm run on real joint mel-cepstra looks like to the kernel."""
    import torch

    import voiceconversion_jl_amd as vc
    import synthdata as npo

    Dj, M, N = args.dj, args.mixtures, args.frames if args.frames != 1_000_000 else 1_250_000
    if variant == "fixture":
        z = np.load(os.path.join(ROOT, "tests", "golden", "model_clb_and_slt_gmm32_order40.npz"))
        w, mu, sig = z["weights"] / z["weights"].sum(), np.ascontiguousarray(z["means"]), z["covars"]
        M, Dj = mu.shape
        var = np.ascontiguousarray(np.stack([np.diag(sig[m]) for m in range(M)]))
        chol = np.linalg.cholesky(sig)

        def shard_frames(r):                         # rank r's frames, drawn from the FULL-covariance model
            rg = np.random.default_rng(2003 + r)
            comp = np.sort(rg.choice(M, size=N, p=w))
            X = np.empty((N, Dj))
            lo = 0
            for m in range(M):
                n = int(np.searchsorted(comp, m, side="right")) - lo
                X[lo:lo + n] = mu[m] + rg.standard_normal((n, Dj)) @ chol[m].T
                lo += n
            return X[rg.permutation(N)]
    else:
        w, mu, _ = npo.synth_model(1003, Dj, M)
        var = np.exp(np.random.default_rng(1003).uniform(np.log(1e-3), 0.0, (M, Dj)))   # the model: same on every rank

        def shard_frames(r):                         # rank r's frames, drawn from the model
            rg = np.random.default_rng(2003 + r)
            comp = rg.choice(M, size=N, p=w)
            return mu[comp] + rg.standard_normal((N, Dj)) * np.sqrt(var[comp])

    X = shard_frames(rank)
    Xd = torch.from_numpy(X).cuda()
    out_t = torch.empty(vc.stats_len(Dj, M), dtype=torch.float64, device="cuda")
    muT, varT = np.asfortranarray(mu.T), np.asfortranarray(var.T)

    def step():
        vc.estep_diag_dev(Xd.t(), w, muT, varT, out=out_t)
        split_mark()
        vc.dist.allreduce_sum_(out_t)

    def step_kernels():
        vc.estep_diag_dev(Xd.t(), w, muT, varT, out=out_t)

    # The very first E-step of this process on these frames: model upload, scratch allocation, the path decision from the call's
    # own frames (csrc/estep_path.hpp) -- what a caller pays before anything is warm (nothing is learnt from
    # earlier calls any more, so the warm-up steps below can no longer hide a path that mis-fires on its first calls)
    # (profiling runs launch nothing but the warm-up and the timed steps: the PMC passes count launches per step)
    cold_first_call_ms = second_call_ms = None
    if not args.profile_run:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_kernels()
        torch.cuda.synchronize()
        cold_first_call_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        step_kernels()
        torch.cuda.synchronize()
        second_call_ms = (time.perf_counter() - t0) * 1e3
    # one set of timed steps prices both parts: `kernel_ms` = start of the step to the mark (the E-step kernels),
    # `allreduce_ms` = the rest (the RCCL all-reduce of the packed statistics; no work with one rank and no process group)
    wall, step_ms = timed_steps(step, args.steps, args.warmup, world, split=True)
    clock_warm = dict(CLOCK_WARM)
    kernel_ms = PER_RANK.pop("first_part_ms")
    allreduce_ms = step_ms - kernel_ms
    fps = world * N * args.steps / wall
    mfma_path = Dj % 2 == 0 and Dj <= 160 and M <= 128           # estep.hip: estep_device
    # what the matrix pipe was given: the kernel counts its own MFMAs in one extra, untimed step (vcmi_debug_estep_mfma) --
    # step B skips k-steps whose responsibilities are all exactly zero, so the algorithmic flop count is not what is issued
    issued_mfma = None
    try:
        if args.profile_run:
            raise RuntimeError("profiling run: nothing but the warm-up and the timed steps is launched")
        import ctypes as C

        from voiceconversion_jl_amd import _lib

        fn = _lib.lib.vcmi_debug_estep_mfma
        fn.argtypes, fn.restype = [C.c_int, C.POINTER(C.c_int64)], C.c_int
        _lib.check(fn(1, None))
        step_kernels()
        torch.cuda.synchronize()
        cnt = C.c_int64(0)
        _lib.check(fn(0, C.byref(cnt)))
        issued_mfma = int(cnt.value) or None
        step_kernels()
        vc.dist.allreduce_sum_(out_t)
    except Exception:  # noqa: BLE001  (an older library: the line keeps its algorithmic figure, labelled)
        issued_mfma = None
    # the hard-assignment path (csrc/estep_hard.hpp): how many frames of the last step still went through the FP64 kernel
    # (-1: the one-kernel path ran) -- asked once, after the timed steps
    try:
        from voiceconversion_jl_amd import _lib as _l

        soft = _l.estep_last_soft()
    except Exception:  # noqa: BLE001  (an older library)
        soft = -1
    alg_tflops = estep_flops_per_frame(Dj, M) * N / (kernel_ms * 1e-3) / 1e12
    # the FP64 kernel of the one-kernel path: M <= 32 (Dj <= 80) runs in workgroups as small as the model (csrc/estep_small.hpp)
    dj_soft = min(d for d in (32, 48, 64, 80, 112, 160) if d >= Dj) if Dj <= 160 else Dj
    small = mfma_path and M <= 32 and dj_soft <= 80 and not (args.debug_force & 1024)
    soft_kernel = f"estep_small_kernel<{dj_soft}, {1 if M <= 16 else 2}>" if small else f"estep_mfma_kernel<{dj_soft}>"
    iss_tflops = issued_mfma * MFMA_FLOP / (kernel_ms * 1e-3) / 1e12 if issued_mfma else None
    achieved = min(alg_tflops, iss_tflops) if iss_tflops else alg_tflops
    out = {"metric": "diag-GMM E-step frames/sec (Dj=%d, M=%d)" % (Dj, M), "value": fps, "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "steady": clock_warm, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "diag E-step on frames drawn from the reference's trained model clb_and_slt_gmm32_order40 (M = 32; not a BASELINE config)"
                      if variant == "fixture" else "diag E-step (BASELINE configs[2])" if (Dj == 80 and M == 128) else
                      f"diag E-step, Dj={Dj}, M={M} ({'MFMA kernel' if mfma_path else 'generic kernels'}; not a BASELINE config)",
                      "Dj": Dj, "M": M, "frames_per_gpu": N,
                      "collective": "all-reduce(sum) of %d doubles per step" % vc.stats_len(Dj, M)},
           "roofline": {"bound": "mfma", "kernel": (soft_kernel if mfma_path else "estep_gamma_kernel + estep_stats_kernel (generic path)") + " (+ all-reduce)", "achieved": achieved,
                        "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                        "traffic": None,
                        "algorithmic_frac": alg_tflops / FP64_PEAK_TFLOPS,
                        "issued_mfma_frac": (iss_tflops / FP64_PEAK_TFLOPS) if iss_tflops else None,
                        "mfma_issued_per_step": issued_mfma,
                        "flop_per_frame": estep_flops_per_frame(Dj, M), "kernel_ms": kernel_ms},
           "collective": {"op": "all-reduce(sum), %d doubles" % vc.stats_len(Dj, M), "allreduce_ms": allreduce_ms,
                          "step_ms_with_allreduce": step_ms, "ranks": world, "backend": BACKEND["name"]}}
    prefixes = ("estep_small_kernel", "estep_mfma_kernel")
    if soft >= 0:
        # Frames one mixture owns never reach the FP64 pipe: what is left is reading X (once algorithmically; this path reads
        # it twice -- the screen, then the sums over the sorted rows) -- an HBM-bound job.  achieved = 8 Dj N bytes over the
        # E-step kernels' time; the FP64-formulation figure (flops of the one-kernel formulation / this time) is kept beside
        # it, labelled: those flops are not performed.
        gbs = 8.0 * Dj * N / (kernel_ms * 1e-3) / 1e9
        dj_inst = min(d for d in (32, 48, 64, 80) if d >= Dj)
        out["roofline"] = {"bound": "hbm", "kernel": f"estep_hard_key_kernel<{dj_inst}> + sort + estep_hard_stats_kernel<{dj_inst}> (+ path decision; + all-reduce)",
                           "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                           "kernel_ms": kernel_ms, "soft_frames": soft, "soft_frac": soft / N,
                           "fp64_formulation_tflops": alg_tflops,
                           "fp64_formulation_note": "flops of the one-kernel formulation over this step's time; owned frames never reach the FP64 pipe",
                           "mfma_issued_per_step": issued_mfma, "algorithmic_bytes_per_frame": 8 * Dj}
        prefixes = ("estep_hard", "estep_path", "gmmmap_group_sc", "estep_mfma_kernel", "estep_small_kernel")
    out["config"]["soft_frames_of_last_step"] = soft          # -1: the one-kernel path ran (the sample of the call's frames found few owners)
    out["cold_first_call_ms"] = cold_first_call_ms
    out["second_call_ms"] = second_call_ms
    attach_traffic(out, "estep_fixture_traffic.json" if variant == "fixture" else "estep_traffic.json", prefixes,
                   standard=(N == 1_250_000 and Dj == 80 and M == (32 if variant == "fixture" else 128)),
                   live=LIVE_PMC.get("estep_fixture" if variant == "fixture" else "estep"), algorithmic_bytes=8.0 * Dj * N)
    if rank == 0 and not args.profile_run:       # (the parity sample launches the kernel once more)
        from oracle import c_oracle as co

        n0 = 20000
        t0 = time.perf_counter()
        co.estep_diag(X[:n0], w, mu, var)
        dt0 = time.perf_counter() - t0
        n = int(min(N, max(n0, args.cpu_seconds / (dt0 / n0))))
        t0 = time.perf_counter()
        r0, r1, r2, rl = co.estep_diag(X[:n], w, mu, var)
        dt = time.perf_counter() - t0
        S0, S1, S2, ll = vc.estep_diag(X[:n].T, w, muT, varT)
        out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {n} frames, C oracle, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        out["parity_max_rel_err_vs_oracle"] = float(np.max(np.abs(S1 - r1.T)) / np.max(np.abs(r1)))
        if args.verify_allreduce:
            # the statistics every rank now holds (after the all-reduce) against ONE process over all ranks' frames
            single = torch.empty_like(out_t)
            allX = np.concatenate([X] + [shard_frames(r) for r in range(1, world)])
            vc.estep_diag_dev(torch.from_numpy(allX).cuda().t(), w, muT, varT, out=single)
            a, b = out_t.cpu().numpy(), single.cpu().numpy()
            out["allreduce_check"] = {"max_rel_err_vs_single_process": float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))),
                                      "frames_total": world * N}
    return out


def full_estep_issued_mfma(step_fn, N, Dj, M):
    """v_mfma_f64_16x16x4 instructions of one full-covariance E-step: the log-density kernel's are a fixed number (two frame
    tiles per wave, four waves per workgroup, every whitening tile of every mixture: U-only tiling of csrc/gmmmap.hip), the
    statistics kernel counts its own (vcmi_debug_estep_full_mfma).  None when the library has no counter / another path."""
    import ctypes as C

    import torch

    from voiceconversion_jl_amd import _lib

    if Dj > 80 or Dj % 4:
        return None
    try:
        fn = _lib.lib.vcmi_debug_estep_full_mfma
        fn.argtypes, fn.restype = [C.c_int, C.POINTER(C.c_int64)], C.c_int
        _lib.check(fn(1, None))
        step_fn()
        torch.cuda.synchronize()
        cnt = C.c_int64(0)
        _lib.check(fn(0, C.byref(cnt)))
    except Exception:  # noqa: BLE001
        return None
    DP, KS = Dj, Dj // 4
    nu = -(-DP // 16)
    steps_u = sum((min(4 * (t + 1), KS) if 16 * t + 15 < DP else KS) for t in range(nu))
    logdens = 2 * steps_u * M * 4 * (-(-N // 128))
    return logdens + int(cnt.value)


def bench_estep_full(args, world, rank, variant="synthetic"):
    """full-covariance E-step (what bin/train_gmm.jl:84-103 runs), Dj=80, M=64, 5e5 frames per
    GPU (weak scaling) + ONE all-reduce of the packed statistics.  Algorithmic flops per frame: triangular
    whitening M*Dj*(Dj+1) + symmetric-half second moments M*Dj*(Dj+1) + first moments 2*M*Dj.
    variant "fixture": the reference's trained joint model (M = 32) and frames drawn from it -- frames that share their
    mixtures, as in a train_gmm run on real joint mel-cepstra."""
    import torch

    import voiceconversion_jl_amd as vc
    import synthdata as npo

    Dj, M, N = args.dj, 64, args.frames if args.frames != 1_000_000 else 500_000
    if variant == "fixture":
        z = np.load(os.path.join(ROOT, "tests", "golden", "model_clb_and_slt_gmm32_order40.npz"))
        w, mu, sig = z["weights"] / z["weights"].sum(), np.ascontiguousarray(z["means"]), np.ascontiguousarray(z["covars"])
        M, Dj = mu.shape
    else:
        w, mu, sig = npo.synth_model(1005, Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(1005 + rank, w, mu, sig, N, 0, Dj)
    Xd = torch.from_numpy(X).cuda()
    out_t = torch.empty(vc.full_stats_len(Dj, M), dtype=torch.float64, device="cuda")
    muT, sgT = np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))

    def step():
        vc.estep_full_dev(Xd.t(), w, muT, sgT, out=out_t)
        vc.dist.allreduce_sum_(out_t)

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)

    clock_warm = dict(CLOCK_WARM)
    flop = 2 * M * Dj * (Dj + 1) + 2 * M * Dj
    alg_tflops = flop * N / (kernel_ms * 1e-3) / 1e12
    issued = full_estep_issued_mfma(step, N, Dj, M) if not args.profile_run else None
    iss_tflops = issued * MFMA_FLOP / (kernel_ms * 1e-3) / 1e12 if issued else None
    achieved = min(alg_tflops, iss_tflops) if iss_tflops else alg_tflops
    out = {"metric": "full-covariance GMM E-step frames/sec (Dj=%d, M=%d)" % (Dj, M), "value": world * N * args.steps / wall,
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "steady": clock_warm,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "full-covariance E-step, the reference's trained model clb_and_slt_gmm32_order40 and frames drawn from it (M = 32)"
                      if variant == "fixture" else "full-covariance E-step (SURVEY 8f rank 1; bin/train_gmm.jl:84-103)", "Dj": Dj, "M": M,
                      "frames_per_gpu": N,
                      "collective": "all-reduce(sum) of %d doubles per step" % vc.full_stats_len(Dj, M)},
           "roofline": {"bound": "mfma", "kernel": ("whole step: gmmmap_mfma_kernel<MODE 1> + estep_full_stats_kernel<%d,1>" % Dj) if Dj <= 80 else
                                  ("whole step: logdens_tiled_kernel + estep_full_stats_kernel<%d,4> (+ host Cholesky of the %d-dim blocks)" % (Dj, Dj)),
                        "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP64_PEAK_TFLOPS,
                        "algorithmic_frac": alg_tflops / FP64_PEAK_TFLOPS,
                        "issued_mfma_frac": (iss_tflops / FP64_PEAK_TFLOPS) if iss_tflops else None, "mfma_issued_per_step": issued,
                        "traffic": None,
                        "flop_per_frame": flop, "kernel_ms": kernel_ms}}
    attach_traffic(out, "estep_full_fixture_traffic.json" if variant == "fixture" else "estep_full_traffic.json",
                   ("gmmmap_mfma_kernel", "estep_full_stats_kernel", "estep_full_softmax_kernel"), standard=(N == 500_000 and Dj == 80))
    if rank == 0 and not args.profile_run:       # (the parity sample launches the kernel once more)
        from oracle import c_oracle as co

        n = 40000 if Dj <= 80 else 10000
        t0 = time.perf_counter()
        r0, r1, r2, rl = co.estep_full(X[:n], w, mu, sig)
        dt = time.perf_counter() - t0
        S0, S1, S2, ll = vc.estep_full(X[:n].T, w, muT, sgT)
        out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {n} frames, C oracle, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        out["parity_max_rel_err_vs_oracle"] = float(np.max(np.abs(S2 - np.transpose(r2, (2, 1, 0)))) / np.max(np.abs(r2)))
    return out


def bench_em_full(args, world, rank):
    """One whole EM iteration of the full-covariance GMM as bin/train_gmm.jl:84-103 trains it (Dj=80, M=64, 5e5 frames
    per GPU): E-step statistics -> ONE all-reduce -> M-step + Cholesky whitening of every mixture, parameters resident
    in HBM throughout (vcmi_gmm_em_*)."""
    import torch

    import voiceconversion_jl_amd as vc
    import synthdata as npo

    Dj, M, N = args.dj, 64, args.frames if args.frames != 1_000_000 else 500_000
    w, mu, sig = npo.synth_model(1005, Dj, M, lam_lo=1e-3)
    X = npo.sample_frames(1005 + rank, w, mu, sig, N, 0, Dj)
    Xd = torch.from_numpy(X).cuda()
    muT, sgT = np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0)))
    em = vc.EMState(w, muT, sgT, min_covar=1e-7)
    stats = torch.empty(vc.full_stats_len(Dj, M), dtype=torch.float64, device="cuda")
    hist = []

    def step():
        em.estep(Xd.t(), out=stats)
        vc.dist.allreduce_sum_(stats)
        hist.append(em.mstep(stats))

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)

    clock_warm = dict(CLOCK_WARM)
    flop = 2 * M * Dj * (Dj + 1) + 2 * M * Dj
    alg_tflops = flop * N / (kernel_ms * 1e-3) / 1e12
    issued = full_estep_issued_mfma(lambda: em.estep(Xd.t(), out=stats), N, Dj, M) if not args.profile_run else None
    iss_tflops = issued * MFMA_FLOP / (kernel_ms * 1e-3) / 1e12 if issued else None
    achieved = min(alg_tflops, iss_tflops) if iss_tflops else alg_tflops
    out = {"metric": "full-covariance GMM EM iteration frames/sec (Dj=%d, M=64)" % Dj, "value": world * N * args.steps / wall,
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "steady": clock_warm,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "EM iteration, full covariance (SURVEY 8f rank 1; bin/train_gmm.jl:84-103)", "Dj": Dj, "M": M,
                      "frames_per_gpu": N, "collective": "all-reduce(sum) of %d doubles per iteration" % vc.full_stats_len(Dj, M)},
           "roofline": {"bound": "mfma", "kernel": "whole iteration: log-densities + second moments + M-step + whitening",
                        "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP64_PEAK_TFLOPS,
                        "algorithmic_frac": alg_tflops / FP64_PEAK_TFLOPS,
                        "issued_mfma_frac": (iss_tflops / FP64_PEAK_TFLOPS) if iss_tflops else None, "mfma_issued_per_step": issued,
                        "traffic": None,
                        "flop_per_frame": flop, "kernel_ms": kernel_ms},
           "loglik_monotone": bool(all(b >= a - 1e-6 * abs(a) for a, b in zip(hist, hist[1:])))}
    attach_traffic(out, "em_full_traffic.json", ("gmmmap_mfma_kernel", "estep_full_stats_kernel", "estep_full_softmax_kernel"),
                   standard=(N == 500_000 and Dj == 80))
    if rank == 0 and args.cpu_seconds > 0:
        from oracle import c_oracle as co

        n = 40000 if Dj <= 80 else 10000          # the E-step is the iteration (the M-step is O(M Dj^2))
        t0 = time.perf_counter()
        co.estep_full(X[:n], w, mu, sig)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"E-step of the first {n} frames, C oracle, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
    return out


# ------------------------------------------------------------------------------------------------ DTW
def _dtw_pairs(seed, n, D):
    rng = np.random.default_rng(seed)
    pairs = []
    for _ in range(n):
        S, T = int(rng.integers(450, 551)), int(rng.integers(450, 551))
        t = rng.standard_normal((S, D))
        idx = np.clip(np.sort(rng.integers(0, S, T)), 0, S - 1)
        pairs.append((t, t[idx] + 0.2 * rng.standard_normal((T, D))))
    return pairs


def bench_dtw(args, world, rank):
    """BASELINE configs[3]: DTW alignment of ~500x500-frame pairs, D=40, bstep=2/fstep=0 (what align uses);
    `--pairs` pairs per GPU (weak scaling, pairs independent, no collective); path-only mode, device-resident."""
    import ctypes as C

    import torch

    import voiceconversion_jl_amd as vc
    from voiceconversion_jl_amd import _lib

    D, n = args.dim, args.pairs
    pairs = _dtw_pairs(1004 + rank, n, D)
    feats, toff, soff, poff, S, T = [], [], [], [], [], []
    fo = po = 0
    for t, s in pairs:
        toff.append(fo); feats.append(t.ravel()); fo += t.size
        soff.append(fo); feats.append(s.ravel()); fo += s.size
        poff.append(po); po += s.shape[0]
        S.append(t.shape[0]); T.append(s.shape[0])
    fd = torch.from_numpy(np.concatenate(feats)).cuda()
    pd = torch.empty(po, dtype=torch.int64, device="cuda")
    arr = lambda a: np.asarray(a, dtype=np.int64)  # noqa: E731
    toff, soff, poff, S, T = arr(toff), arr(soff), arr(poff), arr(S), arr(T)

    def step():
        _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(n, fd.data_ptr(), _lib.iptr(toff), _lib.iptr(S), _lib.iptr(soff), _lib.iptr(T),
                                                   D, 0, 2, pd.data_ptr(), _lib.iptr(poff), torch.cuda.current_stream().cuda_stream))

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)

    clock_warm = dict(CLOCK_WARM)
    cells = float(np.sum(S * T))
    flops = cells * (3 * D + 10)
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    out = {"metric": "DTW aligned pairs/sec (~500x500 frames, D=%d)" % D, "value": world * n * args.steps / wall, "unit": "pairs/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "steady": clock_warm, "ms_per_step": wall / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "DTW fit!+backward, path-only (BASELINE configs[3])" if D == 40 else
                      f"DTW fit!+backward, path-only, D={D}" + (" (order-40 mel-cepstra with c0: bin/mcep.jl:12, src/align.jl:45)" if D == 41 else ""),
                      "D": D, "pairs_per_gpu": n,
                      "fstep": 0, "bstep": 2},
           "roofline": {"bound": "valu", "kernel": ("dtw_fused_%skernel<%d,2> (+ dtw_fused_backward_kernel)" % ("persistent_" if D <= 41 and n > 512 else "", 41 if D == 41 else -(-D // 8) * 8)) if D <= 48 else
                        "dtw_obs_asm_kernel + dtw_rec_kernel (D > 48: observation matrix through HBM)",
                        "achieved": achieved, "peak": FP64_PEAK_TFLOPS / 2, "unit": "TFLOP/s",
                        "frac": achieved / (FP64_PEAK_TFLOPS / 2),
                        "traffic": None,
                        "algorithmic_bytes": float(np.sum((S + T) * D * 8 + T * 8)),
                        "note": "bound: the FP64 VECTOR pipe, not MFMA and not HBM -- the bit-exact contract (src/dtw.jl:33-35: "
                                "sequential in d, separately rounded multiply and add) forbids fused multiply-add and any "
                                "GEMM form, so the roof is one flop per lane-instruction = half the FMA/MFMA figure; "
                                "`achieved` prices the whole step (forward + backward kernels) at 3*D+10 flop per cell",
                        "cells_per_s": cells / (kernel_ms * 1e-3), "kernel_ms": kernel_ms}}
    attach_traffic(out, f"dtw{'' if D == 40 else '_d%d' % D}_traffic.json", "dtw_fused",
                   fetch_factor={"dtw_fused_persistent_kernel": DTW_FUSED_FETCH_FACTOR, "dtw_fused_kernel": DTW_FUSED_FETCH_FACTOR, "": VECTOR_FETCH_FACTOR},
                   standard=(n == 1000), live=LIVE_PMC.get("dtw"), algorithmic_bytes=float(np.sum((S + T) * D * 8 + T * 8)))
    if rank == 0:
        from oracle import c_oracle as co

        t0 = time.perf_counter()
        refs = [co.dtw_fit(t, s, 0, 2, tables=False) for t, s in pairs[:min(n, 40)]]
        dt = time.perf_counter() - t0
        k = int(min(n, max(40, args.cpu_seconds / (dt / len(refs)))))
        if k > len(refs):
            t0 = time.perf_counter()
            refs = [co.dtw_fit(t, s, 0, 2, tables=False) for t, s in pairs[:k]]
            dt = time.perf_counter() - t0
        k = len(refs)
        got = pd.cpu().numpy()
        ok = all(np.array_equal(got[poff[i]:poff[i] + T[i]], refs[i]) for i in range(k))
        out["cpu_baseline"] = {"value": k / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"first {k} pairs, C oracle, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        out["parity_bit_exact_vs_oracle"] = bool(ok)
    return out


# ----------------------------------------------------------------------------------------- trajectory
def bench_traj(args, world, rank, gv=False):
    """BASELINE configs[4]: TrajectoryGMMMap, static D=40 (X dim 80), M=64, T=2000 per utterance, `--utts`
    utterances per GPU (weak scaling, utterance-parallel, no collective); device-resident.
    gv=True: TrajectoryGVGMMMap: the same solve followed by 100 epochs of GV gradient ascent."""
    import torch

    import voiceconversion_jl_amd as vc
    import synthdata as npo
    from voiceconversion_jl_amd import _lib

    D, M, T, n = 40, 64, 2000, args.utts
    w, mu, sig = npo.synth_model(1005, 4 * D, M, lam_lo=1e-3)
    g = vc.GMMMap(*julia_model(w, mu, sig))
    tj = vc.TrajectoryGMMMap(g, T)
    rng = np.random.default_rng(1005 + rank)
    base = []
    for _ in range(min(n, 8)):
        st = npo.sample_frames(int(rng.integers(1 << 30)), w, mu, sig, T, 0, D)
        st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
        base.append(np.ascontiguousarray(vc.push_delta(np.asfortranarray(st.T)).T))   # (T,2D), src/datasets.jl:6-13
    X = np.concatenate([base[i % len(base)] for i in range(n)])                  # (n*T, 2D)
    Xd = torch.from_numpy(X).cuda()
    Yd = torch.empty((n * T, D), dtype=torch.float64, device="cuda")
    xoff = np.arange(n, dtype=np.int64) * T * 2 * D
    yoff = np.arange(n, dtype=np.int64) * T * D
    Ts = np.full(n, T, dtype=np.int64)
    L = int(args.chunk)
    if L > 0 and not gv:
        # the reference CLI's regime: vc(c::TrajectoryConverter, fm) converts every utterance in independent chunks of
        # length(c) = --T frames (bin/vc.jl:18 default 100; src/common.jl:42-57) -> n * ceil(T/L) short solves per batch
        starts = [(u, b) for u in range(n) for b in range(0, T, L)]
        xoff = np.array([(u * T + b) * 2 * D for u, b in starts], dtype=np.int64)
        yoff = np.array([(u * T + b) * D for u, b in starts], dtype=np.int64)
        Ts = np.array([min(L, T - b) for _, b in starts], dtype=np.int64)
    nsolve = len(Ts)

    epochs, alpha = 100, 1.0e-5
    if gv:
        y0 = np.ascontiguousarray(vc.fvconvert(tj, np.asfortranarray(base[0][:200].T)).T)     # target GV statistics from a short conversion
        muv = y0.var(axis=0, ddof=1) * 1.3
        Ar = np.random.default_rng(7).standard_normal((D, D))
        Sv = Ar @ Ar.T / D * np.mean(muv) ** 2 * 0.1 + np.diag(muv ** 2 * 0.05)
        tgv = vc.TrajectoryGVGMMMap(tj, muv, Sv)

    def step():
        if gv:
            _lib.check(_lib.lib.vcmi_trajgv_convert_batch_dev(tgv._h, n, Xd.data_ptr(), _lib.iptr(xoff), _lib.iptr(Ts), epochs,
                                                              alpha, Yd.data_ptr(), _lib.iptr(yoff),
                                                              torch.cuda.current_stream().cuda_stream))
        else:
            _lib.check(_lib.lib.vcmi_traj_convert_batch_dev(tj._h, nsolve, Xd.data_ptr(), _lib.iptr(xoff), _lib.iptr(Ts), Yd.data_ptr(),
                                                            _lib.iptr(yoff), torch.cuda.current_stream().cuda_stream))

    wall, kernel_ms = timed_steps(step, args.steps, args.warmup, world)

    clock_warm = dict(CLOCK_WARM)
    flops_per_utt = 2.0e9 + (epochs * T * 2.0 * (2 * D) ** 2 if gv else 0.0)
    achieved = flops_per_utt * n / (kernel_ms * 1e-3) / 1e12
    out = {"metric": ("trajectory+GV-converted" if gv else "trajectory-converted") + " frames/sec (static D=40, M=64, T=2000)",
           "value": world * n * T * args.steps / wall,
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "steady": clock_warm,
           "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": ("TrajectoryGVGMMMap fvconvert, 100 epochs (SURVEY 8f rank 2)" if gv else
                                   "TrajectoryGMMMap fvconvert (BASELINE configs[4])" if L <= 0 else
                                   f"vc(TrajectoryGMMMap) in chunks of {L} frames (bin/vc.jl:18, src/common.jl:42-57)"),
                      "static_D": D, "M": M, "T": T, "utterances_per_gpu": n, "solves_per_step": nsolve,
                      "frames_per_solve": int(Ts.max())},
           "roofline": {"bound": "mfma", "kernel": "predict + traj_g_mfma_kernel + traj_solve_blk_kernel<40> + traj_backsub_blk_kernel<40>" + (" + traj_gv2_kernel" if gv else ""),
                        "achieved": achieved,
                        "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                        "traffic": None,
                        "flop_per_utterance": flops_per_utt, "kernel_ms": kernel_ms,
                        "note": "whole pipeline (4 kernels); the banded solve is a sequential block recurrence along each "
                                "(sub-)sequence (latency-bound, see DESIGN 3.4: factorisation, then the back substitution as its "
                                "own kernel); chunked conversion has the same number of block steps per CU, in shorter chains"}}
    attach_traffic(out, ("trajgv" if gv else "traj") + "_traffic.json",
                   ("gmmmap_mfma_kernel", "posterior_finish_kernel", "traj_g_mfma_kernel", "traj_solve_blk_kernel", "traj_backsub_blk_kernel",
                    "traj_gv_kernel", "traj_gv2_kernel"), standard=(n == 256 and L <= 0), live=LIVE_PMC.get("trajgv" if gv else "traj"),
                   algorithmic_bytes=8.0 * n * T * 3 * D)
    if rank == 0:
        from oracle import c_oracle as co

        ref = co.TrajectoryGMMMap(co.GMMMap(w, mu, sig))

        def convert_ref(Xu):
            if gv:
                return ref.fvconvert_gv(Xu, muv, Sv, epochs, alpha)
            if L > 0:     # chunk by chunk, as the reference's vc loop does
                return np.concatenate([ref.fvconvert(Xu[b0:b0 + L])[0] for b0 in range(0, T, L)])
            return ref.fvconvert(Xu)[0]

        # utterance 0, then as many more of the distinct utterances as the CPU budget allows; every one is a parity check
        nu, dt, err = 0, 0.0, 0.0
        Yall = Yd[:len(base) * T].cpu().numpy()
        while nu < min(n, len(base)) and (nu == 0 or dt + dt / nu <= args.cpu_seconds):
            t0 = time.perf_counter()
            Yref = convert_ref(base[nu])
            dt += time.perf_counter() - t0
            err = max(err, float(np.max(np.abs(Yall[nu * T:(nu + 1) * T] - Yref)) / np.max(np.abs(Yref))))
            nu += 1
        out["cpu_baseline"] = {"value": nu * T / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{nu} utterance(s) of {T} frames, C oracle, {dt:.1f} s on 1 of {os.cpu_count()} host cores"}
        out["parity_max_rel_err_vs_oracle"] = err
    return out


def bench_selftest(args, world, rank):
    """Launch plumbing only (runs without a GPU): rendezvous, barrier-bracketed timed region with a no-op step, the
    max-over-ranks reduction and one all-reduce(sum) of a rank-dependent statistics-sized vector, checked exactly.  Lets
    a CPU box verify that `bench.py --gpus N` starts N ranks and reports n_gpus = N (tests/test_bench_launch.py)."""
    import torch
    import torch.distributed as dist

    if os.environ.get("VCMI_SELFTEST_DIE_RANK") == str(rank):     # fault injection for tests/test_bench_launch.py
        sys.exit(3)
    n = 20_609                                       # the diag E-step payload at Dj=80, M=128
    v = torch.arange(n, dtype=torch.float64, device=_dev()) * (rank + 1)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    barrier_sync(world)
    wall = max_over_ranks(time.perf_counter() - t0, world)
    if world > 1:
        dist.all_reduce(v)
    expect = torch.arange(n, dtype=torch.float64, device=_dev()) * (world * (world + 1) // 2)
    ok = bool(torch.equal(v, expect))
    PER_RANK["wall_s"] = gather_over_ranks(wall, world)
    return {"metric": "launch self-test (no kernels)", "value": 0.0, "unit": "n/a", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "selftest", "sharding": f"one rank per GPU x{world}; frames / pairs / utterances split by rank, E-step statistics all-reduced"},
            "collective": {"op": "all-reduce(sum), %d doubles" % n, "ranks": world, "backend": BACKEND["name"]},
            "allreduce_exact": ok}


def summarize(out):
    """What the `workloads` table keeps of a workload's line."""
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "steady", "config", "data", "roofline", "cpu_baseline", "collective", "cold_first_call_ms", "second_call_ms",
            "parity_max_rel_err_vs_oracle", "parity_bit_exact_vs_oracle", "speedup_vs_cpu_baseline", "allreduce_check")
    d = {k: out[k] for k in keep if k in out}
    d["kernel_ms"] = out.get("roofline", {}).get("kernel_ms")
    d["per_rank"] = dict(PER_RANK)
    if "cpu_baseline" in d and "value" in d["cpu_baseline"] and "speedup_vs_cpu_baseline" not in d:
        d["speedup_vs_cpu_baseline"] = out["value"] / d["cpu_baseline"]["value"]
    return d


LINE_CAP = 7500           # characters of the ONE stdout line (the driver keeps an 8 KB tail; BENCH_r04 lost a 20.9 KB line)
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_frac", "issued_mfma_frac",
             "traffic_whole_step", "traffic_x_algorithmic", "traffic_GBps", "hbm_frac", "soft_frac", "fp64_formulation_tflops",
             "fp64_formulation_frac")


def _sig(x, n=6):
    """Floats to n significant digits (the line is a record, not an archive: bench_detail has the full values)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, x))


def compact_roofline(r, kernel_chars=80):
    c = {k: _sig(r.get(k)) for k in ROOF_KEYS if k in r or k == "traffic"}
    if isinstance(c.get("kernel"), str):
        c["kernel"] = c["kernel"][:kernel_chars]
    d = r.get("dense")
    if isinstance(d, dict):
        c["dense"] = {"kernel_ms": _sig(d.get("kernel_ms")), "frac": _sig(d.get("frac"))}
    return c


def compact_workload(d):
    """One row of the line's `summary`: numbers only (definitions: profiles/README.md)."""
    c = {k: _sig(d[k]) for k in ("value", "unit", "ms_per_step") if k in d}
    st = d.get("steady") or {}
    if st.get("steady_ms_per_step") is not None:
        c["steady_ms_per_step"] = st["steady_ms_per_step"]
    c["roofline"] = compact_roofline(d.get("roofline", {}), 48)
    if "parity_max_rel_err_vs_oracle" in d:
        c["parity"] = _sig(d["parity_max_rel_err_vs_oracle"], 3)
    elif "parity_bit_exact_vs_oracle" in d:
        c["parity"] = "bit-exact" if d["parity_bit_exact_vs_oracle"] else "MISMATCH"
    b = d.get("cpu_baseline")
    if isinstance(b, dict) and "value" in b:
        c["cpu_baseline"] = {k: _sig(b[k]) for k in ("value", "unit", "cores", "kind") if k in b}
        if "cached" in b:
            c["cpu_baseline"]["cached"] = True
    if d.get("speedup_vs_cpu_baseline") is not None:
        c["speedup_vs_cpu_baseline"] = _sig(d["speedup_vs_cpu_baseline"], 4)
    col = d.get("collective")
    if isinstance(col, dict):
        c["collective"] = {k: _sig(col.get(k)) for k in ("allreduce_ms", "ranks", "backend")}
    if d.get("cold_first_call_ms") is not None:
        c["cold_first_call_ms"] = _sig(d["cold_first_call_ms"], 4)
    return c


def compact_line(out, table):
    """The ONE JSON line the driver parses: the contract's keys first, `roofline` and `cpu_baseline` of the headline workload,
    then one numeric row per workload (`summary`).  Everything else (definitions, notes, per-rank lists, PCIe probe, the full
    `workloads` table) goes to the detail file named in `detail` -- never to stdout."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")
    c = {k: _sig(out.get(k)) for k in head if k in out}
    c["roofline"] = compact_roofline(out.get("roofline", {}))
    b = out.get("cpu_baseline")
    if isinstance(b, dict):
        c["cpu_baseline"] = {k: (_sig(v) if not isinstance(v, str) else v[:160]) for k, v in b.items()}
    for k in ("parity_max_rel_err_vs_oracle", "parity_bit_exact_vs_oracle", "speedup_vs_cpu_baseline", "allreduce_exact", "allreduce_check", "debug_force", "probe_library",
              "collective_backend", "library_source_hash", "loglik_monotone"):
        if out.get(k) is not None:
            c[k] = _sig(out[k])
    st = out.get("steady") or {}
    if st:
        c["steady"] = st
    col = out.get("collective")
    if isinstance(col, dict):
        c["collective"] = {k: _sig(col.get(k)) for k in ("allreduce_ms", "step_ms_with_allreduce", "ranks", "backend")}
    h = out.get("host_inclusive")
    if isinstance(h, dict):
        hc = {"unit": "ms per 10^6-frame ccall-style call on host arrays (PCIe-inclusive; never `value`)"}
        for name, key in (("fresh_output", None), ("reused_output", "reused_output"), ("registered", "registered")):
            d = h if key is None else h.get(key)
            if isinstance(d, dict) and d.get("ms_per_call") is not None:
                hc[name] = {"ms": _sig(d["ms_per_call"], 4), "frac_of_pcie": _sig(d.get("frac_of_pcie"), 3)}
        c["host_inclusive"] = hc
    s1 = out.get("cpu_baseline_strong")
    if isinstance(s1, dict) and "value" in s1:
        c["cpu_baseline_strong"] = {k: _sig(s1[k]) for k in ("value", "unit", "cores", "kind")}
    n1 = out.get("n1_consistency")
    if isinstance(n1, dict):
        c["n1_consistency"] = {k: ({"ratio": _sig(v.get("ratio"), 4), "within_5pct": v.get("within_5pct")} if isinstance(v, dict) else v)
                               for k, v in n1.items()}
    pr = out.get("per_rank")
    if isinstance(pr, dict) and out.get("n_gpus", 1) > 1:
        c["per_rank"] = {k: [_sig(x, 4) for x in v] if isinstance(v, list) else _sig(v, 4) for k, v in pr.items()}
    if len(table) > 1 or (table and next(iter(table.values())) is not out):
        c["summary"] = {name: compact_workload(d) for name, d in table.items() if isinstance(d, dict) and "ms_per_step" in d}
    if out.get("detail"):
        c["detail"] = out["detail"]
    line = json.dumps(c)
    for drop in ("n1_consistency", "per_rank", "host_inclusive", "cpu_baseline_strong", "steady"):   # never expected: the cap holds anyway
        if len(line) <= LINE_CAP:
            break
        c.pop(drop, None)
        line = json.dumps(c)
    if len(line) > LINE_CAP:
        c.pop("summary", None)
        line = json.dumps(c)
    return line


def write_detail(out, name):
    """The full record (every workload's complete object) -> gpurun_out/<name> under the repo, or the temp dir."""
    import tempfile

    for d in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, name)
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


CPU_CACHE = os.path.join(__import__("tempfile").gettempdir(), "vcmi_bench_cpu_baseline.json")


def cpu_baseline_cache_store(table, values=None):
    """N = 1 measured the CPU baselines; keep them for the N > 1 runs that follow on the same box -- and the N = 1 `value`
    of every workload with the hash of the library it came from, which an N > 1 line of the same library compares its
    per-rank rate with (`n1_consistency`)."""
    try:
        json.dump({"measured": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "host_cores": os.cpu_count(),
                   "cpu_baseline": table, "n1_values": values or {}, "library_source_hash": source_hash()}, open(CPU_CACHE, "w"))
    except OSError:
        pass


def n1_consistency(table, world):
    """Weak scaling with no data-path collective: a rank of an N-GPU run does what the N = 1 run did, so value / N must be
    the N = 1 value of the same library on the same box (to a few per cent: clocks, PCIe root sharing).  Reported, not
    enforced -- a failed assertion would cost the line itself."""
    try:
        c = json.load(open(CPU_CACHE))
    except (OSError, ValueError):
        return {"available": False, "why": "no N=1 run of this library on this box before this one"}
    if c.get("library_source_hash") != source_hash():
        return {"available": False, "why": "the cached N=1 run used another build of the library"}
    out = {"available": True, "n1_measured": c.get("measured")}
    for name, d in table.items():
        v1 = c.get("n1_values", {}).get(name)
        if v1 and isinstance(d, dict) and d.get("value"):
            ratio = d["value"] / world / v1
            out[name] = {"n1_value": v1, "per_rank_value": d["value"] / world, "ratio": ratio, "within_5pct": bool(abs(ratio - 1.0) <= 0.05)}
    return out


def cpu_baseline_cached(workload):
    """The CPU baseline is timed at N = 1 only; an N > 1 line carries the cached N = 1 figure, labelled as such (first
    the cache an N = 1 run left on this box, else the committed N = 1 profile of this round)."""
    try:
        c = json.load(open(CPU_CACHE))
        b = dict(c["cpu_baseline"][workload])
        b["cached"] = f"from the N=1 run of {c['measured']} on this box ({c['host_cores']} host cores); not re-timed at N>1"
        return b
    except (OSError, KeyError, ValueError):
        pass
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", f"r06_{workload}_bench.json")))
        b = dict(d["cpu_baseline"])
        b["cached"] = f"from the committed N=1 profile profiles/r06_{workload}_bench.json (another box); not re-timed at N>1"
        return b
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all",
                    choices=["all", "convert", "convert_fixture", "convert_joint", "convert_broad", "estep", "estep_fixture", "estep_full", "estep_full_fixture", "em_full", "dtw", "traj", "trajgv",
                             "selftest"],
                    help="all (default): the headline line of configs[1] plus a `workloads` table over configs[1..4]")
    ap.add_argument("--dim", type=int, default=40, help="dtw: feature dimension (40 = BASELINE; 41 = order-40 mel-cepstra with c0)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: at N=1 measure the headline kernel's HBM traffic in this run (two rocprofv3 --pmc child passes, "
                         "before this process touches the GPU); off: only the committed, source-hash-stamped passes")
    ap.add_argument("--prune", type=float, default=None,
                    help="convert: posterior pruning threshold in nats of the timed kernel (default: the library's own default, 46 = "
                         "regressions of mixtures with posterior < 1e-20 on a whole 16-frame tile are skipped; inf = every mixture "
                         "for every frame; the `dense` object of the line reports that loop as well)")
    ap.add_argument("--cpu-seconds-sub", type=float, default=5.0, help="CPU budget of each non-headline workload's cpu_baseline")
    ap.add_argument("--frames", type=int, default=1_000_000, help="frames per GPU (BASELINE: 10^6)")
    ap.add_argument("--pairs", type=int, default=1000, help="DTW pairs per GPU")
    ap.add_argument("--utts", type=int, default=256, help="trajectory utterances per GPU")
    ap.add_argument("--chunk", type=int, default=0, help="traj: convert in vc() chunks of this many frames "
                    "(bin/vc.jl:18 default --T=100); 0 = whole 2000-frame utterances (BASELINE configs[4])")
    ap.add_argument("--mixtures", type=int, default=128, help="estep: number of mixtures (128 = BASELINE configs[2])")
    ap.add_argument("--dj", type=int, default=80, help="estep: joint feature dimension (80 = BASELINE; 32, 48, 64 and 160 also run the MFMA kernel; others the generic kernels)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU time budget of the cpu_baseline sample")
    ap.add_argument("--debug-force", type=int, default=0, help="A/B experiments: bit mask for the library's vcmi_debug_force test hook "
                    "(csrc/vcmi_common.hpp kDbg*: alternative launch strategies); the line is then marked `debug_force`")
    ap.add_argument("--clock-warm-ms", type=float, default=100.0,
                    help="ms of untimed steps of the same workload enqueued ahead of the W warmup steps so that the timed steps run "
                         "at the steady shader clock (see timed_steps); 0 = only the W warmup steps")
    ap.add_argument("--profile-run", action="store_true",
                    help="for the rocprofv3 passes (tools/pmc_traffic.py, tools/clock_pmc.py, tools/round_profiles.sh): launch NOTHING but "
                         "the W warm-up and the K timed steps (and the steady-clock pass unless --clock-warm-ms 0) -- no MFMA-count launch, "
                         "no dense pass, no parity sample on the device, no host-pointer calls, no CPU baseline; implies --cpu-seconds 0 --pmc off")
    ap.add_argument("--verify-allreduce", action="store_true",
                    help="estep: rank 0 recomputes the statistics of every rank's frames in one process and compares")
    args = ap.parse_args()
    global CLOCK_WARM_MS
    CLOCK_WARM_MS = max(args.clock_warm_ms, 0.0)
    if args.profile_run:
        args.cpu_seconds, args.cpu_seconds_sub, args.pmc = 0.0, 0.0, "off"

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torchrun environment: start the ranks ourselves (child process), before any torch.cuda / HIP call
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    if args.gpus == 1 and args.pmc == "auto" and args.workload in ("all", "convert") and args.cpu_seconds > 0 \
            and "WORLD_SIZE" not in os.environ:
        # before anything in this process has touched torch.cuda / HIP: the PMC passes are child processes
        t = measure_traffic_live("convert", ["--frames", str(args.frames)])
        if t is not None:
            LIVE_PMC["convert"] = t

    world, rank, _ = dist_setup(args.gpus)
    if args.debug_force:
        os.environ["VCMI_TEST_HOOKS"] = "1"        # the hook is inert in a process started without it
        from voiceconversion_jl_amd import _lib

        _lib.debug_force(args.debug_force)
    if world > 1:
        args.cpu_seconds = 0.0        # the CPU baseline is timed at N = 1 only (rank 0 keeps its small parity sample)
        args.cpu_seconds_sub = 0.0
    fns = {"convert": bench_convert, "estep": bench_estep, "estep_full": bench_estep_full, "em_full": bench_em_full,
           "dtw": bench_dtw, "traj": bench_traj, "trajgv": lambda a, w, r: bench_traj(a, w, r, gv=True),
           "convert_fixture": lambda a, w, r: bench_convert(a, w, r, variant="fixture"),
           "convert_broad": lambda a, w, r: bench_convert(a, w, r, variant="broad"),
           "convert_joint": lambda a, w, r: bench_convert(a, w, r, variant="joint"),
           "estep_fixture": lambda a, w, r: bench_estep(a, w, r, variant="fixture"),
           "estep_full_fixture": lambda a, w, r: bench_estep_full(a, w, r, variant="fixture"),
           "selftest": bench_selftest}
    if args.workload == "all":
        import copy
        import gc

        import torch

        out = bench_convert(args, world, rank)
        table = {"convert": summarize(out)}
        sub = copy.copy(args)
        sub.cpu_seconds = args.cpu_seconds_sub
        for name in ("convert_fixture", "convert_broad", "estep", "estep_fixture", "dtw", "traj"):
            gc.collect()
            torch.cuda.empty_cache()
            PER_RANK.clear()
            t0 = time.perf_counter()
            table[name] = summarize(fns[name](sub, world, rank))
            table[name]["bench_wall_s"] = time.perf_counter() - t0
        PER_RANK.clear()
        PER_RANK.update(table["convert"]["per_rank"])
        out["workloads"] = table
    else:
        out = fns[args.workload](args, world, rank)
        table = {args.workload: out}
    if world > 1:
        for k in ("cpu_baseline_strong", "speedup_vs_cpu_baseline", "host_inclusive"):
            out.pop(k, None)
        for name, d in table.items():            # an N > 1 line keeps the N = 1 CPU baseline, labelled as cached
            d.pop("speedup_vs_cpu_baseline", None)
            b = cpu_baseline_cached(name)
            if b is not None:
                d["cpu_baseline"] = b
            else:
                d.pop("cpu_baseline", None)
        if args.workload == "all" and "cpu_baseline" in table["convert"]:
            out["cpu_baseline"] = table["convert"]["cpu_baseline"]
        if rank == 0:
            out["n1_consistency"] = n1_consistency(table, world)
    elif rank == 0 and args.workload != "selftest" and args.cpu_seconds > 0:
        have = {name: d["cpu_baseline"] for name, d in table.items() if isinstance(d.get("cpu_baseline"), dict) and "value" in d["cpu_baseline"]}
        if have:
            try:
                prev = json.load(open(CPU_CACHE))["cpu_baseline"]
            except (OSError, KeyError, ValueError):
                prev = {}
            prev.update(have)
            try:
                prev_vals = json.load(open(CPU_CACHE)).get("n1_values", {})
            except (OSError, KeyError, ValueError):
                prev_vals = {}
            prev_vals.update({name: d["value"] for name, d in table.items() if isinstance(d, dict) and d.get("value")})
            cpu_baseline_cache_store(prev, prev_vals)
    out["n_gpus"] = world
    out["per_rank"] = dict(PER_RANK)
    out["collective_backend"] = BACKEND["name"]
    out["library_source_hash"] = source_hash()
    if args.debug_force:
        out["debug_force"] = args.debug_force      # not a product configuration
    if os.environ.get("LIBVCMI_PROBE"):
        out["probe_library"] = os.environ["LIBVCMI_PROBE"]      # an A/B or probe build, not the in-tree library
    if rank == 0:
        out["detail"] = write_detail(out, "bench_detail_%s_n%d.json" % (args.workload, world))
        print(compact_line(out, table), flush=True)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
