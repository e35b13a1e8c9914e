"""GPU: the multi-rank bench path with real kernels.
  * RCCL ("nccl") with ONE rank: the communicator is created and the E-step statistics go through ncclAllReduce -- the
    most a 1-GPU box can exercise of the collective path;
  * two self-launched ranks sharing the visible GPU (gloo): the all-reduced statistics equal one process over both
    ranks' frames to 1e-12 (summation order differs, so not bitwise);
  * two ranks over RCCL, one per GPU, when the box has two GPUs (skipped with the reason otherwise)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


ESTEP = ["--workload", "estep", "--frames", "60000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--verify-allreduce"]


def test_one_rank_rccl_allreduce():
    out = _bench(["--gpus", "1"] + ESTEP, {"VCMI_BENCH_FORCE_PG": "1"})
    assert out["collective_backend"] == "nccl" and out["n_gpus"] == 1
    assert out["allreduce_check"]["max_rel_err_vs_single_process"] <= 1e-12
    assert out["parity_max_rel_err_vs_oracle"] < 1e-9


def test_two_self_launched_ranks_share_the_gpu_gloo():
    out = _bench(["--gpus", "2"] + ESTEP, {"VCMI_BENCH_BACKEND": "gloo", "VCMI_BENCH_DEVICE": "0"})
    assert out["n_gpus"] == 2 and out["collective_backend"] == "gloo"
    assert out["allreduce_check"]["frames_total"] == 120000
    assert out["allreduce_check"]["max_rel_err_vs_single_process"] <= 1e-12
    assert len(out["per_rank"]["wall_s"]) == 2


def test_two_ranks_rccl_one_per_gpu():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs for one RCCL rank per device; this box has {torch.cuda.device_count()}")
    out = _bench(["--gpus", "2"] + ESTEP, {})
    assert out["n_gpus"] == 2 and out["collective_backend"] == "nccl"
    assert out["allreduce_check"]["max_rel_err_vs_single_process"] <= 1e-12
    conv = _bench(["--gpus", "2", "--frames", "200000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"], {})
    assert conv["n_gpus"] == 2 and conv["parity_max_rel_err_vs_oracle"] < 1e-9
