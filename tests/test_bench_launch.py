"""CPU: `python bench.py --gpus N` starts its own N ranks (child `torch.distributed.run`, before any GPU call), reports
n_gpus = N, and fails loudly when the rank count is wrong or a rank dies.  `--workload selftest` has no kernels, so this
runs without a GPU; the same launch path carries the real workloads (tests/test_gpu_bench_dist.py)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_self_launch_two_ranks_gloo():
    p = _run(["--gpus", "2", "--workload", "selftest", "--steps", "2"], {"VCMI_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["allreduce_exact"] is True and out["collective_backend"] == "gloo"
    assert len(out["per_rank"]["wall_s"]) == 2


def test_rank_count_mismatch_is_refused():
    p = _run(["--gpus", "2", "--workload", "selftest"], {"WORLD_SIZE": "1", "RANK": "0", "VCMI_BENCH_BACKEND": "gloo"})
    assert p.returncode == 2 and "refusing to run" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_dead_rank_fails_the_launch():
    p = _run(["--gpus", "2", "--workload", "selftest"], {"VCMI_BENCH_BACKEND": "gloo", "VCMI_SELFTEST_DIE_RANK": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
