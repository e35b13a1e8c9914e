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


def test_roofline_traffic_is_refused_when_collected_from_other_sources(tmp_path, monkeypatch):
    """bench.py quotes a committed PMC pass only when it was collected from the library sources the loaded library was
    built from (`_meta.source_hash`, tools/pmc_traffic.py); a stale file yields traffic = None and says why
    (a stale file silently reported old traffic)."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    csrc = tmp_path / "voiceconversion.jl_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "k.hip").write_text("// kernel v1\n")
    pm = tmp_path / "profiles" / bench.PMC_DIR
    pm.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    h1 = bench.source_hash()
    table = {"void vcmi::some_kernel<40>": {"FETCH_SIZE_KB_per_step": 1000.0, "WRITE_SIZE_KB_per_step": 500.0},
             "other": {"FETCH_SIZE_KB_per_step": 7.0}, "_meta": {"source_hash": h1, "collected": "now"}}
    (pm / "x_traffic.json").write_text(json.dumps(table))
    v, src = bench.pmc_traffic("x_traffic.json", "some_kernel", 1.0)                 # the counters as they are
    assert v == 1500.0 * 1024 and src["source_hash"] == h1
    v, _ = bench.pmc_traffic("x_traffic.json", "some_kernel")                        # FETCH doubled: vector-memory loads (default)
    assert v == 2500.0 * 1024
    v, _ = bench.pmc_traffic("x_traffic.json", "some_kernel", {"some_kernel<40>": 1.5, "": 2.0})     # per-kernel factor
    assert v == 2000.0 * 1024
    (csrc / "k.hip").write_text("// kernel v2\n")                                    # the sources moved on
    assert bench.source_hash() != h1
    v, src = bench.pmc_traffic("x_traffic.json", "some_kernel")
    assert v is None and "REFUSED" in src["source"]
    v, src = bench.pmc_traffic("missing.json", "some_kernel")
    assert v is None and "missing" in src["source"]
    # a table measured by this run is taken as it is
    v, src = bench.pmc_traffic("x_traffic.json", "some_kernel", 1.0, live=table)
    assert v == 1500.0 * 1024 and "measured in this run" in src["source"]
