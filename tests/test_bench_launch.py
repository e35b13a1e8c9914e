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


def test_self_launch_eight_ranks_gloo():
    """the launch path of `bench.py --gpus 8` -- eight child ranks over gloo, the barrier-bracketed region,
    the max over ranks, the statistics-sized all-reduce -- emits exactly ONE line whose collective names eight ranks and whose
    config names the split, so that on the first 8-GPU node the only unknown left is RCCL itself (nothing in this tree has run
    on two physical GPUs: README)."""
    p = _run(["--gpus", "8", "--workload", "selftest", "--steps", "2"], {"VCMI_BENCH_BACKEND": "gloo"}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["collective"]["ranks"] == 8 and out["collective"]["backend"] == "gloo"
    assert "x8" in out["config"]["sharding"] and out["allreduce_exact"] is True
    assert len(out["per_rank"]["wall_s"]) == 8 and out["scaling"] == "weak"


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


def test_the_one_stdout_line_stays_under_the_cap():
    """BENCH_r04 lost its record because the line had grown to 20.9 KB (the driver keeps an 8 KB tail): the line printed is
    compact_line() of the full record -- contract keys, the headline's roofline and cpu_baseline, one numeric row per
    workload -- and never exceeds LINE_CAP whatever the full record holds."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_all_bench.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000                     # the round-4 record, as it was printed then
    full["detail"] = "gpurun_out/bench_detail_all_n1.json"
    full["n1_consistency"] = {"available": True, "n1_measured": "x", **{k: {"n1_value": 1.0, "per_rank_value": 1.0, "ratio": 1.0,
                                                                         "within_5pct": True} for k in full["workloads"]}}
    full["per_rank"] = {"wall_s": [0.033] * 8, "kernel_ms": [1.6] * 8}
    for n in (1, 8):
        full["n_gpus"] = n
        line = bench.compact_line(full, full["workloads"])
        assert len(line) < bench.LINE_CAP <= 8000, len(line)
        out = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "summary", "detail"):
            assert k in out, k
        assert list(out)[:3] == ["metric", "value", "unit"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in out["roofline"], k
        assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])
        assert set(out["summary"]) == set(full["workloads"])
        assert all("frac" in r["roofline"] and "traffic" in r["roofline"] for r in out["summary"].values())
    # this round's record (seven workloads in the default run) at N = 1 and N = 8 as well
    full5 = json.load(open(os.path.join(ROOT, "profiles", "r05_all_bench_detail.json")))
    assert len(full5["workloads"]) >= 7
    full5["detail"] = "gpurun_out/bench_detail_all_n8.json"
    full5["n1_consistency"] = {"available": True, "n1_measured": "x", **{k: {"n1_value": 1.0, "per_rank_value": 1.0, "ratio": 1.0,
                                                                          "within_5pct": True} for k in full5["workloads"]}}
    full5["per_rank"] = {"wall_s": [0.033] * 8, "kernel_ms": [1.6] * 8, "first_part_ms": [0.5] * 8}
    for n in (1, 8):
        full5["n_gpus"] = n
        line5 = bench.compact_line(full5, full5["workloads"])
        assert len(line5) < bench.LINE_CAP, len(line5)
        assert set(json.loads(line5)["summary"]) == set(full5["workloads"])
    # no prose anywhere in the line: every string value is short
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(out)) <= 160


def test_selftest_line_is_short_and_names_its_detail_file():
    p = _run(["--gpus", "2", "--workload", "selftest", "--steps", "2"], {"VCMI_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][0]
    assert len(line) < 2000
    out = json.loads(line)
    assert out["detail"] and os.path.exists(os.path.join(ROOT, out["detail"]) if not os.path.isabs(out["detail"]) else out["detail"])


def test_traffic_tables_with_wrong_step_accounting_are_refused(tmp_path, monkeypatch):
    """Round 4 divided the counters of ~60 launches by 3 steps; a table whose launch count is not whole per step, or whose
    bytes over the kernel's time exceed the HBM peak, is refused with the reason (traffic = None)."""
    sys.path.insert(0, ROOT)
    import bench
    live = {"void vcmi::k<1>": {"FETCH_SIZE_KB_per_step": 1e6, "WRITE_SIZE_KB_per_step": 1e6, "launches_per_step": 19.0 / 3}, "_meta": {}}
    v, src = bench.pmc_traffic("x", "vcmi::k", 1.0, live=live)
    assert v is None and "REFUSED" in src["source"] and "launches per step" in src["source"]
    live["void vcmi::k<1>"]["launches_per_step"] = 1.0
    out = {"roofline": {"kernel_ms": 0.1}}
    bench.attach_traffic(out, "x", "vcmi::k", 1.0, live=live)                # 2 GB in 0.1 ms = 20 TB/s
    assert out["roofline"]["traffic"] is None and "exceeds" in out["roofline"]["traffic_source"]["source"]
    out = {"roofline": {"kernel_ms": 1.0}}
    bench.attach_traffic(out, "x", "vcmi::k", 1.0, live=live, algorithmic_bytes=1.024e9)
    assert out["roofline"]["traffic"] == 2e6 * 1024 and abs(out["roofline"]["traffic_x_algorithmic"] - 2.0) < 1e-12
    live["_meta"]["problems"] = {"__amd_rocclr_copyBuffer": "5 launches in the FETCH_SIZE pass are not a multiple of the 3 steps"}
    v, src = bench.pmc_traffic("x", "vcmi::k", 1.0, live=live)                # another kernel's problem (a model upload): not ours
    assert v == 2e6 * 1024
    live["_meta"]["problems"]["void vcmi::k<1>"] = "7 launches in the FETCH_SIZE pass are not a multiple of the 3 steps"
    v, src = bench.pmc_traffic("x", "vcmi::k", 1.0, live=live)
    assert v is None and "not a multiple" in src["source"]
