"""CPU, world_size = 2, gloo: the multi-process path -- frame sharding + ONE all-reduce of the packed E-step
statistics reproduces the single-process statistics; pair sharding covers every DTW pair exactly once.
The per-shard arithmetic is done by the oracle here (no GPU in this container); the sharding, packing and
collective code is the product's (voiceconversion_jl_amd.dist)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import voiceconversion_jl_amd as vc
    from oracle import c_oracle as co
    from voiceconversion_jl_amd import dist as vd

    r, w_, _ = vd.init_process_group("gloo")
    assert (r, w_) == (rank, world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "estep_diag_N2000_D80_M16.npz"))
    N = z["X"].shape[0]
    lo, hi = vd.shard_range(N, rank, world)
    S0, S1, S2, ll = co.estep_diag(z["X"][lo:hi], z["w"], z["mu"], z["var"])
    packed = torch.from_numpy(vd.pack_stats(S0, S1.T, S2.T, ll))
    vd.allreduce_sum_(packed)
    g0, g1, g2, gl = vc.unpack_stats(packed.numpy(), 80, 16)
    ok = (np.max(np.abs(g0 - z["S0"])) < 1e-9 * np.max(np.abs(z["S0"])) and
          np.max(np.abs(g1 - z["S1"].T)) < 1e-9 * np.max(np.abs(z["S1"])) and
          np.max(np.abs(g2 - z["S2"].T)) < 1e-9 * np.max(np.abs(z["S2"])) and
          abs(gl - float(z["loglik"])) < 1e-9 * abs(float(z["loglik"])))
    # DTW pairs: each rank aligns its cost-balanced share; gather the paths and compare with golden
    d = np.load(os.path.join(ROOT, "tests", "golden", "dtw_cases.npz"))
    n = int(d["n_random"])
    costs = [d[f"r{k}_tmpl"].shape[0] * d[f"r{k}_seq"].shape[0] for k in range(n)]
    mine = vd.shard_by_cost(costs, world)[rank]
    good = torch.zeros(n, dtype=torch.int32)
    for k in mine:
        fs, bs = (int(x) for x in d[f"r{k}_steps"])
        good[k] = int(np.array_equal(co.dtw_fit(d[f"r{k}_tmpl"], d[f"r{k}_seq"], fs, bs, tables=False), d[f"r{k}_path"]))
    vd.allreduce_sum_(good)
    ok = ok and bool((good == 1).all())            # every pair done exactly once, all correct
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_gloo_estep_allreduce_and_pair_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
