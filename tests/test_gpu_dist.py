"""GPU, world_size = 2 (two processes sharing the one visible GPU, gloo for the collectives because RCCL wants one
device per rank): the distributed product paths end to end --
  * train_gmm: every rank holds half of the frames, the statistics are all-reduced once per EM iteration, and the
    result equals the single-process fit on all frames (E-step is the only path with a collective);
  * fvconvert: contiguous frame shards, no collective, concatenation equals the single-process conversion."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import voiceconversion_jl_amd as vc
    from oracle import np_oracle as npo
    from voiceconversion_jl_amd import dist as vd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    Dj, M, N = 16, 3, 8000
    w, mu, sig = npo.synth_model(21, Dj, M, lam_lo=1e-1)
    X = npo.sample_frames(21, w, mu * 3.0, sig, N, 0, Dj)
    lo, hi = vd.shard_range(N, rank, world)
    Xd = torch.from_numpy(X[lo:hi]).cuda().t()
    start = (np.full(M, 1.0 / M), X[[5, 4000, 7777]].T.copy(), np.repeat(np.cov(X.T)[:, :, None], M, axis=2))
    r = vc.train_gmm(Xd, n_components=M, n_iter=6, tol=0.0, refine=start)
    out = {"ll": r["loglik"], "means": r["means"]}
    dist.barrier()
    # conversion shards
    g = vc.GMMMap(w, np.asfortranarray((mu * 3.0).T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
    Xc = X[:, :Dj // 2]
    lo2, hi2 = vd.shard_range(N, rank, world)
    out["y"] = vc.fvconvert(g, np.asfortranarray(Xc[lo2:hi2].T))
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_rank_train_gmm_and_convert_match_single_process():
    import voiceconversion_jl_amd as vc
    from oracle import np_oracle as npo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all frames
    Dj, M, N = 16, 3, 8000
    w, mu, sig = npo.synth_model(21, Dj, M, lam_lo=1e-1)
    X = npo.sample_frames(21, w, mu * 3.0, sig, N, 0, Dj)
    start = (np.full(M, 1.0 / M), X[[5, 4000, 7777]].T.copy(), np.repeat(np.cov(X.T)[:, :, None], M, axis=2))
    ref = vc.train_gmm(torch.from_numpy(X).cuda().t(), n_components=M, n_iter=6, tol=0.0, refine=start)
    for rank in (0, 1):
        assert np.allclose(res[rank]["ll"], ref["loglik"], rtol=1e-11, atol=0)
        assert np.max(np.abs(res[rank]["means"] - ref["means"])) < 1e-9 * np.max(np.abs(ref["means"]))
    assert np.array_equal(res[0]["means"], res[1]["means"])          # both ranks hold the same model
    g = vc.GMMMap(w, np.asfortranarray((mu * 3.0).T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
    full = vc.fvconvert(g, np.asfortranarray(X[:, :Dj // 2].T))
    assert np.array_equal(np.concatenate([res[0]["y"], res[1]["y"]], axis=1), full)


def _estep_case():
    """first half: frames with owners (far-apart mixtures); second half: frames of a cluster of overlapping mixtures -- contiguous
    shards of two ranks then take DIFFERENT paths of the diagonal E-step"""
    Dj, M, Nh = 80, 64, 70_000
    rg = np.random.default_rng(77)
    w = rg.dirichlet(2.0 * np.ones(M))
    var = np.exp(rg.uniform(np.log(0.05), 0.0, (M, Dj)))
    mu = 3.0 * rg.standard_normal((M, Dj))
    mu[M // 2:] = mu[M // 2] + 0.3 * rg.standard_normal((M - M // 2, Dj)) * np.sqrt(var[M // 2:])
    p_far = w[:M // 2] / w[:M // 2].sum()
    p_near = w[M // 2:] / w[M // 2:].sum()
    c0 = rg.choice(M // 2, size=Nh, p=p_far)
    c1 = M // 2 + rg.choice(M - M // 2, size=Nh, p=p_near)
    comp = np.concatenate([c0, c1])
    X = mu[comp] + rg.standard_normal((2 * Nh, Dj)) * np.sqrt(var[comp])
    return w, mu, var, X


def _estep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VCMI_TEST_HOOKS="1")
    import torch.distributed as dist
    import voiceconversion_jl_amd as vc
    from voiceconversion_jl_amd import _lib, dist as vd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w, mu, var, X = _estep_case()
    lo, hi = vd.shard_range(len(X), rank, world)
    Xd = torch.from_numpy(X[lo:hi]).cuda().t()
    local = vc.estep_diag_dev(Xd, w, mu.T, var.T)
    soft = _lib.estep_last_soft()
    red = vc.unpack_stats(vc.estep_diag_allreduce(Xd, w, mu.T, var.T).cpu().numpy(), mu.shape[1], len(w))
    q.put((rank, {"soft": soft, "stats": [np.array(t) for t in red[:3]] + [float(red[3])]}))
    dist.destroy_process_group()


def test_two_ranks_on_different_estep_paths_match_single_process():
    """the path of the diagonal E-step follows from each rank's OWN frames -- rank 0's shard is all owned (the
    hard-assignment path), rank 1's is shared between overlapping mixtures (the one-kernel path): the all-reduced statistics
    are within 1e-12 of one process on all frames, and both ranks hold the same bits."""
    import voiceconversion_jl_amd as vc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 90)
    procs = [ctx.Process(target=_estep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["soft"] == 0 and res[1]["soft"] == -1, (res[0]["soft"], res[1]["soft"])
    w, mu, var, X = _estep_case()
    vc.estep_set_path(vc.ESTEP_SOFT)
    try:
        S0, S1, S2, ll = vc.estep_diag(X.T, w, mu.T, var.T)
    finally:
        vc.estep_set_path(vc.ESTEP_AUTO)
    for rank in (0, 1):
        g0, g1, g2, gl = res[rank]["stats"]
        assert np.max(np.abs(g0 - S0)) <= 1e-12 * np.max(np.abs(S0)) and np.max(np.abs(g1 - S1)) <= 1e-12 * np.max(np.abs(S1))
        assert np.max(np.abs(g2 - S2)) <= 1e-12 * np.max(np.abs(S2)) and abs(gl - ll) <= 1e-12 * abs(ll)
    assert all(np.array_equal(a, b) for a, b in zip(res[0]["stats"][:3], res[1]["stats"][:3])) and res[0]["stats"][3] == res[1]["stats"][3]
