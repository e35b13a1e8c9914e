#!/usr/bin/env python3
"""What the memory system gives a row gather / scatter of 10^6 x 320-byte rows, by locality of the permutation (run on the GPU
box): identity, random within blocks of B rows (B = 8192 ... 10^6).  torch.index_select / index_copy_ as the probe kernels."""
import json, sys
import torch
N, D = 1_000_000, 40
X = torch.randn(N, D, dtype=torch.float64, device="cuda")
Y = torch.empty_like(X)
g = torch.Generator(device="cpu"); g.manual_seed(1)
def perm_blocks(B):
    p = torch.arange(N)
    for s in range(0, N, B):
        e = min(N, s + B)
        p[s:e] = s + torch.randperm(e - s, generator=g)
    return p.cuda()
def group_sorted(M=64):
    key = torch.randint(0, M, (N,), generator=g)
    return torch.argsort(key, stable=True).cuda()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = {}
res["copy"] = timeit(lambda: Y.copy_(X))
cases = {"identity": torch.arange(N).cuda(), "grouped_by_64_keys_stable": group_sorted(64)}
for B in (2048, 8192, 65536, 262144, 1_000_000):
    cases[f"random_within_{B}"] = perm_blocks(B)
for name, p in cases.items():
    res[name] = {"gather_ms": timeit(lambda: torch.index_select(X, 0, p, out=Y)), "scatter_ms": timeit(lambda: Y.index_copy_(0, p, X))}
for k, v in res.items():
    if isinstance(v, dict):
        v["gather_GBps_2x320B"] = 640.0 * N / (v["gather_ms"] * 1e-3) / 1e9
        v["scatter_GBps_2x320B"] = 640.0 * N / (v["scatter_ms"] * 1e-3) / 1e9
print(json.dumps(res, indent=1))
