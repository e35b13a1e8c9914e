"""Host-side cost of one vcmi_dtw_fit_batch_dev call (descriptor building, uploads, launches) beside the GPU time of the
step: the call returns before the kernels finish, so a step is GPU-bound only while the host part is the shorter one.
usage: python tools/dtw_host_probe.py [D] [pairs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from voiceconversion_jl_amd import _lib  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pairs = bench._dtw_pairs(1004, n, D)
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
own = torch.cuda.Stream() if os.environ.get("PROBE_STREAM") else None
if own is not None:
    torch.cuda.set_stream(own)
st = torch.cuda.current_stream().cuda_stream


def step():
    _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(n, fd.data_ptr(), _lib.iptr(toff), _lib.iptr(S), _lib.iptr(soff), _lib.iptr(T),
                                               D, 0, 2, pd.data_ptr(), _lib.iptr(poff), st))


for _ in range(5):
    step()
torch.cuda.synchronize()
# (a) synchronised after every call: host part + GPU part in series; (b) back to back
enq, tot = [], []
for _ in range(20):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0)
    tot.append(t2 - t0)
def task_times():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[tid] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out


def throttled():
    try:
        return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
    except OSError:
        return {}


import threading  # noqa: E402

main_tid = threading.get_native_id()
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        rec = [time.perf_counter()]
        for f in ("syscall", "wchan", "stat"):
            try:
                v = open("/proc/self/task/%d/%s" % (main_tid, f)).read().strip()
                rec.append(v if f != "stat" else v[v.rindex(")") + 2:].split()[0])
            except OSError as e:
                rec.append("err %s" % e.errno)
        samples.append(rec)
        time.sleep(0.001)


th = threading.Thread(target=sampler, daemon=True)
if os.environ.get("PROBE_SAMPLE"):
    th.start()
if os.environ.get("PROBE_NOGC"):
    import gc

    gc.collect()
    gc.disable()
tt0, th0 = task_times(), throttled()
t0 = time.perf_counter()
each = []
fn = _lib.lib.vcmi_dtw_fit_batch_dev
parts = []
for _ in range(20):
    ta = time.perf_counter()
    a = (n, fd.data_ptr(), _lib.iptr(toff), _lib.iptr(S), _lib.iptr(soff), _lib.iptr(T), D, 0, 2, pd.data_ptr(), _lib.iptr(poff), st)
    tb = time.perf_counter()
    rc = fn(*a)
    tc = time.perf_counter()
    _lib.check(rc)
    td = time.perf_counter()
    each.append(td - ta)
    parts.append((tb - ta, tc - tb, td - tc))
t1 = time.perf_counter()
worst = max(range(20), key=lambda i: each[i])
print("slowest call %d: marshalling %.3f ms, C call %.3f ms, check %.3f ms" % ((worst,) + tuple(1e3 * x for x in parts[worst])))
torch.cuda.synchronize()
t2 = time.perf_counter()
stop[0] = True
if samples:
    print("sampler: %d samples in %.1f ms; main thread (syscall | wchan | state):" % (len(samples), 1e3 * (samples[-1][0] - samples[0][0])))
    last = None
    for r in samples:
        key = (r[1].split()[0] if r[1] else "", r[2], r[3])
        if key != last:
            print("   +%.2f ms  %s | %s | %s" % (1e3 * (r[0] - t0), r[1][:70], r[2], r[3]))
            last = key
tt1, th1 = task_times(), throttled()
busy = sorted(((tt1[k][1] - tt0.get(k, (0, 0))[1], tt1[k][0]) for k in tt1), reverse=True)
print("threads:", len(tt1), "cpu seconds in the back-to-back phase (%.3f s wall):" % (t2 - t0), [(round(b, 3), c) for b, c in busy[:8]],
      "cgroup throttled +%s periods, +%.1f ms" % (int(th1.get("nr_throttled", 0)) - int(th0.get("nr_throttled", 0)),
                                                 (int(th1.get("throttled_usec", 0)) - int(th0.get("throttled_usec", 0))) / 1e3))
print("back to back, ms per call:", " ".join("%.2f" % (1e3 * e) for e in each), "| final sync %.2f" % (1e3 * (t2 - t1)), "| stream", hex(st))
print("D=%d n=%d: host part of a call %.3f ms (min %.3f), call + sync %.3f ms; 20 calls back to back: enqueue %.3f ms per call, "
      "%.3f ms per step in all; loadavg %s" % (D, n, 1e3 * np.median(enq), 1e3 * min(enq), 1e3 * np.median(tot), 1e3 * (t1 - t0) / 20,
                                              1e3 * (t2 - t0) / 20, open("/proc/loadavg").read().split()[0]))
