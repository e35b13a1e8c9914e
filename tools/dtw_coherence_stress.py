"""One-off stress of the fence-free job synchronisation of the fused DTW kernel (csrc/dtw.hip, dtw_wait_flag): the benchmark
batch (1000 pairs of ~500 x 500 frames -> 3279 segment jobs on all eight XCDs) over and over, every result compared with
the whole-length, grid-order run of the same inputs.  usage: python tools/dtw_coherence_stress.py [reps] [D ...]"""
import os
import sys

os.environ.setdefault("VCMI_TEST_HOOKS", "1")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from voiceconversion_jl_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dims = [int(a) for a in sys.argv[2:]] or [40, 41, 24]
for D in dims:
    n = 1000
    pairs = bench._dtw_pairs(7000 + D, n, D)
    feats, toff, soff, poff, S, T = [], [], [], [], [], []
    fo = po = 0
    for t, s in pairs:
        toff.append(fo); feats.append(t.ravel()); fo += t.size
        soff.append(fo); feats.append(s.ravel()); fo += s.size
        poff.append(po); po += s.shape[0]
        S.append(t.shape[0]); T.append(s.shape[0])
    fd = torch.from_numpy(np.concatenate(feats)).cuda()
    arr = lambda a: np.asarray(a, dtype=np.int64)  # noqa: E731
    toff, soff, poff, S, T = arr(toff), arr(soff), arr(poff), arr(S), arr(T)

    def run():
        pd = torch.zeros(po, dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib.vcmi_dtw_fit_batch_dev(n, fd.data_ptr(), _lib.iptr(toff), _lib.iptr(S), _lib.iptr(soff), _lib.iptr(T),
                                                   D, 0, 2, pd.data_ptr(), _lib.iptr(poff), torch.cuda.current_stream().cuda_stream))
        return pd

    _lib.debug_force(_lib.DBG_DTW_NO_SEGMENTS | _lib.DBG_DTW_GRID_ORDER)
    ref = run()
    _lib.debug_force(0)
    bad = 0
    outs = [run() for _ in range(reps)]            # back to back on the stream, compared afterwards on the device
    torch.cuda.synchronize()
    for o in outs:
        bad += int((o != ref).sum().item())
    print("D=%d: %d repetitions of %d pairs, %d path entries differ" % (D, reps, n, bad))
    assert bad == 0
print("STRESS_OK")
