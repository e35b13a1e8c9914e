"""Does a second solver workgroup on a CU pay?  (tools, GPU box.)  At static D = 16 the blocked trajectory solver needs 152
VGPRs and ~67 KB of LDS, so two workgroups fit a CU; at D = 40 (428 VGPRs, 143 KB) only one does.  Times n = 256 and n = 512
utterances of 2000 frames at D = 16: if 512 take about as long as 256, two independent scalar chains per CU overlap --
what a two-solves-per-CU design at D = 40 would buy."""
import os

os.environ.setdefault("VCMI_TEST_HOOKS", "1")       # vcmi_debug_force is inert without it
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import voiceconversion_jl_amd as vc, synthdata as sd
from voiceconversion_jl_amd import _lib

D, M, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 16, 2000
if len(sys.argv) > 2 and sys.argv[2] == "one":
    _lib.debug_force(_lib.DBG_TRAJ_ONE_WG_PER_CU)
w, mu, sig = sd.synth_model(5, 4 * D, M, lam_lo=1e-3)
g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
tj = vc.TrajectoryGMMMap(g, T)
st = sd.sample_frames(6, w, mu, sig, T, 0, D)
st = np.cumsum(st, axis=0) / np.sqrt(np.arange(1, T + 1))[:, None]
X1 = np.ascontiguousarray(vc.push_delta(np.asfortranarray(st.T)).T)
for n in (128, 256, 512, 768, 1024):
    X = torch.from_numpy(np.tile(X1, (n, 1))).cuda()
    Y = torch.empty((n * T, D), dtype=torch.float64, device="cuda")
    xoff = np.arange(n, dtype=np.int64) * T * 2 * D; yoff = np.arange(n, dtype=np.int64) * T * D; Ts = np.full(n, T, dtype=np.int64)
    def step():
        _lib.check(_lib.lib.vcmi_traj_convert_batch_dev(tj._h, n, X.data_ptr(), _lib.iptr(xoff), _lib.iptr(Ts), Y.data_ptr(), _lib.iptr(yoff), torch.cuda.current_stream().cuda_stream))
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); print("D", D, "n", n, "%.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
