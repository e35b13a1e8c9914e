"""Generate the fixtures of the reference's SECOND trained model -- run in the BUILD container only (needs /root/reference
and /opt/conda/bin/h5dump).  TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_golden_joint

test/models/clb_and_slt_gmm32_order40.jld is the joint (non-differential) 32-mixture model test/vc.jl:40-51 loads into a
GMMMap and converts with (the test then needs WORLD and only asserts finiteness).  The tensors are taken from the HDF5 file
as they lie (the raw buffers are the Julia memory image); the expected outputs come from the numpy
restatement (oracle/np_oracle.py), checked here against the C oracle (<= 1e-9) and against third-party code
(oracle/crosscheck.py: sklearn / scipy / LAPACK; 50-digit mpmath for the forward direction) before anything is written:

  tests/golden/model_clb_and_slt_gmm32_order40.npz   weights (32), means (32,80), covars (32,80,80) [m][col][row]
  tests/golden/gmmmap_joint_model.npz                 X / Y / P / idx for 256 frames drawn from the model, both directions,
                                                      and a vc() case (power row kept)
"""
import os
import subprocess
import tempfile

import numpy as np

from . import c_oracle as co
from . import crosscheck as cc
from . import np_oracle as npo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF_MODEL = "/root/reference/test/models/clb_and_slt_gmm32_order40.jld"
H5DUMP = "/opt/conda/bin/h5dump"


def _relmax(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300))


def extract_model():
    with tempfile.TemporaryDirectory() as td:
        arrs = {}
        for name in ("weights", "means", "covars"):
            p = os.path.join(td, name + ".bin")
            subprocess.check_call([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", p, REF_MODEL], stdout=subprocess.DEVNULL)
            arrs[name] = np.fromfile(p, dtype="<f8")
    w = arrs["weights"]
    M = len(w)
    mu = arrs["means"].reshape(M, -1)
    Dj = mu.shape[1]
    return w, mu, arrs["covars"].reshape(M, Dj, Dj)


def main():
    w, mu, sig = extract_model()
    M, Dj = mu.shape
    D = Dj // 2
    assert (M, Dj) == (32, 80) and abs(w.sum() - 1.0) < 1e-9
    np.savez(os.path.join(OUT, "model_clb_and_slt_gmm32_order40.npz"), weights=w, means=mu, covars=sig)
    out = {}
    for swap in (False, True):
        gn, gc = npo.GMMMap(w, mu, sig, swap=swap), co.GMMMap(w, mu, sig, swap=swap)
        lo = D if swap else 0
        X = npo.sample_frames(30261 + swap, w, mu, sig, 256, lo, lo + D)
        Y, P, idx = gn.fvconvert(X), gn.predict_proba(X), gn.predict(X)
        assert _relmax(Y, gc.fvconvert(X)) < 1e-9 and np.max(np.abs(P - gc.predict_proba(X))) < 1e-9
        assert np.array_equal(idx, gc.predict(X)) and _relmax(gn.A, gc.A) < 1e-9
        print("cross-check", "swap" if swap else "fwd", cc.check_conversion(w, mu, sig, X, Y, P, idx, swap=swap))
        if not swap:
            print("cross-check mpmath", cc.check_conversion_mpmath(w, mu, sig, X, Y, P))
        k = "swap" if swap else "fwd"
        out.update({f"X_{k}": X, f"Y_{k}": Y, f"P_{k}": P, f"idx_{k}": idx})
    fm = np.concatenate([np.linspace(-3, 3, 64)[:, None], out["X_fwd"][:64]], axis=1)
    out["vc_fm"], out["vc_out"] = fm, npo.vc_frames(npo.GMMMap(w, mu, sig), fm)
    np.savez(os.path.join(OUT, "gmmmap_joint_model.npz"), **out)
    print("written: model_clb_and_slt_gmm32_order40.npz, gmmmap_joint_model.npz")


if __name__ == "__main__":
    main()
