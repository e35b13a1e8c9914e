#!/usr/bin/env python3
"""Profiling target: BASELINE configs[1] fvconvert, a few calls and nothing else (tools/sq_pmc.py runs it under rocprofv3).
    python3 tools/convert_once.py [calls] [debug_force flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VCMI_TEST_HOOKS"] = "1"
import numpy as np
import torch

import synthdata as sd
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
force = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T = 1_000_000
w, mu, sig = sd.synth_model(1002, 80, 64)
X = sd.sample_frames(1002, w, mu, sig, T, 0, 40)
Xd = torch.from_numpy(X).cuda()
Yd = torch.empty_like(Xd)
g = vc.GMMMap(w, np.asfortranarray(mu.T), np.asfortranarray(np.transpose(sig, (2, 1, 0))))
_lib.debug_force(force)
for _ in range(calls):
    vc.fvconvert(g, Xd.t(), out=Yd.t())
torch.cuda.synchronize()
