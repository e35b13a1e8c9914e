#!/bin/bash
# Builds probe libraries tools/_lib_persexp<N>.so with gmmmap.hip compiled -DVCMI_PERS_EXP=<N> (the other objects as built) and
# times fvconvert with each (LIBVCMI_PROBE): what the persistent screen kernel's time is made of.   tools/pers_exp.sh build | run
R=$(cd $(dirname $0)/.. && pwd); C=$R/voiceconversion.jl_amd/csrc
EXPS="${EXPS:-32}"
if [ "$1" = build ]; then
  for e in $EXPS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -DVCMI_PERS_EXP=$e -c $C/gmmmap.hip -o /tmp/gmmmap_exp$e.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lib_persexp$e.so /tmp/gmmmap_exp$e.o $(ls $C/_obj/*.o | grep -v gmmmap.o) -ldl -lpthread && echo built $e ) &
    if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
  done; wait
else
  for e in 0 $EXPS; do
    if [ $e = 0 ]; then unset LIBVCMI_PROBE; else export LIBVCMI_PROBE=$R/tools/_lib_persexp$e.so; fi
    python3 - <<PY
import os, sys, json
sys.path.insert(0, "$R"); os.environ["VCMI_TEST_HOOKS"]="1"
import numpy as np, torch, synthdata as sd
import voiceconversion_jl_amd as vc
T=1_000_000
w,mu,sig=sd.synth_model(1002,80,64)
X=sd.sample_frames(1002,w,mu,sig,T,0,40)
Xd=torch.from_numpy(X).cuda(); Yd=torch.empty_like(Xd)
g=vc.GMMMap(w,np.asfortranarray(mu.T),np.asfortranarray(np.transpose(sig,(2,1,0))))
for _ in range(5): vc.fvconvert(g,Xd.t(),out=Yd.t())
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): vc.fvconvert(g,Xd.t(),out=Yd.t())
e1.record(); torch.cuda.synchronize()
print("exp $e: %.4f ms per step"%(e0.elapsed_time(e1)/20))
PY
  done
fi
