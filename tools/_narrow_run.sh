for probe in "" tools/_lib_narrow.so; do
for force in 0 268435456; do
LIBVCMI_PROBE=$probe python3 - <<PY
import os, sys
sys.path.insert(0, "."); os.environ["VCMI_TEST_HOOKS"]="1"
import numpy as np, torch, synthdata as sd
import voiceconversion_jl_amd as vc
from voiceconversion_jl_amd import _lib
T=1_000_000
w,mu,sig=sd.synth_model(1002,80,64)
X=sd.sample_frames(1002,w,mu,sig,T,0,40)
Xd=torch.from_numpy(X).cuda(); Yd=torch.empty_like(Xd)
g=vc.GMMMap(w,np.asfortranarray(mu.T),np.asfortranarray(np.transpose(sig,(2,1,0))))
_lib.debug_force($force)
for _ in range(5): vc.fvconvert(g,Xd.t(),out=Yd.t())
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): vc.fvconvert(g,Xd.t(),out=Yd.t())
e1.record(); torch.cuda.synchronize()
print("probe='$probe' force=$force: %.4f ms per step"%(e0.elapsed_time(e1)/20))
PY
done; done
