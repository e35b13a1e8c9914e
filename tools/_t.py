import os, sys
sys.path.insert(0, "."); os.environ["VCMI_TEST_HOOKS"]="1"
import numpy as np, torch, synthdata as sd
import voiceconversion_jl_amd as vc
T=1_000_000
w,mu,sig=sd.synth_model(1002,80,64)
X=sd.sample_frames(1002,w,mu,sig,T,0,40)
Xd=torch.from_numpy(X).cuda(); Yd=torch.empty_like(Xd)
g=vc.GMMMap(w,np.asfortranarray(mu.T),np.asfortranarray(np.transpose(sig,(2,1,0))))
for rep in range(3):
    for _ in range(5): vc.fvconvert(g,Xd.t(),out=Yd.t())
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): vc.fvconvert(g,Xd.t(),out=Yd.t())
    e1.record(); torch.cuda.synchronize()
    print("convert %.4f ms per step"%(e0.elapsed_time(e1)/20))
