import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, sara_amd
from sara_amd.synth import synth_batch
W,H,B=1920,1080,64
frames=torch.from_numpy(synth_batch(W,H,B,unique=8)).to("cuda:0")
p=sara_amd.ImagePyramidParams(0,6,num_octaves_max=4)
with sara_amd.SiftContext(W,H,B,p,device=0) as c:
    ts=[]
    for i in range(80):
        t=time.perf_counter(); c.detect_device(frames.data_ptr(),B,W,H); c.synchronize(); ts.append(time.perf_counter()-t)
    print(" ".join("%.2f"%(1e3*t) for t in ts))
    time.sleep(2.0)
    ts=[]
    for i in range(20):
        t=time.perf_counter(); c.detect_device(frames.data_ptr(),B,W,H); c.synchronize(); ts.append(time.perf_counter()-t)
    print("after 2 s idle:", " ".join("%.2f"%(1e3*t) for t in ts))
