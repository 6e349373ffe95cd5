import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from funscript_flow_amd import _capi
from funscript_flow_amd.pipeline import SMOOTH_RADIUS
from funscript_flow_amd.synth import sine_translate_frames
W,H,B=1920,1080,32
_capi.set_option("lanes", int(os.environ.get("LANES","1")))
fr=sine_translate_frames(B+1,W,H,seed=1)
ctx=_capi.Context(W,H,frame_slots=B+2,flow_slots=3*B,max_batch=B)
ctx.upload_frames(0,list(fr)); ctx.sync()
if os.environ.get("EVENTS"): ctx.profile_enable(["k_blur_solve"])
r=bench.StepRunner(ctx,B,False,SMOOTH_RADIUS)
stamps=[]
orig=ctx.pass1_results
def traced(*a,**k):
    x=orig(*a,**k); stamps.append(time.perf_counter()); return x
ctx.pass1_results=traced
r.run(10); stamps.clear()
t0=time.perf_counter(); r.run(150); dt=time.perf_counter()-t0
d=np.diff(np.array(stamps))*1e3
print("ms/step %.3f  median delta %.3f  p95 %.3f  max %.3f"%(dt/150*1e3, np.median(d), np.percentile(d,95), d.max()))
print("deltas:", " ".join("%.2f"%v for v in d))
