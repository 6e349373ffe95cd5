import os, sys, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from funscript_flow_amd import backend, pipeline
from funscript_flow_amd.synth import sine_translate_frames
W=H=256; N=3000
base = sine_translate_frames(17, W, H, seed=1)
frames = [base[i % 17][:] for i in range(N)]
params = {"backend": "HIP"}
pairs=list(zip(frames[:-1], frames[1:]))
backend.precompute_all(pairs, params)
pr=cProfile.Profile(); pr.enable()
pre=backend.precompute_all(pairs, params)
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2600])
