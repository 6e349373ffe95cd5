import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from funscript_flow_amd import _capi
_capi.set_option("lanes", 2)
out = {}
for name, bgr, pinned in (("gray", False, False), ("bgr", True, False), ("bgr_pinned", True, True)):
    out[name] = round(bench.pcie_inclusive(1920, 1080, 32, 0, 1, 8 * 32 + 1, bgr, None, pinned)["value"])
print(out)
