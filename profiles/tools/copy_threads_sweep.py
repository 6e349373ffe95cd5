"""PCIe-inclusive BGR rate against the number of host threads that share a staging copy (option copy_threads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (HIP runtime first)
import bench
from funscript_flow_amd import _capi
from funscript_flow_amd.synth import sine_translate_frames

W, H, B = 1920, 1080, 32
base = sine_translate_frames(17, W, H, seed=1)
_capi.set_option("lanes", 2)
for n in (1, 2, 4, 6, 8, 12):
    _capi.set_option("copy_threads", n)
    r = [bench.pcie_inclusive(W, H, B, 0, 1, 8 * B + 1, True, base)["value"] for _ in range(2)]
    print("copy_threads", n, "BGR pairs/s", [round(v) for v in r], flush=True)
