"""Detector-only kernel time (8 x 8K frames, 32x32 tiles); PXZ_LIB / PXZ_WPB pick the build and the waves per block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, int(os.environ.get("DIST", "0")))
def timeit(fn, n=200):
    for _ in range(100): fn()  # clocks settle after ~80 launches (tools/exp_ramp.py)
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(n): fn()
    ms = h.last_kernel_ms(); h.enable_timing(False)
    return ms
print(os.environ.get("PXZ_LIB", "default"), "WPB", os.environ.get("PXZ_WPB", "-"), "lod only %.4f ms" % timeit(lambda: h.lod_frames_device(frames, 32, 32, 1, 16.0)))
