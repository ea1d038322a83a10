"""Directional step + detector-only kernel time (8 x 8K frames, 32x32 tiles) under the env knobs given."""
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
lod = timeit(lambda: h.lod_frames_device(frames, 32, 32, 1, 16.0))
out = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
full = timeit(lambda: h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0, out=out))
knobs = {k: v for k, v in os.environ.items() if k.startswith("PXZ_")}
print(knobs, "lod %.4f ms  step %.4f ms" % (lod, full))
