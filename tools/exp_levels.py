"""Kernel time per reduced tile size: 'noise' frames shrink uniformly, the factor picks the level."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 3)
def timeit(fn, n=200):
    for _ in range(100): fn()  # clocks settle after ~80 launches (tools/exp_ramp.py)
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(n): fn()
    ms = h.last_kernel_ms(); h.enable_timing(False)
    return ms
lod = timeit(lambda: h.lod_frames_device(frames, 32, 32, 1, 16.0))
print("lod only", round(lod, 4))
for factor in (16.0, 8.0, 5.6, 4.0, 2.8, 2.0, 1.4, 1.0, 0.7, 0.35, 0.09):
    out = h.shrink_frames_device(frames, 32, 32, 1, 4, factor)
    ms = timeit(lambda: h.shrink_frames_device(frames, 32, 32, 1, 4, factor, out=out))
    ow, oh = out[1], out[2]
    key = (ow.long() * 100 + oh.long()).flatten()
    u, c = torch.unique(key, return_counts=True)
    top = sorted(zip(c.tolist(), u.tolist()), reverse=True)[:2]
    print(f"factor {factor}: {ms:.4f} ms  (+{ms - lod:.4f})  tiles: " + ", ".join(f"{k // 100}x{k % 100}:{n}" for n, k in top))
