"""shrink_by (Oklab detector) at several tile sizes, 8 x 8K RGBA frames: the reference CLI's defaults are 64x64 + shrink_by."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
for bs in (32, 64, 16):
    if 4320 % bs and False: continue
    out = h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0)
    for _ in range(5): h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0, out=out)
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(10): h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0, out=out)
    ms = h.last_kernel_ms(); h.enable_timing(False)
    print("shrink_by %dx%d: %.3f ms per 8 frames" % (bs, bs, ms))
    del out
