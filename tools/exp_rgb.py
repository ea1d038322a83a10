"""RGB against RGBA input, 8 x 8K frames, both callers, by tile size: kernel time of a step (events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
for c in (3, 4):
    frames = h.synth_frames_device(8, 4320, 7680, c, 0, 0)
    for bs in (32, 64, 16):
        for mode, factor, name in ((1, 16.0, "shrink_directionally"), (0, 1.0, "shrink_by")):
            out = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
            for _ in range(40): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(60): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out)
            torch.cuda.synchronize()
            print("c=%d %2dx%-2d %-22s %.3f ms per step (wall clock)" % (c, bs, bs, name, (time.perf_counter() - t0) / 60 * 1e3), flush=True)
            del out
    del frames
