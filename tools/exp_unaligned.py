"""RGBA frames whose width is not a multiple of the tile (ragged right column), rows 16-byte aligned: device times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
full = h.synth_frames_device(1, 4320, 7680, 4, 0, 0)
for width in (7680, 7678):
    frames = full[:, :, :width]  # a view: pitch stays 30720
    for bs in (32, 64):
        for mode, factor in ((1, 16.0), (0, 1.0)):
            out = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
            for _ in range(3): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out)
            torch.cuda.synchronize()
            h.enable_timing(True)
            for _ in range(10): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out)
            ms = h.last_kernel_ms(); h.enable_timing(False)
            print("%dx4320 b%d mode%d: %.3f ms device" % (width, bs, mode, ms), flush=True)
