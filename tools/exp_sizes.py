"""Tile sizes off the fast paths (generic kernel): step times per 8 x 8K RGBA frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
for bw, bh in ((8, 8), (24, 24), (48, 48), (32, 16), (128, 64), (100, 100)):
    for mode, factor, name in ((1, 16.0, "directional"), (0, 1.0, "shrink_by")):
        try:
            out = h.shrink_frames_device(frames, bw, bh, mode, 4, factor)
        except Exception as e:
            print(bw, bh, name, "->", e); continue
        for _ in range(3): h.shrink_frames_device(frames, bw, bh, mode, 4, factor, out=out)
        torch.cuda.synchronize()
        h.enable_timing(True)
        for _ in range(5): h.shrink_frames_device(frames, bw, bh, mode, 4, factor, out=out)
        ms = h.last_kernel_ms(); h.enable_timing(False)
        print("%dx%d %s: %.3f ms per 8 frames (%d tiles)" % (bw, bh, name, ms, out[1].numel()), flush=True)
        del out
