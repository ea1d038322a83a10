"""Size-class histogram of the bench workload (8 x 8K RGBA, 32x32 tiles, Lanczos3), both callers."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
for mode, factor in ((1, 16.0), (0, 1.0)):
    bs = int(os.environ.get("BLOCK", "32"))
    vals, ow, oh, slots = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
    w = ow.cpu().numpy().ravel().astype(int); hh = oh.cpu().numpy().ravel().astype(int)
    c = collections.Counter(zip(w.tolist(), hh.tolist()))
    n = len(w)
    print("mode", mode, "tiles", n)
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print("  %2dx%-2d %7d  %5.1f %%" % (k[0], k[1], v, 100.0 * v / n))
