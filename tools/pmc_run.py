"""Runs a few launches of one kernel variant (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
variant = sys.argv[1] if len(sys.argv) > 1 else "dir_full"
nf = int(os.environ.get("NF", "8"))
n = int(os.environ.get("N", "3"))
dist = int(os.environ.get("DIST", "0"))
bs = int(os.environ.get("BLOCK", "32"))  # tile size (square)
h = P.Handle(0)
frames = h.synth_frames_device(nf, 4320, 7680, 4, 0, dist)
mode, factor = (1, 16.0) if variant.startswith("dir") else (0, 1.0)
if os.environ.get("FACTOR"):
    factor = float(os.environ["FACTOR"])  # with DIST=3 (noise) the factor picks ONE size class for every tile
if variant == "enc":  # shrink once, then the device writer n times
    out = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
    enc = h.encode_frames_device(tuple(frames.shape), 32, 32, *out)
    for _ in range(n - 1): h.encode_frames_device(tuple(frames.shape), 32, 32, *out, out=enc)
elif variant.endswith("lod"):
    for _ in range(n): h.lod_frames_device(frames, bs, bs, mode, factor)
else:
    out = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
    for _ in range(n - 1): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out)
torch.cuda.synchronize()
print("done", variant)
