"""Oklab detector kernel alone (first kernel of a shrink_by step) and the whole step, 8 x 8K frames; PXZ_LIB picks the build,
BLOCK the tile size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
bs = int(os.environ.get("BLOCK", "32"))
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, int(os.environ.get("DIST", "0")))
out = h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0)
for _ in range(60): h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0, out=out)
torch.cuda.synchronize()
h.enable_timing(True)
for _ in range(100): h.shrink_frames_device(frames, bs, bs, 0, 4, 1.0, out=out)
first = h.last_first_kernel_ms(); ms = h.last_kernel_ms(); h.enable_timing(False)
print(os.path.basename(os.environ.get("PXZ_LIB", "default")), "block %d: oklab kernel %.4f ms, step %.4f ms" % (bs, first, ms), flush=True)
