"""A few launches of one tile size on a 16384^2 frame (for rocprofv3 --pmc): BS=16|32|64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
bs = int(os.environ.get("BS", "16"))
frames = h.synth_frames_device(1, 16384, 16384, 4, 0, 0)
out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
for _ in range(2): h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0, out=out)
torch.cuda.synchronize()
print("done", bs, out[1].numel())
