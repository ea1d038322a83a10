"""RGB 64x64 only (for a kernel trace): both callers, 8 x 8K frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 3, 0, 0)
for mode, factor in ((1, 16.0), (0, 1.0)):
    out = h.shrink_frames_device(frames, 64, 64, mode, 4, factor)
    for _ in range(20): h.shrink_frames_device(frames, 64, 64, mode, 4, factor, out=out)
    torch.cuda.synchronize()
