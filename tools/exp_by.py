"""shrink_by (Oklab detector) step time, 8 x 8K frames, 32x32 tiles; PXZ_LIB picks the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, int(os.environ.get("DIST", "0")))
out = h.shrink_frames_device(frames, 32, 32, 0, 4, 1.0)
for _ in range(30): h.shrink_frames_device(frames, 32, 32, 0, 4, 1.0, out=out)
torch.cuda.synchronize()
h.enable_timing(True)
for _ in range(60): h.shrink_frames_device(frames, 32, 32, 0, 4, 1.0, out=out)
ms = h.last_kernel_ms(); h.enable_timing(False)
print(os.environ.get("PXZ_LIB", "default"), "shrink_by step %.4f ms" % ms)
