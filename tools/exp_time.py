"""Kernel time of one flow, by the handle's HIP events: python3 tools/exp_time.py <dir|by> <block> [n_timed] [filter] [dist]
prints the mean of all kernels of the step and of its first (dominant) kernel over n_timed launches after 30 warm-up ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
variant, bs = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
flt = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dist = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ch = int(os.environ.get("CH", "4"))
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, ch, 0, dist)
mode, factor = (1, 16.0) if variant == "dir" else (0, 1.0)
out = h.shrink_frames_device(frames, bs, bs, mode, flt, factor)
for _ in range(30): h.shrink_frames_device(frames, bs, bs, mode, flt, factor, out=out)
torch.cuda.synchronize()
h.enable_timing(True)
for _ in range(n): h.shrink_frames_device(frames, bs, bs, mode, flt, factor, out=out)
torch.cuda.synchronize()
first = h.last_first_kernel_ms()
print(f"{variant} {bs}x{bs} ch{ch} knobs={[k for k in os.environ if k.startswith('PXZ_')]}: step kernels {h.last_kernel_ms():.4f} ms, first kernel {first:.4f} ms")
