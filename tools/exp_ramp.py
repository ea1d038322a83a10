"""Per-launch kernel time of the first launches after start-up (clock / power-state ramp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
out = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "0.5")))
ts = []
for i in range(400):
    h.enable_timing(True)
    h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0, out=out)
    ts.append(h.last_kernel_ms())
    h.enable_timing(False)
for a, b in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, 320), (320, 400)):
    seg = ts[a:b]
    print(f"launches {a:3d}-{b:3d}: mean {sum(seg)/len(seg):.4f} ms  min {min(seg):.4f}")
