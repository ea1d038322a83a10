"""The reference's own Criterion case (benches/bench-00.rs:79-81, log_24-09-26.txt:6: 88.4 ms on its author's CPU):
base.png (1080x1617 RGBA8) -> from_image(64, 64) -> shrink_by(CatmullRom, 0.25).  Host-buffer call (PCIe both
ways) and device-resident call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "base.png"))))
H, W, C = img.shape
for _ in range(5): h.shrink_image(img, 64, 64, 0, 2, 0.25)
t0 = time.perf_counter()
for _ in range(50): h.shrink_image(img, 64, 64, 0, 2, 0.25)
host_ms = (time.perf_counter() - t0) / 50 * 1e3
frames = torch.from_numpy(img)[None].cuda()
out = h.shrink_frames_device(frames, 64, 64, 0, 2, 0.25)
for _ in range(100): h.shrink_frames_device(frames, 64, 64, 0, 2, 0.25, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): h.shrink_frames_device(frames, 64, 64, 0, 2, 0.25, out=out)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / 500 * 1e3
mp = H * W / 1e6
print({"image": f"{W}x{H}x{C}", "host_call_ms": round(host_ms, 3), "host_call_mp_per_s": round(mp / host_ms * 1e3),
       "device_resident_ms": round(dev_ms, 4), "device_resident_mp_per_s": round(mp / dev_ms * 1e3),
       "reference_published_ms": 88.4})
