"""Does the placement of the output buffers (relative to the frames) move the step time?  Same process, same frames;
the four output tensors are re-allocated behind dummy allocations of different sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
def timeit(fn, n=200):
    for _ in range(100): fn()
    torch.cuda.synchronize(); h.enable_timing(True)
    for _ in range(n): fn()
    ms = h.last_first_kernel_ms(); h.last_kernel_ms(); h.enable_timing(False)
    return ms
keep = []
for pad_mb in (0, 1, 3, 5, 17, 64, 129, 0):
    keep.append(torch.empty(max(pad_mb, 0) * (1 << 20) + 256, dtype=torch.uint8, device="cuda"))
    out = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
    ms = timeit(lambda: h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0, out=out))
    print("pad %3d MB: slots at 0x%x (mod 2^21 = 0x%x): kernel %.4f ms" % (pad_mb, out[3].data_ptr(), out[3].data_ptr() & ((1 << 21) - 1), ms), flush=True)
    keep.append(out)
