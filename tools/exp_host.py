"""Host-buffer boundary (pxz_shrink_image): wall clock per call, PCIe both ways included.
Output buffers are allocated and touched once (a caller that reuses its buffers)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
L = P.load_library()
def ptr(a): return C.c_void_p(a.ctypes.data)
for (w, hh) in ((1920, 1080), (7680, 4320)):
    img = h.synth_frames_device(1, hh, w, 4, 0, 0)[0].cpu().numpy()
    # reference points: plain copies of the same bytes
    d = torch.empty(img.nbytes, dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(img.reshape(-1))
    pin = src.pin_memory()
    for name, s in (("pageable", src), ("pinned", pin)):
        for _ in range(2): d.copy_(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): d.copy_(s)
        torch.cuda.synchronize()
        print("  H2D %s %.1f MB: %.2f ms" % (name, img.nbytes / 1e6, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    for bs in (32, 64):
        cols, rows = (w + bs - 1) // bs, (hh + bs - 1) // bs
        n = cols * rows
        vals = np.ones(n, np.float32); ow = np.ones(n, np.uint32); oh = np.ones(n, np.uint32)
        slots = np.ones((n, bs * bs * 4), np.uint8)
        for mode, factor in ((1, 16.0), (0, 1.0)):
            def call(px=True):
                rc = L.pxz_shrink_image(h._h, ptr(img), w, hh, 4, img.strides[0], bs, bs, mode, 4, C.c_float(factor),
                                        ptr(vals), ptr(ow), ptr(oh), ptr(slots) if px else None)
                assert rc == 0, rc
            stream = np.ones(img.nbytes, np.uint8)
            total = C.c_uint64(0)
            def packed():
                rc = L.pxz_shrink_image_packed(h._h, ptr(img), w, hh, 4, img.strides[0], bs, bs, mode, 4, C.c_float(factor),
                                               ptr(vals), ptr(ow), ptr(oh), C.byref(total))
                assert rc == 0, rc
                assert L.pxz_fetch_packed(h._h, ptr(stream), total.value) == 0
            for _ in range(3): packed()
            t0 = time.perf_counter()
            for _ in range(10): packed()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print("%dx%d b%d mode%d packed stream: %.2f ms per call = %.0f MP/s (%.1f MB back)" % (w, hh, bs, mode, ms, w * hh / ms / 1e3, total.value / 1e6), flush=True)
            for px in (True, False):
                for _ in range(3): call(px)
                t0 = time.perf_counter()
                for _ in range(10): call(px)
                ms = (time.perf_counter() - t0) / 10 * 1e3
                print("%dx%d b%d mode%d %s: %.2f ms per call = %.0f MP/s (valid payload %.1f MB of %.1f MB)" % (
                    w, hh, bs, mode, "with pixels" if px else "LOD only", ms, w * hh / ms / 1e3,
                    (ow.astype(np.int64) * oh).sum() * 4 / 1e6, img.nbytes / 1e6), flush=True)
