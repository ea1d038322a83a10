"""The host-buffer boundary (what the Rust shim of INTEGRATION.md calls) on 8K RGBA frames: per-frame wall clock of
pxz_shrink_image / pxz_shrink_image_packed called frame by frame against pxz_shrink_images / pxz_shrink_images_packed over
the list (upload k+1 | kernels k | download k-1).  The caller's buffers are allocated and touched once, outside the clock."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
L = P.load_library()
n = int(os.environ.get("N", "8"))
W, H, Cc, B = 7680, 4320, 4, 32
frames = h.synth_frames_device(n, H, W, Cc, 0, 0).cpu().numpy()
imgs = [np.ascontiguousarray(frames[k]) for k in range(n)]
cols, rows = P.grid(W, H, B, B)
T = cols * rows
vals = [np.ones(T, np.float32) for _ in range(n)]
ow = [np.ones(T, np.uint32) for _ in range(n)]
oh = [np.ones(T, np.uint32) for _ in range(n)]
px = [np.ones(W * H * Cc, np.uint8) for _ in range(n)]   # slots or packed stream: same worst-case size
lens = np.zeros(n, np.uint64)
ptrs = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
vp = lambda a: C.c_void_p(a.ctypes.data)
for mode, factor, name in ((1, 16.0, "shrink_directionally"), (0, 1.0, "shrink_by")):
    f = C.c_float(factor)
    def single(packed, pixels=True):
        for k in range(n):
            if packed:
                ln = C.c_uint64(0)
                assert L.pxz_shrink_image_packed(h._h, vp(imgs[k]), W, H, Cc, W * Cc, B, B, mode, 4, f, vp(vals[k]), vp(ow[k]), vp(oh[k]), C.byref(ln)) == 0
                assert L.pxz_fetch_packed(h._h, vp(px[k]), ln.value) == 0
            else:
                assert L.pxz_shrink_image(h._h, vp(imgs[k]), W, H, Cc, W * Cc, B, B, mode, 4, f, vp(vals[k]), vp(ow[k]), vp(oh[k]), vp(px[k]) if pixels else None) == 0
    def lst(packed, pixels=True):
        if packed:
            assert L.pxz_shrink_images_packed(h._h, ptrs(imgs), n, W, H, Cc, W * Cc, B, B, mode, 4, f, ptrs(vals), ptrs(ow), ptrs(oh), ptrs(px), W * H * Cc, vp(lens)) == 0
        else:
            assert L.pxz_shrink_images(h._h, ptrs(imgs), n, W, H, Cc, W * Cc, B, B, mode, 4, f, ptrs(vals), ptrs(ow), ptrs(oh), ptrs(px) if pixels else None) == 0
    for label, fn in (("frame by frame, slots", lambda: single(False)), ("list, slots", lambda: lst(False)),
                      ("frame by frame, packed", lambda: single(True)), ("list, packed", lambda: lst(True)),
                      ("frame by frame, detector", lambda: single(False, False)), ("list, detector only", lambda: lst(False, False))):
        fn()
        t0 = time.perf_counter(); fn(); fn(); dt = (time.perf_counter() - t0) / 2
        print("%-22s %-26s %.2f ms per frame  %.1f GP/s" % (name, label, dt / n * 1e3, n * 33.1776e-3 / dt), flush=True)
