"""One timing probe for every flow of the library (replaces the eight one-off exp_*.py scripts of round 3):

  python3 tools/exp.py shrink <dir|by> [block=32] [n=60]     all kernels of the step + its first (dominant) kernel, HIP events
  python3 tools/exp.py writer [block=32] [n=100]             pxz_encode_frames_device after one shrink, wall clock
  python3 tools/exp.py reader [block=32] [n=60]              pxz_decode_frames_device on the writer's files, wall clock
  python3 tools/exp.py expand [block=32] [n=100]             pxz_expand_frames_device per filter (FILTERS="0 2 4"), wall clock
  python3 tools/exp.py hist [block=32]                       size-class histogram of the bench frames, both callers
  python3 tools/exp.py classes [factors...]                  noise frames, one size class per factor: step time per class

env: CH (3|4 channels), DIST (0 opaque .. 3 noise), NF (frames, 8), W, H (7680 x 4320), FILTER (4), PXZ_LIB (another build of the
library: A/B on one box), PXZ_* knobs.  8 x 8K frames unless said otherwise; every line names what it measured.
"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
what = sys.argv[1] if len(sys.argv) > 1 else "shrink"
argv = sys.argv[2:]
ch, dist, nf = int(os.environ.get("CH", "4")), int(os.environ.get("DIST", "0")), int(os.environ.get("NF", "8"))
W, H, flt = int(os.environ.get("W", "7680")), int(os.environ.get("H", "4320")), int(os.environ.get("FILTER", "4"))
lib = os.path.basename(os.environ.get("PXZ_LIB", "default"))
knobs = [k for k in os.environ if k.startswith("PXZ_") and k != "PXZ_LIB"]
h = P.Handle(0)


def wall(fn, warm, n):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if what == "shrink":
    variant, bs, n = argv[0], int(argv[1]) if len(argv) > 1 else 32, int(argv[2]) if len(argv) > 2 else 60
    mode, factor = (1, 16.0) if variant == "dir" else (0, 1.0)
    frames = h.synth_frames_device(nf, H, W, ch, 0, dist)
    out = h.shrink_frames_device(frames, bs, bs, mode, flt, factor)
    for _ in range(40): h.shrink_frames_device(frames, bs, bs, mode, flt, factor, out=out)
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(n): h.shrink_frames_device(frames, bs, bs, mode, flt, factor, out=out)
    first = h.last_first_kernel_ms()
    print(f"{lib} {variant} {bs}x{bs} ch{ch} {nf}x{W}x{H} knobs={knobs}: step kernels {h.last_kernel_ms():.4f} ms, first kernel {first:.4f} ms, state {h.state()}", flush=True)
elif what in ("writer", "reader"):
    bs, n = int(argv[0]) if argv else 32, int(argv[1]) if len(argv) > 1 else 100
    frames = h.synth_frames_device(nf, H, W, ch, 0, dist)
    out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
    enc = h.encode_frames_device(tuple(frames.shape), bs, bs, *out)
    if what == "writer":
        ms = wall(lambda: h.encode_frames_device(tuple(frames.shape), bs, bs, *out, out=enc), 20, n)
        print(f"{lib} writer {bs}x{bs} ch{ch}: {ms:.4f} ms per {nf} frames, {int(enc[0][-1])} file bytes", flush=True)
    else:
        dec = h.decode_frames_device(enc[1], enc[0], tuple(frames.shape), bs, bs)
        assert bool((dec[1] == out[1]).all()) and bool((dec[2] == out[2]).all())
        ms = wall(lambda: h.decode_frames_device(enc[1], enc[0], tuple(frames.shape), bs, bs, out=dec), 10, n)
        print(f"{lib} reader {bs}x{bs} ch{ch}: {ms:.4f} ms per {nf} frames", flush=True)
elif what == "expand":
    bs, n = int(argv[0]) if argv else 32, int(argv[1]) if len(argv) > 1 else 100
    frames = h.synth_frames_device(nf, H, W, ch, 0, dist)
    _, ow, oh, slots = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
    for filt in [int(x) for x in os.environ.get("FILTERS", "0 2 4").split()]:
        back = h.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots)
        ms = wall(lambda: h.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots, out=back), 60, n)
        print(f"{lib} expand {bs}x{bs} ch{ch} filter {filt}: {ms:.4f} ms per {nf} frames", flush=True)
        del back
elif what == "hist":
    bs = int(argv[0]) if argv else 32
    frames = h.synth_frames_device(nf, H, W, ch, 0, dist)
    for mode, factor in ((1, 16.0), (0, 1.0)):
        _, ow, oh, _ = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
        c = collections.Counter(zip(ow.cpu().numpy().ravel().tolist(), oh.cpu().numpy().ravel().tolist()))
        print("mode", mode, "tiles", ow.numel())
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print("  %2dx%-2d %7d  %5.1f %%" % (k[0], k[1], v, 100.0 * v / ow.numel()))
elif what == "classes":
    frames = h.synth_frames_device(nf, H, W, 4, 0, 3)
    for factor in [float(x) for x in (argv or "0.25 0.5 1 2 4 8 16 32 64 128 256".split())]:
        out = h.shrink_frames_device(frames, 32, 32, 1, 4, factor)
        c = collections.Counter(zip(out[1].cpu().numpy().ravel().tolist(), out[2].cpu().numpy().ravel().tolist())).most_common(2)
        ms = wall(lambda: h.shrink_frames_device(frames, 32, 32, 1, 4, factor, out=out), 100, 100)
        print("factor %-6g %-40s %.4f ms" % (factor, str(c), ms), flush=True)
else:
    sys.exit(__doc__)
