"""The CPU-baseline rows BASELINE.md section 3 lists, beyond the one frame bench.py times: the oracle (the port of the
reference path) on this box's host cores, synthetic "opaque" frames at 1920x1080, 7680x4320 (32x32 tiles) and 16384^2
(16/32/64), both callers, Lanczos3, all cores (best of 5) and one thread (best of 5; 16384^2: best of 2, a run is 5-20 s),
plus benches/base.png at 64x64 (shrink_by(CatmullRom, 0.25): the reference's own Criterion case).

  python3 tools/cpu_baseline_table.py > gpurun_out/rNN_cpu_baseline.json      (about 4 minutes)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as oracle
oracle.build()
cores = max(1, min(len(os.sched_getaffinity(0)), 16))
MODES = {"shrink_directionally": (1, 16.0), "shrink_by": (0, 1.0)}


def best_of(n, fn):
    fn()
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


rows = []
for (w, h, blocks) in ((1920, 1080, (32,)), (7680, 4320, (32,)), (16384, 16384, (16, 32, 64))):
    img = oracle.synth_frame(w, h, 4, 0, 0)
    mp = w * h / 1e6
    for b in blocks:
        for name, (mode, factor) in MODES.items():
            t_all = best_of(5, lambda: oracle.shrink_image(img, b, b, mode, 4, factor, nthreads=cores))
            t_one = best_of(5 if mp < 100 else 2, lambda: oracle.shrink_image(img, b, b, mode, 4, factor, nthreads=1))
            rows.append({"frame": f"{w}x{h}", "tile": b, "caller": name, "factor": factor, "cores": cores,
                         "all_cores_mp_per_s": round(mp / t_all, 1), "one_thread_mp_per_s": round(mp / t_one, 2)})
            print(rows[-1], file=sys.stderr, flush=True)
from PIL import Image
base = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "base.png")).convert("RGBA")).copy()
mp = base.shape[0] * base.shape[1] / 1e6
for what, filt, k in (("shrink_by(CatmullRom, 0.25)", 2, 0.25), ("shrink_by(CatmullRom, 1.0)", 2, 1.0)):
    t = best_of(5, lambda: oracle.shrink_image(base, 64, 64, 0, filt, k, nthreads=1))
    rows.append({"frame": "benches/base.png 1080x1617", "tile": 64, "caller": what, "cores": 1, "ms": round(t * 1e3, 2),
                 "one_thread_mp_per_s": round(mp / t, 2), "reference_log": "88.4 ms (log_24-09-26.txt:6, hardware unstated)" if k == 0.25 else None})
print(json.dumps({"_how": __doc__.strip(), "host_cores_used": cores, "rows": rows}, indent=1))
