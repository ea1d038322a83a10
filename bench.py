#!/usr/bin/env python3
"""bench.py — pixlzr encode hot path on MI355X: megapixels/s + achieved HBM GB/s vs roofline.

A "step" = one pass of the hot path (per-tile LOD detection + power-of-two down-sampling,
Pixlzr::from_image + shrink_* of the reference) over one batch of synthetic 8K RGBA frames that
is already resident in HBM: ONE fused kernel launch (+ a small worklist kernel) per step and GPU.

  python bench.py --gpus 1 --steps K --warmup W            # single GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # one rank per GPU

Workload (BASELINE.json configs[2] / configs[4]): 7680x4320 RGBA8, 32x32 tiles, 8 frames per GPU
(64 frames over 8 GPUs; 1.06 GB of source per GPU per step, well past the 256 MiB Infinity Cache),
"opaque" synthetic distribution, filter Lanczos3.  `value` is weak scaling: per-GPU work is fixed and the
timed steps are the same at every N (frames are independent: no collective on the data path).

The ONE JSON line also carries, beside `value`:
  modes.shrink_by                            the other caller (Oklab detector), same frames
  modes["shrink_directionally+encode_to_vec"]  the step followed by the device QOI + container writer: frames -> .pixlzr bytes
  strong_scaling (N > 1, or --frames-total)  BASELINE configs[4] as it is stated: a FIXED batch of 64 frames sharded over
                                             the ranks, step = shrink + device writer + gather of the files to rank 0 over
                                             RCCL, the gather of step k running under the shrink of step k + 1
  roofline, cpu_baseline                     as the measurement contract asks
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable

MODES = {"shrink_directionally": (1, 16.0), "shrink_by": (0, 1.0)}  # (pxz_mode, factor) per BASELINE.md §3
ENCODE_MODE = "shrink_directionally+encode_to_vec"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--event-stride", type=int, default=8,
                    help="runs of >= 64 steps: every n-th timed step is bracketed by HIP events (16..63 steps: every fourth, shorter runs: every step)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100,
                    help="a step is ~0.3 ms and the chip needs ~80 launches (~30 ms) of load to reach its steady clocks")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed load before the W warm-up steps so that short runs are measured at steady clocks too")
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--mode", choices=list(MODES) + ["both"], default="both",
                    help="`value` is quoted for --primary; `both` also measures the other detector")
    ap.add_argument("--primary", choices=list(MODES), default="shrink_directionally")
    ap.add_argument("--filter", type=int, default=4)
    ap.add_argument("--dist", type=int, default=0, help="0 opaque, 1 alpha, 2 flat, 3 noise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-sizes", action="store_true", help="skip the 64x64 / 16x16 side measurements")
    ap.add_argument("--no-encode-mode", action="store_true", help="skip the shrink + device writer measurement")
    ap.add_argument("--no-block-sweep", action="store_true", help="skip the 16384x16384 block-size sweep (BASELINE configs[3])")
    ap.add_argument("--frames-total", type=int, default=0,
                    help="strong-scaling leg: a fixed batch of this many frames over all ranks (default: 64 when N > 1, off at N = 1)")
    ap.add_argument("--strong-steps", type=int, default=20)
    ap.add_argument("--strong-timeout", type=float, default=240.0, help="seconds the strong-scaling leg may take before it is abandoned")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling leg (shrink + writer + gather to rank 0)")
    return ap.parse_args()


def histogram(ow, oh):
    import torch
    key = ow.long() * 100000 + oh.long()
    uniq, counts = torch.unique(key, return_counts=True)
    return {f"{int(k) // 100000}x{int(k) % 100000}": int(c) for k, c in zip(uniq.tolist(), counts.tolist())}


def note(msg):
    """progress on stderr (rank 0): a run that stops can be placed"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def agree(dist_mod, world, device, ok):
    """Every rank calls this, whatever happened to it: True only if all ranks are fine (an error on one rank must not
    leave the others inside a collective it will never enter)."""
    if world == 1:
        return ok
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MIN)
    return bool(int(t[0]))


def run_mode(args, handle, frames, mode_name, world, dist_mod, with_writer=False, steps=None):
    """Times K steps of one detector mode; returns a dict of measurements (max over ranks).
    A step is the fused hot-path launch over this rank's frames -- the same at every N: the tiles of
    different frames are independent, so the timed loop has no collective.  with_writer: the device QOI +
    container writer (Pixlzr::encode_to_vec) follows the shrink inside every step."""
    import torch
    pxz_mode, factor = MODES[mode_name]
    N, H, W, C = frames.shape
    bw = bh = args.block
    steps = steps or args.steps
    out = handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor)
    vals, ow, oh, slots = out
    enc_out = handle.encode_frames_device(tuple(frames.shape), bw, bh, vals, ow, oh, slots) if with_writer else None

    def step():
        handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor, out=out)
        if with_writer:
            handle.encode_frames_device(tuple(frames.shape), bw, bh, vals, ow, oh, slots, out=enc_out)

    # untimed: bring the chip to its steady clocks (~80 launches / 30 ms of load, measured in round 1), whatever W is
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    spinup_ms = (time.perf_counter() - t_spin) * 1e3
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    # (three event records cost a step ~2 % of its wall clock: also short runs carry them on every fourth step only)
    stride = 1 if steps < 16 else (4 if steps < 64 else max(1, args.event_stride))
    handle.enable_timing(True, every=stride)  # which steps of the timed region carry the events (~2 us each)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream: the first (dominant) kernel of the shrink alone -- the figure rocprofv3's kernel
    # stats show for it -- and all kernels of the last *_device call of the sampled steps
    first_ms = handle.last_first_kernel_ms()
    kernel_ms = handle.last_kernel_ms()
    handle.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms, first_ms], dtype=torch.float64, device=frames.device)
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        elapsed, kernel_ms, first_ms = float(t[0]), float(t[1]), float(t[2])

    tiles = ow.numel()
    out_bytes = int((ow.long() * oh.long()).sum().item()) * C
    read_bytes = N * H * W * C
    algo_bytes = read_bytes + out_bytes + 12 * tiles  # source once + shrunk pixels + value/w/h
    res = {
        "mode": mode_name, "factor": factor, "steps": steps, "spinup_ms": spinup_ms,
        "event_sampled_steps": (steps + stride - 1) // stride,
        "elapsed_s": elapsed, "ms_per_step": elapsed / steps * 1e3,
        "mp_per_s_per_gpu": N * H * W / 1e6 * steps / elapsed,
        "algo_bytes_per_launch": algo_bytes, "read_bytes": read_bytes, "write_bytes": out_bytes + 12 * tiles,
        "histogram": histogram(ow[0], oh[0]),
        # which kernels the handle picked for the timed steps (read AFTER them: the state the last of them ran in)
        "handle_state": handle.state(),
    }
    if with_writer:
        # (the handle's events bracket the shrink's kernels only; the writer's share is the rest of the step)
        offs = enc_out[0]
        file_bytes = int(offs[-1].item())
        res.update({
            "shrink_kernels_ms": kernel_ms, "writer_ms": elapsed / steps * 1e3 - kernel_ms, "file_bytes": file_bytes,
            # the writer reads the valid slot bytes + value/w/h and writes the files
            "algo_bytes_per_launch": algo_bytes + out_bytes + 12 * tiles + file_bytes,
        })
        res["achieved_gbps"] = res["algo_bytes_per_launch"] / (res["ms_per_step"] * 1e-3) / 1e9  # by wall clock of the step
    else:
        res.update({
            "dominant_kernel_ms": first_ms,     # shrink32_kernel (directional) / oklab2_kernel (shrink_by)
            "step_kernels_ms": kernel_ms,       # every kernel of the step
            "kernel_ms": kernel_ms,
            # the measurement contract prices the DOMINANT kernel: the step's algorithmic bytes over that kernel's own average
            # duration (what rocprofv3's kernel stats show for it).  The small worklist kernel behind it finishes ~1 % of the
            # tiles; the figure over every kernel of the step is kept beside it.
            "achieved_gbps": algo_bytes / (first_ms * 1e-3) / 1e9,
            "achieved_gbps_all_kernels": algo_bytes / (kernel_ms * 1e-3) / 1e9,
        })
    return res


def run_strong(args, handle, product, frames_total, world, rank, dist_mod):
    """BASELINE configs[4] as stated: a FIXED batch of `frames_total` frames sharded over the ranks (contiguous ranges,
    dist.shard_frames).  Step = shrink_directionally + device writer (QOI + container: finished .pixlzr files) + gather
    of the files to the writer rank 0.  Three sets of output buffers (dist.run_pipelined): the kernels of step k + 1
    are enqueued before anything of step k is looked at, the sizes of step k cross to the host while they run, and the
    files of step k travel on a second stream; a set is only overwritten behind the event that ends its sends."""
    import torch
    pdist = product.dist
    pxz_mode, factor = MODES["shrink_directionally"]
    mine = pdist.shard_frames(frames_total, world, rank)
    n = len(mine)
    dev = torch.device("cuda", handle.device_id)
    bw = bh = args.block
    ok, err = True, ""
    try:
        frames = handle.synth_frames_device(max(n, 1), args.height, args.width, 4, first_frame=mine.start, dist=args.dist)
        compute, comm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        bufs = []
        with torch.cuda.stream(compute):
            for _ in range(pdist.PIPELINE_SETS):
                out = handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor)
                enc = handle.encode_frames_device(tuple(frames.shape), bw, bh, *out)
                bufs.append((out, enc))
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001
        ok, err = False, f"{type(exc).__name__}: {exc}"[:300]
    if not agree(dist_mod, world, dev, ok):
        return {"error": err or "another rank failed to set the batch up"}
    state = {"bytes": 0, "files": 0, "last": None}

    def produce(i):  # (run_pipelined puts it on `compute`, behind the sends that still read this buffer set)
        out, enc = bufs[i % len(bufs)]
        handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor, out=out)
        handle.encode_frames_device(tuple(frames.shape), bw, bh, *out, out=enc)

    def begin(i):
        offs, buf = bufs[i % len(bufs)][1]
        if world > 1:
            return pdist.gather_files_begin(offs, buf)
        host = torch.empty(offs.numel(), dtype=torch.int64, pin_memory=True)  # one rank: only the sizes cross to the host
        host.copy_(offs, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return host, ev

    def finish(i, token):
        if world > 1:
            got = pdist.gather_files_finish(token, dst=0)
            if got is not None:
                state["bytes"] = sum(int(g[1].numel()) for g in got)
                state["files"] = sum(int(g[0].numel()) - 1 for g in got)
                state["last"] = got
        else:
            host, ev = token
            ev.synchronize()
            state["bytes"], state["files"] = int(host[-1]), int(host.numel()) - 1

    def loop(k):
        pdist.run_pipelined(k, produce, begin, finish, compute=compute, comm=comm)
        torch.cuda.synchronize()

    ok, err, ms = True, "", 0.0
    try:
        note("strong scaling: buffers ready, warm-up exchange")
        loop(4)  # untimed: allocations, RCCL channels
        note("strong scaling: timed loop")
        if world > 1:
            dist_mod.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(args.strong_steps)
        if world > 1:
            dist_mod.barrier()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.strong_steps
    except Exception as exc:  # noqa: BLE001
        ok, err = False, f"{type(exc).__name__}: {exc}"[:300]
    if not agree(dist_mod, world, dev, ok):
        return {"error": err or "another rank failed in the exchange"}
    # what arrived is what one process writes: rank 0 encodes the shard of the LAST rank itself and compares the bytes of
    # the last step (untimed)
    checked = None
    if world > 1 and rank == 0 and state["last"] is not None:
        try:
            other = pdist.shard_frames(frames_total, world, world - 1)
            f2 = handle.synth_frames_device(len(other), args.height, args.width, 4, first_frame=other.start, dist=args.dist)
            out2 = handle.shrink_frames_device(f2, bw, bh, pxz_mode, args.filter, factor)
            offs2, buf2 = handle.encode_frames_device(tuple(f2.shape), bw, bh, *out2)
            roffs, rbuf = state["last"][world - 1]
            torch.cuda.synchronize()
            nb = int(offs2[-1].item())
            checked = bool(torch.equal(roffs.cpu(), offs2.cpu()) and rbuf.numel() == nb and torch.equal(rbuf, buf2[:nb]))
            del f2, out2, offs2, buf2
        except Exception as exc:  # noqa: BLE001
            checked = f"{type(exc).__name__}: {exc}"[:200]
    if world > 1:
        t = torch.tensor([ms], dtype=torch.float64, device=dev)
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        ms = float(t[0])
    mp = frames_total * args.width * args.height / 1e6
    res = {"scaling": "strong", "frames_total": frames_total, "frames_this_rank": n, "steps": args.strong_steps,
           "ms_per_step": ms, "value": mp / (ms * 1e-3), "unit": "MP/s",
           "what": "shrink_directionally + device QOI/container writer + gather of the .pixlzr files to rank 0; "
                   "the gather of step k overlaps the kernels of step k+1 (second stream, three buffer sets, sizes one step behind)",
           "files_at_rank0": state["files"], "bytes_at_rank0": state["bytes"],
           "gathered_files_equal_single_process": checked,
           "backend": (dist_mod.get_backend() if world > 1 else None),
           "rccl_ranks": (dist_mod.get_world_size() if world > 1 and dist_mod.get_backend() == "nccl" else 0)}
    if checked is not None and checked is not True:
        res["error"] = f"the files gathered from rank {world - 1} differ from a single-process encode: {checked}"
    return res


def cpu_baseline(args, primary, names):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded
    sample: ONE frame of the batch, every caller in `names`.  kind="port": the Rust reference itself cannot be
    built here.  Best of 5 after a warm-up run, on all cores over tile rows and on ONE thread
    (what the reference's sequential shrink_* loops use, pixlzr.rs:163-204); plus the reference's own Criterion case
    (benches/bench-00.rs:55,79-81: base.png, 64x64 tiles, shrink_by(CatmullRom, 0.25)) on one thread, to stand beside
    log_24-09-26.txt:6.  About 25 s of CPU work in all; the other rows of BASELINE.md section 3 (1080p, 16384^2 at
    16/32/64) are in profiles/ (tools/cpu_baseline_table.py), not in this line."""
    from oracle import binding as oracle
    oracle.build()
    img = oracle.synth_frame(args.width, args.height, 4, 0, args.dist)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # the GPU box grants ~16 host cores per GPU
    mp = args.width * args.height / 1e6

    def best_of(n, fn):
        fn()  # warm-up
        best = None
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    per_mode = {}
    for name in names:
        pxz_mode, factor = MODES[name]
        best_all = best_of(5, lambda: oracle.shrink_image(img, args.block, args.block, pxz_mode, args.filter, factor, nthreads=cores))
        best_one = best_of(5, lambda: oracle.shrink_image(img, args.block, args.block, pxz_mode, args.filter, factor, nthreads=1))
        per_mode[name] = {"value": mp / best_all, "unit": "MP/s", "cores": cores, "single_thread_value": mp / best_one,
                          "single_thread_runs": "best of 5 after one warm-up"}
    p = per_mode[primary]
    res = {"value": p["value"], "unit": "MP/s", "cores": cores, "kind": "port",
           "sample": f"1 frame {args.width}x{args.height} RGBA8, {primary}, best of 5, {cores} threads over tile rows",
           "single_thread_value": p["single_thread_value"], "modes": per_mode,
           "reference_published": "88.4 ms / 1.746 MP = 19.8 MP/s single thread (shrink_by, 64x64), hardware unstated (log_24-09-26.txt:6)"}
    try:  # the reference's own bench case on this box's cores, one thread
        import numpy as np
        from PIL import Image
        base = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "base.png")).convert("RGBA")).copy()
        t = best_of(5, lambda: oracle.shrink_image(base, 64, 64, 0, 2, 0.25, nthreads=1))
        res["reference_bench_case"] = {"what": "benches/base.png 1080x1617 RGBA, 64x64 tiles, shrink_by(CatmullRom, 0.25), one thread, best of 5",
                                       "ms": t * 1e3, "mp_per_s": base.shape[0] * base.shape[1] / 1e6 / t,
                                       "reference_log_ms": 88.4}
    except Exception as exc:  # noqa: BLE001
        res["reference_bench_case"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    return res


def load_traffic(mode_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh: one flow per pass, every kernel of
    the library counted), if present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(mode_name)
    except Exception:
        return None


def load_flow_traffic(flow):
    """the same file's per-flow record (dir32, by32, by64, dir64, by16, dir16, enc32): bytes per step and per kernel"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["flows"][flow]
        return {"hbm_bytes_per_step": rec["hbm_bytes_per_step"], "traffic_over_algorithmic": rec["traffic_over_algorithmic"],
                "kernels": {k: v["hbm_bytes_per_step"] for k, v in rec["kernels"].items()}}
    except Exception:
        return None


def load_sq(kernel_prefix):
    """vector-ALU figures of a kernel from the committed SQ counter passes (profiles/sq_summary.json, tools/sq_summary.py)"""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_summary.json")) as f:
            for name, rec in json.load(f)["kernels"].items():
                if name.startswith(kernel_prefix):
                    return dict(rec, kernel=name)
    except Exception:
        pass
    return None


def mode_roofline(name, r):
    """roofline object of a mode other than the headline: the step's algorithmic bytes over the time of ALL its kernels (by
    events), the counter bytes of the committed PMC passes beside them, and -- where the dominant kernel is not bound by
    bytes at all -- what it is bound by"""
    if name == ENCODE_MODE:
        ach = r["achieved_gbps"]
        sq = load_sq("qoi_tiles_kernel")
        out = {"bound": "issue", "bound_is": "the writer's dominant kernel (qoi_tiles_kernel) is bound by vector-instruction issue at the occupancy its LDS "
                                             "index tables allow, not by bytes; hbm_utilisation is the step's algorithmic bytes over its wall clock",
               "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "hbm_utilisation": ach / HBM_PEAK_GBPS,
               "valu_busy": (sq or {}).get("valu_busy_at_4_waves_per_simd"),
               "traffic": load_traffic(ENCODE_MODE), "algorithmic_bytes_per_launch": r["algo_bytes_per_launch"],
               "time_ms": r["ms_per_step"], "time_is": "wall clock of the step (shrink + writer)",
               "kernels": "shrink32_kernel<1> + qoi_bin_* + qoi_tiles_kernel<4> + pack_scan_* + qoi_splice_* + qoi_headers_kernel",
               "writer_traffic": load_flow_traffic("enc32"),
               "what_bounds_it": "qoi_tiles_kernel: a lane per segment of 64..128 pixels, ~100 vector instructions per pixel at 9 waves per CU "
                                 "(its 16 KB index table per wave fills LDS); its 8-byte piece stores reach HBM as partial sectors (see writer_traffic)",
               "sq": sq}
        return out
    ach = r["achieved_gbps_all_kernels"]
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "traffic": load_traffic(name), "algorithmic_bytes_per_launch": r["algo_bytes_per_launch"],
           "time_ms": r["step_kernels_ms"], "time_is": "all kernels of the step, HIP events",
           "dominant_kernel_ms": r["dominant_kernel_ms"]}
    if name == "shrink_by":
        sq = load_sq("oklab2_kernel<32")
        # the dominant kernel is bound by vector-instruction issue: the machine-readable bound says so, `frac` is the share of SIMD
        # cycles with a vector instruction executing (committed SQ counter passes), and the byte rate is kept as hbm_utilisation
        out.update({"bound": "valu", "hbm_utilisation": ach / HBM_PEAK_GBPS, "achieved_gbps": ach})
        busy = (sq or {}).get("valu_busy_at_4_waves_per_simd")  # (the detector's blocks are 16 waves: 4 per SIMD)
        if busy is not None:
            out.update({"achieved": busy, "peak": 1.0, "unit": "share of SIMD cycles with a vector instruction executing", "frac": busy})
        out["kernels"] = "oklab2_kernel<32> (dominant) + shrink32_kernel<0> + worklist kernel"
        out["flow_traffic"] = load_flow_traffic("by32")
        out["what_bounds_it"] = ("vector-instruction issue, not bytes: the detector reproduces glibc's cbrtf bit for bit (three per pixel, f64 path) "
                                 "and replays the reference's sequential f32 sums; sq.valu_busy is the share of SIMD cycles with a vector "
                                 "instruction executing")
        out["sq"] = sq
    return out




def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switch for a ONE-GPU box: all ranks on cuda:0, gloo instead of RCCL (RCCL refuses
    # two ranks on one device).  Never set by the driver.
    same_gpu = os.environ.get("PXZ_BENCH_SAME_GPU") == "1"
    device_index = 0 if same_gpu else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device_index)
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(device_index)

    from __graft_entry__ import load_product
    product = load_product()
    handle = product.Handle(device_index)  # raises if the HIP library / device is missing: no fallback
    nf = args.frames_per_gpu
    frames = handle.synth_frames_device(nf, args.height, args.width, 4, first_frame=rank * nf, dist=args.dist)
    torch.cuda.synchronize()

    names = list(MODES) if args.mode == "both" else [args.mode]
    primary = args.primary if args.primary in names else names[0]
    results = {}
    for name in names:
        note(f"{name}: {args.steps} steps")
        results[name] = run_mode(args, handle, frames, name, world, dist)
    if not args.no_encode_mode and "shrink_directionally" in names:
        note(ENCODE_MODE)
        results[ENCODE_MODE] = run_mode(args, handle, frames, "shrink_directionally", world, dist, with_writer=True,
                                        steps=min(args.steps, 200))

    def timed_shrink(fr, bs, pxz_mode, factor, flow):
        """one configuration off the headline: 20 untimed + 40 timed launches, every launch bracketed by events; a roofline
        object of its own (algorithmic bytes of SURVEY 8d over the time of ALL kernels of the step; the dominant kernel's own
        time beside it; counter bytes from the committed PMC pass of the same flow)"""
        out = handle.shrink_frames_device(fr, bs, bs, pxz_mode, args.filter, factor)
        for _ in range(20):
            handle.shrink_frames_device(fr, bs, bs, pxz_mode, args.filter, factor, out=out)
        torch.cuda.synchronize()
        handle.enable_timing(True)
        for _ in range(40):
            handle.shrink_frames_device(fr, bs, bs, pxz_mode, args.filter, factor, out=out)
        first = handle.last_first_kernel_ms()
        ms = handle.last_kernel_ms()
        handle.enable_timing(False)
        _, ow, oh, _ = out
        algo = fr.numel() + int((ow.long() * oh.long()).sum().item()) * fr.shape[-1] + 12 * ow.numel()
        mp = fr.shape[0] * fr.shape[1] * fr.shape[2] / 1e6
        tr = load_flow_traffic(flow)
        rec = {"kernel_ms": ms, "dominant_kernel_ms": first, "mp_per_s": mp / (ms * 1e-3), "tiles": int(ow.numel()),
               "handle_state": handle.state(),
               "roofline": {"bound": "hbm" if pxz_mode == 1 else "valu", "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "frac_dominant_kernel": algo / (first * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "algorithmic_bytes_per_launch": algo, "time_is": "all kernels of the step, HIP events on the launch stream, 40 launches",
                            "traffic": (tr or {}).get("hbm_bytes_per_step"),
                            "traffic_over_algorithmic": (tr or {}).get("traffic_over_algorithmic"),
                            "traffic_source": f"profiles/pmc_traffic.json flow {flow} (committed rocprofv3 --pmc passes), not measured in this run"}}
        if pxz_mode == 0:
            rec["roofline"]["bound_is"] = ("the Oklab detector in front of the shrink kernel is bound by vector-instruction issue (bit-exact glibc cbrtf), "
                                           "not bytes: frac is the byte rate it leaves, for comparison only")
        del out
        return rec

    # not the metric: the same frames at the reference CLI's default 64x64 tiles and at 16x16
    others = {}
    if world == 1 and not args.no_other_sizes:
        for bs in (64, 16):
            for name, (pxz_mode, factor) in MODES.items():
                others[f"{bs}x{bs} {name}"] = timed_shrink(frames, bs, pxz_mode, factor, ("dir" if pxz_mode == 1 else "by") + str(bs))

    # not the metric either: the way back (Pixlzr::expand + to_image) of what shrink_directionally left, per filter
    # and tile size (32x32; the reference CLI's default 64x64; 16x16), and the reader in front of it
    decode_side = {}

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    if world == 1 and not args.no_other_sizes:
        for bs in (32, 64, 16):
            suffix = "" if bs == 32 else f" {bs}x{bs}"
            vals2, ow, oh, slots = handle.shrink_frames_device(frames, bs, bs, 1, args.filter, MODES["shrink_directionally"][1])
            stored = int((ow.long() * oh.long()).sum().item()) * 4
            for label, filt in (("expand Nearest", 0), ("expand Lanczos3", 4)):
                back = handle.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots)
                ms = timed(lambda: handle.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots, out=back), 20, 40)
                algo = stored + 8 * ow.numel() + back.numel()  # stored pixels + sizes read, frames written
                tr = load_flow_traffic(f"exp{bs}") if filt == args.filter else None
                decode_side[label + suffix] = {
                    "ms_per_step": ms, "time_is": "40 launches between two events on the launch stream",
                    "roofline": {"bound": "hbm", "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": algo,
                                 "traffic": (tr or {}).get("hbm_bytes_per_step"), "traffic_over_algorithmic": (tr or {}).get("traffic_over_algorithmic")}}
                del back
            # the reader (decode_from_vec: index of the records + the QOI decoder) on the files the writer makes
            offs, buf = handle.encode_frames_device(tuple(frames.shape), bs, bs, vals2, ow, oh, slots)
            dec = handle.decode_frames_device(buf, offs, tuple(frames.shape), bs, bs)
            rd_ms = timed(lambda: handle.decode_frames_device(buf, offs, tuple(frames.shape), bs, bs, out=dec), 5, 20)
            rd_algo = int(offs[-1].item()) + stored + 12 * ow.numel()  # files read; pixels, values, sizes written
            rd_tr = load_flow_traffic(f"dec{bs}")
            decode_side["decode (reader)" + suffix] = {
                "ms_per_step": rd_ms, "file_bytes": int(offs[-1].item()), "time_is": "20 launches between two events on the launch stream",
                "roofline": {"bound": "latency", "bound_is": "one lane per tile walks a serial op stream: the launch lasts as long as its longest lanes "
                             f"({bs * bs} dependent pixel steps of a full {bs}x{bs} tile, DESIGN 6c); the byte rate is for comparison only",
                             "achieved": rd_algo / (rd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": rd_algo / (rd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": rd_algo,
                             "traffic": (rd_tr or {}).get("hbm_bytes_per_step"),
                             "traffic_over_algorithmic": (rd_tr or {}).get("traffic_over_algorithmic")}}
            del vals2, ow, oh, slots, offs, buf, dec

    # BASELINE configs[3]: ONE 16384 x 16384 RGBA frame (1.07 GB), tiles of 16 / 32 / 64 px, both callers
    block_sweep = {}
    if world == 1 and not args.no_other_sizes and not args.no_block_sweep:
        note("block sweep on one 16384x16384 frame")
        del frames
        torch.cuda.empty_cache()
        big = handle.synth_frames_device(1, 16384, 16384, 4, first_frame=0, dist=args.dist)
        for bs in (16, 32, 64):
            for name, (pxz_mode, factor) in MODES.items():
                block_sweep[f"{bs}x{bs} {name}"] = timed_shrink(big, bs, pxz_mode, factor, ("sqdir" if pxz_mode == 1 else "sqby") + str(bs))
        del big
        torch.cuda.empty_cache()
        frames = None

    line = None
    if rank == 0:
        r = results[primary]
        total_mp = world * nf * args.width * args.height / 1e6
        keep = ("ms_per_step", "mp_per_s_per_gpu", "steps", "handle_state", "event_sampled_steps", "dominant_kernel_ms", "step_kernels_ms", "kernel_ms",
                "shrink_kernels_ms", "writer_ms", "file_bytes", "achieved_gbps", "achieved_gbps_all_kernels", "algo_bytes_per_launch", "histogram")
        line = {
            "metric": "encode megapixels/sec (per-tile LOD detection + block-wise downsample), 8K RGBA",
            "value": total_mp * r["steps"] / r["elapsed_s"],
            "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_ms": r["spinup_ms"],
            "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if primary == "shrink_directionally" else "u8/f32",
            "data": "synthetic",
            "config": {"workload": f"{nf}x {args.width}x{args.height} RGBA8 frames per GPU, {args.block}x{args.block} tiles, "
                                   f"{primary} factor {r['factor']}, filter {args.filter} (Lanczos3=4), dist {args.dist}, "
                                   f"device-resident, one fused launch per step",
                       "mode": primary, "frames_per_gpu": nf, "tile": args.block,
                       "parallelism": f"{world} ranks x {nf} frames, frames sharded over ranks, no collective in the timed steps"
                                      + ("; see strong_scaling for the fixed 64-frame batch with the gather to rank 0 in the step" if world > 1 else ""),
                       "tile_size_histogram_frame0": r["histogram"]},
            # `achieved` / `frac`: the step's algorithmic bytes over the step's WALL CLOCK (the conservative figure: launch gaps and the
            # worklist kernel are inside it); the dominant kernel's own duration by HIP events -- the figure a rocprofv3 kernel trace
            # shows for it -- is beside it
            "roofline": {"bound": "hbm", "achieved": r["algo_bytes_per_launch"] / (r["ms_per_step"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": r["algo_bytes_per_launch"] / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "frac_is": "algorithmic bytes / wall clock of the timed steps",
                         "achieved_dominant_kernel": r["achieved_gbps"], "frac_dominant_kernel": r["achieved_gbps"] / HBM_PEAK_GBPS,
                         "frac_all_kernels_of_step": r["achieved_gbps_all_kernels"] / HBM_PEAK_GBPS,
                         "handle_state": r["handle_state"],
                         "traffic": load_traffic(primary),
                         "traffic_source": "profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (committed), not measured in this run",
                         "kernel_ms": r["kernel_ms"], "dominant_kernel_ms": r["dominant_kernel_ms"], "step_kernels_ms": r["step_kernels_ms"],
                         "event_sampled_steps": r["event_sampled_steps"],
                         "algorithmic_bytes_per_launch": r["algo_bytes_per_launch"],
                         "kernel": "pxz::shrink32_kernel<1, true> (achieved_dominant_kernel = the step's algorithmic bytes / this kernel's average duration by HIP events; frac_all_kernels_of_step adds the worklist kernel's time)"
                                   if primary == "shrink_directionally" else "pxz::oklab2_kernel<32> + pxz::shrink32_kernel<0, true>"},
            "modes": {k: {kk: v[kk] for kk in keep if kk in v} for k, v in results.items()},
        }
        for k, v in results.items():
            if k != primary:
                line["modes"][k]["roofline"] = mode_roofline(k, v)
        if others:
            line["other_tile_sizes"] = others
        if decode_side:
            line["decode_side"] = decode_side
        if block_sweep:
            line["block_sweep"] = {"workload": "1x 16384x16384 RGBA8 frame (BASELINE configs[3]), factor as in MODES, filter as --filter", **block_sweep}
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(args, primary, names)

    # the strong-scaling leg comes last and under a watchdog: whatever happens to the exchange, the line above is printed
    frames_total = args.frames_total or (64 if world > 1 else 0)
    failed = False
    if frames_total and not args.no_strong:
        def on_expiry():  # a leg stuck in a collective: say what is known, then the process ends non-zero (dist.exit_on_timeout)
            if rank == 0:
                line["strong_scaling"] = {"error": f"no result within {args.strong_timeout} s; the leg was abandoned"}
                print(json.dumps(line), flush=True)

        dog = product.dist.exit_on_timeout(args.strong_timeout, on_expiry, code=3)
        del frames
        torch.cuda.empty_cache()
        note(f"strong scaling: {frames_total} frames over {world} ranks, {args.strong_steps} steps")
        try:
            strong = run_strong(args, handle, product, frames_total, world, rank, dist)
        except Exception as exc:  # noqa: BLE001  (the line with the headline metric is printed whatever this leg does)
            strong = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        dog.cancel()
        note("strong scaling done")
        failed = isinstance(strong, dict) and "error" in strong
        if rank == 0:
            line["strong_scaling"] = strong
    if rank == 0:
        print(json.dumps(line), flush=True)
    if failed:
        # a leg that failed ends the process non-zero, at once: a rank that failed in the exchange may have left the
        # others inside a collective, so there is no barrier to wait in and no teardown to run
        sys.stdout.flush()
        os._exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
