#!/usr/bin/env python3
"""bench.py — pixlzr encode hot path on MI355X: megapixels/s + achieved HBM GB/s vs roofline.

A "step" = one pass of the hot path (per-tile LOD detection + power-of-two down-sampling,
Pixlzr::from_image + shrink_* of the reference) over one batch of synthetic 8K RGBA frames that
is already resident in HBM: ONE fused kernel launch per step and per GPU.

  python bench.py --gpus 1 --steps K --warmup W            # single GPU
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # one rank per GPU

Workload (BASELINE.json configs[2] / configs[4]): 7680x4320 RGBA8, 32x32 tiles, 8 frames per GPU
(64 frames over 8 GPUs; 1.06 GB of source per GPU per step, well past the 256 MiB Infinity Cache),
"opaque" synthetic distribution, filter Lanczos3.  Weak scaling: per-GPU work is fixed; the timed
steps are the same at every N (frames are independent: no collective on the data path).  N > 1:
the final block-stream gather (device-encoded .pixlzr files -> rank 0 over RCCL) runs once after
the timed region and is reported as config.final_gather.  Prints ONE JSON line on rank 0.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--event-stride", type=int, default=8,
                    help="every n-th timed step is bracketed by HIP events (kernel durations for the roofline)")
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
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the final block-stream gather to rank 0")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: encode + gather the files inside every timed step instead of once after the timed region")
    ap.add_argument("--with-bitstream", action="store_true",
                    help="N=1: also run the device QOI + container writer inside every step (always on when gathering)")
    return ap.parse_args()


def histogram(ow, oh):
    import torch
    key = ow.long() * 100000 + oh.long()
    uniq, counts = torch.unique(key, return_counts=True)
    return {f"{int(k) // 100000}x{int(k) % 100000}": int(c) for k, c in zip(uniq.tolist(), counts.tolist())}


def run_mode(args, handle, frames, mode_name, rank, world, dist_mod, pdist):
    """Times K steps of one detector mode; returns a dict of measurements (max over ranks).
    A step is the fused hot-path launch over this rank's frames -- the same at every N: the tiles of
    different frames are independent, so the timed loop has no collective.  N > 1: after the timed
    region the path's one exchange step runs once -- the block streams are encoded on the device
    (QOI + container) and the finished files gathered to the writer rank 0 over RCCL -- and is
    reported beside the metric (`final_gather`).  --gather-every-step puts it inside every step."""
    import torch
    pxz_mode, factor = MODES[mode_name]
    N, H, W, C = frames.shape
    bw = bh = args.block
    out = handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor)
    vals, ow, oh, slots = out
    gather = world > 1 and not args.no_gather
    in_step = gather and args.gather_every_step
    bitstream = in_step or args.with_bitstream
    enc_out = handle.encode_frames_device(tuple(frames.shape), bw, bh, vals, ow, oh, slots) if (bitstream or gather) else None
    gather_state = {"bytes": 0}

    def exchange():
        # encode_to_vec on the device: QOI tiles + container -> finished .pixlzr files, then the file bytes
        # to the writer rank
        offs, buf = handle.encode_frames_device(tuple(frames.shape), bw, bh, vals, ow, oh, slots, out=enc_out)
        got = pdist.gather_files(offs, buf, dst=0)
        if got is not None:
            gather_state["bytes"] = sum(int(g[1].numel()) for g in got)

    def step():
        handle.shrink_frames_device(frames, bw, bh, pxz_mode, args.filter, factor, out=out)
        if in_step:
            exchange()
        elif bitstream:
            handle.encode_frames_device(tuple(frames.shape), bw, bh, vals, ow, oh, slots, out=enc_out)

    # untimed: bring the chip to its steady clocks (~80 launches / 30 ms of load, tools/exp_ramp.py), whatever W is
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    handle.enable_timing(True, every=args.event_stride)  # which steps of the timed region carry the events (~2 us each)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # HIP events on the launch stream, averaged over the K launches: the first (dominant) kernel of the step alone
    # -- the figure rocprofv3's kernel stats show for it -- and all kernels of the step
    first_ms = handle.last_first_kernel_ms()
    kernel_ms = handle.last_kernel_ms()
    handle.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms, first_ms], dtype=torch.float64, device=frames.device)
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        elapsed, kernel_ms, first_ms = float(t[0]), float(t[1]), float(t[2])

    final_gather = None
    if gather:
        # outside the metric: a failure here is reported, it does not take the measured line with it
        try:
            exchange()  # untimed warm-up of the exchange (allocations, RCCL channels)
            torch.cuda.synchronize()
            dist_mod.barrier()
            t1 = time.perf_counter()
            exchange()
            torch.cuda.synchronize()
            dist_mod.barrier()
            gms = (time.perf_counter() - t1) * 1e3
            t = torch.tensor([gms], dtype=torch.float64, device=frames.device)
            dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
            final_gather = {"ms": float(t[0]), "bytes_at_rank0": gather_state["bytes"],
                            "what": "device QOI + container of this rank's frames, then gather of the .pixlzr files to rank 0",
                            "inside_timed_steps": bool(in_step)}
        except Exception as exc:  # noqa: BLE001
            final_gather = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    tiles = ow.numel()
    out_bytes = int((ow.long() * oh.long()).sum().item()) * C
    read_bytes = N * H * W * C
    algo_bytes = read_bytes + out_bytes + 12 * tiles  # source once + shrunk pixels + value/w/h
    mp = N * H * W / 1e6
    return {
        "mode": mode_name, "factor": factor,
        "elapsed_s": elapsed, "ms_per_step": elapsed / args.steps * 1e3,
        "mp_per_s_per_gpu": mp * args.steps / elapsed,
        # directional steps: shrink32_kernel moves all of these bytes (the worklist kernel behind it re-reads a
        # percent of the tiles and finishes the values): its own duration prices the roofline.  shrink_by steps: the
        # detector kernel and the fused kernel together.
        "kernel_ms": first_ms if pxz_mode == 1 else kernel_ms,
        "step_kernels_ms": kernel_ms,
        "algo_bytes_per_launch": algo_bytes, "read_bytes": read_bytes, "write_bytes": out_bytes + 12 * tiles,
        "achieved_gbps": algo_bytes / ((first_ms if pxz_mode == 1 else kernel_ms) * 1e-3) / 1e9,
        "histogram": histogram(ow[0], oh[0]),
        "final_gather": final_gather,
        "bitstream_in_step": bool(bitstream),
    }


def cpu_baseline(args, mode_name):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a bounded
    sample: ONE frame of the batch.  kind="port": the Rust reference itself cannot be built here."""
    from oracle import binding as oracle
    oracle.build()
    pxz_mode, factor = MODES[mode_name]
    img = oracle.synth_frame(args.width, args.height, 4, 0, args.dist)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # the GPU box grants ~16 host cores per GPU
    mp = args.width * args.height / 1e6
    best_all = best_one = None
    for _ in range(2):
        t0 = time.perf_counter()
        oracle.shrink_image(img, args.block, args.block, pxz_mode, args.filter, factor, nthreads=cores)
        dt = time.perf_counter() - t0
        best_all = dt if best_all is None else min(best_all, dt)
    t0 = time.perf_counter()
    oracle.shrink_image(img, args.block, args.block, pxz_mode, args.filter, factor, nthreads=1)
    best_one = time.perf_counter() - t0
    return {"value": mp / best_all, "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"1 frame {args.width}x{args.height} RGBA8, {mode_name}, best of 2, {cores} threads over tile rows",
            "single_thread_value": mp / best_one,
            "reference_published": "88.4 ms / 1.746 MP = 19.8 MP/s single thread, hardware unstated (log_24-09-26.txt:6)"}


def load_traffic(mode_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command, if present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(mode_name)
    except Exception:
        return None


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
        results[name] = run_mode(args, handle, frames, name, rank, world, dist, product.dist)

    # not the metric: the same frames at the reference CLI's default 64x64 tiles and at 16x16 (kernel time only)
    others = {}
    if world == 1 and not args.no_other_sizes:
        for bs in (64, 16):
            for name, (pxz_mode, factor) in MODES.items():
                out = handle.shrink_frames_device(frames, bs, bs, pxz_mode, args.filter, factor)
                for _ in range(20):
                    handle.shrink_frames_device(frames, bs, bs, pxz_mode, args.filter, factor, out=out)
                torch.cuda.synchronize()
                handle.enable_timing(True)
                for _ in range(40):
                    handle.shrink_frames_device(frames, bs, bs, pxz_mode, args.filter, factor, out=out)
                ms = handle.last_kernel_ms()
                handle.enable_timing(False)
                others[f"{bs}x{bs} {name}"] = {"kernel_ms": ms, "mp_per_s": nf * args.width * args.height / 1e6 / (ms * 1e-3)}
                del out
    if rank == 0:
        r = results[primary]
        total_mp = world * nf * args.width * args.height / 1e6
        line = {
            "metric": "encode megapixels/sec (per-tile LOD detection + block-wise downsample), 8K RGBA",
            "value": total_mp * args.steps / r["elapsed_s"],
            "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if primary == "shrink_directionally" else "u8/f32",
            "data": "synthetic",
            "config": {"workload": f"{nf}x {args.width}x{args.height} RGBA8 frames per GPU, {args.block}x{args.block} tiles, "
                                   f"{primary} factor {r['factor']}, filter {args.filter} (Lanczos3=4), dist {args.dist}, "
                                   f"device-resident, one fused launch per step",
                       "mode": primary, "frames_per_gpu": nf, "tile": args.block,
                       "parallelism": f"{world} ranks x {nf} frames, frames sharded over ranks, no collective in the timed steps"
                                      + ("; device-encoded .pixlzr files gathered to rank 0 "
                                         + ("inside every step" if args.gather_every_step else "once after the timed region")
                                         if world > 1 and not args.no_gather else ""),
                       "final_gather": r["final_gather"],
                       "tile_size_histogram_frame0": r["histogram"]},
            "roofline": {"bound": "hbm", "achieved": r["achieved_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": r["achieved_gbps"] / HBM_PEAK_GBPS, "traffic": load_traffic(primary),
                         "kernel_ms": r["kernel_ms"], "step_kernels_ms": r["step_kernels_ms"], "algorithmic_bytes_per_launch": r["algo_bytes_per_launch"],
                         "kernel": "pxz::shrink32_kernel<1, true>" if primary == "shrink_directionally" else "pxz::oklab_kernel<32> + pxz::shrink32_kernel<0, true>"},
            "modes": {k: {kk: v[kk] for kk in ("ms_per_step", "mp_per_s_per_gpu", "kernel_ms", "step_kernels_ms", "achieved_gbps",
                                                "algo_bytes_per_launch", "histogram")} for k, v in results.items()},
        }
        if others:
            line["other_tile_sizes"] = others
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, primary)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
