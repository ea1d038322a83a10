"""Multi-GPU plumbing: one process per GPU, frames sharded across ranks, ONE exchange step.

Tiles (and therefore frames) are independent in the reference — the 3x3 detector window never
leaves its tile (src/operations.rs:220-237) and the resample is per tile — so the path shards
with no data-path collective.  The only exchange is the final gather of each rank's block
stream (shrunk tile bytes + value/w/h per tile) to the writer rank, which assembles the
.pixlzr files (src/encoding/mod.rs:40-89).  Backend-agnostic: "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, world, rank):
    """Contiguous, balanced frame ranges: rank r gets frames [lo, hi)."""
    lo = n_frames * rank // world
    hi = n_frames * (rank + 1) // world
    return range(lo, hi)


def gather_block_streams(values, tile_w, tile_h, packed, packed_len, dst=0, group=None):
    """Variable-length gather of the per-rank block streams to `dst`.

    values f32[n], tile_w/tile_h i32[n], packed u8[>=packed_len] live on this rank's device
    (CUDA under nccl, CPU under gloo).  packed_len may be a Python int or a 0-d tensor.
    Returns on dst a list (rank order) of dicts {values, tile_w, tile_h, packed}; None elsewhere.
    One small all-gather of sizes, then point-to-point sends straight to dst (on MI355X every
    sender has its own xGMI link to the writer rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = values.device
    n = values.numel()
    sizes = torch.zeros(2, dtype=torch.int64, device=dev)
    sizes[0] = packed_len
    sizes[1] = n
    all_sizes = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(world, 2).cpu()  # the one host sync of the step: recv lengths
    meta = torch.stack([values.reshape(-1).view(torch.int32), tile_w.reshape(-1).to(torch.int32),
                        tile_h.reshape(-1).to(torch.int32)])
    my_len = int(all_sizes[rank, 0])
    if rank == dst:
        out = [None] * world
        ops = []
        for r in range(world):
            plen, rn = int(all_sizes[r, 0]), int(all_sizes[r, 1])
            if r == rank:
                out[r] = {"meta": meta, "packed": packed[:my_len]}
                continue
            rmeta = torch.empty((3, rn), dtype=torch.int32, device=dev)
            rpacked = torch.empty(plen, dtype=torch.uint8, device=dev)
            out[r] = {"meta": rmeta, "packed": rpacked}
            ops.append(dist.P2POp(dist.irecv, rmeta, r, group))
            if plen:
                ops.append(dist.P2POp(dist.irecv, rpacked, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return [{"values": o["meta"][0].view(torch.float32), "tile_w": o["meta"][1], "tile_h": o["meta"][2],
                 "packed": o["packed"]} for o in out]
    ops = [dist.P2POp(dist.isend, meta, dst, group)]
    if my_len:
        ops.append(dist.P2POp(dist.isend, packed[:my_len].contiguous(), dst, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return None


class PendingFiles:
    """Step 1 of the file gather (gather_files_begin): the sizes of every rank's files are on their way to the host."""
    __slots__ = ("file_offsets", "buf", "all_sizes", "event", "group")


def gather_files_begin(file_offsets, buf, group=None):
    """First half of gather_files: all-gathers (payload bytes, offset count) of every rank and starts their copy to the
    host WITHOUT waiting for it (pinned memory + an event under CUDA), so the caller can go on enqueueing the next
    step's kernels; gather_files_finish waits for the sizes and moves the files.  Everything is issued on the
    caller's current stream."""
    world = dist.get_world_size(group)
    dev = buf.device
    n = file_offsets.numel()
    sizes = torch.stack([file_offsets[-1].to(torch.int64), torch.tensor(n, dtype=torch.int64, device=dev)])
    all_sizes = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    pend = PendingFiles()
    pend.file_offsets, pend.buf, pend.group, pend.event = file_offsets, buf, group, None
    if dev.type == "cuda":
        host = torch.empty(2 * world, dtype=torch.int64, pin_memory=True)
        host.copy_(all_sizes, non_blocking=True)
        pend.event = torch.cuda.Event()
        pend.event.record(torch.cuda.current_stream(dev))
        pend.all_sizes = host
    else:
        pend.all_sizes = all_sizes
    return pend


def gather_files_finish(pend, dst=0):
    """Second half of gather_files: waits (host) for the sizes, then point-to-point sends straight to `dst`.  Returns on
    dst a list (rank order) of (offsets int64[n_r+1], bytes u8 tensor), both on the device; None elsewhere.  Under
    nccl the returned tensors and the send buffers are only ordered on the CURRENT STREAM: whoever overwrites `buf`
    next must wait for an event recorded on this stream after this call (run_pipelined does)."""
    group, file_offsets, buf = pend.group, pend.file_offsets, pend.buf
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = buf.device
    if pend.event is not None:
        pend.event.synchronize()  # the one host wait of the step: recv lengths (long done when a step of lag is kept)
    all_sizes = pend.all_sizes.view(world, 2)
    my_len = int(all_sizes[rank, 0])
    if rank == dst:
        out, ops = [None] * world, []
        for r in range(world):
            blen, rn = int(all_sizes[r, 0]), int(all_sizes[r, 1])
            if r == rank:
                out[r] = (file_offsets, buf[:my_len])
                continue
            roffs = torch.empty(rn, dtype=torch.int64, device=dev)
            rbuf = torch.empty(blen, dtype=torch.uint8, device=dev)
            out[r] = (roffs, rbuf)
            ops.append(dist.P2POp(dist.irecv, roffs, r, group))
            if blen:
                ops.append(dist.P2POp(dist.irecv, rbuf, r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    ops = [dist.P2POp(dist.isend, file_offsets.contiguous(), dst, group)]
    if my_len:
        ops.append(dist.P2POp(dist.isend, buf[:my_len].contiguous(), dst, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return None


def gather_files(file_offsets, buf, dst=0, group=None):
    """Variable-length gather of finished .pixlzr files (pxz_encode_frames_device output) to `dst`.

    file_offsets int64[n+1] and buf u8[>= file_offsets[-1]] live on this rank's device.  Returns on dst a
    list (rank order) of (offsets int64[n_r+1], bytes u8 tensor on the device); None elsewhere.  One small all-gather
    of sizes (the step's host sync), then direct sends (on MI355X every sender owns an xGMI link to the writer)."""
    return gather_files_finish(gather_files_begin(file_offsets, buf, group), dst)


PIPELINE_SETS = 3


def run_pipelined(n_steps, produce, begin, finish, compute=None, comm=None, n_sets=PIPELINE_SETS):
    """The strong-scaling step loop (bench.py `strong_scaling`, BASELINE configs[4]).

      produce(i)   enqueues the shrink and the device writer of step i into buffer set i % n_sets
      begin(i)     starts the exchange of step i: the sizes travel (gather_files_begin); returns a token
      finish(i, token)  moves the files of step i to the writer rank (gather_files_finish)

    Host order: produce(0); then for every i: produce(i + 1), begin(i), finish(i - 1).  The kernels of step i + 1 are
    enqueued before the host looks at anything of step i, and the sizes of step i are only waited for one iteration
    later, when they have long arrived: the host never stands between two steps of kernels.

    Buffer sets: the files of step i are read by the sends of finish(i), which is ISSUED after produce(i + 2) has been
    enqueued, so two sets are not enough: n_sets >= 3, and produce(i) may not overwrite set i % n_sets before the
    sends of step i - n_sets are complete.  Under nccl a send's `wait()` only orders the comm stream -- it neither
    blocks the host nor says the bytes have left -- so that ordering is made with events here: with `compute` and
    `comm` streams given, produce runs on `compute` behind the `sent` event of the set it is about to overwrite and
    records `ready`; begin/finish run on `comm` behind `ready`, and finish records `sent`.  (Under gloo the sends
    complete inside finish and the streams are None.)"""
    assert n_sets >= 3
    if n_steps <= 0:
        return
    gpu = compute is not None
    ready = [torch.cuda.Event() for _ in range(n_sets)] if gpu else None
    sent = [None] * n_sets

    def do_produce(i):
        s = i % n_sets
        if gpu:
            with torch.cuda.stream(compute):
                if sent[s] is not None:
                    compute.wait_event(sent[s])
                produce(i)
                ready[s].record(compute)
        else:
            produce(i)

    def do_begin(i):
        if gpu:
            with torch.cuda.stream(comm):
                comm.wait_event(ready[i % n_sets])
                return begin(i)
        return begin(i)

    def do_finish(i, token):
        if gpu:
            with torch.cuda.stream(comm):
                finish(i, token)
                ev = torch.cuda.Event()
                ev.record(comm)
                sent[i % n_sets] = ev
        else:
            finish(i, token)

    do_produce(0)
    token_prev = None
    for i in range(n_steps):
        if i + 1 < n_steps:
            do_produce(i + 1)
        token = do_begin(i)
        if i >= 1:
            do_finish(i - 1, token_prev)
        token_prev = token
    do_finish(n_steps - 1, token_prev)


def exit_on_timeout(seconds, on_expiry=None, code=3):
    """Watchdog for a leg that may hang inside a collective (a peer died, a link is down): after `seconds` it calls
    on_expiry() (e.g. print what is known) and ends THIS process with a non-zero code -- no new work is started,
    nothing is re-executed.  Returns the timer; cancel() it when the leg is through."""
    import os
    import sys
    import threading

    def fire():
        try:
            if on_expiry is not None:
                on_expiry()
            print(f"[pixlzr dist] no progress within {seconds} s: giving up", file=sys.stderr, flush=True)
        finally:
            os._exit(code)

    timer = threading.Timer(seconds, fire)
    timer.daemon = True
    timer.start()
    return timer
