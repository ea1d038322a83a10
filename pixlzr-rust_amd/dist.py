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


def gather_files(file_offsets, buf, dst=0, group=None):
    """Variable-length gather of finished .pixlzr files (pxz_encode_frames_device output) to `dst`.

    file_offsets int64[n+1] and buf u8[>= file_offsets[-1]] live on this rank's device.  Returns on dst a
    list (rank order) of (offsets int64[n_r+1] on CPU, bytes u8 tensor on the device); None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = buf.device
    n = file_offsets.numel()
    sizes = torch.stack([file_offsets[-1].to(torch.int64), torch.tensor(n, dtype=torch.int64, device=dev)])
    all_sizes = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes, group=group)
    all_sizes = all_sizes.view(world, 2).cpu()  # the one host sync of the step
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


def run_pipelined(n_steps, produce, exchange):
    """The strong-scaling step loop (bench.py `strong_scaling`, BASELINE configs[4]): produce(i) enqueues the shrink and
    the device writer of step i into buffer set i & 1, exchange(i) ships that step's files to the writer rank.
    produce(i + 1) is issued BEFORE exchange(i), so on a GPU (exchange on its own stream, behind an event of produce(i))
    the files of step i travel while the kernels of step i + 1 run; two buffer sets are enough because exchange(i - 1)
    has returned -- its sends are complete -- before produce(i + 1) overwrites its set."""
    if n_steps <= 0:
        return
    produce(0)
    for i in range(n_steps):
        if i + 1 < n_steps:
            produce(i + 1)
        exchange(i)
