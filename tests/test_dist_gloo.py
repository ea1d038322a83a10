"""N>1 path on CPU: world_size 2 over gloo.  Each rank shrinks its shard of frames (with the oracle
standing in for the GPU kernel, which needs no GPU-side exchange), the block streams are gathered
to rank 0 with the product's dist plumbing, and the writer rank's .pixlzr files must equal the
single-process result byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, B, NF = 160, 96, 32, 5


def _shrink_frames(oracle, frame_ids, mode, factor):
    vals, ws, hs, chunks = [], [], [], []
    for f in frame_ids:
        img = oracle.synth_frame(W, H, 4, f, 1)
        v, ow, oh, slots = oracle.shrink_image(img, B, B, mode, 4, factor)
        vals.append(v)
        ws.append(ow)
        hs.append(oh)
        for t in range(len(ow)):
            chunks.append(slots[t, : int(ow[t]) * int(oh[t]) * 4])
    packed = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return np.concatenate(vals), np.concatenate(ws), np.concatenate(hs), packed


def _files_from_stream(product, values, tw, th, packed, n_frames):
    """Writer rank: cut the gathered stream back into frames and write the containers."""
    tiles = len(values) // n_frames
    files, off = [], 0
    for f in range(n_frames):
        sl = slice(f * tiles, (f + 1) * tiles)
        slots = np.zeros((tiles, B * B * 4), np.uint8)
        for i, (w, h) in enumerate(zip(tw[sl], th[sl])):
            nb = int(w) * int(h) * 4
            slots[i, :nb] = packed[off:off + nb]
            off += nb
        files.append(product.encode_container(W, H, B, B, 4, 0, values[sl], None, tw[sl].astype(np.uint32),
                                              th[sl].astype(np.uint32), slots))
    assert off == len(packed)
    return files


def _worker(rank, world, port, mode, factor, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.conftest import load_product
    from oracle import binding as oracle
    product = load_product()
    from pixlzr_rust_amd import dist as pdist
    mine = pdist.shard_frames(NF, world, rank)
    v, tw, th, packed = _shrink_frames(oracle, mine, mode, factor)
    got = pdist.gather_block_streams(torch.from_numpy(v), torch.from_numpy(tw.astype(np.int32)),
                                     torch.from_numpy(th.astype(np.int32)), torch.from_numpy(packed.copy()),
                                     len(packed), dst=0)
    if rank == 0:
        files = []
        for r, part in enumerate(got):
            nfr = len(pdist.shard_frames(NF, world, r))
            files += _files_from_stream(product, part["values"].numpy(), part["tile_w"].numpy(), part["tile_h"].numpy(),
                                        part["packed"].numpy(), nfr)
        np.save(result_path, np.array([len(f) for f in files]))
        with open(result_path + ".bin", "wb") as fh:
            fh.write(b"".join(files))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,factor", [(1, 8.0), (0, 0.5)])
def test_two_ranks_gather_equals_single_process(tmp_path, product, oracle, mode, factor):
    assert list(product.dist.shard_frames(5, 2, 0)) == [0, 1] and list(product.dist.shard_frames(5, 2, 1)) == [2, 3, 4]
    assert [len(product.dist.shard_frames(64, 8, r)) for r in range(8)] == [8] * 8
    result = str(tmp_path / "sizes.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, mode, factor, result), nprocs=2, join=True)
    sizes = np.load(result)
    blob = open(result + ".bin", "rb").read()
    assert len(sizes) == NF
    off = 0
    for f in range(NF):
        img = oracle.synth_frame(W, H, 4, f, 1)
        v, ow, oh, slots = oracle.shrink_image(img, B, B, mode, 4, factor)
        ref = oracle.encode_container(W, H, B, B, 4, 0, v, None, ow, oh, slots)
        assert blob[off:off + int(sizes[f])] == ref, f"frame {f}"
        off += int(sizes[f])


def _worker_files(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.conftest import load_product
    from oracle import binding as oracle
    product = load_product()
    from pixlzr_rust_amd import dist as pdist
    files = []
    for f in pdist.shard_frames(NF, world, rank):
        img = oracle.synth_frame(W, H, 4, f, 1)
        v, ow, oh, slots = oracle.shrink_image(img, B, B, 1, 4, 8.0)
        files.append(product.encode_container(W, H, B, B, 4, 0, v, None, ow, oh, slots))  # product's writer, host side
    offs = np.concatenate([[0], np.cumsum([len(x) for x in files])]).astype(np.int64)
    buf = np.frombuffer(b"".join(files) + bytes(64), np.uint8).copy()  # capacity larger than the payload
    got = pdist.gather_files(torch.from_numpy(offs), torch.from_numpy(buf), dst=0)
    if rank == 0:
        blob, sizes = b"", []
        for roffs, rbuf in got:
            roffs = roffs.numpy()
            for i in range(len(roffs) - 1):
                blob += rbuf.numpy()[roffs[i]:roffs[i + 1]].tobytes()
                sizes.append(int(roffs[i + 1] - roffs[i]))
        np.save(result_path, np.array(sizes))
        open(result_path + ".bin", "wb").write(blob)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_files(tmp_path, product, oracle):
    """gather_files: each rank's finished .pixlzr files arrive on rank 0 in frame order, byte-exact."""
    result = str(tmp_path / "fsizes.npy")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_files, args=(2, port, result), nprocs=2, join=True)
    sizes = np.load(result)
    blob = open(result + ".bin", "rb").read()
    off = 0
    for f in range(NF):
        img = oracle.synth_frame(W, H, 4, f, 1)
        v, ow, oh, slots = oracle.shrink_image(img, B, B, 1, 4, 8.0)
        ref = oracle.encode_container(W, H, B, B, 4, 0, v, None, ow, oh, slots)
        assert blob[off:off + int(sizes[f])] == ref
        off += int(sizes[f])
    assert off == len(blob)


def _worker_strong(rank, world, port, result_path, n_steps):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.conftest import load_product
    from oracle import binding as oracle
    product = load_product()
    from pixlzr_rust_amd import dist as pdist
    mine = pdist.shard_frames(NF, world, rank)  # the fixed batch of NF frames, contiguous shards
    bufs = [None] * pdist.PIPELINE_SETS         # three buffer sets, as in bench.py
    got_steps = []
    order = []

    def produce(i):
        # step i "shrinks" its frames with a step-dependent factor, so that a stale buffer set would be noticed
        order.append(("produce", i))
        files = []
        for f in mine:
            img = oracle.synth_frame(W, H, 4, f, 1)
            v, ow, oh, slots = oracle.shrink_image(img, B, B, 1, 4, 4.0 + 4.0 * i)
            files.append(product.encode_container(W, H, B, B, 4, 0, v, None, ow, oh, slots))
        offs = np.concatenate([[0], np.cumsum([len(x) for x in files])]).astype(np.int64)
        bufs[i % len(bufs)] = (torch.from_numpy(offs), torch.from_numpy(np.frombuffer(b"".join(files) + bytes(16), np.uint8).copy()))

    def begin(i):
        order.append(("begin", i))
        offs, buf = bufs[i % len(bufs)]
        return pdist.gather_files_begin(offs, buf)

    def finish(i, token):
        order.append(("finish", i))
        got = pdist.gather_files_finish(token, dst=0)
        if rank == 0:
            blob = b""
            for roffs, rbuf in got:
                roffs = roffs.numpy()
                blob += rbuf.numpy()[: roffs[-1]].tobytes()
            got_steps.append(blob)
        else:
            assert got is None

    pdist.run_pipelined(n_steps, produce, begin, finish)
    # the host order the docstring promises: the kernels of step i + 1 are enqueued before anything of step i is looked
    # at, and the files of step i move one iteration after their sizes started travelling
    want = [("produce", 0)]
    for i in range(n_steps):
        if i + 1 < n_steps:
            want.append(("produce", i + 1))
        want.append(("begin", i))
        if i >= 1:
            want.append(("finish", i - 1))
    want.append(("finish", n_steps - 1))
    assert order == want, order
    if rank == 0:
        with open(result_path, "wb") as fh:
            for blob in got_steps:
                fh.write(len(blob).to_bytes(8, "little") + blob)
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_loop_over_two_ranks(tmp_path, product, oracle):
    """bench.py's strong-scaling leg on CPU: a fixed batch of 5 frames over 2 ranks (shards of 2 and 3), five steps
    through run_pipelined with three buffer sets (every set is reused); at every step the writer rank holds all five
    files of THAT step, in frame order, byte for byte what one process writes."""
    result = str(tmp_path / "strong.bin")
    port = 33500 + (os.getpid() % 2000)
    n_steps = 5
    mp.spawn(_worker_strong, args=(2, port, result, n_steps), nprocs=2, join=True)
    data = open(result, "rb").read()
    pos = 0
    for i in range(n_steps):
        n = int.from_bytes(data[pos:pos + 8], "little")
        blob = data[pos + 8:pos + 8 + n]
        pos += 8 + n
        ref = b""
        for f in range(NF):
            img = oracle.synth_frame(W, H, 4, f, 1)
            v, ow, oh, slots = oracle.shrink_image(img, B, B, 1, 4, 4.0 + 4.0 * i)
            ref += oracle.encode_container(W, H, B, B, 4, 0, v, None, ow, oh, slots)
        assert blob == ref, f"step {i}"
    assert pos == len(data)


_FAILING_RANK_SCRIPT = r"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world, port, fail_how = int(sys.argv[1]), 2, sys.argv[2], sys.argv[3]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port
dist.init_process_group("gloo", rank=rank, world_size=world)
from tests.conftest import load_product
product = load_product()
from pixlzr_rust_amd import dist as pdist
dog = pdist.exit_on_timeout(8.0, lambda: print("rank", rank, "abandons the leg", flush=True), code=3)
bufs = [None] * pdist.PIPELINE_SETS
def produce(i):
    if rank == 1 and i == 2:
        if fail_how == "raise":
            raise RuntimeError("injected failure on rank 1")
        time.sleep(3600)  # "hang": a rank that never arrives
    n = 100 + i
    bufs[i % len(bufs)] = (torch.tensor([0, n], dtype=torch.int64), torch.full((n + 8,), i, dtype=torch.uint8))
def begin(i):
    return pdist.gather_files_begin(*bufs[i % len(bufs)])
def finish(i, token):
    pdist.gather_files_finish(token, dst=0)
failed = False
try:
    pdist.run_pipelined(6, produce, begin, finish)
except Exception as exc:
    print("rank", rank, "failed:", type(exc).__name__, flush=True)
    failed = True
dog.cancel()
if failed:
    os._exit(1)   # as bench.py: no barrier, no teardown -- the others may sit in a collective
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("fail_how", ["raise", "hang"])
def test_a_failing_rank_ends_every_rank_non_zero(tmp_path, fail_how):
    """The strong-scaling loop with a rank that fails (raises) or never arrives (hangs) in step 2: every rank comes back
    -- the failing one at once, its peer out of the broken collective or through the watchdog (dist.exit_on_timeout)
    -- and every exit code is non-zero.  This is what bench.py's strong-scaling leg does around the same calls."""
    import subprocess
    script = tmp_path / "failing_rank.py"
    script.write_text(_FAILING_RANK_SCRIPT.format(root=ROOT))
    port = str(35500 + (os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port, fail_how], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    codes = []
    for p in procs:
        try:
            p.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
            pytest.fail("a rank did not return")
        codes.append(p.returncode)
    assert all(c != 0 for c in codes), codes
