"""Child process of tests/test_gpu_rccl.py: ONE rank on ONE GPU over the real `nccl` (= RCCL) backend.

Runs the strong-scaling step loop of bench.py (`dist.run_pipelined`: shrink + device writer on a compute stream, the
file gather through `gather_files_begin` / `gather_files_finish` on a comm stream, three buffer sets, events between
them) for three steps and compares what "arrives" at the writer rank with a plain pxz_encode_frames_device of the same
frames.  With one rank the point-to-point sends do not happen, but everything else does: process-group init,
`all_gather_into_tensor` on int64 device tensors, the pinned copy + event, the stream / event ordering of the sets.
Exits 0 only if every byte matches; a watchdog ends the process with code 3.  (The rendezvous variables are set by the
parent; nothing here touches the GPU before init_process_group.)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    from __graft_entry__ import load_product
    product = load_product()
    pdist = product.dist
    dog = pdist.exit_on_timeout(float(os.environ.get("PXZ_CHILD_TIMEOUT", "240")), code=3)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    handle = product.Handle(0)
    n_steps, bw = 3, 32
    # every step encodes DIFFERENT frames (first_frame = 5 * step), so a set that is read too late or overwritten too
    # early shows up as the wrong step's files
    frames = [handle.synth_frames_device(2, 288, 416, 4, first_frame=5 * i, dist=1) for i in range(n_steps)]
    compute, comm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    bufs = []
    with torch.cuda.stream(compute):
        for _ in range(pdist.PIPELINE_SETS):
            out = handle.shrink_frames_device(frames[0], bw, bw, 1, 4, 16.0)
            enc = handle.encode_frames_device(tuple(frames[0].shape), bw, bw, *out)
            bufs.append((out, enc))
    torch.cuda.synchronize()
    got = {}

    def produce(i):
        out, enc = bufs[i % len(bufs)]
        handle.shrink_frames_device(frames[i], bw, bw, 1, 4, 16.0, out=out)
        handle.encode_frames_device(tuple(frames[i].shape), bw, bw, *out, out=enc)

    def begin(i):
        offs, buf = bufs[i % len(bufs)][1]
        return pdist.gather_files_begin(offs, buf)

    def finish(i, token):
        res = pdist.gather_files_finish(token, dst=0)
        assert res is not None and len(res) == 1
        offs, data = res[0]
        got[i] = (offs.clone(), data.clone())  # (on the comm stream: behind the exchange of step i)

    pdist.run_pipelined(n_steps, produce, begin, finish, compute=compute, comm=comm)
    torch.cuda.synchronize()
    # the same through the collective-free path: a plain shrink + writer per step, default stream
    bad = 0
    for i in range(n_steps):
        out = handle.shrink_frames_device(frames[i], bw, bw, 1, 4, 16.0)
        offs, buf = handle.encode_frames_device(tuple(frames[i].shape), bw, bw, *out)
        torch.cuda.synchronize()
        n = int(offs[-1].item())
        g_offs, g_data = got[i]
        ok = torch.equal(g_offs.cpu(), offs.cpu()) and g_data.numel() == n and torch.equal(g_data, buf[:n])
        print(f"step {i}: {n} bytes, equal={ok}", flush=True)
        bad += 0 if ok else 1
    # one more collective on device tensors, as gather_block_streams issues it
    sizes = torch.tensor([7, 11], dtype=torch.int64, device=dev)
    all_sizes = torch.empty(2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes)
    torch.cuda.synchronize()
    bad += 0 if all_sizes.tolist() == [7, 11] else 1
    dog.cancel()
    handle.close()
    dist.destroy_process_group()
    if bad:
        print(f"MISMATCH in {bad} checks", flush=True)
        sys.exit(1)
    print("rccl single-rank ok", flush=True)


if __name__ == "__main__":
    main()
