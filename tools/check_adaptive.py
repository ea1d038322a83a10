"""The adaptive kernel choice on frames full of transparency (no hint): the first call takes the generic kernel for list A,
later calls the four-plane kernels -- every call must give the oracle's bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_product
import test_gpu_parity as T
from oracle import binding as oracle
oracle.build()
P = load_product()
gpu = P.Handle(0)
frames = gpu.synth_frames_device(2, 2160, 3840, 4, first_frame=3, dist=1)
f0 = frames[0].cpu().numpy()
for bs in (32, 64):
    for mode, factor in ((1, 16.0), (0, 1.0)):
        exp = oracle.shrink_image(f0, bs, bs, mode, 4, factor, nthreads=8)
        for call in range(3):
            vals, ow, oh, slots = gpu.shrink_frames_device(frames, bs, bs, mode, 4, factor)
            torch.cuda.synchronize()
            got = (vals[0].cpu().numpy(), ow[0].cpu().numpy().astype(np.uint32), oh[0].cpu().numpy().astype(np.uint32), slots[0].cpu().numpy())
            T.assert_same_tiles(got, exp, 4, f"b{bs} mode{mode} call {call}")
        print("b%d mode%d: 3 calls equal the oracle" % (bs, mode), flush=True)
