"""The `nccl` (= RCCL) branch of pixlzr-rust_amd/dist.py, executed: one rank on one GPU in a fresh child process.

The multi-GPU legs of bench.py (`strong_scaling`, N > 1) can only run on a multi-GPU node, which this build never sees;
the gloo tests (tests/test_dist_gloo.py) cover the protocol but not a single line of the RCCL path.  This test runs the
same step loop -- `dist.run_pipelined` with real compute / comm streams, `gather_files_begin` / `gather_files_finish` --
under `init_process_group("nccl", world_size=1)`: process-group init with a device id, `all_gather_into_tensor` on int64
device tensors, the pinned size copy behind an event, the stream / event ordering of the three buffer sets.  Only the
point-to-point sends need a second GPU.  (The row table the gather preserves: encoding/mod.rs:60,77-82.)
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_single_rank_rccl_step_loop():
    with socket.socket() as s:  # a free port for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PXZ_CHILD_TIMEOUT="240")
    # a child of its own: the process group is initialised before anything in that process touches the GPU; nothing is re-executed
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, f"exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    assert "rccl single-rank ok" in r.stdout
    assert r.stdout.count("equal=True") == 3
