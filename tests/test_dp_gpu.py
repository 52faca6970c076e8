"""Data-parallel path on the device: two ranks share the one GPU of the test box (gloo instead of RCCL, which
refuses two ranks on one device), see tests/dp_gpu_worker.py for what is checked."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_gradient_allreduce_and_graph_replay_on_device():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "tests", "dp_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("identical_across_ranks=True") == 2, r.stdout[-2000:]


def test_rccl_backend_world_size_one_step():
    """The real RCCL path of the product (world_size 1): uz_comm_* over librccl.so, bucket events inside the backward
    hipGraph, per-bucket ncclAllReduce(avg) on the communication stream (the collective IS issued), Adam behind the last
    all-reduce; parameters stay bit-identical to a non-DP run (tools/nccl_world1_check.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", UZ_CHECK_BATCH="8")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_world1_check.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "nccl world 1: ms/step" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
