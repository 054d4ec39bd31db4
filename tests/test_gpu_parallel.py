"""Data parallelism with the HIP kernels in the loop: two ranks share the box's one GPU and talk over gloo (RCCL
refuses two ranks on one device; the collective code path is the same torch.distributed API).  Sync-BN mode must
reproduce ONE reference step on the concatenated batch (SURVEY §8(e) mode ii, encoders.py:1048-1052, 1326-1331)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case", ["small", "packed"])
def test_sync_bn_two_ranks_match_the_oracle_on_the_concatenated_batch(case):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_sync_bn_worker.py"), case]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert r.stdout.count("equals the oracle on the concatenated batch") == 2, r.stdout
