"""N > 1 on the real kernels: two processes share cuda:0 over gloo (the RCCL path differs only in the backend string);
see tests/two_rank_worker.py for what is asserted."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_stay_bit_identical_idqn_qmix_a2c():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "tests", "two_rank_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "TWO_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
