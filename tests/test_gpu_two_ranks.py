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


def _torchrun(args, extra_env, port, timeout=900):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("algo,extra", [("idqn", ["algorithm.updates_per_round=4", "algorithm.update_batch_size=64"]),
                                        ("idqn", ["algorithm.updates_per_round=4", "algorithm.update_batch_size=64", "algorithm.standardise_returns=True"]),
                                        ("vdn", ["algorithm.updates_per_round=2", "algorithm.update_batch_size=64"]),
                                        ("ia2c", ["algorithm.standardise_returns=True"])])
def test_drop_in_entry_point_shards_under_torchrun(tmp_path, algo, extra):
    """`torchrun --nproc-per-node 2 -m codebase_amd.run +algorithm=...` IS the sharded job: two ranks on this box's one
    GPU over gloo.  tests/two_rank_entry.py wraps run.main and checks on every rank: replicas bit-identical at the end, the shards
    different, rank 0 alone wrote results.csv with whole-job step counts."""
    out = _torchrun([os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_entry.py"), str(tmp_path), f"+algorithm={algo}"] + extra,
                    dict(MARLHIP_DIST_BACKEND="gloo", MARLHIP_ONE_DEVICE="1"), 29541)
    assert out.returncode == 0 and "ENTRY_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-5000:]
