"""N > 1 on the real kernels: two processes share cuda:0 over gloo (the RCCL path differs only in the backend string);
see tests/two_rank_worker.py for what is asserted."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_stay_bit_identical_idqn_qmix_a2c():
    """once with the in-library peer-to-peer exchange (marlhip_p2p_allreduce: IPC-shared buffers, the default on GPU ranks) and once
    with torch.distributed's collective (MARLHIP_P2P=0): replicas identical within each run, and - two ranks, so the rank-ordered sum
    r0 + r1 is the collective's sum bit for bit - the SAME final parameters in both runs"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for p2p, port in (("1", "29533"), ("0", "29535")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MARLHIP_P2P=p2p, MARLHIP_P2P_TIMEOUT_MS="20000",
                   MARLHIP_P2P_SHARED_DEVICE="1")  # two ranks on this box's one GPU: the exchange is refused there unless asked for
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(root, "tests", "two_rank_worker.py")]
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        if p2p == "1" and out.returncode != 0 and "ran into its peer timeout" in out.stderr:
            # Two processes time-sharing ONE device (this rig only; a job's ranks own a GPU each): until marlhip_p2p_allreduce capped its
            # grid, ~100 spinning workgroups of the early rank could leave the late rank's 128-wide learner kernels (a SIMD's whole register
            # file each) no compute unit to start on - 9 of 80 runs ran into the 20 s bound; 0 of 128 since (gpurun
            # r6W - r6Y).  The worker asserts the lanes' status itself, so a timeout can never pass as a result; kept: one more attempt.
            print("[two ranks] a lane of the in-library exchange timed out on the shared device; second attempt\n" + out.stderr[-1500:])
            out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "TWO_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
        digests[p2p] = [ln for ln in out.stdout.splitlines() if ln.startswith("TWO_RANK_OK")][-1].split()[1]
    assert digests["1"] == digests["0"], "the peer-to-peer exchange and the collective left different parameters"


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
                    dict(MARLHIP_DIST_BACKEND="gloo", MARLHIP_ONE_DEVICE="1", MARLHIP_P2P_SHARED_DEVICE="1", MARLHIP_P2P_TIMEOUT_MS="20000"), 29541)
    assert out.returncode == 0 and "ENTRY_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-5000:]
