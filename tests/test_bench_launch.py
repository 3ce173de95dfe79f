"""`python bench.py --gpus N` must run N ranks by itself (the driver's literal command form) and also accept being
launched under torch.distributed.run; a --gpus / WORLD_SIZE mismatch is an error, not a silent 1-rank run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_self_launches_n_ranks_dryrun():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         env=_env(MARLHIP_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["ranks_in_allreduce"] == 2


def test_gpus_flag_under_torchrun_and_mismatch_dryrun():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", "29547", BENCH, "--steps", "1", "--warmup", "0"]
    ok = subprocess.run(base + ["--gpus", "2"], cwd=ROOT, env=_env(MARLHIP_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-3000:]
    assert _json_line(ok.stdout)["n_gpus"] == 2
    bad = subprocess.run(base + ["--gpus", "4"], cwd=ROOT, env=_env(MARLHIP_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["idqn", "ia2c"])
def test_bench_gpus_2_runs_two_ranks_on_the_real_kernels(algo):
    """Two ranks share cuda:0 over gloo (the RCCL path differs only in the backend string): the JSON line must say 2."""
    def run():
        return subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs", "256", "--algo", algo, "--no-cpu-baseline"],
                              cwd=ROOT, env=_env(MARLHIP_BENCH_BACKEND="gloo", MARLHIP_BENCH_ONE_DEVICE="1", MARLHIP_P2P_SHARED_DEVICE="1",
                                                 MARLHIP_BENCH_SMALL_ROWS="1", MARLHIP_P2P_TIMEOUT_MS="20000"), capture_output=True, text=True, timeout=900)

    out = run()
    if "did not publish its gradient in time" in out.stdout + out.stderr:
        # the one-device rig's rare lane timeout (tests/test_gpu_two_ranks.py has the numbers): the run says so itself - once more
        print("[bench --gpus 2] a lane of the in-library exchange timed out on the shared device; second attempt")
        out = run()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = _json_line(out.stdout)
    rr = line["rccl_ranks"]
    assert line["n_gpus"] == 2 and rr["world_size"] == 2 and rr["backend"] == "gloo"
    assert rr["exchange"].startswith("p2p"), rr  # the in-library exchange came up between the two processes and never timed out
    assert line["value"] > 0 and line["scaling"] == "weak"
    if algo == "ia2c":
        # ia2c.yaml trains without a joint clip: the critics' half of the update runs beside the next rollout ALSO next to a gradient
        # exchange - their slice goes through the exchange's second lane on their own stream
        assert line["roofline"]["critic_backward_overlaps_next_rollout"] is True and rr["critics_slice_on_their_own_lane"] is True, (line["roofline"], rr)
        assert rr["side_lane"]["exchange"].startswith("p2p")
        return
    # the multi-GPU line describes itself: the headline under BOTH exchanges in one process group, the per-update exchange
    # time from the library's timers, and - two ranks: the rank-ordered sum is the collective's sum - the same final parameters
    rows = rr["exchange_rows"]
    kinds = [k for k in rows if k != "same_final_parameters"]
    assert len(kinds) == 2 and any("p2p" in k for k in kinds) and any(k.startswith("MARLHIP_P2P=0: collective") for k in kinds), rows
    for k in kinds:
        assert rows[k]["value"] > 0 and rows[k]["per_update_exchange_us"] > 0 and rows[k]["exchanges_timed"] > 0 and rows[k]["params_sha16_rank0"]
    assert rows["same_final_parameters"] is True, rows
    assert next(v for k, v in rows.items() if k.startswith("MARLHIP_P2P=0"))["fallback_reason"] == "MARLHIP_P2P=0"
    c4 = next(v for k, v in line["modes"].items() if k.startswith("BASELINE config 4"))
    c5 = next(v for k, v in line["modes"].items() if k.startswith("BASELINE config 5"))
    assert c4["value"] > 0 and c4["critic_backward_overlaps_next_rollout"] is True and c4["critics_slice_on_their_own_lane"] is True, c4
    assert c5["value"] > 0 and c5["per_update_exchange_us"] > 0, c5


def test_cpu_baseline_all_cores_leg_forks_and_reports():
    """bench.py --cpu-all-cores N (the internal leg of cpu_baseline that a fresh interpreter runs): N forked 1-thread copies of the
    reference loop, [total env-steps, slowest window, copies that reported]"""
    out = subprocess.run([sys.executable, BENCH, "--cpu-all-cores", "2", "--cpu-seconds", "1.0"], cwd=ROOT, env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert got[2] == 2 and got[0] > 0 and 0.9 <= got[1] <= 30.0
