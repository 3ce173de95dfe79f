"""bench.py quotes `roofline.traffic` from a committed PMC profile only when the kernel sources of this tree hash to the ones the
profile was measured on.  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KEY = "idqn:lbforaging:Foraging-8x8-2p-3f-v3:N4096:H64:B4096:T25:rnn0"


def test_traffic_is_quoted_only_for_matching_sources(tmp_path, monkeypatch):
    t, src = bench.traffic_from_profile(KEY)
    now = bench.kernel_source_hash(bench.TRAFFIC_SOURCES[""])
    if t is not None:  # the committed profile matches this tree
        assert src["source_hash"] == now and t > 0
    else:
        assert "reason" in src
    # a profile measured on other sources is not quoted
    prof = tmp_path / "profiles"
    prof.mkdir()
    fake = {"head": "deadbeef", "workloads": {KEY: {"kernel": "k", "traffic_bytes": 1.0, "source_files": list(bench.TRAFFIC_SOURCES[""]),
                                                    "source_hash": "0" * 16}}}
    (prof / "r04_pmc_traffic.json").write_text(json.dumps(fake))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda files: now)
    t, src = bench.traffic_from_profile(KEY)
    assert t is None and "not quoted" in src["reason"]
    fake["workloads"][KEY]["source_hash"] = now
    (prof / "r04_pmc_traffic.json").write_text(json.dumps(fake))
    t, src = bench.traffic_from_profile(KEY)
    assert t == 1.0 and src["file"] == "profiles/r04_pmc_traffic.json"
    assert bench.traffic_from_profile("no:such:workload")[0] is None


def test_headline_defaults_are_the_reference_config(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.hparams == "reference" and a.hidden == 64 and a.envs == 4096 and a.cadence == "ratio" and a.gpus == 1


def test_needed_flops_leave_out_exactly_the_first_layer_input_gradient():
    """`roofline.frac_needed` = `frac` without the products `backward = 2 x forward` counts and nothing runs (the gradient w.r.t. a
    network's input rows): 2 D H per row and backward pass, for the actor and the (centralised) critic"""
    import bench

    P, D, A, H, T, N = 8, 39, 6, 128, 25, 4096
    flops, parts = bench.dqn_update_flops("idqn", False, 2, 15, 6, 64, 25, 4096)
    assert parts["first_layer_input_gradient (counted, never run)"] == 2.0 * 15 * 64 * 2 * 4096 * 25
    assert 0.04 < parts["first_layer_input_gradient (counted, never run)"] / flops < 0.05  # 15-wide rows: a small share
    full = bench.ac_update_flops(False, P, D, A, H, T, N, True)
    dx = bench.ac_unneeded_flops(P, D, H, T, N, True)
    assert dx == (2.0 * D * H + 2.0 * P * D * H) * P * N * T
    assert 0.14 < dx / full < 0.17  # 312-wide centralised critics: the share that makes the two fractions differ visibly
    # PPO: per launch group (1 prepare + E epochs), only the E epoch groups run a backward pass
    assert bench.ac_unneeded_flops(P, D, H, T, N, True, epochs=4) == dx * 4 / 5
    # A2C on a rollout whose collector kept the actors' forward pass: the step runs one actor forward less (T rows per agent and env)
    assert full - bench.ac_update_flops(False, P, D, A, H, T, N, True, actor_forward_kept=True) == bench.mlp_fwd_flops(D, H, A) * P * N * T
