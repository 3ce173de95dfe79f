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
