"""Host-side regressions: results.csv keeps ONE header when a metric (loss) first appears after the first row; overrides such as
`1e6` are numbers (Hydra / OmegaConf typing), not YAML-1.1 strings."""
import os

from codebase_amd import config as C
from codebase_amd.utils.loggers import FileSystemLogger


def test_results_csv_header_grows_when_loss_appears_later(tmp_path):
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        lg = FileSystemLogger("t", None)
        base = {"updates": 0, "environment_steps": 100, "epsilon": 0.9}
        ep = [{"episode_returns": [0.5, 0.25], "episode_length": 25}, {"episode_returns": [0.0, 0.25], "episode_length": 20}]
        lg.log_metrics([base] + ep)                                                  # evaluation before the first update: no loss
        lg.log_metrics([dict(base, updates=3, environment_steps=200, loss=0.125)] + ep)
        lg.log_metrics([dict(base, updates=6, environment_steps=300, loss=0.0625)] + ep)
        df = lg.get_state()
        assert list(df.index) == [100, 200, 300]
        assert "loss" in df.columns and df["loss"].isna().tolist() == [True, False, False]
        assert df["loss"].iloc[2] == 0.0625 and df["mean_episode_returns"].iloc[0] == 0.5
        # a logger re-opened on the same run directory keeps appending under the same header
        lg2 = FileSystemLogger("t", None)
        lg2.log_metrics([dict(base, updates=9, environment_steps=400, loss=0.03125)] + ep)
        df = lg2.get_state()
        assert list(df.index) == [100, 200, 300, 400] and df["loss"].iloc[3] == 0.03125
    finally:
        os.chdir(cwd)


def test_exponent_literals_are_numbers():
    cfg = C.compose(["+algorithm=idqn", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "algorithm.total_steps=1e6",
                     "algorithm.target_update_interval_or_tau=1e-2", "algorithm.eps_end=5e-2", "algorithm.lr=3e-4",
                     "algorithm.model.layers=[64,64]", "seed=3"])
    a = cfg.algorithm
    assert a.total_steps == 1000000.0 and isinstance(a.total_steps, float)
    assert a.target_update_interval_or_tau == 0.01 and a.eps_end == 0.05 and a.lr == 3e-4
    assert a.model.layers == [64, 64] and cfg.seed == 3 and isinstance(cfg.seed, int)
    assert cfg.env.name == "lbforaging:Foraging-8x8-2p-3f-v3"      # strings that merely contain digits stay strings
    # only the YAML-1.1 gap is re-typed (exponent form without a dot, which safe_load leaves a string); what safe_load hands over as a
    # string otherwise was QUOTED in the file ("7", "1.5", "007") and stays a string, as it does under OmegaConf
    assert C._numbers({"a": ["1e3", "x1e3", "-2E-2", "7", "1.5", "007", "v3", 7, 1.5]}) == {"a": [1000.0, "x1e3", -0.02, "7", "1.5", "007", "v3", 7, 1.5]}
    # dotted mantissas with an UNSIGNED exponent are in the same gap (safe_load("1.5e6") == "1.5e6"); ADVICE r3
    assert C._numbers(["1.5e6", "2.5E5", ".5e3", "1.e3", "+.5e-3", "1.5e", "e3", "1.5e6x"]) == [1.5e6, 2.5e5, 500.0, 1000.0, 0.0005, "1.5e", "e3", "1.5e6x"]
    cfg = C.compose(["+algorithm=idqn", "env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "algorithm.total_steps=1.5e6", "algorithm.lr=2.5e-4"])
    assert cfg.algorithm.total_steps == 1.5e6 and isinstance(cfg.algorithm.total_steps, float) and cfg.algorithm.lr == 2.5e-4
