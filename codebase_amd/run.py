"""Entry point with the reference's command line (marlbase/run.py:14-47):

    python -m codebase_amd.run +algorithm=idqn env.name="lbforaging:Foraging-8x8-2p-3f-v3" env.time_limit=25 \\
        [env.parallel_envs=4096] [algorithm.model.layers=[64,64]] [seed=0] [--config-dir /path/to/marlbase/configs]

Same sequence as the reference: logger, env, eval_env (parallel_envs dropped), seeding of torch and
numpy (python `random` left unseeded, as there), then the algorithm's `_target_`.
"""
import logging
import os
import sys

import numpy as np
import torch

from . import config as C


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    config_dir = None
    if "--config-dir" in argv:
        i = argv.index("--config-dir")
        config_dir = argv[i + 1]
        del argv[i:i + 2]
    cfg = C.compose(argv, config_dir)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(levelname)s] - %(message)s")
    out = os.environ.get("MARLHIP_RUN_DIR")
    if out:
        os.makedirs(out, exist_ok=True)
        os.chdir(out)  # hydra.job.chdir: True (configs/default.yaml:14-15)
    logger = C.instantiate(cfg.logger, cfg=cfg)
    env = C.call(cfg.env, seed=cfg.seed)
    eval_cfg = C.to_cfg({k: v for k, v in cfg.env.items() if k != "parallel_envs"})
    eval_env = C.call(eval_cfg, seed=cfg.seed) if not cfg.env.get("parallel_envs") else None
    torch.set_num_threads(1)
    if cfg.seed is not None:
        torch.manual_seed(cfg.seed)
        np.random.seed(cfg.seed)
    else:
        logger.warning("No seed has been set.")
    assert cfg.env.time_limit is not None, "Time limit must be set."
    C.call(cfg.algorithm, env, eval_env, logger, time_limit=cfg.env.time_limit)
    return logger.get_state()


if __name__ == "__main__":
    main()
