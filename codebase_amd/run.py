"""Entry point with the reference's command line (marlbase/run.py:14-47):

    python -m codebase_amd.run +algorithm=idqn env.name="lbforaging:Foraging-8x8-2p-3f-v3" env.time_limit=25 \\
        [env.parallel_envs=4096] [algorithm.model.layers=[64,64]] [seed=0] [--config-dir /path/to/marlbase/configs]

Same sequence as the reference: logger, env, eval_env (parallel_envs dropped), seeding of torch and
numpy (python `random` left unseeded, as there), then the algorithm's `_target_`.

Multi-GPU (the reference is one process): the same command line under torchrun,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m codebase_amd.run +algorithm=idqn \
        env.name=... env.parallel_envs=4096

is the sharded drop-in: one process per GPU (LOCAL_RANK picks the device), env.parallel_envs batched envs and a replay shard PER
RANK with their own Philox streams (`parallel.rank_env_seed`), identical initial weights from `seed`, one RCCL all-reduce of the
flat gradient per update inside the algorithm's `main`, rank 0 logging / saving with whole-job step counts and gathered returns.
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
    from .parallel import init_distributed, rank_env_seed, setup_device

    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:  # torchrun: bind the device first, then rendezvous (backend nccl == RCCL; MARLHIP_DIST_BACKEND=gloo for 1-GPU test boxes)
        setup_device(os.environ.get("LOCAL_RANK", 0))
    dist, rank, world, _ = init_distributed(os.environ.get("MARLHIP_DIST_BACKEND"))
    if rank == 0:
        logger = C.instantiate(cfg.logger, cfg=cfg)
    else:  # ranks > 0 keep the interface (watch / info / ...) but write no results.csv; the algorithm logs on rank 0 only
        from .utils.loggers import Logger

        logger = Logger(cfg=cfg)
        logging.getLogger().setLevel(logging.WARNING)
    env_seed = cfg.seed if dist is None or cfg.seed is None else rank_env_seed(cfg.seed, rank)  # a rank's env shard has its own stream
    env = C.call(cfg.env, seed=env_seed)
    eval_cfg = C.to_cfg({k: v for k, v in cfg.env.items() if k != "parallel_envs"})
    eval_env = C.call(eval_cfg, seed=env_seed) if not cfg.env.get("parallel_envs") else None
    torch.set_num_threads(1)
    if cfg.seed is not None:
        torch.manual_seed(cfg.seed)
        np.random.seed(cfg.seed)
    else:
        logger.warning("No seed has been set.")
    assert cfg.env.time_limit is not None, "Time limit must be set."
    C.call(cfg.algorithm, env, eval_env, logger, time_limit=cfg.env.time_limit)
    state = logger.get_state()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return state


if __name__ == "__main__":
    main()
