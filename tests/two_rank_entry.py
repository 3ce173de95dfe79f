"""Worker of tests/test_gpu_two_ranks.py::test_drop_in_entry_point_shards_under_torchrun: `codebase_amd.run.main` with the
reference's command line, launched by torch.distributed.run with 2 processes (gloo, both on cuda:0).  The algorithm's `main` is
wrapped so that the model it returns can be inspected before run.main tears the process group down."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, args = sys.argv[1], sys.argv[2:]
    rank = int(os.environ["RANK"])
    os.environ["MARLHIP_RUN_DIR"] = os.path.join(out_dir, f"rank{rank}")
    import torch.distributed as dist

    from codebase_amd import run
    from codebase_amd.ac import train as ac_train
    from codebase_amd.dqn import train as dqn_train

    seen = {}

    def wrap(mod):
        inner = mod.main

        def main_and_check(env, eval_env, logger, time_limit, **cfg):
            model = inner(env, eval_env, logger, time_limit, **cfg)
            assert dist.is_initialized() and dist.get_world_size() == 2
            blocks = [model.params, model.target_params] if hasattr(model, "params") else [model.updater.block, model.updater.target_critic]
            for t in blocks:
                ref = t.detach().cpu().clone()
                dist.broadcast(ref, src=0)
                assert torch.equal(ref, t.detach().cpu()), f"rank {rank}: replicas diverged"
            st = getattr(model.updater, "ret_stats", None)
            if st is not None and st.columns == 0:
                for t in (st.mean, st.var, st.count_t):
                    ref = t.detach().cpu().clone()
                    dist.broadcast(ref, src=0)
                    assert torch.equal(ref, t.detach().cpu()), f"rank {rank}: return statistics diverged"
                assert st.count > 1.0
            seed = torch.tensor([int(env.cfg.seed) & 0x7FFFFFFF])
            other = seed.clone()
            dist.broadcast(other, src=0)
            assert rank == 0 or int(other) != int(seed), "both ranks drive the same env stream"
            seen["updates"] = getattr(model, "updates", None) or model.updater.step
            return model

        mod.main = main_and_check

    wrap(dqn_train)
    wrap(ac_train)
    argv = args + ["env.name=lbforaging:Foraging-8x8-2p-3f-v3", "env.time_limit=25", "env.parallel_envs=256", "seed=3",
                   "algorithm.total_steps=60000", "algorithm.eval_interval=20000"]
    state = run.main(argv)
    assert seen.get("updates"), "the wrapped main never ran"
    csv = os.path.join(out_dir, "rank%d", "results.csv")
    if rank == 0:
        assert state is not None and len(state) >= 2, state
        steps = [int(i) for i in state.index]
        assert steps[-1] > 40000, steps  # whole-job env-steps: two shards of 256 envs
        print("ENTRY_OK", seen["updates"], steps)
    else:
        assert not os.path.exists(csv % rank), "a non-zero rank wrote results.csv"


if __name__ == "__main__":
    main()
