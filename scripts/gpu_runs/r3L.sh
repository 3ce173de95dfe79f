# round 3 learning parity: the vectorised reference-like cadence (N=64 envs, 64 sequential updates of 32 episodes per round) with the SAME
# total_steps (18 M -> the same epsilon schedule) and evaluation checkpoints as the CPU oracle loop of profiles/r03_learning/oracle_*.jsonl
O=$GRAFT_REPO_ROOT/gpurun_out/r3L; mkdir -p $O; cd $GRAFT_REPO_ROOT
for s in 0 1 2; do timeout 600 python tests/tools/learning_parity.py vec $s 18000000 64 64 32 > $O/vec64x64x32_18M_seed$s.jsonl 2> $O/vec_seed$s.err; tail -n 1 $O/vec64x64x32_18M_seed$s.jsonl | cut -c1-200; done
for s in 0 1 2; do timeout 600 python tests/tools/learning_parity.py vec $s 18000000 8 8 32 > $O/vec8x8x32_18M_seed$s.jsonl 2> $O/vec8_seed$s.err & done; wait
for s in 0 1 2; do tail -n 1 $O/vec8x8x32_18M_seed$s.jsonl | cut -c1-200; done
