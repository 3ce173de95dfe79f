mkdir -p gpurun_out/learn5
V="python tests/tools/learning_parity.py vec"
i=0
for cfg in "0 1.5e6 64 64 32" "1 1.5e6 64 64 32" "2 1.5e6 64 64 32" "0 1.5e6 1 1 32" \
           "1 3e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1" "2 3e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1" \
           "0 1e8 4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50 algorithm.eps_decay_over=0.15" \
           "0 3e7 4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50 algorithm.use_proper_termination=True" \
           "0 3e7 64 64 32 algorithm.use_proper_termination=True"; do
  i=$((i+1)); timeout 400 $V $cfg 2>/dev/null | grep '^{' > gpurun_out/learn5/run_$i.jsonl; echo "run $i: $cfg"; tail -1 gpurun_out/learn5/run_$i.jsonl | cut -c1-160
done
