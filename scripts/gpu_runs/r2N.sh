mkdir -p gpurun_out/learn3; O=gpurun_out/learn3
V="python tests/tools/learning_parity.py vec"
i=0
for cfg in "4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "4096 128 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.05" \
           "4096 256 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=100" "4096 256 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.05" \
           "4096 512 1024 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=100" "4096 1024 512 algorithm.lr=1e-3" "4096 1024 1024 algorithm.lr=1e-3" \
           "4096 2048 256 algorithm.lr=3e-4" "4096 2048 256 algorithm.lr=1e-3" "4096 4096 128" "4096 4096 128 algorithm.lr=1e-3" \
           "1024 1024 128" "1024 256 512 algorithm.lr=1e-3" "256 256 128" "256 256 128 algorithm.lr=1e-3"; do
  i=$((i+1)); timeout 300 $V 0 3e7 $cfg 2>/dev/null | grep '^{' > $O/sweep_$i.jsonl; echo "sweep $i: $cfg"; tail -1 $O/sweep_$i.jsonl | cut -c1-160
done
