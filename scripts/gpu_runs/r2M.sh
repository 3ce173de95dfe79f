mkdir -p gpurun_out/learn2; O=gpurun_out/learn2
V="python tests/tools/learning_parity.py vec"
i=0
for cfg in "64 64 32" "256 256 32" "32 32 32" \
           "4096 32 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=20" "4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=20" \
           "4096 32 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=10" "4096 64 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=20" \
           "4096 128 1024 algorithm.lr=1e-3" "4096 128 1024 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "4096 64 2048 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" \
           "1024 32 1024 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "1024 32 1024 algorithm.lr=3e-3" "2048 32 2048 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" \
           "4096 32 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=0.05" "4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1" ; do
  i=$((i+1)); timeout 200 $V 0 3e7 $cfg 2>/dev/null | grep '^{' > $O/sweep_$i.jsonl; echo "sweep $i: $cfg"; tail -1 $O/sweep_$i.jsonl | cut -c1-160
done
