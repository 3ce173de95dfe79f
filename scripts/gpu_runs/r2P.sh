mkdir -p gpurun_out/r2p gpurun_out/learn4; O=gpurun_out/r2p
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
V="python tests/tools/learning_parity.py vec"
i=0
for cfg in "0 1e8 4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "0 1e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1" \
           "0 1e8 4096 64 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=20" "0 1e8 4096 32 4096" \
           "1 3e7 64 64 32" "2 3e7 64 64 32" "1 3e7 4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "2 3e7 4096 128 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" \
           "0 3e8 4096 32 4096 algorithm.lr=3e-3 algorithm.target_update_interval_or_tau=0.1"; do
  i=$((i+1)); timeout 400 $V $cfg 2>/dev/null | grep '^{' > gpurun_out/learn4/run_$i.jsonl; echo "run $i: $cfg"; tail -1 gpurun_out/learn4/run_$i.jsonl | cut -c1-160
done
