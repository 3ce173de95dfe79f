mkdir -p gpurun_out/learn; O=gpurun_out/learn
L="python tests/tools/learning_parity.py"
for s in 0 1 2; do timeout 300 $L scalar $s 400000 2>/dev/null | grep '^{' > $O/scalar_s$s.jsonl; tail -1 $O/scalar_s$s.jsonl; done
V="$L vec"
for s in 0 1 2; do timeout 120 $V $s 3e7 4096 32 4096 2>/dev/null | grep '^{' > $O/vec_default_s$s.jsonl; tail -1 $O/vec_default_s$s.jsonl; done
i=0
for cfg in "4096 32 4096 algorithm.lr=1e-3" "4096 32 4096 algorithm.lr=3e-3" "4096 32 4096 algorithm.target_update_interval_or_tau=50" "4096 32 4096 algorithm.target_update_interval_or_tau=20" \
           "4096 32 4096 algorithm.lr=1e-3 algorithm.target_update_interval_or_tau=50" "4096 256 512" "4096 256 512 algorithm.lr=1e-3" "1024 128 256" "1024 32 1024" "1024 32 1024 algorithm.lr=1e-3" \
           "4096 32 4096 algorithm.use_proper_termination=True" "4096 32 4096 algorithm.use_proper_termination=True algorithm.lr=1e-3" "4096 64 4096" "4096 128 4096 algorithm.target_update_interval_or_tau=800" ; do
  i=$((i+1)); timeout 120 $V 0 3e7 $cfg 2>/dev/null | grep '^{' > $O/sweep_$i.jsonl; echo "sweep $i: $cfg"; tail -1 $O/sweep_$i.jsonl | cut -c1-200
done
