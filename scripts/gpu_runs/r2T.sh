# two-wave recurrent backward at hidden 128 (streamed gate matrices): parity tests + A/B against the one-wave form
O=gpurun_out/r2T; mkdir -p $O
timeout 600 python -m pytest tests/test_gru.py tests/test_gpu_standardise.py tests/test_gpu_ac_update.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?"; tail -5 $O/tests.log | cut -c1-300
for mode in two one; do
  if [ $mode = one ]; then export MARLHIP_GRU_BWD_ONE_WAVE=1; else unset MARLHIP_GRU_BWD_ONE_WAVE; fi
  timeout 200 python bench.py --steps 5 --warmup 1 --rnn --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode idqn gru128', d['value'], d['ms_per_step'])"
  timeout 200 python bench.py --steps 20 --warmup 2 --rnn --algo ia2c --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode ia2c gru128', d['value'], d['ms_per_step'])"
done
unset MARLHIP_GRU_BWD_ONE_WAVE
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/stats_gru128 --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --rnn --hidden 128 > /root/repo/$O/stats_gru128.log 2>&1
