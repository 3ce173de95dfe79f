# hidden-128 recurrent forward: second chunk buffer filled by global_load_lds while the current gate product runs
O=gpurun_out/r2V; mkdir -p $O
timeout 600 python -m pytest tests/test_gru.py tests/test_gpu_ac_update.py tests/test_gpu_standardise.py tests/test_gpu_qmix.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log | cut -c1-300
timeout 200 python bench.py --steps 5 --warmup 1 --rnn --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('idqn gru128', d['value'], d['ms_per_step'], d['kernels'])"
timeout 200 python bench.py --steps 20 --warmup 2 --rnn --algo ia2c --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ia2c gru128', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/stats_gru128 --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --rnn --hidden 128 > /root/repo/$O/stats_gru128.log 2>&1
