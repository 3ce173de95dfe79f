# gate nonlinearities on v_exp_f32 / v_rcp_f32: the recurrent parity tests, then the recurrent IDQN line
mkdir -p gpurun_out/r2Y
timeout 40 python -m pytest tests/test_gru.py tests/test_gpu_ac_update.py tests/test_gpu_standardise.py tests/test_gpu_qmix.py -x -q -m gpu > gpurun_out/r2Y/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2Y/tests.log | cut -c1-400
timeout 30 python bench.py --steps 10 --warmup 2 --rnn --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('idqn gru64', d['value'], d['ms_per_step'], {k: round(v['avg_us']) for k, v in d['kernels'].items()})"
