timeout 900 python -m pytest tests/test_gru.py tests/test_gpu_standardise.py -q -m gpu 2>&1 | tail -8 | cut -c1-250
B="python bench.py --no-cpu-baseline --steps 6 --warmup 2 --rnn"
for v in 1 0; do
  export MARLHIP_GRU_BWD_ONE_WAVE=$v; [ $v = 0 ] && unset MARLHIP_GRU_BWD_ONE_WAVE
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('gru64 idqn one_wave=$v', round(r['value']/1e6,3), round(r['ms_per_step'],2), {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})"
done
timeout 300 $B --algo ia2c --hidden 64 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('gru64 ia2c', round(r['value']/1e6,3), round(r['ms_per_step'],2))"
