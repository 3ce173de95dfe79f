timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_qmix.py tests/test_gpu_ac_update.py -q -m gpu 2>&1 | tail -6 | cut -c1-250
B="python bench.py --no-cpu-baseline"
for v in 1 0; do
  export MARLHIP_WIDE_GEMM64=$v; [ $v = 0 ] && unset MARLHIP_WIDE_GEMM64
  echo "== MARLHIP_WIDE_GEMM64=$v"
  timeout 300 $B --steps 3 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('mappo rware', round(r['value']/1e6,2), round(r['ms_per_step'],1), round(r['roofline']['frac'],3))"
  timeout 300 $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('maa2c 8p', round(r['value']/1e6,2), round(r['ms_per_step'],1), round(r['roofline']['frac'],3))"
done
