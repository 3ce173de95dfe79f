mkdir -p gpurun_out/r2Z
timeout 900 python -m pytest tests/test_ac_collector.py tests/test_gpu_rware.py tests/test_gpu_ac_update.py -q -m gpu > gpurun_out/r2Z/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2Z/tests.log | cut -c1-200
B="python bench.py --no-cpu-baseline"
for nw in 1 0; do
  export MARLHIP_ACOL_NW=$nw; [ $nw = 0 ] && unset MARLHIP_ACOL_NW
  echo "== MARLHIP_ACOL_NW=$nw"
  timeout 200 $B --steps 50 --warmup 5 --algo ia2c 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c lbf H64', round(r['value']/1e6,2), r['ms_per_step'])"
  timeout 200 $B --steps 50 --warmup 5 --algo ia2c --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c lbf H128', round(r['value']/1e6,2), r['ms_per_step'])"
  timeout 300 $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H128', round(r['value']/1e6,2), r['ms_per_step'], r.get('collector'))"
  timeout 300 $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H64', round(r['value']/1e6,2), r['ms_per_step'])"
done
