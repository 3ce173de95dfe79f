mkdir -p gpurun_out/r2AA
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2AA/tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2AA/tests.log | cut -c1-200
B="python bench.py --no-cpu-baseline"
for nw in 1 0; do
  export MARLHIP_COL_NW=$nw; [ $nw = 0 ] && unset MARLHIP_COL_NW
  echo "== MARLHIP_COL_NW=$nw"
  timeout 200 $B --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('idqn lbf H64', round(r['value']/1e6,2), r['ms_per_step'], r.get('collector'))"
  timeout 200 $B --steps 60 --warmup 5 --cadence env-only 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('env-only', round(r['value']/1e6,2), r['ms_per_step'])"
  timeout 200 $B --steps 20 --warmup 3 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('idqn lbf H128', round(r['value']/1e6,2), r['ms_per_step'])"
  timeout 300 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('idqn rware H64', round(r['value']/1e6,2), r['ms_per_step'])"
done
