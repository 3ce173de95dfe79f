mkdir -p gpurun_out/r2AC
timeout 900 python -m pytest tests/test_ac_collector.py tests/test_gpu_rware.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r2AC/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2AC/tests.log | cut -c1-200
B="python bench.py --no-cpu-baseline"
timeout 300 $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H128', round(r['value']/1e6,2), r['ms_per_step'])"
timeout 300 $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H64', round(r['value']/1e6,2), r['ms_per_step'])"
timeout 300 $B --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('idqn rware H64', round(r['value']/1e6,2), r['ms_per_step'])"
timeout 300 $B --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 16384 --hidden 128 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ia2c rware H128 16384 envs', round(r['value']/1e6,2), r['ms_per_step'])"
