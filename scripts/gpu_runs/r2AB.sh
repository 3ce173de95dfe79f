mkdir -p gpurun_out/r2AB; cd /tmp; export TMPDIR=/tmp
for h in 128 64; do
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2AB/prof_rware_$h --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden $h > /root/repo/gpurun_out/r2AB/bench_$h.json 2>/dev/null
f=$(find /root/repo/gpurun_out/r2AB/prof_rware_$h -name '*kernel_stats.csv' | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:7]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3,1), 'us', round(float(r['TotalDurationNs'])/tot*100,1),'%')
PY
done
