# round 5, call B: the whole GPU suite on the tree with the generic QMIX mixer stage, the critics' own sharing map, the reordered reduce / Adam
# loads; the reference-cadence row before/after; the HBM micro-benchmarks under the traffic counters (WRITE_SIZE / FETCH_SIZE per kernel)
O=$GRAFT_REPO_ROOT/gpurun_out/r5B; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --durations=12 ) 2>&1 | tail -120 | tee $O/pytest_gpu.log
B="python $R/bench.py --no-cpu-baseline --no-modes"
for a in "--cadence reference --steps 3 --warmup 1" "--steps 40 --warmup 5" "--cadence reference --steps 3 --warmup 1 --hidden 128"; do
  timeout 300 $B $a 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('$a','->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms', {k[:14]:round(v['avg_us'],2) for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/rows.txt
timeout 300 python $R/scripts/ubench_hbm.py 2>/dev/null | python -c "
import sys,json
for k,v in json.loads(sys.stdin.read()).items(): print('UBENCH',k,round(v['us'],1),'us',round(v['achieved_GBs']),'GB/s',round(v['frac_of_8TBs'],3))" | tee $O/ubench.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_ref --output-format csv -- $B --cadence reference --steps 2 --warmup 1 --no-kernel-timing > $O/stats_ref.log 2>&1
# where do the fills / copies of the forced one-rank data-parallel profile come from?  HIP API statistics of the same run
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/stats_dist --output-format csv -- $B --steps 10 --warmup 2 --no-kernel-timing > $O/stats_dist.log 2>&1
for f in $(find $O/stats_dist -name "*hip_api_stats.csv" -o -name "*kernel_stats.csv" | head -4); do echo "== $f"; head -14 $f | cut -c1-150; done
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 $B --steps 40 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l);print('forced-dist','->',round(d['value']/1e6,3),'M', round(d['ms_per_step'],3),'ms', d['rccl_ranks'])" | tee -a $O/rows.txt
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_hbm_$c --output-format csv -- python $R/scripts/ubench_hbm.py > $O/pmc_hbm_$c.log 2>&1
done
python - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5B"
for c in ("WRITE_SIZE","FETCH_SIZE"):
    fs=glob.glob(O+f"/pmc_hbm_{c}/**/*_counter_collection.csv",recursive=True)
    if not fs: print("no counters for",c); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if any(s in k for s in ("replay","lbf_step")): print(c,k,"launches",len(v),"mean",sum(v)/len(v),"max",max(v))
f=glob.glob(O+"/stats_ref/**/*_kernel_stats.csv",recursive=True)
if f:
    for i,l in enumerate(open(f[0])):
        if i<8: print(l.strip()[:200])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
