#!/bin/bash
# round 6, run AI: the gradient reduce's record loop with 16 / 32 loads in flight per thread instead of 8 (MARL_REDUCE_BATCH: explicit batches; #pragma unroll 16 / 32 serialised every load: 5.7 -> 17.7 us), alternating
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6AI"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
B="python bench.py --no-cpu-baseline --no-modes --steps 60 --warmup 5"
: > $O/rows.txt
for rep in 1 2; do
  for v in base rb16 rb32; do
    if [ $v = base ]; then unset MARLHIP_LIB; else export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_$v.so; fi
    timeout 400 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['avg_launch_us'])" >> $O/rows.txt
  done
done
cat $O/rows.txt
cd /tmp; export TMPDIR=/tmp
for v in base rb32; do
  if [ $v = base ]; then unset MARLHIP_LIB; else export MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$v --output-format csv -- python $R/bench.py --no-cpu-baseline --no-modes --steps 20 --warmup 3 > $O/stats_$v.log 2>&1
  f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -4 $f | cut -c1-160
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
