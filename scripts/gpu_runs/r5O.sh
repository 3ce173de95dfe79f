#!/bin/bash
# round 5, run O: a profiled process with the half-chip stream exits cleanly (the stream is destroyed at interpreter exit); smoke; kept / overlap tests
O=$GRAFT_REPO_ROOT/gpurun_out/r5O; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_rware_ia2c --output-format csv -- python $R/bench.py --no-cpu-baseline --no-modes --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/stats_rware_ia2c.log 2>&1
echo "rocprofv3 exit code: $?" | tee $O/rocprof_rc.txt
tail -2 $O/stats_rware_ia2c.log | cut -c1-200
cd $R
python bench.py --no-cpu-baseline --no-modes --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/plain.json 2> $O/plain.err; echo "plain bench exit code: $?" | tee -a $O/rocprof_rc.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_ac_keep.py -q 2>&1 | tail -3
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
