#!/bin/bash
# Round 6, after csrc/a2c_core.h changed (stacked GRU layers: the workspace layout takes a depth): the actor-critic workload's evidence again -
# PMC traffic passes and kernel stats of BASELINE config 4 (ia2c, rware-tiny-4ag, 2048 envs x 500 steps, 128-128) only; the other five
# workloads' kernel sources are unchanged (their source hashes in profiles/r06_pmc_traffic.json still match).  Lands in gpurun_out/prof6/
# next to the earlier passes; `python scripts/profiles_post.py prof6 r06` then rewrites the committed files, this workload under its own head.
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof6"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/scripts/_bin/head.txt $O/head_ia2c_rware.txt 2>/dev/null
B="python $R/bench.py --no-cpu-baseline --no-modes"
RW="--algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"
pmc() { rm -rf $O/pmc_$2; timeout 300 rocprofv3 --kernel-trace --pmc $1 -d $O/pmc_$2 --output-format csv -- $B --steps ${4:-4} --warmup 1 --no-kernel-timing $3 > $O/pmc_$2.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do pmc $c ${c}_ia2c_rware "$RW" 3; done
rm -rf $O/stats_rware_ia2c
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_rware_ia2c --output-format csv -- $B --steps 4 --warmup 1 $RW > $O/stats_rware_ia2c.log 2>&1
cd $R
: > $O/matrix_ia2c.jsonl
for i in 1 2; do timeout 400 $B --steps 20 --warmup 2 $RW 2>/dev/null | grep '^{' >> $O/matrix_ia2c.jsonl; done
MARLHIP_AC_NO_OVERLAP=1 timeout 400 $B --steps 20 --warmup 2 $RW 2>/dev/null | grep '^{' >> $O/matrix_ia2c.jsonl
timeout 400 $B --steps 10 --warmup 2 --rnn 2>/dev/null | grep '^{' >> $O/matrix_ia2c.jsonl
timeout 400 $B --steps 100 --warmup 5 --algo ia2c 2>/dev/null | grep '^{' >> $O/matrix_ia2c.jsonl
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; ls $O | head -50; cat $O/matrix_ia2c.jsonl | cut -c1-200
