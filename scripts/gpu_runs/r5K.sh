#!/bin/bash
# round 5, run K: evidence refresh after the kept forward pass and the critics' half-chip stream (C-ABI 213 / 214) - kernel stats of the actor-critic rows, their bench lines, the
# default line (its `modes` carry BASELINE config 4), then the whole GPU suite on this tree.  `python scripts/profiles_merge_ac.py prof5k r05`
# merges the result into profiles/.
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof5k"; mkdir -p "$O"; rm -rf "$O"/stats* "$O/matrix_ac.jsonl"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
st() { timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats$1 --output-format csv -- $B $2 > $O/stats$1.log 2>&1; }
st _rware_ia2c "--steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"
st _maa2c8p "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128"
st _ia2c64 "--steps 20 --warmup 3 --algo ia2c"
cd $R
: > $O/matrix_ac.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix_ac.jsonl; }
run --steps 100 --warmup 5 --algo ia2c
run --steps 100 --warmup 5 --algo ia2c --hidden 128
run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128
run --steps 3 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 40 --warmup 5 --algo ippo
MARLHIP_AC_NO_OVERLAP=1 run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
MARLHIP_AC_NO_OVERLAP=1 MARLHIP_AC_NO_KEEP=1 run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 60 --warmup 5 --algo ia2c --envs 2048 --hidden 128
wc -l $O/matrix_ac.jsonl
( timeout 600 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err )
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -6 | tee $O/gpu_suite.txt
