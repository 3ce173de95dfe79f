#!/bin/bash
# Round 6, after the gradient reduce's record loop changed (csrc/dqn_update_kernels.h: 32 loads in flight): the evidence of the two workloads whose
# kernel sources hash that file again - the default line (PMC traffic + TCC + SQ / instruction counters + kernel stats + the bench line) and
# VDN 15x15-4p at hidden 64 (PMC traffic).  Lands in gpurun_out/prof6/ next to the earlier passes, each under its own head.
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof6"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/scripts/_bin/head.txt $O/head_vdn64.txt 2>/dev/null; cp $R/scripts/_bin/head.txt $O/head_default.txt 2>/dev/null
B="python $R/bench.py --no-cpu-baseline --no-modes"
V4="--algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192"
pmc() { rm -rf $O/pmc_$2; timeout 300 rocprofv3 --kernel-trace --pmc $1 -d $O/pmc_$2 --output-format csv -- $B --steps ${4:-4} --warmup 1 --no-kernel-timing $3 > $O/pmc_$2.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do pmc $c $c ""; pmc $c ${c}_vdn64 "$V4" 2; done
pmc "TCC_HIT_sum TCC_MISS_sum" TCC ""
rm -rf $O/pmc_SQ $O/pmc_INST $O/stats
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_SQ --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_INST --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_INST.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $B --steps 20 --warmup 3 > $O/stats.log 2>&1
cd $R
( timeout 900 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err )
: > $O/matrix_h64.jsonl
for a in "--steps 60 --warmup 5" "--steps 60 --warmup 5 --hparams tuned" "--steps 30 --warmup 3 --hidden 128" "--steps 10 --warmup 2 $V4" "--steps 6 --warmup 2 $V4 --hidden 128" "--steps 4 --warmup 1 --cadence reference" "--steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128" "--steps 20 --warmup 2 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128" "--steps 10 --warmup 2 --rnn"; do
  timeout 400 $B $a 2>/dev/null | grep '^{' >> $O/matrix_h64.jsonl
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
python - <<'PY'
import json,os
o=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/prof6")
d=json.loads(open(o+"/bench_default_line.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for l in open(o+"/matrix_h64.jsonl"):
    r=json.loads(l); print(r["metric"][-32:], round(r["value"]/1e6,3), r["ms_per_step"], (r.get("roofline") or {}).get("frac"))
PY
