#!/bin/bash
# Round-6 evidence, collected on the GPU box (gpurun) on the FINAL tree: PMC traffic (TCC passes) of the default line, the hidden-128 line AND the
# `modes` rows of BASELINE configs 3 / 4 / 5, SQ counters of the default and the hidden-128 line, kernel stats of the BASELINE
# configs and the reference cadence, the forced one-rank data-parallel profile, the bench matrix, the trained policy's episode lengths.
# Everything lands under gpurun_out/prof6/; `python scripts/profiles_post.py prof6 r06` turns it into the committed files under profiles/.
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof6"; mkdir -p "$O"; rm -rf "$O"/pmc_* "$O"/stats* "$O/matrix.jsonl"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/scripts/_bin/head.txt $O/head.txt 2>/dev/null
B="python $R/bench.py --no-cpu-baseline --no-modes"
pmc() { timeout 300 rocprofv3 --kernel-trace --pmc $1 -d $O/pmc_$2 --output-format csv -- $B --steps ${4:-4} --warmup 1 --no-kernel-timing $3 > $O/pmc_$2.log 2>&1; }
V4="--algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192"
Q8="--algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128"
RW="--algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128"
for c in FETCH_SIZE WRITE_SIZE; do
  pmc $c $c ""; pmc $c ${c}_h128 "--hidden 128"
  pmc $c ${c}_vdn64 "$V4" 2; pmc $c ${c}_vdn128 "$V4 --hidden 128" 2; pmc $c ${c}_qmix8p "$Q8" 2; pmc $c ${c}_ia2c_rware "$RW" 3
done
pmc "TCC_HIT_sum TCC_MISS_sum" TCC ""; pmc "TCC_HIT_sum TCC_MISS_sum" TCC_h128 "--hidden 128"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_SQ --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_INST --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_INST.log 2>&1
st() { timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats$1 --output-format csv -- $B $2 > $O/stats$1.log 2>&1; }
st "" "--steps 20 --warmup 3"
st _h128 "--steps 10 --warmup 2 --hidden 128"
st _reference "--steps 2 --warmup 1 --cadence reference"
st _vdn4p "--steps 4 --warmup 1 $V4 --hidden 128"
st _rware_ia2c "--steps 4 --warmup 1 $RW"
st _qmix8p "--steps 2 --warmup 1 $Q8"
st _maa2c8p "--steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128"
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_forcedist --output-format csv -- $B --steps 20 --warmup 3 > $O/stats_forcedist.log 2>&1
cd $R
: > $O/matrix.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix.jsonl; }
run --steps 60 --warmup 5
MARLHIP_NO_PLAN=1 run --steps 60 --warmup 5
run --steps 60 --warmup 5 --hparams tuned
run --steps 30 --warmup 3 --hidden 128
run --steps 4 --warmup 1 --cadence reference
run --steps 60 --warmup 5 --cadence env-only
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05
run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05 --clear-stale
MARLHIP_NO_PLAN=1 run --steps 20 --warmup 2 --hparams tuned --pretrain-rounds 1500 --eps-fixed 0.05 --clear-stale
run --steps 10 --warmup 2 $V4
run --steps 6 --warmup 2 $V4 --hidden 128
run --steps 20 --warmup 3 --algo qmix
run --steps 4 --warmup 1 $Q8
run --steps 4 --warmup 1 $Q8 --mixer-fp16
run --steps 100 --warmup 5 --algo ia2c
run --steps 100 --warmup 5 --algo ia2c --hidden 128
run --steps 20 --warmup 2 $RW
MARLHIP_AC_NO_OVERLAP=1 run --steps 20 --warmup 2 $RW
run --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128
run --steps 3 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 10 --warmup 2 --rnn
wc -l $O/matrix.jsonl
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 $B --steps 60 --warmup 5 2>/dev/null | grep '^{' > $O/forcedist_line.json
timeout 300 python $R/scripts/episode_length_hist.py 1500 > $O/episode_lengths.json 2>/dev/null
timeout 300 python $R/scripts/episode_length_hist.py 1500 --clear-stale > $O/episode_lengths_clear_stale.json 2>/dev/null
( timeout 900 python $R/bench.py > $O/bench_default_line.json 2> $O/bench_default_line.err )
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
