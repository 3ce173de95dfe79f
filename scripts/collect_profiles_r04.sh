#!/bin/bash
# Round-4 evidence, collected on the GPU box (gpurun): PMC traffic (TCC passes) and SQ counters of the default line, the hidden-128 line and
# the opt-in split16 line, kernel stats of the BASELINE configs, the bench matrix.  Everything lands under gpurun_out/prof4/;
# `python scripts/profiles_post.py prof4 r03` turns it into the committed files under profiles/.
O="${GRAFT_REPO_ROOT:?}/gpurun_out/prof4"; mkdir -p "$O"; rm -rf "$O"/pmc_* "$O"/stats* "$O/matrix.jsonl"; cd /tmp; export TMPDIR=/tmp  # (no leftovers of an earlier collection)
R=$GRAFT_REPO_ROOT
cp $R/scripts/_bin/head.txt $O/head.txt 2>/dev/null
B="python $R/bench.py --no-cpu-baseline --no-modes"
pmc() { timeout 300 rocprofv3 --kernel-trace --pmc $1 -d $O/pmc_$2 --output-format csv -- $B --steps 4 --warmup 1 --no-kernel-timing $3 > $O/pmc_$2.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do pmc $c $c ""; pmc $c ${c}_h128 "--hidden 128"; pmc $c ${c}_s16 "--split16"; done
pmc "TCC_HIT_sum TCC_MISS_sum" TCC ""; pmc "TCC_HIT_sum TCC_MISS_sum" TCC_h128 "--hidden 128"; pmc "TCC_HIT_sum TCC_MISS_sum" TCC_s16 "--split16"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_SQ --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_INST --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_INST.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $B --steps 20 --warmup 3 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 10 --warmup 2 --hidden 128 > $O/stats_h128.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_gru64 --output-format csv -- $B --steps 4 --warmup 1 --rnn > $O/stats_gru64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_rware_ia2c --output-format csv -- $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/stats_rware_ia2c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_rware_ia2c64 --output-format csv -- $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64 > $O/stats_rware_ia2c64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_ia2c64 --output-format csv -- $B --steps 50 --warmup 5 --algo ia2c > $O/stats_ia2c64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_envonly --output-format csv -- $B --steps 50 --warmup 5 --cadence env-only > $O/stats_envonly.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix2p --output-format csv -- $B --steps 10 --warmup 2 --algo qmix > $O/stats_qmix2p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_maa2c8p --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/stats_maa2c8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_mappo_rware --output-format csv -- $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/stats_mappo_rware.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix8p --output-format csv -- $B --steps 2 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/stats_qmix8p.log 2>&1
cd $R
: > $O/matrix.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix.jsonl; }
run --steps 60 --warmup 5
run --steps 60 --warmup 5 --split16
run --steps 60 --warmup 5 --hparams tuned
run --steps 30 --warmup 3 --hidden 128
run --steps 4 --warmup 1 --cadence reference
run --steps 60 --warmup 5 --cadence env-only
run --steps 30 --warmup 3 --updates-per-round 128 --update-batch 4096
run --steps 10 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128
run --steps 20 --warmup 3 --algo qmix
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
run --steps 6 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
run --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --mixer-fp16
run --steps 100 --warmup 5 --algo ia2c
run --steps 100 --warmup 5 --algo ia2c --hidden 128
run --steps 50 --warmup 5 --algo ippo --hidden 128
run --steps 20 --warmup 3 --algo ia2c --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 16384 --hidden 128
run --steps 100 --warmup 5 --algo maa2c --hidden 128
run --steps 50 --warmup 5 --algo mappo --hidden 128
run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 3 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 16384 --hidden 128
run --steps 3 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128
run --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128
run --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 64
run --steps 5 --warmup 1 --algo ia2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128
run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 10 --warmup 2 --rnn
run --steps 5 --warmup 1 --rnn --hidden 128
run --steps 10 --warmup 2 --rnn --algo qmix
run --steps 20 --warmup 2 --rnn --algo ia2c --hidden 128
run --steps 20 --warmup 2 --rnn --algo ippo
wc -l $O/matrix.jsonl
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete; du -sh $O
