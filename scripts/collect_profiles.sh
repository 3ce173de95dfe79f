#!/bin/bash
# Round-2 evidence, collected on the GPU box (gpurun): kernel stats, PMC traffic (two TCC passes), SQ counters, the bench matrix,
# the MFMA / VALU micro-benchmarks and the HBM micro-benchmark with its own rocprof stats.  Everything lands under gpurun_out/prof2/;
# scripts/profiles_post.py turns it into the committed files under profiles/.
O=$GRAFT_REPO_ROOT/gpurun_out/prof2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $B --steps 20 --warmup 3 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c --output-format csv -- $B --steps 4 --warmup 1 --no-kernel-timing > $O/pmc_$c.log 2>&1; done
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_TCC --output-format csv -- $B --steps 4 --warmup 1 --no-kernel-timing > $O/pmc_TCC.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_SQ --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_INST --output-format csv -- $B --steps 3 --warmup 1 --no-kernel-timing > $O/pmc_INST.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 10 --warmup 2 --hidden 128 > $O/stats_h128.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_gru64 --output-format csv -- $B --steps 4 --warmup 1 --rnn > $O/stats_gru64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_rware_ia2c --output-format csv -- $B --steps 4 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/stats_rware_ia2c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_qmix8p --output-format csv -- $B --steps 2 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 > $O/stats_qmix8p.log 2>&1
cd $R
./scripts/_bin/mfma_ubench > $O/mfma_ubench.txt 2>&1; ./scripts/_bin/mfma_ubench2 > $O/mfma_ubench2.txt 2>&1; ./scripts/_bin/mfma_ubench3 > $O/mfma_ubench3.txt 2>&1; ./scripts/_bin/mfma_ubench4 > $O/mfma_ubench4.txt 2>&1
: > $O/matrix.jsonl
run() { timeout 400 $B "$@" 2>/dev/null | grep '^{' >> $O/matrix.jsonl; }
run --steps 60 --warmup 5
run --steps 30 --warmup 3 --hidden 128
run --steps 4 --warmup 1 --cadence reference
run --steps 60 --warmup 5 --cadence env-only
run --steps 30 --warmup 3 --updates-per-round 128 --update-batch 4096
run --steps 10 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run --steps 6 --warmup 2 --algo vdn --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --hidden 128
run --steps 20 --warmup 3 --algo qmix
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
run --steps 5 --warmup 1 --algo ia2c --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 3 --warmup 1 --algo idqn --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 3 --warmup 1 --algo qmix --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 64
run --steps 10 --warmup 2 --rnn
run --steps 5 --warmup 1 --rnn --hidden 128
run --steps 10 --warmup 2 --rnn --algo qmix
run --steps 20 --warmup 2 --rnn --algo ia2c --hidden 128
run --steps 20 --warmup 2 --rnn --algo ippo
wc -l $O/matrix.jsonl
timeout 200 python scripts/ubench_hbm.py > $O/hbm_ubench.txt 2>&1; tail -12 $O/hbm_ubench.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_hbm --output-format csv -- python $R/scripts/ubench_hbm.py > $O/stats_hbm.log 2>&1
cd $R; git rev-parse HEAD > $O/head.txt 2>/dev/null || true
ls $O | head -30
