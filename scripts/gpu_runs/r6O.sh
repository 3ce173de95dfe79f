#!/bin/bash
# round 6, run O: pass F with THREE row blocks per step (MARL_TP_NBF=3: the step's fixed latency chains amortised over 1.5 x the matrix work;
# 470 registers, 147 KB of LDS) against the product's two, same tree, same box; goldens on the variant
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6O"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
V=$R/codebase_amd/csrc/variants/libmarlhip_nbf3.so
row() { $B --steps 10 --warmup 2 --hidden 128 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$1', round(d['value']/1e6,3), round(d['kernels']['dqn_lossgrad_kernel']['avg_us'],1))"; }
row nbf2
MARLHIP_LIB=$V row nbf3
row nbf2
MARLHIP_LIB=$V row nbf3
MARLHIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path_vs_oracle.py -x -q -m gpu -k "128 or H128 or hidden128" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
MARLHIP_LIB=$V timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 6 --warmup 2 --hidden 128 --no-kernel-timing > $O/stats_h128.log 2>&1
f=$(find $O/stats_h128 -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-60,150-260
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
