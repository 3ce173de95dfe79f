#!/bin/bash
# round 6, run N: tp_fwd with the epilogue deferred into the next step's layer-2 region: phase counters (MARL_TP_PROF variant), timing against the
# non-deferred variant of the same tree, H128 goldens / at-size
O="${GRAFT_REPO_ROOT:?}/gpurun_out/r6N"; mkdir -p "$O"; R=$GRAFT_REPO_ROOT; cd $R
B="python $R/bench.py --no-cpu-baseline --no-modes"
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_tpprof.so $B --steps 2 --warmup 1 --hidden 128 2>/dev/null | grep TPPROF | tail -4
row() { $B --steps 10 --warmup 2 --hidden 128 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('$1', round(d['value']/1e6,3), round(d['kernels']['dqn_lossgrad_kernel']['avg_us'],1))"; }
row deferred
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_nodefer.so row not-deferred
row deferred
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_nodefer.so row not-deferred
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path_vs_oracle.py tests/test_gpu_qmix.py tests/test_gpu_standardise.py tests/test_gpu_sharing.py tests/test_action_masks.py "tests/test_gpu_at_size_vs_oracle.py::test_config3_vdn_15x15_4p5f_H128_B8192_vs_oracle_port" -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_h128 --output-format csv -- $B --steps 6 --warmup 2 --hidden 128 --no-kernel-timing > $O/stats_h128.log 2>&1
f=$(find $O/stats_h128 -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-60,150-260
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
