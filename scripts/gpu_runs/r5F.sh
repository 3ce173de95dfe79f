# round 5, call F: which of qmix_wgrad_kernel's two groups takes the time (MARLHIP_QMIX_WG_ONLY=1|2 runs one alone; timing only, the gradient is incomplete)
O=$GRAFT_REPO_ROOT/gpurun_out/r5F; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-modes --no-kernel-timing"
for only in 0 1 2; do
MARLHIP_QMIX_WG_ONLY=$only timeout 300 rocprofv3 --kernel-trace --stats -d $O/s8p_$only --output-format csv -- $B --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --steps 1 --warmup 1 > $O/s8p_$only.log 2>&1
MARLHIP_QMIX_WG_ONLY=$only timeout 300 rocprofv3 --kernel-trace --stats -d $O/s4p_$only --output-format csv -- $B --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192 --steps 2 --warmup 1 > $O/s4p_$only.log 2>&1
done
for f in $(find $O -name "*kernel_stats.csv" | sort); do echo "== $f"; grep "qmix_wgrad" $f | sed 's/(marl.*)"/"/' | cut -c1-200; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
