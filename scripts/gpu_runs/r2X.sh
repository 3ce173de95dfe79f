mkdir -p gpurun_out/r2X
B="python bench.py --no-cpu-baseline --steps 40 --warmup 4"
timeout 300 $B 2>/dev/null | tail -1 | cut -c60-125
MARLHIP_NO_FUSED_LOOP=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c60-125
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 $B 2>gpurun_out/r2X/force_dist.err | tail -1 | cut -c60-125; tail -3 gpurun_out/r2X/force_dist.err
cd /tmp; export TMPDIR=/tmp
MARLHIP_BENCH_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2X/prof_dist --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
cd /root/repo; f=$(find gpurun_out/r2X/prof_dist -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-50,150-260
