mkdir -p gpurun_out/r2V
timeout 900 python -m pytest tests/test_gru.py -x -q -m gpu > gpurun_out/r2V/gru_tests.log 2>&1; echo "gru tests rc=$?"; tail -4 gpurun_out/r2V/gru_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2V/prof_gru64 --output-format csv -- python /root/repo/bench.py --rnn --steps 4 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r2V/bench_gru64.json 2>/root/repo/gpurun_out/r2V/bench_gru64.err
cd /root/repo
tail -1 gpurun_out/r2V/bench_gru64.json | cut -c1-260
f=$(find gpurun_out/r2V/prof_gru64 -name '*kernel_stats.csv' | head -1); head -9 $f | cut -c1-60,200-330
timeout 300 python bench.py --rnn --algo qmix --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
