mkdir -p gpurun_out/r2AK
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gru.py tests/test_gpu_host_api.py -q -m gpu 2>&1 | tail -3 | cut -c1-200
timeout 200 python scripts/ubench_hbm.py > gpurun_out/r2AK/hbm_ubench.txt 2>&1; tail -3 gpurun_out/r2AK/hbm_ubench.txt | cut -c1-900
cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2AK/stats_hbm --output-format csv -- python /root/repo/scripts/ubench_hbm.py > /dev/null 2>&1
