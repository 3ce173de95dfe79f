# after the fused wide critics: the other users of a2c_core.h (recurrent AC, standardised returns, sharing, layers, host API, two ranks) + region
# counters of the 8-agent hidden-128 LBF collector (now the largest kernel of the MAA2C 15x15-8p round)
O=$GRAFT_REPO_ROOT/gpurun_out/r4P; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gru.py tests/test_gpu_standardise.py tests/test_gpu_sharing.py tests/test_gpu_layers.py tests/test_gpu_host_api.py tests/test_gpu_two_ranks.py tests/test_gpu_checkpoints.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 4096 lbforaging:Foraging-15x15-8p-5f-v3 2>&1 | tail -11 | tee $O/prof_lbf8p_128.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 4096 lbforaging:Foraging-15x15-8p-5f-v3 2>&1 | tail -11 | tee $O/prof_lbf8p_64.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 4096 lbforaging:Foraging-15x15-4p-5f-v3 2>&1 | tail -11 | tee $O/prof_lbf4p_128.txt
