# warehouse env core rewritten branch-free (rw_step / rw_resolve / bit-mask observation): parity tests + region counters
O=$GRAFT_REPO_ROOT/gpurun_out/r4D; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_rware.py tests/test_gpu_collector_variants.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 2048 2>&1 | tail -9 | tee $O/prof64.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 2048 2>&1 | tail -9 | tee $O/prof128.txt
