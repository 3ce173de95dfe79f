# round 3: parity of the stored-hidden-layer actor-critic passes and of the multi-step fused QMIX mixer kernels against the oracle ports
O=$GRAFT_REPO_ROOT/gpurun_out/r3M; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_ac_update.py tests/test_gpu_qmix.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -40 $O/tests.log | cut -c1-400
