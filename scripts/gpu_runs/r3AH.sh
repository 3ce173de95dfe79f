# round 3: recurrent centralised critics for 3 agents x 24 observations, and the port comparison of the compiled pairs
O=$GRAFT_REPO_ROOT/gpurun_out/r3AH; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gru.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-300
