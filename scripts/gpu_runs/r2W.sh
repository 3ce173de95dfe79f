# last validation of the round's final tree: the whole GPU suite + smoke
mkdir -p gpurun_out/r2W
timeout 140 python -m pytest tests -x -q -m gpu > gpurun_out/r2W/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2W/tests.log | cut -c1-200
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
