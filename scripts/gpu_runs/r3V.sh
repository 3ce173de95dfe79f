# round 3: in-kernel cycle counters of the learner kernel at B = 32 (reference cadence) and B = 4096
cd $GRAFT_REPO_ROOT; timeout 300 python scripts/prof_small_update.py 2>&1 | tail -5
