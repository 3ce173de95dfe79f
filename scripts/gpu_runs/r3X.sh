# round 3: learning sanity of the DQN family on the final tree (new mixer kernels, stored two-pass learner) with round 1's protocol
# (profiles/r01_learning_curves.md: IDQN 0.27, VDN 0.10, QMIX 0.75 at 90 % of 30 M steps, seed 0)
O=$GRAFT_REPO_ROOT/gpurun_out/r3X; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python scripts/dqn_family_curves.py 3e7 1024 > $O/curves.log 2>&1; grep CURVE $O/curves.log
timeout 600 python scripts/dqn_family_curves.py 2e7 1024 lbforaging:Foraging-10x10-3p-3f-v3 algorithm.gamma=0.99 vdn,qmix > $O/curves_3p.log 2>&1; grep CURVE $O/curves_3p.log
