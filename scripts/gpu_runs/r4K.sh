# end-to-end sanity on the round's tree: the drop-in drivers learn as before (returns at the end of short runs)
O=$GRAFT_REPO_ROOT/gpurun_out/r4K; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
export MARLHIP_RUN_DIR=$O/ia2c
( time python -m codebase_amd.run +algorithm=ia2c env.name="lbforaging:Foraging-8x8-2p-3f-v3" env.time_limit=25 env.parallel_envs=4096 algorithm.model.actor.layers=[64,64] algorithm.model.critic.layers=[64,64] algorithm.total_steps=200000000 algorithm.eval_interval=50000000 seed=0 ) > $O/ia2c.log 2>&1
tail -3 $O/ia2c.log; python - <<'PY'
import pandas as pd, os, glob
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4K/ia2c/**/results.csv", recursive=True):
    df=pd.read_csv(f); print(f.split("/")[-3:], df[["environment_steps","mean_episode_returns"]].tail(4).to_string())
PY
export MARLHIP_RUN_DIR=$O/ia2c_rw
( time python -m codebase_amd.run +algorithm=ia2c env.name="rware:rware-tiny-4ag-v2" env.time_limit=500 env.parallel_envs=2048 algorithm.model.actor.layers=[64,64] algorithm.model.critic.layers=[64,64] algorithm.total_steps=60000000 algorithm.eval_interval=20000000 seed=0 ) > $O/ia2c_rw.log 2>&1
tail -3 $O/ia2c_rw.log; python - <<'PY'
import pandas as pd, os, glob
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4K/ia2c_rw/**/results.csv", recursive=True):
    df=pd.read_csv(f); print(f.split("/")[-3:], df[["environment_steps","mean_episode_returns"]].tail(4).to_string())
PY
