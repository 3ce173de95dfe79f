# LBF rollout collector: region counters (IA2C 8x8-2p-3f at 4096 envs, 64-64 and 128-128; 15x15-4p-5f) + dynamic instruction counts
O=$GRAFT_REPO_ROOT/gpurun_out/r4F; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 4096 lbforaging:Foraging-8x8-2p-3f-v3 2>&1 | tail -9 | tee $O/prof_lbf64.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 4096 lbforaging:Foraging-8x8-2p-3f-v3 2>&1 | tail -9 | tee $O/prof_lbf128.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 8192 lbforaging:Foraging-15x15-4p-5f-v3 2>&1 | tail -9 | tee $O/prof_lbf4p.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d $O/pmc_inst --output-format csv -- python $R/scripts/prof_ac_collect.py 64 4096 lbforaging:Foraging-8x8-2p-3f-v3 > $O/pmc_inst.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4F"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc_inst/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "ac_collect" in r["Kernel_Name"]: acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k,{c:sum(x)/len(x) for c,x in v.items()})
PY
find $O -name "*.db" -delete
