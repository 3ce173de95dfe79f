# in-kernel region counters + dynamic instruction counts of the warehouse actor-critic collector (BASELINE config 4), BEFORE
O=$GRAFT_REPO_ROOT/gpurun_out/r4C; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
V=$R/codebase_amd/csrc/variants/libmarlhip_acolprof.so
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 2048 2>&1 | tee $O/prof64.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 128 2048 2>&1 | tee $O/prof128.txt
MARLHIP_LIB=$V timeout 120 python scripts/prof_ac_collect.py 64 16384 2>&1 | tee $O/prof64_16k.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_inst --output-format csv -- python $R/scripts/prof_ac_collect.py 64 2048 > $O/pmc_inst.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d $O/pmc_wait --output-format csv -- python $R/scripts/prof_ac_collect.py 64 2048 > $O/pmc_wait.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4C"
for d in ("pmc_inst","pmc_wait"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O+"/"+d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "ac_collect" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(d,k,{c:sum(x)/len(x) for c,x in v.items()})
PY
find $O -name "*.db" -delete
