# round 5, call D: SQ counters of the 8-agent QMIX mixer stage (BASELINE config 5) - what bounds qmix_l1 / qmix_wgrad / qmix_mix at 0.43
O=$GRAFT_REPO_ROOT/gpurun_out/r5D; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-modes --no-kernel-timing --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128 --steps 1 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_SQ --output-format csv -- $B > $O/pmc_SQ.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc_INST --output-format csv -- $B > $O/pmc_INST.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_TCP --output-format csv -- $B > $O/pmc_TCP.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $O/pmc_MEM --output-format csv -- $B > $O/pmc_MEM.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- $B > $O/stats.log 2>&1
python - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5D"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("SQ","INST","TCP","MEM"):
    for f in glob.glob(O+f"/pmc_{tag}/**/*_counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("void marl::","").replace("marl::","").split("(")[0]
            if "qmix" in k or "tp_" in k: acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    n=max(len(x) for x in v.values())
    print(k, "launches", n)
    print("   ", {c: f"{sum(x)/len(x):.4g}" for c,x in sorted(v.items())})
f=glob.glob(O+"/stats/**/*_kernel_stats.csv",recursive=True)
if f:
    for i,l in enumerate(open(f[0])):
        if i<12: print(l.strip()[:60], l.strip().split('",')[-1][:80] if '",' in l else "")
for t in ("SQ","INST","TCP","MEM"):
    print(t, open(O+f"/pmc_{t}.log").read()[-300:].replace("\n"," | ")[-200:])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
