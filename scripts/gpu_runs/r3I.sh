# round 3: QMIX mixer - fused per-instance kernels vs the split form per shape (default: fused for one K chunk; variant: up to three)
O=$GRAFT_REPO_ROOT/gpurun_out/r3I; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -q -m gpu -k "qmix or standardise or rware or two_ranks or gru or host_api or layers" > $O/tests_qmix.log 2>&1; echo "qmix tests rc=$?"; tail -4 $O/tests_qmix.log | cut -c1-300
MARLHIP_LIB=$R/codebase_amd/csrc/variants/libmarlhip_fuse3.so timeout 900 python -m pytest tests -q -m gpu -k "qmix" > $O/tests_qmix_v.log 2>&1; echo "qmix tests (variant) rc=$?"; tail -4 $O/tests_qmix_v.log | cut -c1-300
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
run() { # name lib args...
  n=$1; lib=$2; shift 2
  MARLHIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/st_$n --output-format csv -- $B "$@" > $O/$n.log 2>&1
}
D=$R/codebase_amd/csrc/libmarlhip.so; V=$R/codebase_amd/csrc/variants/libmarlhip_fuse3.so
run q2p_def $D --steps 6 --warmup 1 --algo qmix
run q4p_def $D --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run q4p_var $V --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-4p-5f-v3 --envs 8192
run q3p_def $D --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
run q3p_var $V --steps 4 --warmup 1 --algo qmix --env-name lbforaging:Foraging-10x10-3p-3f-v3 --envs 8192
run qrw2_def $D --steps 2 --warmup 1 --algo qmix --env-name rware:rware-tiny-2ag-v2 --time-limit 500 --envs 1024
run qrw2_var $V --steps 2 --warmup 1 --algo qmix --env-name rware:rware-tiny-2ag-v2 --time-limit 500 --envs 1024
run q8p_def $D --steps 3 --warmup 1 --algo qmix --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 8192 --hidden 128
cd $R; python - <<'PY'
import csv,glob,os,json
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r3I"
for d in sorted(glob.glob(O+"/st_*")):
    n=os.path.basename(d)[3:]
    try:
        l=[x for x in open(O+"/"+n+".log").read().splitlines() if x.startswith("{")][-1]; j=json.loads(l); print("==",n,"%.3f M"%(j["value"]/1e6))
    except Exception as e: print("==",n,"ERR",e)
    for f in glob.glob(d+"/*/*kernel_stats.csv"):
        for r in list(csv.DictReader(open(f)))[:40]:
            if "qmix" in r["Name"]: print("   %-70s calls %5s avg_us %9.2f"%(r["Name"].replace("marl::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
