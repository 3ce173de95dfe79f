# GEMM path: per-launch breakdown (kernel x grid) of the MAA2C 15x15-8p and MAPPO rware rounds
O=$GRAFT_REPO_ROOT/gpurun_out/r4N; mkdir -p $O; R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu-baseline --no-modes"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/tr_maa2c --output-format csv -- $B --steps 3 --warmup 1 --algo maa2c --env-name lbforaging:Foraging-15x15-8p-5f-v3 --envs 4096 --hidden 128 > $O/maa2c8p.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/tr_mappo --output-format csv -- $B --steps 2 --warmup 1 --algo mappo --env-name rware:rware-tiny-4ag-v2 --time-limit 500 --envs 2048 --hidden 128 > $O/mappo.log 2>&1
cd $R; python - <<'PY'
import csv,glob,os,collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4N"
for tag in ("tr_maa2c","tr_mappo"):
    fs=glob.glob(O+"/"+tag+"/*/*kernel_trace.csv")
    out=open(O+"/"+tag+".txt","w")
    for f in fs:
        rows=list(csv.DictReader(open(f)))
        if not rows: continue
        agg=collections.OrderedDict(); tot=0
        for r in rows:
            k=(r["Kernel_Name"].replace("marl::","")[:110], r.get("Grid_Size_X",r.get("Grid_Size","?")), r.get("Grid_Size_Y","?"), r.get("Grid_Size_Z","?"), r.get("Workgroup_Size_X","?"))
            d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
            a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d; tot+=d
        print(tag, "total kernel us", round(tot), "launches", len(rows), file=out)
        for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]:
            print("%-110s grid %8s %5s %4s wg %4s calls %6d avg_us %9.2f pct %5.1f"%(k[0],k[1],k[2],k[3],k[4],a[0],a[1]/a[0],100*a[1]/tot), file=out)
    out.close()
    print(open(O+"/"+tag+".txt").read()[:6000])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
