"""Aggregate a rocprofv3 --kernel-trace csv by (kernel, grid): `python scripts/trace_by_grid.py <out dir> <trace subdir> ...` writes
<out dir>/<subdir>.txt (calls, average us, share of the kernel time) and prints its head."""
import collections
import csv
import glob
import sys

O = sys.argv[1]
for tag in sys.argv[2:]:
    out = open(f"{O}/{tag}.txt", "w")
    for f in glob.glob(f"{O}/{tag}/*/*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        if not rows:
            continue
        agg, tot = collections.OrderedDict(), 0.0
        for r in rows:
            k = (r["Kernel_Name"].replace("marl::", "")[:100], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += d
            tot += d
        print(tag, "total kernel us", round(tot), "launches", len(rows), file=out)
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
            print("%-100s grid %8s %5s %4s calls %6d avg_us %9.2f pct %5.1f" % (k[0], k[1], k[2], k[3], a[0], a[1] / a[0], 100 * a[1] / tot), file=out)
    out.close()
    print(open(f"{O}/{tag}.txt").read()[:3600])
