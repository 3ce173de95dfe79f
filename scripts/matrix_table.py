#!/usr/bin/env python
"""The DESIGN.md section-6 table from profiles/<prefix>_bench_matrix.jsonl (one bench.py line per row).

    python scripts/matrix_table.py r03            # prints the markdown rows
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pfx = sys.argv[1] if len(sys.argv) > 1 else "r03"
    print("| Workload | cadence | env-steps/s | ms / round | roofline of the loss/grad (or update) stage |")
    print("|---|---|---|---|---|")
    for line in open(os.path.join(ROOT, "profiles", pfx + "_bench_matrix.jsonl")):
        d = json.loads(line)
        c, r = d["config"], d.get("roofline") or {}
        w = c["workload"]
        if "split" in d.get("dtype", ""):
            w += " **[opt-in split16 learner]**"
        if "mixer first layers" in d.get("dtype", ""):
            w += " [opt-in fp16 first mixer layers]"
        if c.get("cadence"):
            cad = "%s (U=%s, B=%s)" % (c["cadence"], c.get("updates_per_round"), c.get("update_batch_episodes"))
            if c.get("hparams") == "reference" and c["cadence"] == "ratio":
                cad += ", idqn.yaml lr / target"
        else:
            cad = "one update per rollout"
        roof = "%.3f of %s %s (%s, %.0f µs per launch group)" % (r.get("frac") or 0, r.get("peak"), r.get("unit"), r.get("kernel"), r.get("avg_launch_us") or 0)
        if r.get("frac_needed"):
            roof = roof.replace(" of ", " (%.3f needed) of " % r["frac_needed"], 1)
        print("| %s | %s | %.2f M | %.2f | %s |" % (w, cad, d["value"] / 1e6, d["ms_per_step"], roof))


if __name__ == "__main__":
    main()
