"""Merge a partial evidence run (scripts/gpu_runs/r5K.sh: the actor-critic rows after the kept forward pass) into profiles/:
    python scripts/profiles_merge_ac.py prof5k r05
kernel stats of the re-profiled workloads replace the files of the same name, bench lines replace the matrix rows with the same
(metric, workload, dtype) key - rows of a workload the matrix did not have are appended - and the default line is replaced."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, PFX = os.path.join(ROOT, "gpurun_out", sys.argv[1]), sys.argv[2]
DST = os.path.join(ROOT, "profiles")


def key(d):
    c = d.get("config", {})
    r = d.get("roofline") or {}
    return (d.get("metric"), c.get("workload"), d.get("dtype"), bool(r.get("actor_forward_kept", True)), c.get("note"))


for tag, out in (("stats_rware_ia2c", "_rware_ia2c_tiny4ag_H128_kernel_stats.csv"), ("stats_maa2c8p", "_maa2c_15x15_8p5f_H128_kernel_stats.csv"),
                 ("stats_ia2c64", "_ia2c_8x8_2p3f_H64_kernel_stats.csv")):
    f = sorted(glob.glob(os.path.join(SRC, tag, "**", "*_kernel_stats.csv"), recursive=True))
    if f:
        shutil.copy(f[0], os.path.join(DST, PFX + out))
        print("stats:", PFX + out)
mp = os.path.join(DST, PFX + "_bench_matrix.jsonl")
rows = [json.loads(l) for l in open(mp) if l.strip()]
new = [json.loads(l) for l in open(os.path.join(SRC, "matrix_ac.jsonl")) if l.strip()]
for d in new:
    r = d.get("roofline") or {}
    n_envs = d["config"].get("envs_per_gpu", 0)
    if not r.get("actor_forward_kept", True):
        d["config"]["note"] = "MARLHIP_AC_NO_KEEP=1 MARLHIP_AC_NO_OVERLAP=1: the step runs the actors' forward pass itself, on one stream (the A/B row)"
    elif "rware" in d["metric"] and "IA2C" in d["metric"] and n_envs <= 2048 and not r.get("critic_backward_overlaps_next_rollout", False):
        d["config"]["note"] = "MARLHIP_AC_NO_OVERLAP=1: the critics' backward pass stays on the caller's stream (the A/B row of the half-chip stream)"
    for i, r in enumerate(rows):
        if key(r) == key(d):
            rows[i] = d
            break
    else:
        rows.append(d)
with open(mp, "w") as o:
    for r in rows:
        o.write(json.dumps(r) + "\n")
print("matrix rows:", len(rows))
p = os.path.join(SRC, "bench_default_line.json")
if os.path.exists(p) and os.path.getsize(p) > 0:
    shutil.copy(p, os.path.join(DST, PFX + "_bench_default_line.json"))
    print("default line replaced")
