#!/usr/bin/env python
"""profiles/r03_learning_parity.md from profiles/r03_learning/*.jsonl (tests/tools/learning_parity.py writes those: one JSON line per
evaluation point).  The criterion is coarse on purpose (three and six runs) and the report says pass or fail.  Its two numbers (0.20 =
about the seed-to-seed standard deviation of the return past the take-off, one run in three) were chosen with the curves in view, so it is
a sanity bound a broken learner would fail, not a pre-registered test:

  at every checkpoint from 30 % of the run on, the mean return over the HIP runs lies within `TOL` of the mean over the CPU-oracle
  runs, and at the last common checkpoint the share of runs that left the 0.40 plateau (return >= 0.45) differs by at most one run
  in three between the two groups.

    python scripts/learning_parity_report.py > profiles/r03_learning_parity.md
"""
import glob
import json
import os
import statistics as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "profiles", "r03_learning")
TOL = 0.20
PLATEAU = 0.45


def load(pattern):
    runs = []
    for f in sorted(glob.glob(os.path.join(SRC, pattern))):
        rows = [json.loads(l) for l in open(f) if l.strip()]
        if rows:
            runs.append((os.path.basename(f), rows))
    return runs


def main():
    groups = [("CPU oracle loop (`oracle/dqn_port` + `oracle/lbf.py`), N = 1, one update of 32 episodes per episode", load("oracle_refcadence_18M_seed*.jsonl")),
              ("HIP, `codebase_amd.run` vectorised, N = 8 envs, 8 sequential updates of 32 per round", load("vec8x8x32_18M_seed*.jsonl")),
              ("HIP, `codebase_amd.run` vectorised, N = 64 envs, 64 sequential updates of 32 per round", load("vec64x64x32_18M_seed*.jsonl"))]
    ncp = min(len(r) for _, runs in groups for _, r in runs)
    print("# Learning parity at the reference's cadence, to 16-18 M env-steps (round 3; VERDICT r2 weak 11 / next 7)\n")
    print("IDQN, feed-forward 64-64, `lbforaging:Foraging-8x8-2p-3f-v3`, `time_limit` 25, the reference's `idqn.yaml` hyper-parameters (lr 3e-4,")
    print("hard target copy every 200 updates, batch 32, eps 1.0 -> 0.05 over the first half of 18 M steps, gamma 0.99, Double-Q, clip 1.0),")
    print("evaluation at eps = 0.05 every tenth of the run.  All three configurations make ONE optimiser step of 32 sampled episodes per")
    print("collected episode (the reference's cadence: 157x the optimiser steps per env-step of the headline `ratio` cadence); they differ in how")
    print("many episodes are collected between update groups (1, 8, 64) and in who does the arithmetic (torch on one CPU core: 3.3 hours per")
    print("seed; the HIP library: 22-76 s per seed).  Three seeds each; generator `tests/tools/learning_parity.py`, data")
    print("`profiles/r03_learning/*.jsonl`, this file `scripts/learning_parity_report.py`.\n")
    print("| configuration | seed | return at 10 ... %d %% of 18 M steps | updates | wall |" % (10 * ncp))
    print("|---|---|---|---|---|")
    means = []
    finals = []
    for name, runs in groups:
        per_cp = [[] for _ in range(ncp)]
        for fn, rows in runs:
            seed = rows[0]["seed"]
            print("| %s | %d | %s | %d | %.0f s |" % (name, seed, " ".join("%.2f" % r["mean_return"] for r in rows[:ncp]), rows[ncp - 1]["updates"],
                                                   rows[ncp - 1]["wall_s"]))
            for i in range(ncp):
                per_cp[i].append(rows[i]["mean_return"])
        means.append([st.mean(c) for c in per_cp])
        finals.append(per_cp[ncp - 1])
    hip = [st.mean(groups[1][1][k][1][i]["mean_return"] for k in range(len(groups[1][1]))) for i in range(ncp)]
    hip_all = [[r[1][i]["mean_return"] for g in (1, 2) for r in groups[g][1]] for i in range(ncp)]
    hip_mean = [st.mean(c) for c in hip_all]
    print("\n| mean over seeds | " + " | ".join("%d %%" % (10 * (i + 1)) for i in range(ncp)) + " |")
    print("|---|" + "---|" * ncp)
    print("| CPU oracle (3 runs) | " + " | ".join("%.2f" % m for m in means[0]) + " |")
    print("| HIP N = 8 (3 runs) | " + " | ".join("%.2f" % m for m in means[1]) + " |")
    print("| HIP N = 64 (3 runs) | " + " | ".join("%.2f" % m for m in means[2]) + " |")
    print("| HIP, both (6 runs) | " + " | ".join("%.2f" % m for m in hip_mean) + " |")
    diffs = [abs(hip_mean[i] - means[0][i]) for i in range(2, ncp)]
    left_o = sum(v >= PLATEAU for v in finals[0]) / len(finals[0])
    left_h = sum(v >= PLATEAU for v in finals[1] + finals[2]) / len(finals[1] + finals[2])
    ok = max(diffs) <= TOL and abs(left_o - left_h) <= 1.0 / 3.0 + 1e-9
    print("\n## Criterion (`scripts/learning_parity_report.py`; thresholds chosen with the curves in view - a sanity bound, not a pre-registered test)\n")
    print("At every checkpoint from 30 %% on, |mean(HIP) - mean(oracle)| <= %.2f: largest difference **%.3f** (at %d %%).  " % (
        TOL, max(diffs), 10 * (3 + diffs.index(max(diffs)))))
    print("Share of runs above the 0.40 plateau (return >= %.2f) at %d %%: oracle %d of %d, HIP %d of %d.  " % (
        PLATEAU, 10 * ncp, sum(v >= PLATEAU for v in finals[0]), len(finals[0]), sum(v >= PLATEAU for v in finals[1] + finals[2]),
        len(finals[1] + finals[2])))
    print("\n**%s.**\n" % ("PASS" if ok else "FAIL"))
    print("## Reading\n")
    print("* Round 2's comparison stopped at 0.36 / 1.35 M steps, where both sides sit on the 0.02-0.07 noise floor and agreement says nothing.  Here the")
    print("  CPU restatement of the reference loop was run to where the task is learned: it leaves the floor between 3.6 and 5.4 M steps (0.22-0.33),")
    print("  reaches the 0.40 plateau (agents that only load the food one of them can lift alone) by 7.2 M and two seeds in three leave the plateau")
    print("  before 16.2 M (0.57 / 0.80 / 0.39 at 16.2 M).  The HIP path at the same cadence does the same things at the same env-steps: 0.25-0.34 at")
    print("  5.4 M, 0.35-0.40 at 7.2 M, five of six runs above the plateau at 16.2 M (0.56 / 0.79 / 0.85 with 8 envs, 0.78 / 0.38 / 0.62 with 64).")
    print("  Seed-to-seed spread (0.39-0.80 on the CPU, 0.38-0.85 on HIP) is larger than any difference between the groups.")
    print("* The streams differ by construction (Philox draws vs torch's generator, different episode interleaving with N > 1), so curves are compared")
    print("  as distributions over seeds, not point by point.  With three and six runs this criterion can detect a broken learner (flat at the floor, stuck")
    print("  on the plateau in every seed, or a shifted take-off) but not a 10 % difference in sample efficiency.")
    print("* Wall-clock for the same 16.2 M steps and ~0.7 M optimiser steps: 3.3 hours on one CPU core vs 22-76 s, evaluation included.")


if __name__ == "__main__":
    main()
