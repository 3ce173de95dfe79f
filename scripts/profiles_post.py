#!/usr/bin/env python
"""Turn gpurun_out/prof<N>/ (scripts/collect_profiles*.sh, run on the MI355X box) into the committed evidence under profiles/:
    python scripts/profiles_post.py prof4 r04"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "prof2")  # python scripts/profiles_post.py prof3 r03
PFX = sys.argv[2] if len(sys.argv) > 2 else "r02"
DST = os.path.join(ROOT, "profiles")


def first(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)  # gpurun_out/ accumulates the files of every collection run: take the newest
    return max(g, key=os.path.getmtime) if g else None


def counters(tag):
    f = first(f"pmc_{tag}/**/*_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f is None:
        return acc
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    n = name.replace("void marl::", "").replace("marl::", "")
    return n.split("(")[0]


def hbm_pmc_table():
    """profiles/rNN_hbm_ubench_pmc.md: scripts/ubench_hbm.py under the memory-side counters - sector traffic per launch of the HBM-side API
    kernels next to their algorithmic bytes (WRITE_SIZE in KiB as reported; FETCH_SIZE doubled, MI355X_MICROARCH.md "HBM")"""
    wr, fe = counters("hbm_WRITE_SIZE"), counters("hbm_FETCH_SIZE")
    if not wr and not fe:
        return
    timed = {}
    try:
        timed = json.load(open(os.path.join(SRC, "hbm_ubench.txt")))
    except (OSError, ValueError):
        pass
    with open(os.path.join(DST, PFX + "_hbm_ubench_pmc.md"), "w") as o:
        o.write("# HBM-side API kernels under the memory-side counters (rocprofv3 --pmc WRITE_SIZE | FETCH_SIZE -- python scripts/ubench_hbm.py; " + PFX + ")\n\n"
                "Per launch, mean over the launches of the run (several problem sizes share a kernel name: the max column is the largest one).  "
                "read MB = 2 x FETCH_SIZE KiB (gfx950 tallies 128-B read requests at 64 B), write MB = WRITE_SIZE KiB.\n\n"
                "| kernel | launches | write MB mean / max | read MB mean / max |\n|---|---|---|---|\n")
        for k in sorted(set(wr) | set(fe)):
            if "marl::" not in k:
                continue
            w, f = wr.get(k, {}).get("WRITE_SIZE", []), fe.get(k, {}).get("FETCH_SIZE", [])
            o.write(f"| {short(k)[:60]} | {max(len(w), len(f))} | " + (f"{sum(w) / len(w) * 1.024e-3:.1f} / {max(w) * 1.024e-3:.1f}" if w else "-") + " | "
                    + (f"{2 * sum(f) / len(f) * 1.024e-3:.1f} / {2 * max(f) * 1.024e-3:.1f}" if f else "-") + " |\n")
        if timed:
            o.write("\nTimed run of the same script (algorithmic bytes / measured time):\n\n| row | us | achieved GB/s | of 8 TB/s |\n|---|---|---|---|\n")
            for k, v in timed.items():
                o.write(f"| {k} | {v.get('us', 0):.1f} | {v.get('achieved_GBs', 0):.0f} | {v.get('frac_of_8TBs', 0):.3f} |\n")


def main():
    hp = os.path.join(SRC, "head.txt") if os.path.exists(os.path.join(SRC, "head.txt")) else os.path.join(ROOT, "scripts", "_bin", "head.txt")
    head = open(hp).read().strip() if os.path.exists(hp) else None
    for tag, out in (("stats", PFX + "_bench_ratio_kernel_stats.csv"), ("stats_h128", PFX + "_bench_ratio_H128_kernel_stats.csv"),
                     ("stats_hbm", PFX + "_hbm_ubench_kernel_stats.csv"), ("stats_gru64", PFX + "_gru_H64_kernel_stats.csv"),
                     ("stats_rware_ia2c", PFX + "_rware_ia2c_tiny4ag_H128_kernel_stats.csv"),
                     ("stats_qmix8p", PFX + "_qmix_15x15_8p5f_H128_kernel_stats.csv"),
                     ("stats_maa2c8p", PFX + "_maa2c_15x15_8p5f_H128_kernel_stats.csv"), ("stats_mappo_rware", PFX + "_mappo_rware_tiny4ag_H128_kernel_stats.csv"),
                     ("stats_rware_ia2c64", PFX + "_rware_ia2c_tiny4ag_H64_kernel_stats.csv"), ("stats_ia2c64", PFX + "_ia2c_8x8_2p3f_H64_kernel_stats.csv"),
                     ("stats_envonly", PFX + "_env_only_kernel_stats.csv"), ("stats_forcedist", PFX + "_forcedist_1rank_kernel_stats.csv"),
                     ("stats_reference", PFX + "_reference_cadence_kernel_stats.csv"), ("stats_vdn4p", PFX + "_vdn_15x15_4p5f_H128_kernel_stats.csv")):
        f = first(f"{tag}/**/*_kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(DST, out))
    if os.path.exists(os.path.join(SRC, "matrix.jsonl")):
        shutil.copy(os.path.join(SRC, "matrix.jsonl"), os.path.join(DST, PFX + "_bench_matrix.jsonl"))
    if os.path.exists(os.path.join(SRC, "matrix_h64.jsonl")):  # every BASELINE-config row after the reduce kernel's record loop changed
        shutil.copy(os.path.join(SRC, "matrix_h64.jsonl"), os.path.join(DST, PFX + "_bench_rows_final_tree.jsonl"))
    if os.path.exists(os.path.join(SRC, "matrix_ia2c.jsonl")):  # the actor-critic rows again on the tree that carries stacked GRU layers
        shutil.copy(os.path.join(SRC, "matrix_ia2c.jsonl"), os.path.join(DST, PFX + "_bench_matrix_ac_rows_final_tree.jsonl"))
    if os.path.exists(os.path.join(SRC, "mfma_ubench.txt")):
        with open(os.path.join(DST, PFX + "_mfma_valu_ubench.txt"), "w") as o:
            for f in ("mfma_ubench.txt", "mfma_ubench2.txt", "mfma_ubench3.txt", "mfma_ubench4.txt"):
                p = os.path.join(SRC, f)
                if os.path.exists(p):
                    o.write(f"==== scripts/{f.replace('.txt', '.hip')} (MI355X, one wave per SIMD unless stated, 256 workgroups x 256 threads) ====\n{open(p).read()}\n")
            o.write("-> mfma_ubench4 (read its TFLOP/s column): a second / fourth wave on the SIMD recovers only ~10-13 % (69 -> 76 -> 78 TFLOP/s at 4 VALU\n"
                    "   per MFMA): the VALU work occupies the pipe the f32 MFMA runs on (about 7 cycles per v_fma even when another wave's MFMAs are\n"
                    "   ready), it is not an in-order-issue stall of one wave.  Restructuring the learner for 2 waves per SIMD (it needs 444 of 512\n"
                    "   registers today) would not pay.\n")
    if os.path.exists(os.path.join(SRC, "hbm_ubench.txt")):
        shutil.copy(os.path.join(SRC, "hbm_ubench.txt"), os.path.join(DST, PFX + "_hbm_ubench.txt"))
    for name in ("bench_default_line.json", "forcedist_line.json", "episode_lengths.json", "episode_lengths_clear_stale.json"):
        if os.path.exists(os.path.join(SRC, name)) and os.path.getsize(os.path.join(SRC, name)) > 0:
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, PFX + "_" + name))
    hbm_pmc_table()
    # ---- HBM traffic of the learner kernels (two TCC passes per workload; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM")
    P, D, T, B = 2, 15, 25, 4096
    alg_read = B * (4 * P * D * (T + 1) + P * T * 5 + (T + 1) + T)
    out = {"source": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum> -- python bench.py --steps 4 --warmup 1 "
                     "--no-cpu-baseline --no-modes --no-kernel-timing [--hidden 128 | --split16] (one counter set per run), MI355X, " + PFX,
           "head": head,
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B "
                    "read requests at 64 B), write bytes = WRITE_SIZE (the reduce kernel's known 5.7 MB record read calibrates the read side)",
           "kernels": {}, "workloads": {}}
    def replay_bytes(p_, d_, b_):  # one sampled episode = 4 P D (T + 1) + P T 5 + (T + 1) + T bytes
        return b_ * (4 * p_ * d_ * (T + 1) + p_ * T * 5 + (T + 1) + T)

    # round 6: every `modes` row of the default line that a PMC pass was collected for; `files` = the kernel sources the
    # figure is keyed to (bench.traffic_from_profile refuses to quote it when they hash differently)
    extra = (
        ("_vdn64", "vdn:lbforaging:Foraging-15x15-4p-5f-v3:N8192:H64:B8192:T25:rnn0", ("dqn_lossgrad_kernel", "vdn_mix_kernel"),
         {"replay_read_two_passes": 2 * replay_bytes(4, 27, 8192), "hidden_layers_write_plus_read": 2 * 4 * T * 8192 * 2 * 64 * 4,
          "mixer_planes": 13 * T * 8192 * 4}, ("dqn_update_kernels.h", "mlp.h")),
        ("_vdn128", "vdn:lbforaging:Foraging-15x15-4p-5f-v3:N8192:H128:B8192:T25:rnn0", ("tp_fwd_kernel", "tp_mix_kernel", "tp_bwd_kernel"),
         {"replay_read_two_passes": 2 * replay_bytes(4, 27, 8192), "h1_h2_activations_write_plus_read": 2 * 2 * 4 * T * 8192 * 128 * 4},
         ("dqn_update_tp.h", "mlp.h")),
        ("_qmix8p", "qmix:lbforaging:Foraging-15x15-8p-5f-v3:N8192:H128:B8192:T25:rnn0", ("tp_fwd_kernel", "tp_bwd_kernel", "qmix_"),
         {"replay_read_two_passes": 2 * replay_bytes(8, 39, 8192), "h1_h2_activations_write_plus_read": 2 * 2 * 8 * T * 8192 * 128 * 4,
          "mixer_state_rows_read (online + target + weight gradient)": 3 * T * 8192 * 8 * 39 * 4}, ("dqn_update_tp.h", "mlp.h", "qmix.h")),
        ("_ia2c_rware", "ia2c:rware:rware-tiny-4ag-v2:N2048:H128:T500", ("mlp_rows_fwd_kernel", "ac_elem_kernel", "ac_metrics_kernel", "tp_bwd_kernel", "dqn_reduce_kernel"),  # (the update stage: not the collector)
         {"rollout_rows_read (critic, target critic, actor backward)": 3 * 4 * 500 * 2048 * 71 * 4,
          "h1_h2_records_write_plus_read (critics)": 2 * 2 * 4 * 500 * 2048 * 128 * 4,
          "h1_h2_records_read (actors: written by the collector, outside this stage)": 2 * 4 * 500 * 2048 * 128 * 4,
          "logits_values_gradients_planes": 4 * 500 * 2048 * (5 + 5 + 1 + 1 + 1 + 1) * 4}, ("a2c_core.h", "dqn_update_tp.h", "mlp.h", "mlp_keep.h")))
    for row in (
            ("", "idqn:lbforaging:Foraging-8x8-2p-3f-v3:N4096:H64:B4096:T25:rnn0", ("dqn_lossgrad_kernel",),
             {"replay_read": alg_read, "partial_records_write": 256 * (5574 + 2) * 4}),
            ("_h128", "idqn:lbforaging:Foraging-8x8-2p-3f-v3:N4096:H128:B4096:T25:rnn0", ("tp_fwd_kernel", "tp_mix_kernel", "tp_bwd_kernel"),
             {"replay_read_two_passes": 2 * alg_read, "h2_activations_write_plus_read": 2 * P * T * B * 128 * 4, "partial_records_write": 256 * (19334 + 2) * 4}),
            ("_s16", "idqn:lbforaging:Foraging-8x8-2p-3f-v3:N4096:H64:B4096:T25:rnn0:split16", ("dqn_lossgrad_h16_kernel",),
             {"replay_read": alg_read, "partial_records_write": 256 * (5574 + 2) * 4})) + extra:
        sfx, key, match, alg = row[:4]
        files_override = row[4] if len(row) > 4 else None
        fe, wr, tcc = counters("FETCH_SIZE" + sfx), counters("WRITE_SIZE" + sfx), counters("TCC" + sfx)
        kernels = {}
        for k in sorted(set(fe) | set(wr)):
            if "marl::" not in k:
                continue
            f = fe[k].get("FETCH_SIZE", [])
            w = wr[k].get("WRITE_SIZE", [])
            ent = {"launches_profiled": max(len(f), len(w))}
            if f:
                ent["FETCH_SIZE_KiB_per_launch"] = sum(f) / len(f)
                ent["hbm_read_bytes_corrected"] = 2.0 * 1024.0 * sum(f) / len(f)
            if w:
                ent["WRITE_SIZE_KiB_per_launch"] = sum(w) / len(w)
                ent["hbm_write_bytes"] = 1024.0 * sum(w) / len(w)
            if f and w:
                ent["traffic_bytes"] = ent["hbm_read_bytes_corrected"] + ent["hbm_write_bytes"]
            if k in tcc:
                h, m = tcc[k].get("TCC_HIT_sum", []), tcc[k].get("TCC_MISS_sum", [])
                if h and m:
                    ent["l2_hit_rate"] = sum(h) / max(1.0, sum(h) + sum(m))
            kernels[short(k)] = ent
        if not kernels:
            continue
        out["kernels"]["default" if not sfx else sfx[1:]] = kernels
        parts = [v for k, v in kernels.items() if any(k.startswith(mm) for mm in match) and "traffic_bytes" in v]
        if parts:
            sys.path.insert(0, ROOT)
            import bench  # kernel_source_hash: the key bench.py checks before quoting the figure (run this script on the PROFILED tree)

            files = files_override or bench.TRAFFIC_SOURCES["split16" if sfx == "_s16" else ("H128" if sfx == "_h128" else "")]
            # per UPDATE (= per launch of the group bench.py's timer brackets): a kernel launched k times per update counts k times - the
            # matched kernel with the fewest launches in the run is the once-per-update one
            n_upd = max(1, min(v["launches_profiled"] for v in parts))
            out["workloads"][key] = {"kernel": " + ".join(k for k in kernels if any(k.startswith(mm) for mm in match)),
                                     "traffic_bytes": sum(v["traffic_bytes"] * v["launches_profiled"] for v in parts) / n_upd,
                                     "updates_profiled": n_upd, "algorithmic_bytes": alg,
                                     "source_files": list(files), "source_hash": bench.kernel_source_hash(files)}
            hp2 = os.path.join(SRC, f"head{sfx or '_default'}.txt")  # a workload profiled again later than the rest (scripts/collect_profiles_r06_*.sh)
            if os.path.exists(hp2):
                out["workloads"][key]["head"] = open(hp2).read().strip()
    json.dump(out, open(os.path.join(DST, PFX + "_pmc_traffic.json"), "w"), indent=1)
    # ---- SQ counters
    sq, inst = counters("SQ"), counters("INST")
    rows = []
    for k, v in sq.items():
        if "marl::" not in k or "SQ_WAVE_CYCLES" not in v:
            continue
        w = sum(v["SQ_WAVE_CYCLES"])
        g = lambda n: sum(v.get(n, [0.0])) / w  # noqa: E731
        lds = sum(v.get("SQ_LDS_IDX_ACTIVE", [0.0]))
        row = [short(k)[:70], len(v["SQ_WAVE_CYCLES"]), f"{w / len(v['SQ_WAVE_CYCLES']):.3g}", f"{g('SQ_WAIT_ANY'):.2f}", f"{g('SQ_WAIT_INST_ANY'):.2f}",
               f"{g('SQ_ACTIVE_INST_ANY'):.2f}", f"{g('SQ_WAIT_INST_LDS'):.3f}", f"{sum(v.get('SQ_VALU_MFMA_BUSY_CYCLES', [0.0])) / (4 * w):.3f}",
               f"{sum(v.get('SQ_LDS_BANK_CONFLICT', [0.0])) / lds:.2f}" if lds else "-"]
        if k in inst:
            iv = inst[k]
            n = max(1, len(iv.get("SQ_INSTS_MFMA", [1])))
            row.append(" / ".join(f"{sum(iv.get(c, [0.0])) / n:.3g}" for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD")))
        else:
            row.append("-")
        rows.append(row)
    with open(os.path.join(DST, PFX + "_sq_pmc_summary.md"), "w") as o:
        o.write("# SQ counters (rocprofv3 --pmc, MI355X; " + PFX + ")\n\n"
                f"git head of the profiled tree: `{head}`.  Fractions are of SQ_WAVE_CYCLES (quad-cycles); `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 x "
                "SQ_WAVE_CYCLES).  Instruction counts are per launch (all waves): VALU / MFMA / LDS / SALU / VMEM_RD.\n\n"
                "| kernel | launches | wave quad-cycles / launch | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | WAIT_INST_LDS | mfma_busy | LDS bank-conflict / LDS active | instructions per launch |\n"
                "|---|---|---|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r[2])):
            o.write("| " + " | ".join(str(x) for x in r) + " |\n")
    print("profiles written:", sorted(f for f in os.listdir(DST) if f.startswith(PFX + "_")))


if __name__ == "__main__":
    main()
