#!/usr/bin/env python
"""Static look at a gfx950 kernel's instruction stream (no GPU needed): per basic block, the order of MFMA / LDS / VMEM /
VALU / SALU / waitcnt instructions as a compact string, the issue slots between consecutive MFMAs, and a crude in-order
estimate of the cycles one wave per SIMD needs (v_mfma_f32_16x16x4_f32 = 32 cycles of matrix pipe, any other instruction
one 4-cycle issue slot, ds_read_b128 8, so a gap with more than 7 fillers leaves the pipe idle).

    hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S x.hip -o x.s ; python scripts/isa_sched.py x.s <kernel substring>
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "L"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "W"
    if op.startswith("ds_"):
        return "X"  # bpermute / swizzle
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "G"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("global_atomic"):
        return "T"
    if op.startswith("s_waitcnt"):
        return "w"
    if op.startswith("s_nop"):
        return "n"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "j"
    if op.startswith("s_"):
        return "s"
    if op.startswith("v_accvgpr") :
        return "a"
    if op.startswith("v_"):
        return "v"
    return "?"


COST = {"M": 32, "L": 8, "W": 8, "X": 8, "G": 8, "T": 8, "w": 4, "n": 4, "B": 4, "j": 4, "s": 4, "a": 4, "v": 4, "?": 4}


def blocks(path, key):
    name, cur, out, on = None, None, [], False
    for ln in open(path):
        ln = ln.rstrip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            on = key in m.group(1)
            name = m.group(1)
            if on:
                cur = ["entry", []]
                out.append(cur)
            continue
        if not on:
            continue
        if ln.startswith(".Lfunc_end"):
            on = False
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur = [m.group(1), []]
            out.append(cur)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", ln)
        if m and not ln.strip().startswith("."):
            cur[1].append((classify(m.group(1)), ln.strip()))
    return out


def estimate(seq):
    """in-order issue: an MFMA waits until the matrix pipe is free; everything else takes its slot"""
    t = pipe_free = 0
    for c in seq:
        if c == "M":
            t = max(t, pipe_free)
            pipe_free = t + 32
            t += 4
        else:
            t += COST[c] if c != "n" else 4
    return max(t, pipe_free)


def main():
    path, key = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 4 and sys.argv[3] == "--gaps":
        return gaps_report(path, key, sys.argv[4])
    verbose = len(sys.argv) > 3
    for name, ins in blocks(path, key):
        seq = "".join(c for c, _ in ins)
        nm = seq.count("M")
        if nm == 0 and not verbose:
            continue
        gaps = [len(g) for g in re.split("M", seq)]
        est = estimate(seq)
        print(f"{name}: {len(seq)} instr, {nm} MFMA, est {est} cyc (MFMA floor {32 * nm}, eff {32 * nm / max(est, 1):.2f}); "
              f"gaps>7: {sum(1 for g in gaps if g > 7)} max {max(gaps)}")
        if verbose:
            # run-length compress
            print("   " + re.sub(r"(.)\1{3,}", lambda m: f"{m.group(1)}{len(m.group(0))}", seq))




def gaps_report(path, key, block):
    """per-gap filler cost for one block: python isa_sched.py x.s key --gaps .LBBn_m"""
    for name, ins in blocks(path, key):
        if name != block:
            continue
        seq = "".join(c for c, _ in ins)
        parts = re.split("M", seq)
        out = []
        for i, g in enumerate(parts):
            cost = sum(COST[c] for c in g)
            mark = "!" if cost > 28 else ""
            out.append(f"{g or '.'}{mark}")
        print(" ".join(out))


if __name__ == "__main__":
    main()
