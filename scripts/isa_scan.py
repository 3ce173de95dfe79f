"""Static scan of the device ISA for two things that cost without showing up in a source read (cdna_hip_programming.md, the
".s-level traps"): global loads SERIALISED by a full `s_waitcnt vmcnt(0)` between one load and the next inside a loop (a dependent
memory round trip per load - what a per-element `if (uniform condition) load` compiles to), and scratch (spill) traffic.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S csrc/X.hip -o /tmp/isa/X.hip.s     (one per translation unit)
    python scripts/isa_scan.py /tmp/isa/*.hip.s

Per kernel: loads, the longest chain of loads each separated from the next by vmcnt(0) with no barrier or MFMA in between, scratch
instructions.  Sorted by chain length; chains of <= 2 are normal (a load, its use, the next load)."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def scan(path):
    rows, name, st = [], None, None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            st = dict(loads=0, chain=0, best=0, scratch=0, waited=False, mfma=0)
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            rows.append((name, st))
            name = None
            continue
        if t.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in t.split()[0]:
            st["loads"] += 1
            st["chain"] = st["chain"] + 1 if st["waited"] else 1
            st["best"] = max(st["best"], st["chain"])
            st["waited"] = False
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            st["waited"] = True
        elif t.startswith(("s_barrier", "v_mfma", "s_endpgm")):
            st["mfma"] += t.startswith("v_mfma")
            st["chain"], st["waited"] = 0, False
        elif t.startswith("scratch_"):
            st["scratch"] += 1
    return rows


def main(paths):
    rows = [(p.split("/")[-1], n, s) for p in paths for n, s in scan(p)]
    names = demangle([n for _, n, _ in rows])
    rows.sort(key=lambda r: (-r[2]["best"], -r[2]["scratch"]))
    print(f"{'chain':>5} {'loads':>5} {'scratch':>7} {'mfma':>5}  kernel")
    for f, n, s in rows:
        if s["best"] > 2 or s["scratch"]:
            print(f"{s['best']:5d} {s['loads']:5d} {s['scratch']:7d} {s['mfma']:5d}  {names[n][:150]}  [{f}]")


if __name__ == "__main__":
    main(sys.argv[1:])
