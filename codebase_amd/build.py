"""Build libmarlhip.so (gfx950 HIP kernels + C-ABI) in-tree with hipcc.

    python -m codebase_amd.build            # incremental
    python -m codebase_amd.build --force

One object per .hip file (compiled in parallel), linked into
codebase_amd/csrc/libmarlhip.so.  No torch, no CUDA shims: plain hipcc for gfx950.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(CSRC, "libmarlhip.so")
SOURCES = ["api.hip", "lbf_kernels.hip", "replay_kernels.hip", "collect.hip", "collect_oid.hip", "dqn_update.hip", "dqn_update_h64.hip", "dqn_update_h64_oid.hip", "dqn_update_h128.hip", "dqn_update_h128_oid.hip", "dqn_update_rware.hip", "dqn_update_h16.hip", "round.hip", "p2p.hip", "rware_kernels.hip", "rware_collect.hip", "rware_collect_big.hip", "rware_collect_p8.hip", "ac_collect.hip", "ac_collect_oid.hip", "a2c.hip", "gru.hip", "gru_ac.hip", "mixed_ac.hip", "wide.hip", "qmix_gen.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "marlhip.h"))
    return hdr


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _dep_list(o, fallback):
    """headers a translation unit really includes, from the depfile its last compile left (hipcc -MMD); all headers when there is none"""
    d = o[:-2] + ".d"
    if not os.path.exists(d) or not os.path.exists(o):
        return fallback
    txt = open(d).read().replace("\\\n", " ")
    deps = []
    for part in txt.split(":", 1)[-1].split():
        if part.endswith((".h", ".hip")) and os.path.exists(part):
            deps.append(part)
    return deps or fallback


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _deps()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + _dep_list(o, hdr)):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        if verbose:
            print(f"[marlhip] hipcc {os.path.basename(s)}", flush=True)
        subprocess.check_call([hipcc] + FLAGS + ["-MMD", "-MF", o[:-2] + ".d", "-c", s, "-o", o])

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        if verbose:
            print("[marlhip] link libmarlhip.so", flush=True)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
