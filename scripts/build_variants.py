#!/usr/bin/env python
"""Build experiment variants of libmarlhip.so: the learner translation unit of the bench shape (dqn_update_h64.hip) recompiled
with extra -D flags, linked with the product objects.  Variants land in codebase_amd/csrc/variants/libmarlhip_<name>.so
(git-ignored; they travel to the GPU box) and are selected with MARLHIP_LIB=<path> (codebase_amd/_lib.py).

    python scripts/build_variants.py name1:-DMARL_BURST=0 name2:-DMARL_STEP_PROF=1,-DFOO=2
    python scripts/build_variants.py flat:-DMARLHIP_WIDE_FLATLOAD=1,-DMARLHIP_GRU_WGRAD_FLAT=1:a2c.hip+wide.hip+gru.hip+gru_ac.hip

A third field names the translation units to recompile (default: the learner of the bench shape).
"""
from concurrent.futures import ThreadPoolExecutor
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codebase_amd import build as b

VARIANT_SOURCES = ["dqn_update_h64.hip"]


def main():
    b.build(verbose=False)
    out_dir = os.path.join(b.CSRC, "variants")
    os.makedirs(out_dir, exist_ok=True)
    for spec in sys.argv[1:]:
        name, _, rest = spec.partition(":")
        flags, _, srcs = rest.partition(":")
        flags = [f for f in flags.split(",") if f]
        variant_sources = srcs.split("+") if srcs else VARIANT_SOURCES
        odir = os.path.join(b.OBJ, "variant_" + name)
        os.makedirs(odir, exist_ok=True)
        objs, jobs = [], []
        for src in b.SOURCES:
            o = os.path.join(b.OBJ, src.replace(".hip", ".o"))
            if src in variant_sources:
                o = os.path.join(odir, src.replace(".hip", ".o"))
                jobs.append([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", o])
            objs.append(o)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(subprocess.check_call, jobs))
        lib = os.path.join(out_dir, f"libmarlhip_{name}.so")
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
        print(lib, flags)


if __name__ == "__main__":
    main()
