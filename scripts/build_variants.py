#!/usr/bin/env python
"""Build experiment variants of libmarlhip.so: the learner translation unit of the bench shape (dqn_update_h64.hip) recompiled
with extra -D flags, linked with the product objects.  Variants land in codebase_amd/csrc/variants/libmarlhip_<name>.so
(git-ignored; they travel to the GPU box) and are selected with MARLHIP_LIB=<path> (codebase_amd/_lib.py).

    python scripts/build_variants.py name1:-DMARL_BURST=0 name2:-DMARL_STEP_PROF=1,-DFOO=2
"""
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
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        odir = os.path.join(b.OBJ, "variant_" + name)
        os.makedirs(odir, exist_ok=True)
        objs = []
        for src in b.SOURCES:
            o = os.path.join(b.OBJ, src.replace(".hip", ".o"))
            if src in VARIANT_SOURCES:
                o = os.path.join(odir, src.replace(".hip", ".o"))
                subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", o])
            objs.append(o)
        lib = os.path.join(out_dir, f"libmarlhip_{name}.so")
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
        print(lib, flags)


if __name__ == "__main__":
    main()
