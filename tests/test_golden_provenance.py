"""Every fixture under tests/golden/ names the committed script that regenerates it from the reference's own classes
(three recurrent fixtures had only a comment).  CPU only; the regeneration itself needs /root/reference and is
run by hand (`python -m oracle.make_golden*`), bit-identical for all 38 files as of round 4."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _patterns():
    pats = []
    for path in sorted(glob.glob(os.path.join(ROOT, "oracle", "make_golden*.py"))):
        src = open(path).read()
        for lit in re.findall(r'f?"([^"\n]*\.(?:npz|pt))"', src):
            pats.append((re.compile("^" + re.sub(r"\\\{[^}]*\\\}", ".+", re.escape(lit)) + "$"), os.path.basename(path)))
    return pats


def test_every_golden_has_a_generator_call():
    pats = _patterns()
    orphans = []
    for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden"))):
        if not any(p.match(f) for p, _ in pats):
            orphans.append(f)
    assert not orphans, f"fixtures without a generator under oracle/make_golden*.py: {orphans}"
