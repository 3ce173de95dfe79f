"""Recipe for oracle/_ref/: the reference's OWN learner sources, placed where the GPU box can import them.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

    python -m oracle.make_ref          # needs /root/reference (the build container); __graft_entry__.build() runs it there

The reference is pure Python, so "building" it is a copy: the files of SURVEY.md 8(a)'s learner rows (marlbase/dqn/{model,train}.py,
marlbase/utils/{models,standardise_stream,utils,video}.py and the package markers) go, byte for byte, into oracle/_ref/marlbase/ - a
directory that is git-ignored (no reference source enters the history) but NOT gpurun-ignored, so it travels to the GPU box the way
a built .so does.  There `bench.py`'s cpu_baseline leg times the reference's unmodified QNetwork / ReplayBuffer / _epsilon_schedule
(`"kind": "reference"`); without oracle/_ref it times the port (`"kind": "port"`).  Nothing under codebase_amd/ may import it."""
import os
import shutil
import sys

REF = os.environ.get("MARLHIP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
FILES = ["marlbase/__init__.py", "marlbase/dqn/__init__.py", "marlbase/dqn/model.py", "marlbase/dqn/train.py",
         "marlbase/utils/__init__.py", "marlbase/utils/models.py", "marlbase/utils/standardise_stream.py", "marlbase/utils/utils.py",
         "marlbase/utils/video.py"]  # video.py: imported by dqn/train.py at module level (imageio is stubbed; nothing records)


def make(verbose=True):
    """copy FILES from the reference checkout into oracle/_ref/; returns False (and leaves oracle/_ref alone) when there is none"""
    if not os.path.isdir(os.path.join(REF, "marlbase")):
        return False
    for rel in FILES:
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
    with open(os.path.join(OUT, "README"), "w") as f:
        f.write("copied from the reference checkout by oracle/make_ref.py; git-ignored, never edit, never import from codebase_amd/\n")
    if verbose:
        print(f"[oracle] oracle/_ref: {len(FILES)} reference files from {REF}")
    return True


if __name__ == "__main__":
    sys.exit(0 if make() else 1)
