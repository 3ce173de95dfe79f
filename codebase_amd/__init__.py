"""codebase_amd - MI355X-native (gfx950 HIP) implementation of marlbase's independent-learner
hot path, behind marlbase's own Python surface.  See DESIGN.md / INTEGRATION.md.

The compute lives in csrc/libmarlhip.so (C-ABI: include/marlhip.h).  Sub-modules import it on
first use and fail loudly when it has not been built (python -m codebase_amd.build)."""

__version__ = "0.1.0"


def __getattr__(name):
    if name == "hip":
        import importlib

        return importlib.import_module(".hip", __name__)
    raise AttributeError(name)
