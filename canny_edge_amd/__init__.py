"""canny_edge_amd -- MI355X-native Canny edge detection behind the reference's stage API.

The product is the C-ABI shared library built from ``canny_edge_amd/csrc`` (hand-written HIP
kernels for gfx950); this package is the thin Python host binding over it (ctypes, no torch types
in any signature).  There is no CPU fallback: every entry point raises if the HIP library is
missing or no GPU is present.
"""
from .synth import synth_batch, synth_frame  # noqa: F401

__all__ = ["synth_frame", "synth_batch", "capi"]
__version__ = "0.1.0"


def __getattr__(name):
    if name == "capi":
        import importlib
        return importlib.import_module(".capi", __name__)
    raise AttributeError(name)
