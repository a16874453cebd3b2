"""maria_amd: MI355X (gfx950) implementation of maria's atmosphere -> TOD hot path.

The compute lives in ``libmrx.so`` (hand-written HIP, C ABI in ``include/mrx.h``);
this package is the thin Python host side that mirrors the reference's
``Atmosphere`` / ``Simulation`` seams.  There is no CPU fallback: every entry
point that computes raises if the library or a gfx950 device is missing.
"""

from ._lib import LIB_PATH, Context, MrxError, load  # noqa: F401

__version__ = "0.1.0"
