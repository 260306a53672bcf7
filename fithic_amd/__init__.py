"""fithic_amd - MI355X-native Fit-Hi-C significance engine (drop-in for the hot path of ay-lab/fithic).

Only what the hot path needs lives here: the HIP kernels + C ABI (csrc/, libfithic_mi355x.so), its ctypes
binding (_capi), the host driver (engine), the three input tables + output writer (tables), the reference's
function signatures for the path (fithic), the CLI (cli) and the multi-GPU sharding (dist).
"""
from ._version import __version__  # noqa: F401
