"""Loading of libfithic_mi355x.so, free of numpy so that it can run while the interpreter is still importing.

The command line starts `warm_in_background` first thing: dlopen of the library (0.06-0.09 s) and the first touch of the
HIP runtime (0.15-0.25 s on the MI355X box) then overlap the imports of numpy and of the package (0.2 s) instead of standing
in front of the contacts file."""
import ctypes
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FHX_LIB") or os.path.join(_PKG, "libfithic_mi355x.so")   # FHX_LIB: A/B experiments only

_lock = threading.Lock()
_cdll = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7 (+ HSA, comgr) next to libtorch and can only
    work with that copy; if the system copy (/opt/rocm, what this library is linked against, same soname) is loaded first, a
    later `import torch` finds no device.  So when torch is installed, its copy is loaded first - without importing torch -
    and this library binds to it; without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None


def load():
    """the CDLL (argtypes are set by _capi.lib()); raises (loudly) when the library has not been built - there is no fallback"""
    global _cdll
    with _lock:
        if _cdll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError("libfithic_mi355x.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                                   "g.build()'` or fithic_amd._capi.build(); fithic_amd has no CPU fallback." % LIB_PATH)
            _share_torch_hip_runtime()
            _cdll = ctypes.CDLL(LIB_PATH)
        return _cdll


def warm_in_background(device=0):
    """library + HIP runtime + the device's primary context, on a thread; errors are left for the real calls to report"""
    def work():
        try:
            fn = load().fhx_warmup
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_int]
            fn(int(device))
        except Exception:                      # noqa: BLE001 - a missing library or GPU is reported by the engine, with context
            pass
    t = threading.Thread(target=work, daemon=True)
    t.start()
    return t
