"""ctypes loader for liboffthip.so (the C-ABI product library).

The library carries no DT_NEEDED on the HIP runtime (see Makefile), so the
loader first makes ONE libamdhip64 globally visible: PyTorch's bundled copy
when torch is importable (PyTorch is the plumbing for device memory, streams and
torch.distributed), otherwise /opt/rocm's.  Loading fails loudly if the shared
object is missing -- there is no Python or CPU fallback.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFFT_AMD_LIB") or os.path.join(_HERE, "liboffthip.so")  # override: developer A/B builds
_lib = None
_by_path = {}


def _preload_hip():
    cands = []
    try:
        import torch  # noqa: F401  (loads its bundled ROCm libraries)
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        cands.append(os.path.join(tl, "libamdhip64.so"))
    except Exception:  # torch absent: C-only environment
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    last = None
    for c in cands:
        try:
            return ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # try the next candidate
            last = e
    raise OSError(f"offt_amd: no HIP runtime (libamdhip64) could be loaded: {last}")


def load(path=None):
    """Load the product library (default) or another build of it given by `path` (the test build with the
    test-only seams, a developer A/B build).  Every build is linked -Bsymbolic and loaded RTLD_LOCAL, so two of
    them can live in one process without binding to each other's symbols."""
    global _lib
    path = path or LIB_PATH
    if path in _by_path:
        return _by_path[path]
    if not os.path.exists(path):
        raise OSError(
            f"offt_amd: {path} is missing -- run `make` (or __graft_entry__.build()); "
            "there is no fallback implementation")
    _preload_hip()
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    _by_path[path] = L
    if path == LIB_PATH:
        _lib = L
    return L
