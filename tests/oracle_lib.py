"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package offt_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "liboracle.so")


class OrcComm(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("p1", "p2", "rank_x", "rank_y", "M1", "M2", "M3", "M4", "F1", "F2", "F3", "F4",
                                       "m1", "m2", "m3", "m4", "b1", "b2", "b3", "b4")] + \
               [(k, C.c_int * 3) for k in ("istart", "isize", "istride", "ostart", "osize", "ostride")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", ROOT, "oracle/liboracle.so"])
    L = C.CDLL(SO)
    L.orc_fft_plan_create.restype = C.c_void_p
    L.orc_fft_plan_create.argtypes = [C.c_int]
    L.orc_fft_plan_destroy.argtypes = [C.c_void_p]
    L.orc_fft_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]
    L.orc_comm_build.argtypes = [C.POINTER(OrcComm)] + [C.c_int] * 9
    L.orc_params_default.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    L.orc_world_create.restype = C.c_void_p
    L.orc_world_create.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)]
    L.orc_world_destroy.argtypes = [C.c_void_p]
    L.orc_world_comm.restype = C.POINTER(OrcComm)
    L.orc_world_comm.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_params.restype = C.POINTER(C.c_int)
    L.orc_world_params.argtypes = [C.c_void_p]
    L.orc_world_local_elems.restype = C.c_long
    L.orc_world_local_elems.argtypes = [C.c_void_p]
    L.orc_world_buffer.restype = C.POINTER(C.c_double)
    L.orc_world_buffer.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_fill.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_execute.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_gather.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_hash_val.restype = C.c_double
    L.orc_hash_val.argtypes = [C.c_int] * 4
    _lib = L
    return L


PARAM_NAMES = ["P1", "T1", "W1", "Px1", "Py1", "Fz", "FP1", "Ux1", "Uz1", "FU1", "Fy1", "Ry",
               "T2", "W2", "Pz2", "Px2", "Fy2", "FP2", "Uz2", "Uy2", "FU2", "Fx", "V", "S"]


def custom(**kw):
    v = (C.c_int * 24)(*([-1] * 24))
    for k, val in kw.items():
        v[PARAM_NAMES.index(k)] = int(val)
    return v


def comm_to_dict(c):
    d = {k: getattr(c, k) for k, _ in OrcComm._fields_[:20]}
    for k in ("istart", "isize", "istride", "ostart", "osize", "ostride"):
        d[k] = list(getattr(c, k))
    return d


def params_default(Nx, Ny, Nz, p, is_r2c=0, is_W0=0, is_notest=0):
    v = (C.c_int * 24)()
    lib().orc_params_default(Nx, Ny, Nz, p, is_r2c, is_W0, is_notest, v)
    return list(v)


def comm(Nx, Ny, Nz, p, rank, p1, is_r2c=0, is_equalxy=0, S=0):
    c = OrcComm()
    lib().orc_comm_build(C.byref(c), Nx, Ny, Nz, p, rank, p1, is_r2c, is_equalxy, S)
    return comm_to_dict(c)


def hash_field(Nx, Ny, Nz, x0=0, y0=0, z0=0):
    """seeded position hash of SURVEY.md Appendix D, vectorised."""
    x = (np.arange(Nx, dtype=np.uint64) + np.uint64(x0))[:, None, None]
    y = (np.arange(Ny, dtype=np.uint64) + np.uint64(y0))[None, :, None]
    z = (np.arange(Nz, dtype=np.uint64) + np.uint64(z0))[None, None, :]
    M = np.uint64(0xffffffff)

    def val(c):
        h = ((x * np.uint64(73856093)) & M) ^ ((y * np.uint64(19349663)) & M) ^ ((z * np.uint64(83492791)) & M) \
            ^ np.uint64((c * 2654435761) & 0xffffffff)
        h ^= h >> np.uint64(13)
        h = (h * np.uint64(0x5bd1e995)) & M
        h ^= h >> np.uint64(15)
        return (h & np.uint64(0xffffff)).astype(np.float64) / 8388608.0 - 1.0
    return val(0) + 1j * val(1)


def ramp_field(Nx, Ny, Nz):
    x = np.arange(Nx)[:, None, None]
    y = np.arange(Ny)[None, :, None]
    z = np.arange(Nz)[None, None, :]
    return (z + 10 * y + 100 * x).astype(np.complex128)


def world_fft(Nx, Ny, Nz, p, kind=1, is_r2c=0, is_oned=0, is_equalxy=0, nthreads=1, **params):
    """Run the restated reference pipeline on p simulated ranks; returns (global result, per-rank comm dicts, v)."""
    L = lib()
    cv = custom(**params) if params else None
    w = L.orc_world_create(Nx, Ny, Nz, p, is_r2c, is_oned, is_equalxy, cv)
    try:
        L.orc_world_fill(w, kind)
        L.orc_world_execute(w, nthreads)
        Nzn = Nz // 2 + 1 if is_r2c else Nz
        g = np.zeros((Nx, Ny, Nzn), dtype=np.complex128)
        L.orc_world_gather(w, g.ctypes.data_as(C.c_void_p))
        comms = [comm_to_dict(L.orc_world_comm(w, r).contents) for r in range(p)]
        v = [L.orc_world_params(w)[i] for i in range(24)]
        return g, comms, v
    finally:
        L.orc_world_destroy(w)


def fft1d(a):
    """1-D forward DFT of a complex128 vector through the oracle's FFT."""
    L = lib()
    a = np.ascontiguousarray(a, dtype=np.complex128).copy()
    n = a.shape[0]
    pl = L.orc_fft_plan_create(n)
    scr = np.zeros(6 * n + 8)
    L.orc_fft_execute(pl, a.ctypes.data_as(C.c_void_p), 1, 0, 1, scr.ctypes.data_as(C.c_void_p))
    L.orc_fft_plan_destroy(pl)
    return a
