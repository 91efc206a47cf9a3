"""Python mirror of the offt.h plan/execute interface, over the C ABI.

Same names and argument meaning as the reference's C API (rchyena/offt
offt.h:235-244): ``offt_3d_init`` / ``offt_3d_execute`` / ``offt_3d_fin`` /
``print_params`` / ``offt_print_time``, plus the extensions of
``include/offt_hip.h``.  Everything here is plumbing (ctypes structs, pointer
passing); all arithmetic happens in ``liboffthip.so``.
"""
import ctypes as C

from . import _lib

PARAM_COUNT = 24
GES = 16
(P1, T1, W1, Px1, Py1, Fz, FP1, Ux1, Uz1, FU1, Fy1, Ry,
 T2, W2, Pz2, Px2, Fy2, FP2, Uz2, Uy2, FU2, Fx, V, S) = range(PARAM_COUNT)
PARAM_NAMES = ["P1", "T1", "W1", "Px1", "Py1", "Fz", "FP1", "Ux1", "Uz1", "FU1", "Fy1", "Ry",
               "T2", "W2", "Pz2", "Px2", "Fy2", "FP2", "Uz2", "Uy2", "FU2", "Fx", "V", "S"]
# timer slots, offt.h:171-188
(ALL, INIT1, WAIT1, TEST1, INIT2, WAIT2, TEST2, FFTz, FFTy1, FFTy2, FFTx,
 TRANSPOSE, PACK1, UNPACK1, PACK2, UNPACK2) = range(GES)
FFTW_ESTIMATE = 1 << 6
F64, F32 = 0, 1


class OfftParams(C.Structure):
    _fields_ = [("is_converged", C.c_int), ("is_infeasible", C.c_int), ("is_in_database", C.c_int),
                ("v", C.c_int * PARAM_COUNT)]


class OfftComm(C.Structure):
    _fields_ = [("p1", C.c_int), ("p2", C.c_int),
                ("comm1", C.c_void_p), ("comm2", C.c_void_p), ("group1", C.c_void_p), ("group2", C.c_void_p),
                ("M1", C.c_int), ("M2", C.c_int), ("M3", C.c_int), ("M4", C.c_int),
                ("F1", C.c_int), ("F2", C.c_int), ("F3", C.c_int), ("F4", C.c_int),
                ("m1", C.c_int), ("m2", C.c_int), ("m3", C.c_int), ("m4", C.c_int),
                ("b1", C.c_int), ("b2", C.c_int), ("b3", C.c_int), ("b4", C.c_int),
                ("istart", C.c_int * 3), ("isize", C.c_int * 3), ("istride", C.c_int * 3),
                ("ostart", C.c_int * 3), ("osize", C.c_int * 3), ("ostride", C.c_int * 3)]


class OfftPlan(C.Structure):
    _fields_ = [("p", C.c_int), ("rank", C.c_int), ("Nx", C.c_int), ("Ny", C.c_int), ("Nz", C.c_int),
                ("is_r2c", C.c_int), ("fftw_flag", C.c_int), ("ah_strategy", C.c_int), ("max_loop", C.c_int),
                ("tuning_mode", C.c_int), ("is_W0", C.c_int), ("extrapolation_window", C.c_int),
                ("is_oned", C.c_int), ("is_a2a", C.c_int), ("is_equalxy", C.c_int), ("is_notest", C.c_int),
                ("t_init", C.c_double * 4), ("t", C.c_double * GES),
                ("point_database_file", C.c_char * 256), ("user_vertex_file", C.c_char * 256),
                ("params", C.POINTER(OfftParams)), ("comm", C.POINTER(OfftComm)),
                ("buffer_chunk", C.c_void_p), ("buffers1", C.c_void_p), ("buffers2", C.c_void_p),
                ("pt_transpose", C.c_void_p), ("pt_transpose_list", C.c_void_p),
                ("pt_transpose_list_size", C.c_int),
                ("p1d_x", C.c_void_p), ("p1d_y", C.c_void_p), ("p1d_z", C.c_void_p),
                ("p1d_x_t", C.c_void_p), ("p1d_y_t", C.c_void_p),
                ("p1d_x_s_list", C.c_void_p), ("p1d_y_s_list", C.c_void_p), ("p1d_xy_s_list_size", C.c_int),
                ("hip_state", C.c_void_p)]


_bound = None


def lib():
    """The loaded C-ABI library with argtypes/restypes set."""
    global _bound
    if _bound is None:
        _bound = bind(_lib.load())
    return _bound


def use_library(path=None):
    """Route this module through another build of the library (`path`), or back to the product (None).
    The test-suite uses it for the build that carries the test-only seams (tests/liboffthip_test.so)."""
    global _bound
    _bound = bind(_lib.load(path))
    return _bound


def bind(L):
    """set argtypes/restypes on a loaded build of the library"""
    if getattr(L, "_offt_bound", False):
        return L
    PP = C.POINTER(OfftPlan)
    i = C.c_int
    L.offt_3d_init.restype = PP
    L.offt_3d_init.argtypes = [i, i, i, C.c_void_p, C.c_void_p] + [i] * 11 + [C.POINTER(OfftParams)]
    L.offt_3d_init_ex.restype = PP
    L.offt_3d_init_ex.argtypes = [i, i, i, C.c_void_p, C.c_void_p] + [i] * 11 + [C.POINTER(OfftParams), i]
    L.offt_3d_execute.restype = None
    L.offt_3d_execute.argtypes = [PP, C.c_void_p, C.c_void_p, i]
    L.offt_3d_execute_dir.restype = None
    L.offt_3d_execute_dir.argtypes = [PP, C.c_void_p, C.c_void_p, i]
    L.offt_3d_fin.restype = None
    L.offt_3d_fin.argtypes = [PP]
    L.print_params.restype = None
    L.print_params.argtypes = [C.POINTER(C.c_int)]
    L.offt_print_time.restype = None
    L.offt_print_time.argtypes = [C.POINTER(C.c_double)]
    L.offt_hip_get_unique_id.argtypes = [C.c_void_p]
    L.offt_hip_set_world.argtypes = [i, i, C.c_void_p, i]
    L.offt_hip_set_stream.restype = None
    L.offt_hip_set_stream.argtypes = [PP, C.c_void_p]
    L.offt_hip_set_async.restype = None
    L.offt_hip_set_async.argtypes = [PP, i]
    L.offt_hip_set_variant.restype = None
    L.offt_hip_set_variant.argtypes = [PP, i, i]
    L.offt_hip_set_output_scale.restype = None
    L.offt_hip_set_output_scale.argtypes = [PP, C.c_double]
    L.offt_hip_local_bytes.restype = C.c_longlong
    L.offt_hip_local_bytes.argtypes = [PP]
    L.offt_hip_last_device_seconds.restype = C.c_double
    L.offt_hip_last_device_seconds.argtypes = [PP]
    L.offt_hip_last_pass_seconds.restype = None
    L.offt_hip_last_pass_seconds.argtypes = [PP, C.POINTER(C.c_double)]
    L.offt_hip_last_passes_paired.restype = i
    L.offt_hip_last_passes_paired.argtypes = [PP]
    L.offt_hip_last_error.restype = C.c_char_p
    L.offt_hip_malloc.restype = C.c_void_p
    L.offt_hip_malloc.argtypes = [C.c_longlong]
    L.offt_hip_free.restype = None
    L.offt_hip_free.argtypes = [C.c_void_p]
    L.offt_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    L.offt_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    L.offt_hip_fill_input.argtypes = [PP, C.c_void_p, i]
    L.offt_hipk_variant_count.argtypes = [i, i]
    L.offt_hipk_variant_name.restype = C.c_char_p
    L.offt_hipk_variant_name.argtypes = [i, i, i]
    L.offt_hipk_has_fast_path.argtypes = [i, i]
    L.offt_hip_world_count.restype = i
    L.offt_hip_link_probe.restype = C.c_double
    L.offt_hip_link_probe.argtypes = [i, i, C.c_longlong, i]
    L.offt_hip_wait.restype = i
    L.offt_hip_wait.argtypes = [PP]
    if hasattr(L, "offt_hip_set_debug_skip"):  # diagnostics / test builds only, never the product library
        L.offt_hip_set_debug_skip.restype = None
        L.offt_hip_set_debug_skip.argtypes = [PP, i]
    L.offt_hip_set_exchange.argtypes = [PP, i]
    L.offt_hip_get_exchange.argtypes = [PP]
    L.offt_hip_set_option.argtypes = [PP, i, C.c_longlong]
    L.offt_hip_get_option.restype = C.c_longlong
    L.offt_hip_get_option.argtypes = [PP, i]
    L._offt_bound = True
    return L


def make_params(**kw):
    """custom_params as run-fft.c builds them: -1 = keep default (run-fft.c:163-167)."""
    cp = OfftParams()
    for k in range(PARAM_COUNT):
        cp.v[k] = -1
    for name, val in kw.items():
        cp.v[PARAM_NAMES.index(name)] = int(val)
    return cp


def offt_3d_init(Nx, Ny, Nz, inp=None, out=None, is_r2c=0, fftw_flag=FFTW_ESTIMATE, is_oned=0, is_a2a=0,
                 is_equalxy=0, is_notest=0, ah_strategy=0, max_loop=0, tuning_mode=0, is_W0=0,
                 extrapolation_window=0, custom_params=None, precision=F64):
    L = lib()
    cp = C.byref(custom_params) if custom_params is not None else None
    po = L.offt_3d_init_ex(Nx, Ny, Nz, inp, out, is_r2c, fftw_flag, is_oned, is_a2a, is_equalxy, is_notest,
                           ah_strategy, max_loop, tuning_mode, is_W0, extrapolation_window, cp, precision)
    if not po:
        raise RuntimeError("offt_3d_init failed: " + L.offt_hip_last_error().decode())
    return po


def offt_3d_execute(po, inp, out, is_tuning=0):
    L = lib()
    L.offt_3d_execute(po, inp, out, is_tuning)
    if po.contents.t[ALL] >= 99999999.0:  # the reference's failure marker (offt-compute.c:3881)
        raise RuntimeError("offt_3d_execute failed: " + L.offt_hip_last_error().decode())


def offt_3d_execute_dir(po, inp, out, direction):
    L = lib()
    L.offt_3d_execute_dir(po, inp, out, direction)
    if po.contents.t[ALL] >= 99999999.0:
        raise RuntimeError("offt_3d_execute_dir failed: " + L.offt_hip_last_error().decode())


def offt_3d_fin(po):
    lib().offt_3d_fin(po)


def comm_dict(po):
    c = po.contents.comm.contents
    d = {k: getattr(c, k) for k in ("p1", "p2", "M1", "M2", "M3", "M4", "F1", "F2", "F3", "F4",
                                     "m1", "m2", "m3", "m4", "b1", "b2", "b3", "b4")}
    for k in ("istart", "isize", "istride", "ostart", "osize", "ostride"):
        d[k] = list(getattr(c, k))
    return d


def local_elems(po):
    c = po.contents.comm.contents
    return (c.M1 * c.M2 * c.M3 * c.p2) if (c.M2 * c.p2 > c.M4 * c.p1) else (c.M1 * c.M3 * c.M4 * c.p1)
