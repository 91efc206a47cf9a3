"""The C-ABI shared library loads without a GPU and exports every symbol include/*.h declares.
No compute calls here."""
import ctypes as C
import os
import re

from offt_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"static __inline__ int (max|min)\(.*?\n}", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*(?:\([^()]*\)[^;{}()]*)*\)\s*;", txt)
    return sorted(set(n for n in names if n.startswith(("offt_", "print_params"))))


def test_library_loads_and_exports_everything(built):
    L = _lib.load()
    want = declared_functions("offt.h") + declared_functions("offt_hip.h")
    assert {"offt_3d_init", "offt_3d_execute", "offt_3d_fin", "print_params", "offt_print_time"} <= set(want)
    assert len(want) >= 20
    missing = [n for n in want if not hasattr(L, n)]
    assert not missing, missing


def test_no_dt_needed_on_hip_and_no_oracle_symbols(built):
    """the product library must not pull in the oracle, and leaves the HIP runtime choice to the host process"""
    import subprocess
    out = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "liboracle" not in out
    syms = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    assert "orc_" not in syms


def test_struct_mirror_sizes(built):
    # the ctypes mirror must agree with the C layout: compile-time facts only
    assert C.sizeof(api.OfftParams) == 4 * 27
    assert api.OfftComm.M1.offset == 8 + 4 * 8
    assert api.OfftPlan.t.offset % 8 == 0


def test_print_formats(built, capfd):
    """stdout formats the reference harness relies on (offt-compute.c:3239-3294)"""
    L = api.lib()
    C.CDLL(None).fflush(None)  # drop whatever earlier tests left in the C stdio buffer
    capfd.readouterr()
    v = (C.c_int * 24)(*range(24))
    v[3] = -1
    L.print_params(v)
    t = (C.c_double * 16)(*[i / 8 for i in range(16)])
    L.offt_print_time(t)
    C.CDLL(None).fflush(None)
    out = capfd.readouterr().out.splitlines()
    assert out[0] == "P1 0 T1 1 W1 2 Py1 4 Fz 5 FP1 6 Ux1 7 Uz1 8 FU1 9 Fy1 10 Ry 11 T2 12 W2 13 Pz2 14 Px2 15 Fy2 16 FP2 17 Uz2 18 Uy2 19 FU2 20 Fx 21 V 22 S 23 "
    assert out[1] == "0.00000  0.12500 0.25000 0.37500 0.50000 0.62500 0.75000  1.37500  1.50000 1.62500 1.75000 1.87500  0.87500 1.00000 1.12500 1.25000"


def test_fails_loudly_without_gpu(built):
    """no CPU fallback: without a HIP device init must refuse (skipped where a GPU exists)"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    try:
        api.offt_3d_init(8, 8, 8)
    except RuntimeError as e:
        assert "no HIP device" in str(e) or "hip" in str(e).lower()
    else:
        raise AssertionError("offt_3d_init succeeded without a GPU")


def test_product_library_has_no_test_seams(built):
    """the test-only entry points of offt_backend.h exist in tests/liboffthip_test.so only"""
    import subprocess
    syms = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    assert "offt_hip_test_" not in syms
    # ... nor the diagnostic switch that makes an execute skip its passes or its exchanges (tools/liboffthip_diag.so has it)
    assert "offt_hip_set_debug_skip" not in syms
    dsyms = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "tools", "liboffthip_diag.so")]).decode()
    assert "offt_hip_set_debug_skip" in dsyms and "offt_hip_test_" not in dsyms
    tsyms = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "tests", "liboffthip_test.so")]).decode()
    assert all(n in tsyms for n in ("offt_hip_test_set_backend", "offt_hip_test_set_transport", "offt_hip_test_set_transport_async", "offt_hip_test_set_p2p"))
    # the negative-control switches of the ordering tests are compiled into the test build only
    import mmap
    with open(_lib.LIB_PATH, "rb") as f:
        blob = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        assert blob.find(b"OFFT_TEST_DROP_EDGE") < 0 and blob.find(b"OFFT_TEST_SLOW_PASS_MS") < 0
    with open(os.path.join(ROOT, "tests", "liboffthip_test.so"), "rb") as f:
        blob = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        assert blob.find(b"OFFT_TEST_DROP_EDGE") >= 0


def test_mpi_harness_compiles_and_links(built, tmp_path):
    """`make harness MPI=1` (rank/size from MPI, RCCL id by MPI_Bcast -- the reference's launch model,
    run-fft.c:158-160): compile and link against the image's MPICH; it cannot run without GPUs"""
    import shutil
    import subprocess
    import pytest
    if not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("no MPI in this image")
    exe = tmp_path / "run-fft-mpi"
    cmd = ["gcc", "-std=gnu11", "-O2", "-Wall", "-DOFFT_HARNESS_MPI", "-I" + os.path.join(ROOT, "include"), "-I/opt/conda/include",
           "-o", str(exe), os.path.join(ROOT, "harness", "run-fft.c"), "-L" + os.path.join(ROOT, "offt_amd"), "-loffthip",
           "-L/opt/rocm/lib", "-lamdhip64", "/opt/conda/lib/libmpi.so", "-lm", "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + os.path.join(ROOT, "offt_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/conda/lib"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)]).decode()
    assert "MPI_Bcast" in syms and "offt_hip_set_world" in syms and "offt_3d_execute" in syms
    # without -d the harness must leave the mesh to the library (run-fft.c:290-293 only sizes the buffer with p1 = p)
    src = open(os.path.join(ROOT, "harness", "run-fft.c")).read()
    assert "p1 = cp->v[_P1_] = p;" not in src


def test_kernel_routing_without_a_gpu(built):
    """which kernel family a pass descriptor resolves to is host logic of the kernel library and needs no device:
    single-precision column pairs only for descriptors they can take (even column count, 16-B pairs on a strided side),
    cache-keeping twins only where one is registered (so that the host alternates y and x launches only there)"""
    import ctypes as C
    from test_gpu_descriptors import Desc
    L = api.lib()
    L.offt_hipk_kernel_name.restype = C.c_char_p
    L.offt_hipk_kernel_name.argtypes = [C.POINTER(Desc)]
    L.offt_hipk_keeps_output.argtypes = [C.POINTER(Desc)]

    def desc(n, prec, ncols=64, in_contig=1, out_contig=0):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = n, prec, -1, ncols, 4, 1
        if in_contig:
            d.in_axis_stride, d.in_col_stride = 1, n
        else:
            d.in_axis_stride, d.in_col_stride = ncols, 1
        if out_contig:
            d.out_axis_stride, d.out_col_stride = 1, n
        else:
            d.out_axis_stride, d.out_col_stride = ncols, 1
        d.in_b1_stride = d.out_b1_stride = n * ncols
        d.in_contig, d.out_contig, d.variant, d.scale = in_contig, out_contig, -1, 1.0
        return d

    name = lambda d: L.offt_hipk_kernel_name(C.byref(d)).decode()
    for n in (512, 1024, 2048, 4096):
        assert name(desc(n, api.F32)) == "fft_panel_k<pairs>"
        assert name(desc(n, api.F32, ncols=63)) == "fft_panel_k"        # odd column count
        assert name(desc(n, api.F64)) == "fft_panel_k"
    d = desc(1024, api.F32, in_contig=0)
    assert name(d) == "fft_panel_k<pairs>"
    d.in_b1_stride += 1                                                     # a pair would straddle the 16-B grid
    assert name(d) == "fft_panel_k"
    assert name(desc(256, api.F32)) == "fft_panel_k"                        # no pair kernel registered at 256
    assert name(desc(768, api.F64)) == "fft_panelx_k"
    # cache-keeping twins: the power-of-two contig-in defaults (y pass of the forward, x pass of the inverse), nothing else
    for n, prec, want in ((1024, api.F64, 1), (512, api.F64, 1), (1024, api.F32, 1), (2048, api.F32, 1), (768, api.F64, 0), (1016, api.F64, 0)):
        assert L.offt_hipk_keeps_output(C.byref(desc(n, prec))) == want, (n, prec)
    assert L.offt_hipk_keeps_output(C.byref(desc(1024, api.F64, out_contig=1))) == 1
    assert L.offt_hipk_keeps_output(C.byref(desc(1024, api.F64, in_contig=0, out_contig=1))) == 0
    assert L.offt_hipk_keeps_output(C.byref(desc(1024, api.F64, in_contig=0, out_contig=0))) == 0
