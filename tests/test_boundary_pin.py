"""The drop-in boundary pinned against the reference's OWN files (container only: skipped where /root/reference or an MPI
header is absent -- nothing of the reference travels to the GPU box).

1. The reference's caller, run-fft.c (run-fft.c:49-57 fills through istart/isize/istride, 314-321 calls offt_3d_init with
   17 arguments, 361-364 / 477-478 read ostart/osize/ostride and po->t), is checked with `gcc -fsyntax-only` from its
   mounted location against include/offt.h: every name, field, macro and call signature it uses must exist with a
   compatible type.  FFTW is absent from the image; the FFTW *names* come from a declarations-only header under
   tests/compile_shim/ (compile check only -- nothing is built, linked or run).
2. offsetof / sizeof of every public field of _offt_params, _offt_comm and _offt_plan are compared between include/offt.h
   and the reference's offt.h preprocessed with its Hopper flags (-DA2AV -DSTRIDE -DSHSONG_HOPPER, Makefile:27-29)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SHIM = os.path.join(ROOT, "tests", "compile_shim")
MPI_INC = next((d for d in ("/opt/conda/include", "/usr/include/mpich", "/usr/lib/x86_64-linux-gnu/mpich/include", "/usr/include/openmpi")
                if os.path.exists(os.path.join(d, "mpi.h"))), None)
HOP = ["-DA2AV", "-DSTRIDE", "-DSHSONG_HOPPER", "-DSHSONG_MPICH"]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "run-fft.c")) or MPI_INC is None,
                                reason="needs the mounted reference and an MPI header (development container only)")

PUBLIC = {
    "_offt_params": ["is_converged", "is_infeasible", "is_in_database", "v"],
    "_offt_comm": ["p1", "p2", "M1", "M2", "M3", "M4", "F1", "F2", "F3", "F4", "m1", "m2", "m3", "m4", "b1", "b2", "b3", "b4",
                   "istart", "isize", "istride", "ostart", "osize", "ostride"],
    "_offt_plan": ["p", "rank", "Nx", "Ny", "Nz", "is_r2c", "fftw_flag", "ah_strategy", "max_loop", "tuning_mode", "is_W0",
                   "extrapolation_window", "is_oned", "is_a2a", "is_equalxy", "is_notest", "t_init", "t", "point_database_file",
                   "user_vertex_file", "params", "comm"],
}
MACROS = ["PARAM_COUNT", "LOG0", "_P1_", "_T1_", "_W1_", "_Px1_", "_Py1_", "_Fz_", "_FP1_", "_Ux1_", "_Uz1_", "_FU1_", "_Fy1_", "_Ry_",
          "_T2_", "_W2_", "_Pz2_", "_Px2_", "_Fy2_", "_FP2_", "_Uz2_", "_Uy2_", "_FU2_", "_Fx_", "_V_", "_S_", "INIT_ALL", "INIT_FFTW",
          "INIT_AH", "INIT_BUFFER", "T_INIT_COUNT", "ALL", "INIT1", "WAIT1", "TEST1", "INIT2", "WAIT2", "TEST2", "FFTz", "FFTy1", "FFTy2",
          "FFTx", "TRANSPOSE", "PACK1", "UNPACK1", "PACK2", "UNPACK2", "GES", "TUNING_REPS", "SUBTILE_SIZE", "BUFFER_SIZE_LIMIT"]


def test_reference_caller_compiles_against_our_header():
    # `-include include/offt.h` comes first and shares the reference header's include guard (OFFT_INCLUDE), so the
    # `#include "offt.h"` inside run-fft.c -- which would find the file next to it -- adds nothing: the caller sees OUR header
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-w"] + HOP + ["-I" + SHIM, "-I" + MPI_INC, "-include", os.path.join(ROOT, "include", "offt.h"),
                                                     os.path.join(REF, "run-fft.c")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    # the guard trick must really have kept the reference's header out: its struct has MPI_Comm members, ours void *
    probe = subprocess.run(["gcc", "-E"] + HOP + ["-I" + SHIM, "-I" + MPI_INC, "-include", os.path.join(ROOT, "include", "offt.h"),
                                                  os.path.join(REF, "run-fft.c")], capture_output=True, text=True)
    assert probe.returncode == 0 and "MPI_Comm *comm1" not in probe.stdout and "void *comm1" in probe.stdout


def _layout_program(header_args, tmp_path, tag):
    lines = ["#include <stdio.h>", "#include <stddef.h>"]
    body = []
    for st, fields in PUBLIC.items():
        body.append(f'  printf("sizeof {st} %zu\\n", sizeof(struct {st}));') if st != "_offt_plan" else None
        for f in fields:
            body.append(f'  printf("{st}.{f} %zu %zu\\n", offsetof(struct {st}, {f}), sizeof(((struct {st} *)0)->{f}));')
    for m in MACROS:
        body.append(f'  printf("macro {m} %ld\\n", (long)({m}));')
    src = tmp_path / f"layout_{tag}.c"
    src.write_text("\n".join(lines) + "\nint main(void) {\n" + "\n".join(body) + "\n  return 0;\n}\n")
    exe = tmp_path / f"layout_{tag}"
    p = subprocess.run(["gcc", "-std=gnu99", "-w"] + header_args + [str(src), "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()


def test_public_field_layout_equals_the_reference_header(tmp_path):
    ours = _layout_program(["-include", os.path.join(ROOT, "include", "offt.h")], tmp_path, "ours")
    # the reference's header from where it lies; it pulls fftw3.h / fftw3-mpi.h / mpi.h for the types of its PRIVATE members
    ref = _layout_program(HOP + ["-I" + SHIM, "-I" + MPI_INC, "-include", os.path.join(REF, "offt.h")], tmp_path, "ref")
    assert len(ours) == len(ref) > 90
    for a, b in zip(ours, ref):
        assert a == b, (a, b)
    # prototypes: the five entry points with the reference's argument lists (a conflicting redeclaration is an error)
    both = tmp_path / "both.c"
    both.write_text('#include "%s"\n#undef OFFT_INCLUDE\n' % os.path.join(REF, "offt.h") +
                    "struct _offt_plan* offt_3d_init(int Nx, int Ny, int Nz, double* in, double* out, int is_r2c, int fftw_flag, int is_oned, "
                    "int is_a2a, int is_equalxy, int is_notest, int ah_strategy, int max_loop, int tuning_mode, int is_W0, "
                    "int extrapolation_window, struct _offt_params *custom_params);\n")
    p = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-w"] + HOP + ["-I" + SHIM, "-I" + MPI_INC, str(both)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    protos = [ln.strip() for ln in open(os.path.join(ROOT, "include", "offt.h")) if ln.startswith(("struct _offt_plan* offt_3d_init", "void offt_3d_", "void print_params", "void offt_print_time"))]
    refp = [ln.strip() for ln in open(os.path.join(REF, "offt.h")) if ln.startswith(("struct _offt_plan* offt_3d_init", "void offt_3d_", "void print_params", "void offt_print_time"))]
    norm = lambda s: "".join(s.split())
    assert len(protos) == 5 and all(any(norm(a) == norm(b) for b in refp) for a in protos), (protos, refp)
