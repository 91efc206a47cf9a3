"""TEST INFRASTRUCTURE: run the product's host logic (liboffthip.so) on the test-only CPU backend.

Single process: the all-to-all callback copies blocks locally (world of one rank).
Multi process: one `gloo` rank per process, the callback exchanges blocks with isend/irecv.
"""
import ctypes as C
import os

import numpy as np

import oracle_lib as O
from offt_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the TEST build of the library: same kernel objects, host compiled with -DOFFT_TEST_SEAMS (Makefile); the product
# library offt_amd/liboffthip.so carries neither offt_hip_test_set_backend nor offt_hip_test_set_transport
TEST_LIB = os.environ.get("OFFT_AMD_TEST_LIB") or os.path.join(ROOT, "tests", "liboffthip_test.so")
A2A_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_keep = {}


def _cb_lib():
    so = os.path.join(ROOT, "tests", "libcpubackend.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "tests/libcpubackend.so"])
    L = C.CDLL(so)
    L.cpu_backend_table.restype = C.c_void_p
    L.cpu_backend_pass_count.restype = C.c_long
    return L


def test_lib():
    """route offt_amd.api through the test build (api.use_library(None) goes back to the product)"""
    if not os.path.exists(TEST_LIB):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "tests/liboffthip_test.so"])
    L = api.use_library(TEST_LIB)
    L.offt_hip_test_set_backend.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.offt_hip_test_set_backend.restype = None
    L.offt_hip_test_set_transport.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.offt_hip_test_set_transport.restype = None
    return L


def group_peer(which, g, rank, size, p1):
    """world rank of member g of exchange group `which` as seen from `rank`: 1 = row group (p2 ranks sharing rank_x),
    2 = column group (p1 ranks sharing rank_y), 0 = the whole world (offt_backend.h)"""
    p2 = size // p1 if p1 else size
    rx, ry = rank // p2, rank % p2
    return g if which == 0 else (rx * p2 + g if which == 1 else g * p2 + ry)


def install(rank=0, size=1, p1=None, dist=None, fail_after=None):
    """Install the CPU backend into the test build of the library for (rank, size).  The mesh shape is asked from
    the library per exchange (offt_hip_test_current_p1; `p1` is kept for callers that pass it); fail_after = N makes
    the N-th exchange fail (communication-failure tests)."""
    CB = _cb_lib()
    L = test_lib()
    calls = {"n": 0}

    def a2a(which, npeers, peer_in_group, sendp, sendbytes, recvp, recvbytes):
        try:
            import torch
            calls["n"] += 1
            if fail_after is not None and calls["n"] >= fail_after:
                return -1
            cur_p1 = L.offt_hip_test_current_p1()
            reqs, keep = [], []
            for a in range(npeers):
                peer = group_peer(which, peer_in_group[a], rank, size, cur_p1)
                sb, rb = sendbytes[a], recvbytes[a]
                if peer == rank:
                    # a schedule exchange (which 1 / 2) of a real world never carries this rank's own block: the packing
                    # pass stores it where the next pass reads it (offt_pass_desc::out_block_tab) -- unless switched off
                    assert which == 0 or size == 1 or os.environ.get("OFFT_SELF_BYPASS") == "0", "self block in an exchange"
                    assert sb == rb
                    C.memmove(recvp[a], sendp[a], sb)
                    continue
                if dist is None:
                    continue  # a lone rank of a larger world (plan-only tests: decomposition, defaults): nothing moves
                if rb:
                    buf = (C.c_char * rb).from_address(recvp[a])
                    t = torch.frombuffer(buf, dtype=torch.uint8)
                    keep.append(t)
                    reqs.append(dist.irecv(t, src=peer, tag=which))
                if sb:
                    buf = (C.c_char * sb).from_address(sendp[a])
                    t = torch.frombuffer(buf, dtype=torch.uint8)
                    keep.append(t)
                    reqs.append(dist.isend(t, dst=peer, tag=which))
            for r in reqs:
                r.wait()
            return 0
        except Exception as e:  # surfaced as a failed execute
            print("a2a callback failed:", repr(e), flush=True)
            return -1

    cb = A2A_CB(a2a)
    _keep["cb"] = cb
    _keep["lib"] = CB
    CB.cpu_backend_set_a2a(cb)
    L.offt_hip_test_set_backend(CB.cpu_backend_table(), rank, size)
    return CB


def uninstall():
    L = test_lib()
    L.offt_hip_test_set_backend(None, 0, 1)
    api.use_library(None)


def run_rank(Nx, Ny, Nz, kind=1, is_equalxy=0, precision=api.F64, direction=-1, is_r2c=0, roundtrip=False, max_loop=0, **params):
    """init + fill this rank's block on the host + execute; returns (comm dict, params, local result array)."""
    cp = api.make_params(**params)
    po = api.offt_3d_init(Nx, Ny, Nz, custom_params=cp, is_equalxy=is_equalxy, precision=precision, is_r2c=is_r2c,
                          max_loop=max_loop)
    c = api.comm_dict(po)
    v = list(po.contents.params.contents.v)
    n = api.local_elems(po)
    ct = np.complex128 if precision == api.F64 else np.complex64
    buf = np.zeros(n, dtype=ct)
    i0, i1, i2 = c["isize"]
    if i0 and i1 and i2:
        f = O.hash_field(i0, i1, i2, c["istart"][0], c["istart"][1], c["istart"][2]) if kind else \
            O.ramp_field(Nx, Ny, Nz)[c["istart"][0]:c["istart"][0] + i0, c["istart"][1]:c["istart"][1] + i1, :]
        s0, s1, s2 = c["istride"]
        if is_r2c:  # real rows: scalar index z + 2*istride1*y + 2*istride0*x (run-fft.c:54)
            rv = buf.view(np.float64 if precision == api.F64 else np.float32)
            idx = np.arange(i0)[:, None, None] * 2 * s0 + np.arange(i1)[None, :, None] * 2 * s1 + np.arange(i2)[None, None, :]
            rv[idx.ravel()] = f.real.ravel()
        else:
            idx = np.arange(i0)[:, None, None] * s0 + np.arange(i1)[None, :, None] * s1 + np.arange(i2)[None, None, :] * s2
            buf[idx.ravel()] = f.astype(ct).ravel()
    ptr = buf.ctypes.data_as(C.c_void_p)
    api.offt_3d_execute_dir(po, ptr, ptr, direction)
    back = None
    if roundtrip:  # inverse of the result: must reproduce the input block times Nx*Ny*Nz
        back = buf.copy()
        bp = back.ctypes.data_as(C.c_void_p)
        api.offt_3d_execute_dir(po, bp, bp, +1)
    api.offt_3d_fin(po)
    if roundtrip:
        return c, v, buf, back
    return c, v, buf


def scatter_out(c, buf, G):
    """place a rank's output block (ostart/osize/ostride) into the global array G."""
    o0, o1, o2 = c["osize"]
    if not (o0 and o1 and o2):
        return
    s0, s1, s2 = c["ostride"]
    idx = np.arange(o0)[:, None, None] * s0 + np.arange(o1)[None, :, None] * s1 + np.arange(o2)[None, None, :] * s2
    G[c["ostart"][0]:c["ostart"][0] + o0, c["ostart"][1]:c["ostart"][1] + o1, c["ostart"][2]:c["ostart"][2] + o2] = \
        buf[idx.ravel()].reshape(o0, o1, o2)


def input_block(c, buf):
    """a rank's block read back through istart/isize/istride -> (array, global slices)"""
    i0, i1, i2 = c["isize"]
    s0, s1, s2 = c["istride"]
    idx = np.arange(i0)[:, None, None] * s0 + np.arange(i1)[None, :, None] * s1 + np.arange(i2)[None, None, :] * s2
    return buf[idx.ravel()].reshape(i0, i1, i2)
