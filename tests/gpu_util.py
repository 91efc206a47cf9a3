"""helpers for the -m gpu parity tests: everything goes through the C ABI (offt_amd.api)."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O
from offt_amd import api


def make_input(po, field, precision=api.F64):
    """host array in this rank's input layout (istart/isize/istride) -> device tensor"""
    c = api.comm_dict(po)
    ct = np.complex128 if precision == api.F64 else np.complex64
    buf = np.zeros(api.local_elems(po), dtype=ct)
    s0, s1, s2 = c["istride"]
    n0, n1, n2 = field.shape
    idx = (np.arange(n0)[:, None, None] * s0 + np.arange(n1)[None, :, None] * s1 + np.arange(n2)[None, None, :] * s2).ravel()
    buf[idx] = field.astype(ct).ravel()
    dev = torch.from_numpy(buf.view(np.float64 if precision == api.F64 else np.float32)).cuda()
    return dev, idx


def read_output(po, dev, shape, precision=api.F64):
    c = api.comm_dict(po)
    ct = np.complex128 if precision == api.F64 else np.complex64
    res = dev.cpu().numpy().view(ct)
    s0, s1, s2 = c["ostride"]
    n0, n1, n2 = shape
    idx = (np.arange(n0)[:, None, None] * s0 + np.arange(n1)[None, :, None] * s1 + np.arange(n2)[None, None, :] * s2).ravel()
    return res[idx].reshape(shape)


def gpu_fft(shape, field=None, is_equalxy=0, precision=api.F64, is_r2c=0, **params):
    field = O.hash_field(*shape) if field is None else field
    po = api.offt_3d_init(*shape, custom_params=api.make_params(**params), is_equalxy=is_equalxy, precision=precision,
                          is_r2c=is_r2c)
    try:
        if is_r2c:
            c = api.comm_dict(po)
            ct = np.complex128 if precision == api.F64 else np.complex64
            buf = np.zeros(api.local_elems(po), dtype=ct)
            rv = buf.view(np.float64 if precision == api.F64 else np.float32)
            s0, s1, _ = c["istride"]
            n0, n1, n2 = shape
            idx = (np.arange(n0)[:, None, None] * 2 * s0 + np.arange(n1)[None, :, None] * 2 * s1 + np.arange(n2)[None, None, :]).ravel()
            rv[idx] = field.real.ravel()
            dev = torch.from_numpy(rv).cuda()
            api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            return read_output(po, dev, (n0, n1, n2 // 2 + 1), precision), c
        dev, _ = make_input(po, field, precision)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        return read_output(po, dev, shape, precision), api.comm_dict(po)
    finally:
        api.offt_3d_fin(po)


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def gpu_roundtrip(shape, precision=api.F64, is_equalxy=0, **params):
    """forward, then the inverse on the result: rel-L2 distance of (result / N) from the input block"""
    field = O.hash_field(*shape)
    po = api.offt_3d_init(*shape, custom_params=api.make_params(**params), is_equalxy=is_equalxy, precision=precision)
    try:
        dev, idx = make_input(po, field, precision)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
        ct = np.complex128 if precision == api.F64 else np.complex64
        back = dev.cpu().numpy().view(ct)[idx].reshape(shape) / np.prod(shape)
        return rel(back, field)
    finally:
        api.offt_3d_fin(po)
