#!/usr/bin/env python3
"""Developer smoke: GPU result vs numpy.fft.fftn for a few shapes/layouts (not a pytest)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from offt_amd import api

def hash_field(Nx, Ny, Nz):
    x = np.arange(Nx, dtype=np.uint64)[:, None, None]; y = np.arange(Ny, dtype=np.uint64)[None, :, None]
    z = np.arange(Nz, dtype=np.uint64)[None, None, :]
    M = np.uint64(0xffffffff)
    def val(c):
        h = ((x * np.uint64(73856093)) & M) ^ ((y * np.uint64(19349663)) & M) ^ ((z * np.uint64(83492791)) & M) ^ np.uint64((c * 2654435761) & 0xffffffff)
        h ^= h >> np.uint64(13); h = (h * np.uint64(0x5bd1e995)) & M; h ^= h >> np.uint64(15)
        return (h & np.uint64(0xffffff)).astype(np.float64) / 8388608.0 - 1.0
    return val(0) + 1j * val(1)

def run(Nx, Ny, Nz, S=0, eq=0, prec=api.F64, inverse=False):
    cp = api.make_params(S=S)
    po = api.offt_3d_init(Nx, Ny, Nz, custom_params=cp, is_equalxy=eq, precision=prec)
    c = api.comm_dict(po)
    n = api.local_elems(po)
    f = hash_field(Nx, Ny, Nz)
    ctype = np.complex128 if prec == api.F64 else np.complex64
    host = np.zeros(n, dtype=ctype)
    is0, is1, is2 = c["istride"]
    idx = (np.arange(Nx)[:, None, None] * is0 + np.arange(Ny)[None, :, None] * is1 + np.arange(Nz)[None, None, :] * is2)
    host[idx.ravel()] = f.astype(ctype).ravel()
    dev = torch.from_numpy(host.view(np.float64 if prec == api.F64 else np.float32)).cuda()
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    res = dev.cpu().numpy().view(ctype)
    os0, os1, os2 = c["ostride"]
    oidx = (np.arange(Nx)[:, None, None] * os0 + np.arange(Ny)[None, :, None] * os1 + np.arange(Nz)[None, None, :] * os2)
    G = res[oidx.ravel()].reshape(Nx, Ny, Nz)
    F = np.fft.fftn(f)
    err = np.linalg.norm(G - F) / np.linalg.norm(F)
    msg = f"N=({Nx},{Ny},{Nz}) S={S} eq={eq} prec={prec}: relL2={err:.3e} dev_s={api.lib().offt_hip_last_device_seconds(po):.6f}"
    if inverse:
        api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
        back = dev.cpu().numpy().view(ctype)[idx.ravel()].reshape(Nx, Ny, Nz) / (Nx * Ny * Nz)
        msg += f" roundtrip={np.linalg.norm(back - f) / np.linalg.norm(f):.3e}"
    print(msg, flush=True)
    api.offt_3d_fin(po)
    return err

if __name__ == "__main__":
    bad = 0
    tol = {api.F64: 1e-13, api.F32: 5e-6}
    cases = [(8, 8, 8), (16, 16, 16), (32, 32, 32), (64, 64, 64), (128, 128, 128), (256, 256, 256),
             (64, 32, 16), (16, 128, 64), (20, 20, 20), (18, 12, 30), (512, 8, 8), (8, 1024, 16), (8, 8, 2048), (4096, 8, 8)]
    for prec in (api.F64, api.F32):
        for (a, b, c) in cases:
            for (S, eq) in ((1, 0), (0, 0), (0, 1)):
                if eq and a != b: continue
                e = run(a, b, c, S=S, eq=eq, prec=prec, inverse=True)
                bad += e > tol[prec]
    print("FAILED" if bad else "ALL OK", bad)
    sys.exit(1 if bad else 0)
