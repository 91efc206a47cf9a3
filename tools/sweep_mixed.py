#!/usr/bin/env python3
"""Static sweep of the mixed-radix panel kernel (fft_panelx_k) shapes.

  tools/sweep_mixed.py gen  <f64|f32> N [N ...]   -> candidate list (stdout) + build/dev/msweep_<prec>/ library
  tools/sweep_mixed.py run  <f64|f32> N [N ...]   -> (on the GPU box) time every candidate, print the ranking
  tools/sweep_mixed.py emit <f64|f32> <log>       -> reg_variantx lines for the winners of a `run` log

A candidate is (TPL threads per line, radix order R0 x R1 x R2, COLS columns).  The generator enumerates ordered
factorisations into {2,3,5}-smooth radices <= 32 and thread counts, scores them by register footprint and idle
butterfly slots, and keeps the best few per length; the GPU decides between those.
"""
import itertools
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOOTH = [r for r in range(2, 33) if all(p in (2, 3, 5, 7, 11, 13) for p in
                                          [q for q in range(2, r + 1) if r % q == 0 and all(q % d for d in range(2, q))])]


def cdiv(a, b):
    return (a + b - 1) // b


def cfg(N, tpl, r0, r1, r2, cols, split, esz):
    rs = [r for r in (r0, r1, r2) if r > 1]
    nb = [cdiv(N // r, tpl) for r in rs]
    emax = max(b * r for b, r in zip(nb, rs))
    swz = r0 % 16 == 0 and N % 16 == 0
    paddiv = r0 if (r0 % 2 == 0 and not swz) else 0
    npad = N + N // paddiv if paddiv else N
    lstride = (npad + 31) // 32 * 32 + 4
    ex = cols * lstride * esz * (1 if split else 2) if len(rs) > 1 else 0
    qt = N // 4 + 1 if N % 4 == 0 else N
    lds = (ex + 15) // 16 * 16 + qt * 2 * esz if len(rs) > 1 else 0
    eff = sum(N / (tpl * b * r) for b, r in zip(nb, rs)) / len(rs)  # live fraction of butterfly slots
    return dict(emax=emax, lds=lds, nt=tpl * cols, eff=eff, nstage=len(rs))


def candidates(N, prec, keep=5):
    esz = 8 if prec == "f64" else 4
    emax_cap = int(os.environ.get("EMAX_CAP", "24" if prec == "f64" else "32"))
    out = []
    facts = set()
    for r0 in SMOOTH:
        if N % r0:
            continue
        m = N // r0
        if m == 1:
            facts.add((r0, 1, 1))
        for r1 in SMOOTH:
            if m % r1:
                continue
            r2 = m // r1
            if r2 == 1:
                facts.add((r0, r1, 1))
            elif r2 in SMOOTH:
                facts.add((r0, r1, r2))
    for (r0, r1, r2) in facts:
        for cols in ([int(os.environ["COLS"])] if os.environ.get("COLS") else (4, 8, 16)):
            for tpl in range(1, 257):
                if tpl * cols > 1024 or tpl * cols < 64:
                    continue
                c = cfg(N, tpl, r0, r1, r2, cols, True, esz)
                if (tpl * cols) % 64 and not (c["eff"] == 1.0 and (tpl * cols) % 16 == 0):
                    continue  # partial waves only for shapes without idle butterfly slots
                if c["emax"] > emax_cap or c["emax"] < 6 or c["lds"] > 160 * 1024 or c["eff"] < 0.74:
                    continue
                wg = max(1, min(4, (160 * 1024) // max(c["lds"], 1)))
                waves = wg * cdiv(c["nt"], 64)
                if waves < 4:
                    continue
                # heuristic score: live slots, enough waves per CU, wide panels (>= 128-B segments), few stages
                score = c["eff"] * min(1.0, waves / 8.0) ** 0.5 * (1.0 if cols * esz * 2 >= 128 or os.environ.get("COLS") else 0.8)
                score *= (1.0 if c["emax"] >= 12 or os.environ.get("EMAX_CAP") else 0.85) * (0.97 ** (c["nstage"] - 2) if c["nstage"] > 2 else 1.0)
                score *= 0.92 if max(r0, r1, r2) > 16 else 1.0      # radix > 16: > 64 live VGPR pairs in one butterfly
                if os.environ.get("R0"):
                    if r0 != int(os.environ["R0"]):
                        continue
                score *= (64 * cdiv(tpl * cols, 64)) and (tpl * cols) / (64 * cdiv(tpl * cols, 64))  # idle lanes of the last wave
                out.append((score, tpl, r0, r1, r2, cols, c))
    out.sort(key=lambda t: (-t[0], max(t[2:5]), -t[2]))
    # keep the best few, spread over radix multisets / thread counts / panel widths so the GPU sees different shapes
    picked, per_set, per_shape = [], {}, set()
    for t in out:
        ms = tuple(sorted(t[2:5]))
        shape = (ms, t[1], t[5])
        orders = sum(1 for q in picked if (tuple(sorted(q[2:5])), q[1], q[5]) == shape)
        if orders >= 2 or per_set.get(ms, 0) >= 3:
            continue
        if orders == 0 and any((tuple(sorted(q[2:5])) == ms and abs(q[1] - t[1]) < 16 and q[5] == t[5]) for q in picked):
            continue  # nearly the same thread count for the same radices
        per_set[ms] = per_set.get(ms, 0) + 1
        picked.append(t)
        if len(picked) >= keep:
            break
    return picked


def gen(prec, sizes, keep):
    T = "double" if prec == "f64" else "float"
    d = os.path.join(ROOT, "build", "dev", f"msweep_{prec}" + os.environ.get("TAG", ""))
    os.makedirs(d, exist_ok=True)
    groups = [[] for _ in range(8)]
    listing = []
    k = 0
    for N in sizes:
        for vid, (score, tpl, r0, r1, r2, cols, c) in enumerate(candidates(N, prec, keep)):
            line = f"  reg_variantx<{T}, {N}, {tpl}, {r0}, {r1}, {r2}, {cols}, true>({vid}, {'F_ALL' if vid == 0 else 0});"
            groups[k % 8].append(line)
            k += 1
            listing.append(f"N={N} v{vid} tpl={tpl} {r0}x{r1}x{r2} cols={cols} emax={c['emax']} lds={c['lds']} nt={c['nt']} eff={c['eff']:.2f} score={score:.3f}")
    open(os.path.join(d, "candidates.txt"), "w").write("\n".join(listing) + "\n")
    print("\n".join(listing))
    srcs = []
    for g, lines in enumerate(groups):
        src = os.path.join(d, f"reg_dev_{g}.hip")
        open(src, "w").write('#include "offt_panel.hpp"\nnamespace offtk {\nvoid reg_dev_%d() {\n%s\n}\n}\n' % (g, "\n".join(lines)))
        srcs.append(src)
    src = os.path.join(d, "reg_dev.hip")
    open(src, "w").write('#include "offt_panel.hpp"\nnamespace offtk {\n' + "".join(f"void reg_dev_{g}();\n" for g in range(8)) +
                         "void reg_dev() {\n" + "".join(f"  reg_dev_{g}();\n" for g in range(8)) + "}\n}\n")
    srcs.append(src)
    srcs.append(os.path.join(ROOT, "offt_amd", "csrc", "offt_kernels.hip"))
    flags = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "offt_amd", "csrc"),
             "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "build"), "-DOFFT_DEV_REGISTRY", "-Rpass-analysis=kernel-resource-usage"]

    def cc(s):
        o = os.path.join(d, os.path.basename(s).replace(".hip", ".o"))
        r = subprocess.run(flags + ["-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-3000:])
            raise SystemExit(f"compile failed: {s}")
        open(o + ".res.txt", "w").write(r.stderr)
        return o
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.check_call(["g++", "-shared", "-o", os.path.join(d, "liboffthip.so")] + objs + [os.path.join(ROOT, "build", "offt_host.o"),
                          "-Wl,--allow-shlib-undefined", "-ldl", "-lm", "-lpthread"])
    for o in objs:
        os.remove(o)
    print("built", os.path.join(d, "liboffthip.so"))


def run(prec, sizes):
    if os.environ.get("TAG") != "prod" and not os.environ.get("OFFT_AMD_LIB"):  # TAG=prod: the product library's variants
        os.environ["OFFT_AMD_LIB"] = os.path.join(ROOT, "build", "dev", f"msweep_{prec}" + os.environ.get("TAG", ""), "liboffthip.so")
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    from offt_amd import api
    L = api.lib()
    P = api.F64 if prec == "f64" else api.F32
    esz = 16 if prec == "f64" else 8
    for N in sizes:
        # a cube when two copies of it fit comfortably, else a slab with N on the z and x axes
        shape = (N, N, N) if 2 * esz * N ** 3 < 150e9 else (N, 256, N)
        nv = L.offt_hipk_variant_count(N, P)
        res = []
        for v in range(nv):
            po = api.offt_3d_init(*shape, custom_params=api.make_params(S=0), precision=P)
            for ax in range(3):
                L.offt_hipk_variant_count(N, P)
                L.offt_hip_set_variant(po, ax, v)
            n = api.local_elems(po)
            dev = torch.zeros(n * 2, dtype=torch.float64 if prec == "f64" else torch.float32, device="cuda")
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                L.offt_hip_fill_input(po, dev.data_ptr(), 1)
                api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
                t = (C.c_double * 3)()
                L.offt_hip_last_pass_seconds(po, t)
                tt = t[0] + t[2] if shape[1] != N else t[0] + t[1] + t[2]
                if best is None or tt < best:
                    best, per = tt, (t[0], t[1], t[2])
            api.offt_3d_fin(po)
            del dev
            npass = 2 if shape[1] != N else 3
            E = shape[0] * shape[1] * shape[2]
            frac = npass * 2 * esz * E / best / 8e12
            res.append((frac, v))
            print(f"N={N} {prec} v{v} {L.offt_hipk_variant_name(N, P, v).decode()}: {best * 1e3:.3f} ms over {npass} passes of {shape} = {frac * 100:.1f}% of 8 TB/s"
                  f"  [z/y/x {per[0] * 1e3:.3f}/{per[1] * 1e3:.3f}/{per[2] * 1e3:.3f} ms]", flush=True)
        if res:
            frac, v = max(res)
            print(f"BEST N={N} {prec} v{v} {frac * 100:.1f}% {L.offt_hipk_variant_name(N, P, v).decode()}", flush=True)


def emit(prec, log):
    T = "double" if prec == "f64" else "float"
    for line in open(log):
        m = re.match(r"BEST N=(\d+) (\w+) v\d+ ([\d.]+)% .*radix=(\d+)x(\d+)x(\d+) threads/line=(\d+) .*cols=(\d+)", line)
        if m and m.group(2) == prec:
            N, _, frac, r0, r1, r2, tpl, cols = m.groups()
            print(f"  reg_variantx<{T}, {N}, {tpl}, {r0}, {r1}, {r2}, {cols}, true>(0);  // {frac} %")


if __name__ == "__main__":
    mode, prec = sys.argv[1], sys.argv[2]
    if mode == "emit":
        emit(prec, sys.argv[3])
    else:
        sizes = [int(x) for x in sys.argv[3:]]
        if mode == "gen":
            gen(prec, sizes, int(os.environ.get("KEEP", "5")))
        else:
            run(prec, sizes)
