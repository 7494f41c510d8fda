#!/usr/bin/env python3
"""Algorithmic FLOPs of one stage evaluation + derivative + Gauss-Newton assembly, counted by instrumentation in the oracle
(SURVEY 8d): builds oracle/_flops/liborc_flops.so with -DORC_COUNT_FLOPS (every V3 / M3 / dense-matrix primitive of the oracle
adds its operation count, an FMA = 2; structural zeros skipped by the oracle's sparse-aware products are not counted), runs the
stage models at seeded states and writes profiles/flop_counts.json, which bench.py uses for the FP64 side of the roofline of
the rigid-body kernels.  The proximal Riccati recursion is counted the same way as a cross-check of SURVEY's F_ric formula.

    python tools/count_flops.py
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT_DIR = os.path.join(ROOT, "oracle", "_flops")
LIB = os.path.join(OUT_DIR, "liborc_flops.so")


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-DORC_COUNT_FLOPS", "-Wno-unused-function",
                           os.path.join(ROOT, "oracle", "orc_capi.cpp"), "-o", LIB])
    import oracle_lib as O

    O.LIB_PATH = LIB
    O.build = lambda force=False: LIB
    L = O.lib()
    L.orc_flops_get.restype = C.c_double
    os.environ["OMP_NUM_THREADS"] = "1"

    def count(fn):
        L.orc_flops_reset()
        fn()
        return float(L.orc_flops_get())

    out = {}
    rng = np.random.default_rng(1)
    # ---- Go2 kinodynamics ----
    rb = O.Robot("go2_like")
    K = O.Kino(rb, O.go2_kino_settings(rb))
    sg = np.concatenate([np.ones(3) * 0.02, np.ones(3) * 0.05, np.ones(12) * 0.1, np.ones(3) * 0.1, np.ones(3) * 0.2, np.ones(12) * 0.5])
    x = rb.integrate(rb.x_ref, rng.normal(size=36) * sg)
    u = np.concatenate([np.tile([0, 0, 37.0], 4) + rng.normal(size=12), rng.normal(size=12)])
    u_ref = np.concatenate([np.tile([0, 0, rb.mass * 9.81 / 4], 4), np.zeros(12)])
    fr = rb.centroidal(rb.x_ref)["feet"]
    ev = {m: count(lambda: K.eval(m, u_ref, rb.x_ref, fr, x, u)) for m in (15, 6)}
    de = {m: count(lambda: K.deriv(m, u_ref, rb.x_ref, fr, x, u)) for m in (15, 6)}
    # trot: 20 of 80 cycle stages have four feet in contact, 60 have two
    out["kinodynamics"] = {"eval_flops_per_stage": 0.25 * ev[15] + 0.75 * ev[6], "deriv_flops_per_stage": 0.25 * de[15] + 0.75 * de[6],
                           "by_mask": {"15": {"eval": ev[15], "deriv": de[15]}, "6": {"eval": ev[6], "deriv": de[6]}},
                           "note": "deriv = evaluation + derivatives + Gauss-Newton Hessian assembly of one stage (orc_kino.hpp KinoModel::deriv); trot mix 25 % / 75 %"}
    # ---- Go2 / Talos full dynamics ----
    for name, robot, fs, masks, mix in (("fulldynamics_go2", "go2_like", 3, (15, 6), (0.25, 0.75)), ("fulldynamics_talos", "talos_like", 6, (3, 1), (0.2, 0.8))):
        r = O.Robot(robot)
        s = O.go2_full_settings(r) if fs == 3 else O.talos_full_settings(r)
        F = O.Full(r, s)
        sgr = np.concatenate([np.ones(3) * 0.02, np.ones(3) * 0.05, np.ones(r.nv - 6) * 0.1, np.ones(3) * 0.1, np.ones(3) * 0.2, np.ones(r.nv - 6) * 0.5])
        xx = r.integrate(r.x_ref, rng.normal(size=r.ndx) * sgr * 0.5)
        uu = rng.normal(size=F.nu) * 3
        fz = np.zeros(fs)
        fz[2] = r.mass * 9.81 / r.nf
        ur = np.concatenate([np.zeros(F.nu), np.tile(fz, r.nf)])
        ft = r.centroidal(r.x_ref)["feet"]
        e = [count(lambda: F.eval(m, ur, r.x_ref, ft, xx, uu)) for m in masks]
        d = [count(lambda: F.deriv(m, ur, r.x_ref, ft, xx, uu)) for m in masks]
        out[name] = {"eval_flops_per_stage": mix[0] * e[0] + mix[1] * e[1], "deriv_flops_per_stage": mix[0] * d[0] + mix[1] * d[1],
                     "by_mask": {str(m): {"eval": e[i], "deriv": d[i]} for i, m in enumerate(masks)}}
    # ---- proximal Riccati recursion (cross-check of F_ric) ----
    def f_ric(ndx, nu, nc):
        return (4 * ndx**3 + 4 * ndx**2 * nu + 2 * ndx * nu**2 + (nu + nc) ** 3 / 3 + 2 * (nu + nc) ** 2 * (ndx + 1) + 2 * ndx**2 * (nu + nc) + 2 * ndx * (nu + nc))

    ric = {}
    for ndx, nu, nc in ((36, 24, 24), (36, 12, 24), (56, 22, 78)):
        H = 4
        A = rng.normal(size=(H, ndx, ndx)) * 0.1 + np.eye(ndx)
        Bm = rng.normal(size=(H, ndx, nu))
        Q = np.tile(np.eye(ndx), (H, 1, 1)) + 0.0
        R = np.tile(np.eye(nu), (H, 1, 1)) + 0.0
        Sx = rng.normal(size=(H, ndx, nu)) * 0.01
        Cm, Dm = rng.normal(size=(H, nc, ndx)), rng.normal(size=(H, nc, nu))
        args = (Q, Sx, R, rng.normal(size=(H, ndx)), rng.normal(size=(H, nu)), A, Bm, rng.normal(size=(H, ndx)) * 0.01, Cm, Dm,
                rng.normal(size=(H, nc)) * 0.01, np.eye(ndx), rng.normal(size=ndx), 1e-8)
        n = count(lambda: O.riccati(*args)) / H
        ric["%d,%d,%d" % (ndx, nu, nc)] = {"counted_per_stage": n, "F_ric": f_ric(ndx, nu, nc)}
    out["riccati_cross_check"] = ric
    with open(os.path.join(ROOT, "profiles", "flop_counts.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        print(k, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items() if a != "by_mask" and a != "note"})


if __name__ == "__main__":
    main()
