#!/usr/bin/env python3
"""bench_side.py -- the single-GPU workloads measured beside bench.py's headline (the other BASELINE configurations, the inverse-dynamics QPs,
the control stacks, the B = 1 latency).  bench.py calls run_all() after its timed region and keeps only the flat cfg* values in its line;
the full entries (rooflines, CPU legs) go to stderr, one short JSON line per workload, and to bench_side.json next to this file.

    python bench_side.py [--iters 3] [--no-cpu-baseline] [--only name,name]
"""
import json
import os
import sys
import time

import bench_common as C
from bench_common import ROOT, _oracle_imports, both_bounds, f_ric, flop_counts, loop_stream, make_mpc, pmc_traffic, step_sync


def constraint_dynamics_line(gm, mh, batch, horizon):
    """Constrained forward dynamics kernel of the full-dynamics model alone, on batch x horizon states, all feet in contact --
    one launch, timed around the launch itself."""
    import numpy as np
    from simple_mpc import presets as P

    n = batch * horizon
    X = np.tile(P.random_states(mh, 512, seed=3), ((n + 511) // 512, 1))[:n]
    tau, mask = np.zeros((n, mh.nv - 6)), np.full(n, (1 << mh.getFeetNb()) - 1, np.uint32)
    ms = [gm.constraintDynamics(X, tau, mask)["kernel_ms"] for _ in range(3)]
    return {"metric": "constrained forward dynamics (full-dynamics model), states/sec", "value": n / (min(ms) * 1e-3), "unit": "states/s",
            "kernel_ms": min(ms), "states": n, "dtype": "f64", "note": "smpc_full_forward_dynamics: the forward dynamics kernel alone"}


def inverse_dynamics_line(batch, device_id, with_cpu=True):
    """Whole-body inverse-dynamics QP (KinodynamicsID): one QP per robot and control tick, `batch` robots per call;
    all tasks on, the 1 kHz settings of the reference's tests.  Timed around solve() -- host copies of the states and torques included."""
    import numpy as np
    import torch  # (before the HIP library initialises the runtime: torch's own copy of it then finds the devices)
    import simple_mpc
    from simple_mpc import presets as P

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like"), "standing", "root_joint")
    for n in P.GO2_FEET:
        mh.addPointFoot(n, "root_joint")
    eff, vmax = np.array([23.7, 23.7, 45.43] * 4), np.array([30.1, 30.1, 15.7] * 4)
    st = dict(kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)
    kid = simple_mpc.KinodynamicsID(mh, 1e-3, st, eff, vmax, batch=batch, device_id=device_id, admm_iters=100, admm_tol=-1.0)  # fixed work per QP
    X = P.random_states(mh, batch, scale=0.3)
    q, v = X[:, : mh.nq], X[:, mh.nq :]
    for _ in range(3):
        kid.solve(0.0, q, v)
    t0, n = time.perf_counter(), 20
    for _ in range(n):
        kid.solve(0.0, q, v)
    dt_host = (time.perf_counter() - t0) / n
    Xd = torch.from_numpy(np.ascontiguousarray(X)).to(torch.device("cuda", device_id))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        kid.solve_device(Xd.data_ptr())
    kid.wait()
    dt = (time.perf_counter() - t0) / n
    out = {"metric": "whole-body inverse-dynamics QPs/sec (KinodynamicsID: 30 variables, 76 rows, 100 ADMM iterations)", "value": batch / dt,
           "unit": "QPs/s", "ms_per_call": dt * 1e3, "batch": batch, "dtype": "f64", "max_residual": float(kid.resid.max()),
           "host_buffers": {"value": batch / dt_host, "unit": "QPs/s", "ms_per_call": dt_host * 1e3},
           "note": "smpc_id_solve_device: rigid-body quantities + QP assembly + ADMM, three kernels, states and torques resident in HBM; "
                   "host_buffers = smpc_id_solve with the copies of states, torques, accelerations and forces"}
    # the solver of record (stop on residuals <= 1e-7, checked every 20 iterations, cap 400) on states that move between ticks
    kid2 = simple_mpc.KinodynamicsID(mh, 1e-3, st, eff, vmax, batch=batch, device_id=device_id)
    rng = np.random.default_rng(5)
    Xs = [X + np.concatenate([np.zeros((batch, 7)), rng.normal(0.0, 2e-3, (batch, X.shape[1] - 7))], axis=1) for _ in range(8)]
    for k in range(3):
        kid2.solve(0.0, Xs[k][:, : mh.nq], Xs[k][:, mh.nq :])
    Xds = [torch.from_numpy(np.ascontiguousarray(x)).to(torch.device("cuda", device_id)) for x in Xs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        kid2.solve_device(Xds[k % 8].data_ptr())
    kid2.wait()
    dt2 = (time.perf_counter() - t0) / n
    kid2.solve(0.0, Xs[0][:, : mh.nq], Xs[0][:, mh.nq :])  # (the residuals come back with the host-buffer call)
    # the same at 1e-5, the absolute tolerance ProxQP runs with by default (proxsuite Settings::eps_abs, as recalled: the library is absent)
    kid3 = simple_mpc.KinodynamicsID(mh, 1e-3, st, eff, vmax, batch=batch, device_id=device_id, admm_tol=1e-5)
    for k in range(3):
        kid3.solve(0.0, Xs[k][:, : mh.nq], Xs[k][:, mh.nq :])
    t0 = time.perf_counter()
    for k in range(n):
        kid3.solve_device(Xds[k % 8].data_ptr())
    kid3.wait()
    dt3 = (time.perf_counter() - t0) / n
    kid3.solve(0.0, Xs[0][:, : mh.nq], Xs[0][:, mh.nq :])
    out["default_stopping_rule"] = {"value": batch / dt2, "unit": "QPs/s", "ms_per_call": dt2 * 1e3, "max_residual": float(kid2.resid.max()),
                                    "note": "residuals <= 1e-7 checked every 20 iterations (cap 400), warm start, joint states perturbed by N(0, 2e-3) per tick",
                                    "at_tolerance_1e-5": {"value": batch / dt3, "unit": "QPs/s", "ms_per_call": dt3 * 1e3, "max_residual": float(kid3.resid.max())}}
    if with_cpu:
        S, O = _oracle_imports()
        threads = O.use_effective_cpus()
        rbc = O.Robot("go2_like")
        Bc = 8 * threads
        ok = O.OracleKinoID(rbc, O.id_settings(rbc, 1e-3, admm_iters=100, admm_tol=-1.0, **st), Bc)
        Xc = S.random_states(rbc, Bc, scale=0.3)
        ok.solve(Xc)
        t0, n = time.time(), 0
        while time.time() - t0 < 3.0:
            ok.solve(Xc)
            n += 1
        out["cpu_baseline"] = {"value": Bc * n / (time.time() - t0), "unit": "QPs/s", "cores": threads, "kind": "port",
                               "sample": "CPU restatement (oracle/, not TSID / ProxQP): %d robots x %d ticks, same ADMM" % (Bc, n)}
    return out


def inverse_dynamics_quad_line(batch, device_id, with_cpu=True):
    """KinodynamicsID of a biped with flat feet (tsid Contact6d: 12 corner forces per foot, 52 variables / 126 rows): the gains of the
    reference's contactQuad test, 100 ADMM iterations of fixed work per QP, states resident in HBM."""
    import numpy as np
    import torch
    import simple_mpc
    from simple_mpc import presets as P

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("talos_like"), "half_sitting", "root_joint")
    for n in P.TALOS_FEET:
        mh.addQuadFoot(n, "root_joint", P.TALOS_QUAD)
    st = dict(kp_base=1.0, kp_posture=1.0, kp_contact=10.0, w_base=1.0, w_posture=0.05, w_contact_motion=10.0, w_contact_force=1.0)
    kid = simple_mpc.KinodynamicsID(mh, 1e-3, st, P.TALOS_EFFORT, P.TALOS_VMAX, batch=batch, device_id=device_id, admm_iters=100, admm_tol=-1.0)
    X = P.random_states(mh, batch, scale=0.2)
    for _ in range(3):
        kid.solve(0.0, X[:, : mh.nq], X[:, mh.nq :])
    Xd = torch.from_numpy(np.ascontiguousarray(X)).to(torch.device("cuda", device_id))
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        kid.solve_device(Xd.data_ptr())
    kid.wait()
    dt = (time.perf_counter() - t0) / n
    out = {"metric": "whole-body inverse-dynamics QPs/sec, flat feet (KinodynamicsID with Contact6d: 52 variables, 126 rows, 100 ADMM iterations)",
           "value": batch / dt, "unit": "QPs/s", "ms_per_call": dt * 1e3, "batch": batch, "dtype": "f64", "max_residual": float(kid.resid.max())}
    # roofline of the dominant kernel (qp6_admm_body) over the whole call (three kernels; the ADMM kernel is > 80 % of it).  FP64 side, counted on
    # the STRUCTURE the kernel uses (round 5): per ADMM iteration one product with K^-1 (n x n, n = 52) and two with the constraint matrix, whose
    # non-zeros are the 40 dense rows (dynamics 6, contact motion 12, actuation 22) x n, the 34 friction rows of Contact6d (32 pyramid rows of 2
    # entries, 2 normal-force rows of 4) and the n box rows -> 2 (n^2 + 2 (40 n + 72 + n)) FLOPs; x 100 iterations.  (Counted dense over all 126
    # rows, as rounds 3 - 4 did, the same work would read 2.2 x higher.)  HBM side = the assembled QP it reads once (H, C, bounds) + the solution
    n_, m_, it_ = 52, 126, 100
    fl = batch * it_ * 2.0 * (n_ * n_ + 2 * (40 * n_ + 72 + n_))
    by = batch * 8.0 * (n_ * n_ + m_ * n_ + 3 * m_ + 2 * n_)
    out["roofline"] = both_bounds(fl, by, dt, "mfma")
    out["roofline"].update({"kernel": "id6_assemble_body + qp6_admm_body (whole solve_device call)",
                            "note": "algorithmic FLOPs = B x 100 x 2 (n^2 + 2 (40 n + 72 + n)), n = 52: K^-1 and the non-zeros of C (40 dense rows, "
                                    "34 friction rows with 72 entries, n box rows) per ADMM iteration", "traffic": None})
    if with_cpu:
        S, O = _oracle_imports()
        threads = O.use_effective_cpus()
        rbc = O.Robot("talos_like")
        Bc = 4 * threads
        ok = O.OracleKinoID(rbc, O.talos_id_settings(rbc, 1e-3, admm_iters=100, admm_tol=-1.0, **st), Bc)
        Xc = S.talos_random_states(rbc, Bc, scale=0.2)
        ok.solve(Xc)
        t0, nn = time.time(), 0
        while time.time() - t0 < 3.0:
            ok.solve(Xc)
            nn += 1
        out["cpu_baseline"] = {"value": Bc * nn / (time.time() - t0), "unit": "QPs/s", "cores": threads, "kind": "port",
                               "sample": "CPU restatement (oracle/, not TSID / ProxQP): %d robots x %d ticks, same ADMM" % (Bc, nn)}
    return out


def single_robot_latency(iters, device_id, steps=50):
    """One robot (B = 1): wall time of MPC::iterate through host buffers, the reference's own use case (its control loop calls iterate once
    per 10 ms period).  The stage kernels have 51 wavefronts of work and the Riccati sweeps one: this is a latency, not a throughput."""
    import numpy as np
    from simple_mpc import presets as P

    out = {}
    for kind in ("kinodynamics", "centroidal", "fulldynamics"):
        gm, mh = make_mpc(kind, 1, iters, device_id)
        X = np.tile(mh.getReferenceState(), (1, 1))
        for _ in range(5):
            gm.iterate(X)
            X = gm.xs[:, 1, :].copy() if kind != "centroidal" else X
        lat = []
        for _ in range(steps):
            t0 = time.perf_counter()
            gm.iterate(X)
            lat.append(time.perf_counter() - t0)
            X = gm.xs[:, 1, :].copy() if kind != "centroidal" else X
        out[kind] = {"median_ms": 1e3 * float(np.median(lat)), "p99_ms": 1e3 * float(np.quantile(lat, 0.99))}
        del gm
    out["note"] = "B = 1, %d ProxDDP iterations per step, host buffers in and out, Go2, H = 50" % iters
    return out


def control_stack_line(batch, device_id, mpc_steps=30):
    """The control stack of the reference's examples/go2_kinodynamics.py for `batch` simulated robots, nothing crossing the host inside
    the loop: MPC (1 ProxDDP iteration, as the example runs it) at 100 Hz, interpolated targets + KinodynamicsID at 1 kHz, constrained
    forward dynamics + semi-implicit Euler as the simulator (examples/go2_stack_resident.py)."""
    import numpy as np
    import torch
    import simple_mpc
    from simple_mpc import presets as P

    gm, mh = make_mpc("kinodynamics", batch, 1, device_id)
    eff, vmax = np.array([23.7, 23.7, 45.43] * 4), np.array([30.1, 30.1, 15.7] * 4)
    ids = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=1.0, w_contact_motion=1.0)
    kid = simple_mpc.KinodynamicsID(mh, 1e-3, ids, eff, vmax, batch=batch, device_id=device_id)
    kid.shareStream(gm)  # one in-order queue for the MPC step, the targets, the QP solves and the simulator steps
    X = torch.from_numpy(np.tile(mh.getReferenceState(), (batch, 1))).to(torch.device("cuda", device_id))
    torch.cuda.synchronize()

    def period():
        gm.iterate_device(X.data_ptr())
        gm.wait()
        contact = gm.ocp_handler.getContactState(0)
        for sub in range(10):
            kid.setTargetsFromMPC(gm, sub * 1e-3)
            kid.solve_device(X.data_ptr())
            gm.simStepDevice(X.data_ptr(), kid.tau_device_ptr(), contact, 1e-3, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])

    for _ in range(3):
        period()
    gm.wait()
    t0 = time.perf_counter()
    for _ in range(mpc_steps):
        period()
    gm.wait()
    dt = (time.perf_counter() - t0) / mpc_steps
    Xh = X.cpu().numpy()
    kid.shareStream(None)
    ok = bool(np.all(np.isfinite(Xh)) and np.all(np.abs(Xh[:, 2] - mh.getReferenceState()[2]) < 0.05))
    return {"metric": "simulated robot-seconds per second, MPC (100 Hz, 1 iteration) + KinodynamicsID (1 kHz) + forward-dynamics simulator",
            "value": batch * 0.01 / dt, "unit": "robot-seconds/s", "ms_per_mpc_period": dt * 1e3, "batch": batch, "robots_upright": ok,
            "note": "states, targets and torques resident in HBM; one MPC period = 1 iterate + 10 x (targets from the MPC, ID QP, simulator step)"}


def control_stack_talos_line(batch, device_id, mpc_steps=20):
    """The control stack of the reference's examples/talos_kinodynamics.py for `batch` simulated bipeds, nothing crossing the host inside
    the loop: kinodynamics MPC with 6-D feet (1 ProxDDP iteration) at 100 Hz, interpolated targets + KinodynamicsID with flat feet at 1 kHz,
    constrained forward dynamics with 6-D contacts + semi-implicit Euler as the simulator (a full-dynamics handle of the same robot)."""
    import numpy as np
    import torch
    import simple_mpc
    from simple_mpc import presets as P

    gm, mh = make_mpc("talos_kinodynamics", batch, 1, device_id, horizon=100)
    focp = simple_mpc.FullDynamicsOCP(P.talos_full_settings(mh), mh)
    focp.createProblem(mh.getReferenceState(), 2, 6, -9.81, False)
    ms = P.talos_mpc_settings(mh, max_iters=1)
    sim = simple_mpc.BatchedMPC({k: ms[k] for k in P.MPC_KEYS}, focp, batch, device_id=device_id)
    ids = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=0.001, w_contact_motion=1.0)
    kid = simple_mpc.KinodynamicsID(mh, 1e-3, ids, P.TALOS_EFFORT, P.TALOS_VMAX, batch=batch, device_id=device_id)
    kid.shareStream(gm)
    X = torch.from_numpy(np.tile(mh.getReferenceState(), (batch, 1))).to(torch.device("cuda", device_id))
    torch.cuda.synchronize()

    def period():
        gm.iterate_device(X.data_ptr())
        gm.wait()
        contact = gm.ocp_handler.getContactState(0)
        for sub in range(10):
            kid.setTargetsFromMPC(gm, sub * 1e-3)
            kid.solve_device(X.data_ptr())
            kid.wait()  # (the simulator runs on its own handle's stream: the torques must be complete)
            sim.simStepDevice(X.data_ptr(), kid.tau_device_ptr(), contact, 1e-3, Kp=[0.0] * 6, Kd=[50.0] * 6)
            sim.wait()

    for _ in range(3):
        period()
    gm.wait()
    t0 = time.perf_counter()
    for _ in range(mpc_steps):
        period()
    gm.wait()
    dt = (time.perf_counter() - t0) / mpc_steps
    Xh = X.cpu().numpy()
    kid.shareStream(None)
    ok = bool(np.all(np.isfinite(Xh)) and np.all(np.abs(Xh[:, 2] - mh.getReferenceState()[2]) < 0.05))
    return {"metric": "simulated robot-seconds per second, biped: kinodynamics MPC with 6-D feet (100 Hz, 1 iteration) + KinodynamicsID with flat feet "
                      "(1 kHz) + forward-dynamics simulator with 6-D contacts",
            "value": batch * 0.01 / dt, "unit": "robot-seconds/s", "ms_per_mpc_period": dt * 1e3, "batch": batch, "robots_upright": ok,
            "note": "states, targets and torques resident in HBM; the simulator is a second handle (host-side waits between its stream and the controller's)"}


def centroidal_line(batch, iters, steps, warmup, device_id, with_cpu=True):
    """BASELINE config "Go2 centroidal (9-dim state), H=50, batch=4096": same step definition on the centroidal OCP
    (round 5: a pipeline of kernels per ProxDDP iteration, smpc_cent_split.h).  Measured states: x_ref (+) N(0, sigma^2), resident in HBM, re-drawn on the
    device every step (the centroidal solution has no multibody state to feed back)."""
    import numpy as np
    import torch
    from simple_mpc import presets as P

    gm, mh = make_mpc("centroidal", batch, iters, device_id)
    dev = torch.device("cuda", device_id)
    X0 = torch.from_numpy(P.random_states(mh, batch)).to(dev)
    X = X0.clone()
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)

    on_stream = loop_stream(gm, dev)

    def step():
        gm.iterate_device(X.data_ptr())
        if C.SYNC_STEPS:
            gm.wait()
        with on_stream():
            X.copy_(X0)
            X[:, :3].add_(torch.randn((batch, 3), generator=gen, device=dev, dtype=torch.float64) * 1e-3)
        step_sync(gm)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-kernel durations: a short profiled loop of the same closed loop AFTER the timed one (HIP events around every launch; the engine
    # runs the batch as ONE part while profiling -- the timed loop above overlaps the launches of two parts on two streams)
    gm.set_profiling(True)
    gm.reset_kernel_times()
    for _ in range(min(steps, 10)):
        step()
    torch.cuda.synchronize()
    kt = gm.kernel_times()
    gm.set_profiling(False)
    ms = {k: v[0] / max(v[1], 1) for k, v in kt.items() if k != "-"}
    per_step = {k: v[0] / min(steps, 10) for k, v in kt.items() if k != "-"}  # ms of a control step spent in each kernel
    fused = ms.get("backward", 0.0) == 0.0  # (SMPC_CENT_FUSED=1: the one-kernel control step)
    BH = batch * gm.H
    fr = f_ric(9, gm.nu, gm.nc)
    io_bytes = batch * 8 * (2 * (gm.H + 1) * 9 + 2 * gm.H * gm.nu + gm.nu * 9 + (gm.H + 1) * 9 + gm.H * gm.nc)  # compulsory I/O per step
    if fused:
        avg = ms["step"] * 1e-3
        rl = both_bounds(batch * gm.H * iters * fr, io_bytes, avg, "hbm")
        tr, src = pmc_traffic("cent_step_body", batch == 4096 and iters == 3)
        rl.update({"kernel": "cent_step_body (whole control step: recede + %d ProxDDP iterations)" % iters, "traffic": tr, "traffic_source": src})
    else:
        # dominant kernel of the pipeline: the backward sweep (FP64: B*H*F_ric(9,12,8) per launch; bytes: the stage record read + gains written)
        REC, GAIN = 256 * 8, 174 * 8  # CentRec::STRIDE doubles read; [K k | P~ packed | p+] doubles written per stage ([Z z] only with active cone rows)
        rl = both_bounds(BH * fr, BH * (REC + GAIN), ms["backward"] * 1e-3, "mfma")
        tr, src = pmc_traffic("cent_bwd_body", batch == 4096 and iters == 3)
        rl.update({"kernel": "cent_bwd_body (proximal Riccati recursion of one ProxDDP iteration; %.0f %% of the step's kernel time)"
                   % (100.0 * per_step["backward"] / max(sum(per_step.values()), 1e-12)), "traffic": tr, "traffic_source": src})
        # the memory-bound kernels of the pipeline: algorithmic bytes per launch
        ITER = 8 * (9 * 2 + gm.nu + gm.nc * 2 + 9 * 3 + 3 * 4 + 6 + gm.nu + 3)  # iterate + references a stage evaluation reads
        STEPB = 8 * (9 + gm.nu + gm.nc + 9)                                      # dx, du, dnu, dlam of a stage
        other = {
            "pre": both_bounds(None, BH * (ITER + 256 * 8), ms["pre"] * 1e-3, "hbm"),
            "forward": both_bounds(None, BH * (192 * 8 + 128 * 8 + STEPB), ms["forward"] * 1e-3, "hbm"),  # gains quarters 0-2, record quarters 2-3, steps out
            "line_search": both_bounds(None, BH * (ITER + STEPB + 3 * STEPB), ms["line_search"] * 1e-3, "hbm"),  # evaluation inputs once + the accept axpy
        }
        for k in other:
            t2, _ = pmc_traffic("cent_%s_body" % {"pre": "pre", "forward": "fwd", "line_search": "ls"}[k], batch == 4096 and iters == 3)
            other[k]["traffic"] = t2
        rl["pipeline"] = other
        rl["step"] = both_bounds(batch * gm.H * iters * fr, io_bytes, dt / steps, "hbm")
        rl["step"]["note"] = "whole control step: B*H*k*F_ric(9,12,8) and the compulsory I/O of a step (SURVEY 8d, 24.7 KB per instance) over the measured step time"
    rl["note"] = "FP64 side: F_ric(9,12,8) of SURVEY 8d per (instance, stage); HBM side: algorithmic bytes of the launch"
    out = {
        "metric": "MPC control-steps/sec at fixed ProxDDP iters, Go2 centroidal H=50",
        "value": batch * steps / dt, "unit": "control-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": "f64",
        "config": {"workload": "Go2 centroidal (go2_like table), H=%d, %d ProxDDP iters/step, batch=%d, trot 10/30/10/30, "
                   "x_meas = x_ref (+) N(0, sigma^2)" % (gm.H, iters, batch), "finite": bool(np.all(np.isfinite(gm.info)))},
        "kernel_ms": {k: round(v, 4) for k, v in ms.items()},
        "roofline": rl,
    }
    if with_cpu:
        S, O = _oracle_imports()
        threads = O.use_effective_cpus()  # hardware threads capped by the cgroup CPU quota
        Bc = max(threads * 4, 16)
        om, rbc, _ = S.make_cent_oracle(Bc, max_iters=iters)
        om.generateCycleHorizon(O.trot_cycle())
        om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
        Xc = S.random_states(rbc, Bc)
        om.iterate(Xc)
        t0 = time.time()
        n = 0
        while time.time() - t0 < 5.0 and n < 200:
            om.iterate(Xc)
            n += 1
        out["cpu_baseline"] = {"value": Bc * n / (time.time() - t0), "unit": "control-steps/s", "cores": threads, "kind": "port",
                               "sample": "CPU restatement (oracle/, not Aligator): %d instances x %d steps, k=%d" % (Bc, n, iters)}
    return out


def fulldynamics_line(batch, iters, steps, warmup, device_id, with_cpu=True, robot="go2"):
    """Full-dynamics OCP: robot = "go2" (reference examples/go2_fulldynamics.py, 3-D contacts, H = 50) or "talos" (BASELINE
    configs[3]: examples/talos_fulldynamics.py, 6-D contacts + wrench cones, H = 100).  Same step definition and closed loop as
    the headline, joint torques as controls, dense A / B."""
    import numpy as np
    import torch
    from simple_mpc import presets as P

    talos = robot == "talos"
    gm, mh = make_mpc("talos" if talos else "fulldynamics", batch, iters, device_id, horizon=100 if talos else 50)
    dev = torch.device("cuda", device_id)
    X = torch.from_numpy(P.random_states(mh, batch, scale=0.7 if talos else 1.0)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)

    on_stream = loop_stream(gm, dev)

    def step():
        gm.iterate_device(X.data_ptr())
        gm.get_x_device(1, X.data_ptr())
        if C.SYNC_STEPS:
            gm.wait()
        with on_stream():
            X.add_(torch.randn(X.shape, generator=gen, device=dev, dtype=torch.float64) * 1e-3)
            q = X[:, 3:7]
            q.div_(q.norm(dim=1, keepdim=True))
        step_sync(gm)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # per-kernel durations: a short profiled loop of the same closed loop AFTER the timed one (HIP events around every launch; the engine runs
    # the batch as ONE part while profiling, whatever SMPC_FULL_PARTS says)
    gm.set_profiling(True)
    gm.reset_kernel_times()
    for _ in range(min(steps, 5)):
        step()
    torch.cuda.synchronize()
    kt = gm.kernel_times()
    gm.set_profiling(False)
    H, ndx, nu, nc = gm.H, gm.ndx, gm.nu, gm.nc
    name = "Talos" if talos else "Go2"
    out = {
        "metric": "MPC control-steps/sec at fixed ProxDDP iters, %s fulldynamics H=%d" % (name, H),
        "value": batch * steps / dt, "unit": "control-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": "f64",
        "config": {"workload": ("Talos full dynamics (talos_like table, nq 29 / nv 28, two 6-D feet, wrench cones), H=%d, %d ProxDDP iters/step, "
                                "batch=%d, walk 20/80/20/80, closed loop x_meas = xs[1] + N(0,1e-3^2)" if talos else
                                "Go2 full dynamics (go2_like table, 3-D contacts), H=%d, %d ProxDDP iters/step, batch=%d, trot 10/30/10/30, "
                                "closed loop x_meas = xs[1] + N(0,1e-3^2)") % (H, iters, batch), "finite": bool(np.all(np.isfinite(gm.info)))},
        "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items() if k != "-"},
    }
    if kt.get("riccati", (0, 0))[1] and kt.get("deriv", (0, 0))[1]:
        tag = "talos" if talos else "go2"
        at_record = batch == (1024 if talos else 4096) and iters == 3  # the configuration the committed PMC summary was taken on
        ncd = nc - 2 * nu  # dense (wrench-cone) rows
        lq_bytes = 8 * (2 * ndx * ndx + 2 * ndx * nu + nu * nu + ncd * (ndx + nu) + 2 * ndx + nu + 2 * nc)  # dense knot (box rows are selectors)
        avg = kt["riccati"][0] / kt["riccati"][1] * 1e-3
        ric = both_bounds(batch * H * f_ric(ndx, nu, nc), batch * H * lq_bytes, avg, "mfma")
        tr, src = pmc_traffic("riccati_dense_body_" + tag, at_record)
        ric.update({"kernel": "riccati_dense_body (proximal Riccati backward sweep, dense A / B)",
                    "note": "algorithmic FLOPs = B*H*F_ric(%d,%d,%d) per launch (SURVEY 8d)" % (ndx, nu, nc), "traffic": tr, "traffic_source": src})
        # the stage kernel (evaluation + derivatives of the constrained dynamics + Gauss-Newton assembly): FP64 side = the FLOPs the
        # instrumented oracle counts per stage (profiles/flop_counts.json), HBM side = the knot it writes
        fl = flop_counts().get("fulldynamics_" + tag, {}).get("deriv_flops_per_stage")
        avgd = kt["deriv"][0] / kt["deriv"][1] * 1e-3
        der = both_bounds(None if fl is None else batch * (H + 1) * fl, batch * H * lq_bytes, avgd, "mfma" if fl is not None else "hbm")
        tr, src = pmc_traffic("fdyn_deriv_body_" + tag, at_record)
        der.update({"kernel": "fdyn_deriv_body (constrained dynamics, derivatives, Gauss-Newton knot)",
                    "note": "FP64 bound: algorithmic FLOPs per stage counted by instrumentation in the oracle (profiles/flop_counts.json)"
                            + ("; `traffic` above the algorithmic bytes (the knot) is the per-block device slice of the derivative blocks R1 / JT "
                               "(24.4 KB per block, written and re-read three times by the solve chain: DESIGN 3.9a) -- scratch that buys the third "
                               "resident block per CU, not re-reads of the inputs" if talos else ""),
                    "traffic": tr, "traffic_source": src})
        dom_deriv = kt["deriv"][0] >= kt["riccati"][0]  # the dominant kernel carries the line's roofline
        out["roofline"] = der if dom_deriv else ric
        out["roofline_other"] = {"riccati": ric} if dom_deriv else {"deriv": der}
    if with_cpu:
        S, O = _oracle_imports()
        threads = O.use_effective_cpus()  # hardware threads capped by the cgroup CPU quota
        Bc = max(threads, 8)
        if talos:
            om, _, rbc = None, None, None
            rbc = O.Robot("talos_like")
            ms = O.talos_mpc_settings(rbc, max_iters=iters)
            om = O.OracleFullMPC(O.Full(rbc, O.talos_full_settings(rbc)), ms, Bc)
            om.generateCycleHorizon(O.walk_cycle())
            om.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
            Xc = S.talos_random_states(rbc, Bc, scale=0.7)
        else:
            om, rbc = S.make_full_oracle(Bc, max_iters=iters)
            Xc = S.random_states(rbc, Bc)
        om.iterate(Xc)
        t0 = time.time()
        n = 0
        while time.time() - t0 < 10.0 and n < 50:
            om.iterate(Xc)
            Xc = om.xs[:, 1, :].copy()
            n += 1
        out["cpu_baseline"] = {"value": Bc * n / (time.time() - t0), "unit": "control-steps/s", "cores": threads, "kind": "port",
                               "sample": "CPU restatement (oracle/, not Aligator): %d instances x %d steps, k=%d" % (Bc, n, iters)}
    return out


def talos_flat_feet_line(kind, batch, iters, steps, warmup, device_id, with_cpu=True):
    """The kinodynamics / centroidal OCPs of a Talos-class biped with 6-D feet (reference examples/talos_kinodynamics.py,
    talos_centroidal.py; H = 100, walk 20/80/20/80): the same control step as the other lines, brief."""
    import numpy as np
    import torch
    from simple_mpc import presets as P

    gm, mh = make_mpc(kind, batch, iters, device_id, horizon=100)
    dev = torch.device("cuda", device_id)
    X0 = torch.from_numpy(P.random_states(mh, batch, scale=0.7)).to(dev)
    X = X0.clone()
    gen = torch.Generator(device=dev)
    gen.manual_seed(13)
    on_stream = loop_stream(gm, dev)
    cent = kind == "talos_centroidal"

    def step():
        gm.iterate_device(X.data_ptr())
        if not cent:
            gm.get_x_device(1, X.data_ptr())
        if C.SYNC_STEPS:
            gm.wait()
        with on_stream():
            if cent:  # (no multibody state to feed back: x_ref (+) noise on the base position, as the Go2 centroidal line)
                X.copy_(X0)
                X[:, :3].add_(torch.randn((batch, 3), generator=gen, device=dev, dtype=torch.float64) * 1e-3)
            else:
                X.add_(torch.randn(X.shape, generator=gen, device=dev, dtype=torch.float64) * 1e-3)
                q = X[:, 3:7]
                q.div_(q.norm(dim=1, keepdim=True))
        step_sync(gm)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # (per-kernel durations from a short profiled loop after the timed one: the kinodynamics engine runs the batch as one part while profiling)
    gm.set_profiling(True)
    gm.reset_kernel_times()
    for _ in range(min(steps, 5)):
        step()
    torch.cuda.synchronize()
    kt = gm.kernel_times()
    gm.set_profiling(False)
    out = {
        "metric": "MPC control-steps/sec at fixed ProxDDP iters, Talos %s (6-D feet) H=%d" % (kind.split("_")[1], gm.H),
        "value": batch * steps / dt, "unit": "control-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": "f64",
        "config": {"workload": "Talos %s OCP (talos_like table, two 6-D feet, wrench cones), H=%d, %d ProxDDP iters/step, batch=%d, walk 20/80/20/80"
                   % (kind.split("_")[1], gm.H, iters, batch), "sizes": {"ndx": gm.ndx, "nu": gm.nu, "nc": gm.nc},
                   "finite": bool(np.all(np.isfinite(gm.info)))},
        "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items() if k != "-" and v[1]},
    }
    # rooflines: the dense Riccati sweep on the FP64 side (F_ric of SURVEY 8d with the rows the sweep pivots explicitly: the 2 x 17 wrench-cone
    # rows), the stage kernel on the HBM side (the knot it writes); the dominant one carries the line's `roofline`
    H, ndx, nu, nc = gm.H, gm.ndx, gm.nu, gm.nc
    ncd = 34
    if kt.get("riccati", (0, 0))[1] and kt.get("deriv", (0, 0))[1]:
        lq_bytes = 8 * (2 * ndx * ndx + 2 * ndx * nu + nu * nu + ncd * (ndx + nu) + 2 * ndx + nu + 2 * nc)
        tag = "cent6" if cent else "taloskino"
        at_record = batch == 1024 and iters == 3
        ric = both_bounds(batch * H * f_ric(ndx, nu, ncd), batch * H * lq_bytes, kt["riccati"][0] / kt["riccati"][1] * 1e-3, "mfma")
        tr, src = pmc_traffic("riccati_dense_body_" + tag, at_record)
        ric.update({"kernel": "riccati_dense_body (dense proximal Riccati sweep, %d states / %d controls / %d explicit multiplier rows)" % (ndx, nu, ncd),
                    "note": "algorithmic FLOPs = B*H*F_ric(%d,%d,%d) per launch (SURVEY 8d); stages without an active cone row run the light grid" % (ndx, nu, ncd),
                    "traffic": tr, "traffic_source": src})
        der = both_bounds(None, batch * H * lq_bytes, kt["deriv"][0] / kt["deriv"][1] * 1e-3, "hbm")
        tr, src = pmc_traffic(("cent6_deriv_body" if cent else "fdyn_deriv_body_taloskino"), at_record)
        der.update({"kernel": ("cent6_deriv_body" if cent else "fdyn_deriv_body<kinodynamics variant>") + " (stage evaluation, derivatives, Gauss-Newton knot)",
                    "note": "HBM side: the dense knot the stage kernel writes (the oracle's FLOP count exists for the full-dynamics stage only)"
                            + ("" if cent else "; `traffic` above it is the per-block device slice of the derivative blocks R1 / JT / Cv (DESIGN 3.9a)"),
                    "traffic": tr, "traffic_source": src})
        dom_deriv = kt["deriv"][0] >= kt["riccati"][0]
        out["roofline"] = der if dom_deriv else ric
        out["roofline_other"] = {"riccati": ric} if dom_deriv else {"deriv": der}
    if with_cpu:
        S, O = _oracle_imports()
        threads = O.use_effective_cpus()
        Bc = max(threads, 8)
        rbc = O.Robot("talos_like")
        ms = O.talos_mpc_settings(rbc, max_iters=iters)
        om = O.OracleCentMPC(O.Cent(rbc, O.talos_centroidal_settings(rbc)), ms, Bc) if cent else O.OracleMPC(O.Kino(rbc, O.talos_kino_settings(rbc)), ms, Bc)
        om.generateCycleHorizon(O.walk_cycle())
        om.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        Xc = S.talos_random_states(rbc, Bc, scale=0.7)
        om.iterate(Xc)
        t0 = time.time()
        n = 0
        while time.time() - t0 < 8.0 and n < 50:
            om.iterate(Xc)
            if not cent:
                Xc = om.xs[:, 1, :].copy()
            n += 1
        out["cpu_baseline"] = {"value": Bc * n / (time.time() - t0), "unit": "control-steps/s", "cores": threads, "kind": "port",
                               "sample": "CPU restatement (oracle/, not Aligator): %d instances x %d steps, k=%d" % (Bc, n, iters)}
    return out



def run_all(batch, iters, device_id, with_cpu=True, only=None):
    """Every workload beside the headline, at the sizes of record (BASELINE configs[1], configs[3] and the rows 8f)."""
    table = [
        ("centroidal", lambda: centroidal_line(batch, iters, 40, 5, device_id, with_cpu)),
        ("fulldynamics_go2", lambda: fulldynamics_line(min(batch, 4096), iters, 20, 3, device_id, with_cpu)),
        ("fulldynamics_talos", lambda: fulldynamics_line(1024, iters, 10, 2, device_id, with_cpu, robot="talos")),
        ("talos_kinodynamics_6d", lambda: talos_flat_feet_line("talos_kinodynamics", 1024, iters, 8, 2, device_id, with_cpu)),
        ("talos_centroidal_6d", lambda: talos_flat_feet_line("talos_centroidal", 1024, iters, 20, 3, device_id, with_cpu)),
        ("inverse_dynamics_qp", lambda: inverse_dynamics_line(batch, device_id, with_cpu)),
        ("inverse_dynamics_qp_flat_feet", lambda: inverse_dynamics_quad_line(batch, device_id, with_cpu)),
        ("control_stack", lambda: control_stack_line(batch, device_id)),
        ("control_stack_talos", lambda: control_stack_talos_line(1024, device_id)),
        ("single_robot_latency", lambda: single_robot_latency(iters, device_id)),
    ]
    return {name: fn() for name, fn in table if only is None or name in only}


def report(other, stream=None):
    """Full entries to bench_side.json (next to this file) and, one JSON line per workload, to `stream` (stderr by default)."""
    stream = stream or sys.stderr
    for name, e in other.items():
        stream.write(json.dumps({"side_workload": name, **e}) + "\n")
    stream.flush()
    try:
        with open(os.path.join(ROOT, "bench_side.json"), "w") as f:
            json.dump(other, f, indent=1)
    except OSError:
        pass


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated workload names")
    args = ap.parse_args()
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench_side.py needs a GPU: the engines have no CPU path")
    import __graft_entry__ as g

    g.build_hip()
    g.build_oracle()
    C.measure_fp64_peak()
    other = run_all(args.batch, args.iters, 0, not args.no_cpu_baseline, None if args.only is None else set(args.only.split(",")))
    report(other, sys.stdout)


if __name__ == "__main__":
    main()
