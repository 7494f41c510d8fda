#!/usr/bin/env python3
"""bench.py -- MPC control-steps/sec at fixed ProxDDP iterations, Go2 kinodynamics, H=50 (BASELINE.json).

A "step" is one batched MPC::iterate (reference src/mpc.cpp:189-218) of B instances per GPU with exactly
k ProxDDP iterations each; value = (instances x steps) / wall time, whole job.  Inputs (measured states)
are resident in HBM when the timed region starts: the closed loop feeds back xs[1] + N(0, 1e-3^2) noise
generated on the device.  Multi-GPU: one process per GPU, batch sharded by instance, no
collective on the solve path (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-per-gpu B] [--iters 3]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts N fresh ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`), waits and relays rank 0's JSON line.  Under a launcher
(WORLD_SIZE set) it is one rank.  `--dry-run` exercises the same launch / shard / barrier / reduce path on CPU (gloo, the
test-only emulation build of the kernel bodies, a tiny batch): launch-path test infrastructure, never a measurement.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
# (tests/ holds the oracle binding: it goes on the path only inside the cpu_baseline legs)

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix dense peak (public spec)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6300 GB/s achievable)
MEASURED_PEAK = None     # {"fma_tflops", "mfma_tflops", ...} of THIS device, from tools/micro/fp64_peak.bin ("quote the measured peak")


def loop_stream(gm, dev):
    """torch work of a closed loop on the handle's own stream (smpc_get_stream): one in-order queue, no host-side wait between control steps.
    --sync-steps (SYNC_STEPS) restores a host synchronisation after every step."""
    import contextlib

    import torch

    ptr = 0 if SYNC_STEPS else gm.stream()
    if not ptr:
        return contextlib.nullcontext
    ext = torch.cuda.ExternalStream(ptr, device=dev)
    return lambda: torch.cuda.stream(ext)


def step_sync(gm):
    if SYNC_STEPS:
        import torch

        gm.wait()
        torch.cuda.synchronize()


SYNC_STEPS = False


def measure_fp64_peak(seconds=0.5):
    """Dependency-free v_fma_f64 / v_mfma_f64_16x16x4 loops on all CUs (tools/micro/fp64_peak.hip, built by __graft_entry__.build):
    the peak this device actually reaches, carried beside the spec value in every FP64 roofline entry."""
    global MEASURED_PEAK
    exe = os.path.join(ROOT, "tools", "micro", "fp64_peak.bin")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, str(seconds)], stdout=subprocess.PIPE, text=True, timeout=60).stdout.strip().splitlines()[-1]
        MEASURED_PEAK = json.loads(out)
    except Exception:  # (a diagnostic: the bench line stands without it)
        MEASURED_PEAK = None
    return MEASURED_PEAK


def f_ric(ndx, nu, nc):
    """Algorithmic FLOPs of the proximal Riccati backward+forward per (instance, stage, iteration): SURVEY 8(d)."""
    return (4 * ndx**3 + 4 * ndx**2 * nu + 2 * ndx * nu**2 + (nu + nc) ** 3 / 3 + 2 * (nu + nc) ** 2 * (ndx + 1)
            + 2 * ndx**2 * (nu + nc) + 2 * ndx * (nu + nc))


def flop_counts():
    """Algorithmic FLOPs of the stage evaluation / derivative passes, counted by instrumentation in the oracle
    (tools/count_flops.py -> profiles/flop_counts.json).  {} if the file is absent."""
    p = os.path.join(ROOT, "profiles", "flop_counts.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


def pmc_traffic(kernel, want):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary of this command (PMC counters cannot
    be collected from inside this process); None unless the summary was taken on the configuration `want` describes."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")))
    if not files or not want:
        return None, None
    with open(files[-1]) as f:
        doc = json.load(f)
    src = os.path.basename(files[-1]) + ("@" + doc["git_commit"] if doc.get("git_commit") else "")  # (the tree the profile was taken on)
    return doc.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch_corrected"), src


def both_bounds(flops, bytes_, avg_s, primary):
    """Roofline entry with BOTH fractions: `achieved/peak/unit/frac` are those of the primary bound."""
    fp = None if flops is None else {"achieved": flops / avg_s / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / FP64_PEAK_TFLOPS,
                                     "algorithmic_flops": flops}
    if fp is not None and MEASURED_PEAK:
        pm = max(MEASURED_PEAK.get("fma_tflops", 0.0), MEASURED_PEAK.get("mfma_tflops", 0.0))
        fp.update({"peak_measured": pm, "frac_of_measured": fp["achieved"] / pm, "peak_measured_detail": MEASURED_PEAK})
    hb = None if bytes_ is None else {"achieved": bytes_ / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_ / avg_s / 1e9 / HBM_PEAK_GBS,
                                      "algorithmic_bytes": bytes_}
    pr = fp if (primary == "mfma" and fp is not None) else hb
    out = {"bound": "mfma" if pr is fp else "hbm", "achieved": pr["achieved"], "peak": pr["peak"], "unit": pr["unit"], "frac": pr["frac"],
           "avg_launch_ms": avg_s * 1e3, "fp64": fp, "hbm": hb}
    if pr is fp and "peak_measured" in fp:
        out["peak_measured"] = fp["peak_measured"]
    return out


def rooflines(kt, B, H, ndx, nu, nc, nx, at_record_size):
    """Roofline entries of the kernels that carry the kinodynamics step: average launch duration from the HIP
    events the engine records on ITS stream around every launch inside the timed region."""
    out = {}
    fc = flop_counts().get("kinodynamics", {})
    nfc = 12                 # 3 * nf force components (Go2: 4 point feet)
    na = nu - nfc            # actuated joints = box rows
    tl = [min(16, ndx - 16 * i) for i in range((ndx + 15) // 16)]
    q_upper = sum(tl[i] * tl[j] for i in range(len(tl)) for j in range(i, len(tl)))  # upper 16x16 tiles of Q (912 of 1296)
    # state-dependent part of the knot written per iteration + the iterate read
    knot_w = 8 * (q_upper + ndx * nfc + nfc * nfc + na + (nc - na) * ndx + na + 12 * (ndx + nu) + 4 * ndx + 2 * nu + 2 * nc)
    iter_r = 8 * (nx + nu + 2 * nc + 4 * ndx)
    if "deriv" in kt and kt["deriv"][1]:
        # the derivative pass of a launch = lane_tree_body (lane-per-problem evaluation, hand-over) + deriv2_body (wavefront per problem);
        # SMPC_LANE_DERIV=0: deriv_body alone ("tree" then only counts the line-search launches)
        two = kt.get("tree", (0.0, 0))[1] > kt.get("trial", (0.0, 0))[1]
        avg2 = kt["deriv"][0] / kt["deriv"][1] * 1e-3
        avgt = kt["tree"][0] / kt["tree"][1] * 1e-3 if two else 0.0
        avg = avg2 + avgt
        fl = fc.get("deriv_flops_per_stage")
        ho = 8 * 688 if two else 0  # hand-over stream written by the tree pass and read by the derivative kernel (EvStream::STRIDE on Go2: 64 + 12 * 32 + 4 * 16 + 176)
        e = both_bounds(None if fl is None else B * H * fl, B * H * (knot_w + iter_r + 2 * ho), avg, "mfma")
        kname = "deriv2_body" if two else "deriv_body"
        tr, src = pmc_traffic(kname, at_record_size)
        tr2, _ = pmc_traffic("lane_tree_body", at_record_size) if two else (0.0, None)
        e.update({"kernel": ("derivative pass = lane_tree_body + deriv2_body" if two else "deriv_body") + " (stage evaluation + derivatives + LQ knot)",
                  "avg_launch_ms_parts": {"lane_tree_body": avgt * 1e3, kname: avg2 * 1e3},
                  "traffic": None if tr is None else tr + (tr2 or 0.0), "traffic_source": src,
                  "note": "FP64 bound: algorithmic FLOPs of one stage evaluation + derivative + Gauss-Newton assembly counted by "
                          "instrumentation in the oracle (profiles/flop_counts.json) x B*H; HBM side: B*H*%d bytes per launch" % (knot_w + iter_r + 2 * ho)})
        out["deriv"] = e
    if "riccati" in kt and kt["riccati"][1]:
        avg = kt["riccati"][0] / kt["riccati"][1] * 1e-3
        # bytes: the knot read (full A B Q S R C + vectors as the structured sweep reads them) + gains written
        gains_w = 8 * (nu * (ndx + 1) + ndx * (ndx + 1) // 2 + ndx)
        e = both_bounds(B * H * f_ric(ndx, nu, nc), B * H * (knot_w + gains_w), avg, "mfma")
        tr, src = pmc_traffic("riccati_kino_body", at_record_size)
        e.update({"kernel": "riccati_kino_body (proximal Riccati backward sweep)", "traffic": tr, "traffic_source": src,
                  "note": "FP64 dense peak (vector = matrix on MI355X); algorithmic FLOPs = B*H*F_ric(36,24,24) of the "
                          "unstructured recursion (SURVEY 8d) per launch"})
        out["riccati"] = e
    if "forward" in kt and kt["forward"][1]:
        avg = kt["forward"][0] / kt["forward"][1] * 1e-3
        rd = 8 * (nu * (ndx + 1) + ndx * (ndx + 1) // 2 + ndx + 12 * (ndx + nu) + (nc - na) * ndx + 6 * ndx + 2 * nu + 2 * nc)
        wr = 8 * (2 * ndx + nu + nc)
        fl = 2 * (nu * ndx + nc * ndx + 12 * (ndx + nu) + ndx * ndx)
        e = both_bounds(B * H * fl, B * H * (rd + wr), avg, "hbm")
        tr, src = pmc_traffic("forward_kino_body", at_record_size)
        e.update({"kernel": "forward_kino_body (gains -> Newton step)", "traffic": tr, "traffic_source": src})
        out["forward"] = e
    return out


def _oracle_imports():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mpc_setup as S
    import oracle_lib as O
    return S, O


def cpu_baseline(iters, seconds_budget=14.0):
    """Oracle (CPU restatement, not Aligator) on the host cores, bounded sample of the same workload; plus the single-thread
    latency of one control step at B = 1."""
    import numpy as np
    S, O = _oracle_imports()

    threads = O.use_effective_cpus()  # hardware threads capped by the cgroup CPU quota
    B = max(threads * 4, 16)
    om, rb, _ = S.make_oracle(B, max_iters=iters)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, B)
    om.iterate(X)  # warm-up
    X = om.xs[:, 1, :].copy()
    t0 = time.time()
    n = 0
    while True:
        om.iterate(X)
        X = om.xs[:, 1, :].copy()
        n += 1
        if time.time() - t0 > seconds_budget or (n >= 400 and time.time() - t0 > 12.0):
            break
    dt = time.time() - t0
    # B = 1: one instance = one OpenMP work item = one thread
    o1, rb1, _ = S.make_oracle(1, max_iters=iters)
    o1.generateCycleHorizon(O.trot_cycle())
    o1.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X1 = S.random_states(rb1, 1)
    o1.iterate(X1)
    lat = []
    for _ in range(5):
        X1 = o1.xs[:, 1, :].copy()
        t1 = time.time()
        o1.iterate(X1)
        lat.append(time.time() - t1)
    # BASELINE configs[0] ("Go2 kinodynamics, H=50, 1 ProxDDP iter, batch=1 -- CPU reference, plumbing"): the reference's own operating
    # point, one robot, one iteration per control step, one core
    c1, rbc1, _ = S.make_oracle(1, max_iters=1)
    c1.generateCycleHorizon(O.trot_cycle())
    c1.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    Xc1 = S.random_states(rbc1, 1)
    c1.iterate(Xc1)
    lat1 = []
    for _ in range(20):
        Xc1 = c1.xs[:, 1, :].copy()
        t1 = time.time()
        c1.iterate(Xc1)
        lat1.append(time.time() - t1)
    return {
        "value": B * n / dt,
        "unit": "control-steps/s",
        "cores": threads,
        "cfg1_k1_b1": {"ms_per_step": 1e3 * float(np.median(lat1)), "control_steps_per_s": 1.0 / float(np.median(lat1)),
                       "note": "BASELINE configs[0]: Go2 kinodynamics, H=50, 1 ProxDDP iteration, batch 1, one CPU thread (median of 20 steps)"},
        "seconds": dt,
        "host_hw_threads": os.cpu_count(),
        "kind": "port",
        "b1_latency_ms": 1e3 * min(lat),
        "sample": "CPU restatement (oracle/, not Aligator): %d instances x %d steps, k=%d, OpenMP over instances; b1_latency_ms = one "
                  "control step of one instance on one thread (best of 5)" % (B, n, iters),
    }


def make_mpc(kind, batch, iters, device_id, lib=None, horizon=50):
    """BatchedMPC on the settings of record (simple_mpc.presets): kind in kinodynamics / centroidal / fulldynamics / talos."""
    import numpy as np
    import simple_mpc
    from simple_mpc import presets as P

    if kind.startswith("talos"):
        mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("talos_like", lib), "half_sitting", "root_joint")
        for n in P.TALOS_FEET:
            mh.addQuadFoot(n, "root_joint", P.TALOS_QUAD)
        if kind == "talos_kinodynamics":  # (6-D feet in the kinodynamics / centroidal OCPs: round 4)
            ocp = simple_mpc.KinodynamicsOCP(P.talos_kino_settings(mh), mh)
            ocp.createProblem(mh.getReferenceState(), horizon, 6, -9.81, False)
        elif kind == "talos_centroidal":
            ocp = simple_mpc.CentroidalOCP(P.talos_centroidal_settings(mh), mh)
            ocp.createProblem(np.zeros(9), horizon, 6, -9.81, False)
        else:
            ocp = simple_mpc.FullDynamicsOCP(P.talos_full_settings(mh), mh)
            ocp.createProblem(mh.getReferenceState(), horizon, 6, -9.81, False)
        ms = P.talos_mpc_settings(mh, max_iters=iters)
        gm = simple_mpc.BatchedMPC({k: ms[k] for k in P.MPC_KEYS}, ocp, batch, device_id=device_id, lib=lib)
        gm.generateCycleHorizon(P.walk_cycle())
        gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        return gm, mh
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in P.GO2_FEET:
        mh.addPointFoot(n, "root_joint")
    if kind == "kinodynamics":
        ocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)
        ocp.createProblem(mh.getReferenceState(), horizon, 3, -9.81, False)
    elif kind == "centroidal":
        ocp = simple_mpc.CentroidalOCP(P.go2_centroidal_settings(mh), mh)
        ocp.createProblem(np.zeros(9), horizon, 3, -9.81, False)
    else:
        ocp = simple_mpc.FullDynamicsOCP(P.go2_full_settings(mh), mh)
        ocp.createProblem(mh.getReferenceState(), horizon, 3, -9.81, False)
    ms = P.go2_mpc_settings(mh, max_iters=iters)
    gm = simple_mpc.BatchedMPC({k: ms[k] for k in P.MPC_KEYS}, ocp, batch, device_id=device_id, lib=lib)
    gm.generateCycleHorizon(P.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    return gm, mh


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
        if SYNC_STEPS:
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
        if SYNC_STEPS:
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
        if SYNC_STEPS:
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


def launch_ranks(args, argv):
    """Parent of a multi-GPU run: starts N ranks with torch.distributed.run as a CHILD process (this process has not touched a
    GPU and never re-execs), relays rank 0's JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        env.setdefault("OMP_NUM_THREADS", "2")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        raise SystemExit("bench.py: the %d-rank run failed (exit code %d)" % (args.gpus, p.returncode))
    out = json.loads(line)
    assert out["n_gpus"] == args.gpus, (out["n_gpus"], args.gpus)
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", "--batch", dest="batch", type=int, default=None,
                    help="instances per GPU (default 4096; 8192 at 8 GPUs = BASELINE's 65536-instance configuration)")
    ap.add_argument("--iters", type=int, default=3, help="ProxDDP iterations per control step")
    ap.add_argument("--streams", type=int, default=1,
                    help="kinodynamics: run the iterations of N parts of the batch on N HIP streams (SMPC_STREAMS; tails of one part's "
                    "launches are filled by the next part's). Per-kernel rooflines then refer to launches of batch/N instances")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the other single-GPU workloads measured beside the headline (profiling passes "
                    "whose per-kernel means must be those of the headline launches)")
    ap.add_argument("--sync-steps", action="store_true", help="host synchronisation after every control step (default: the closed loop is one in-order queue "
                    "on the handle's stream, synchronised only at the ends of the timed region)")
    ap.add_argument("--no-gather", action="store_true", help="leave the return set [x1 | u0 | K0] of every control step on the devices (default at N > 1: "
                    "every step packs it on the device and moves it, overlapped with the next step, into ONE pinned host buffer on rank 0, inside the timed "
                    "region; at N = 1 there is no exchange between devices and the device-to-host copy is PCIe traffic, which `value` never includes: the "
                    "same loop is then measured beside the timed region and reported in `gather`)")
    ap.add_argument("--gather", action="store_true", help="N = 1: put the return-set copy inside the timed region as well")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launch path (gloo + emulated kernel bodies): not a measurement")
    ap.add_argument("--workload", default="kinodynamics", choices=["kinodynamics", "centroidal", "fulldynamics", "talos"],
                    help="kinodynamics = the headline metric (with the other single-GPU configurations measured briefly beside it at 1 GPU)")
    args = ap.parse_args()
    global SYNC_STEPS
    SYNC_STEPS = bool(args.sync_steps)
    if args.streams > 1:
        os.environ["SMPC_STREAMS"] = str(args.streams)  # read by the engine when the handle is created
    if args.batch is None:
        args.batch = 2 if args.dry_run else (1024 if args.workload == "talos" else (8192 if args.gpus >= 8 else 4096))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])
        return

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    dry = args.dry_run
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the MPC engine has no CPU path")
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g

    if rank == 0:
        # rank 0 (re)builds what is stale, the others wait: no concurrent compiler / make runs on the shared tree
        if dry:
            g.build_emu()
        else:
            g.build_hip()
            g.build_oracle()
    if dist is not None:
        dist.barrier()
    from simple_mpc import presets as P

    B = args.batch
    sync = (lambda: None) if dry else torch.cuda.synchronize
    if args.workload != "kinodynamics":
        if world > 1 or dry:
            raise SystemExit("--workload %s is a single-GPU line" % args.workload)
        measure_fp64_peak()
        if args.workload == "centroidal":
            line = centroidal_line(B, args.iters, args.steps, args.warmup, local_rank, not args.no_cpu_baseline)
        else:
            line = fulldynamics_line(B, args.iters, args.steps, args.warmup, local_rank, not args.no_cpu_baseline,
                                     robot="talos" if args.workload == "talos" else "go2")
        line.update({"n_gpus": 1, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
        print(json.dumps(line))
        return
    lib = None
    if dry:
        from simple_mpc._capi import SmpcLib

        lib = SmpcLib(g.EMU_LIB)
    if rank == 0 and not dry:
        measure_fp64_peak()
    gm, mh = make_mpc("kinodynamics", B, args.iters, local_rank, lib)

    X0 = P.random_states(mh, B, seed=20240529 + rank)  # contiguous block of the global batch: rank r owns instances [r B, (r+1) B)
    if dry:
        rng = np.random.default_rng(20240529 + rank)
        Xh = X0.copy()

        def step():
            gm.iterate(Xh)
            Xh[:] = gm.xs[:, 1, :] + rng.normal(0.0, 1e-3, Xh.shape)
            Xh[:, 3:7] /= np.linalg.norm(Xh[:, 3:7], axis=1, keepdims=True)
    else:
        dev = torch.device("cuda", local_rank)
        X = torch.from_numpy(X0).to(dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(20240529 + rank)

        on_stream = loop_stream(gm, dev)

        def step():
            gm.iterate_device(X.data_ptr())
            gm.get_x_device(1, X.data_ptr())  # x_meas <- xs[1] (same stream, ordered after the solve)
            if SYNC_STEPS:
                gm.wait()
            with on_stream():  # the noise kernels queue behind the solve on the handle's stream: no host-side wait inside the timed region
                noise = torch.randn(X.shape, generator=gen, device=dev, dtype=torch.float64) * 1e-3
                X.add_(noise)
                q = X[:, 3:7]
                q.div_(q.norm(dim=1, keepdim=True))
            step_sync(gm)

    # ---- return set of every control step -> one pinned host buffer on rank 0, inside the timed region ----
    # What leaves the devices per step is what the controllers consume: rows [x1 | u0 | K0] (nx + nu + nu ndx doubles per instance).  Each rank
    # packs its rows on the device (one kernel on the solve stream), then a SIDE stream moves them while the next control step runs: ranks > 0
    # send theirs to rank 0 (torch.distributed gather = RCCL send / recv over xGMI -- the one exchange this path has, off the solve path), rank
    # 0 copies the whole [N B][row] block into pinned memory.  Two slots: the solve stream waits for the move of step k - 2 before it repacks.
    row = gm.nx + gm.nu + gm.nu * gm.ndx
    gather_on = not args.no_gather and args.streams == 1
    gather_timed = gather_on and (world > 1 or args.gather or dry)
    g_events = []
    if gather_on and not dry:
        main_s = torch.cuda.ExternalStream(gm.stream(), device=dev)
        side_s = torch.cuda.Stream(device=dev)
        pack = [torch.empty((B, row), dtype=torch.float64, device=dev) for _ in range(2)]
        ev_packed = [torch.cuda.Event() for _ in range(2)]
        ev_moved = [torch.cuda.Event() for _ in range(2)]
        if rank == 0:
            recv = [torch.empty((world, B, row), dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
            pinned = torch.empty((2, world * B, row), dtype=torch.float64).pin_memory()
    elif gather_on:
        pack_h = np.zeros((B, row))
        gathered_h = [torch.zeros((B, row), dtype=torch.float64) for _ in range(world)] if rank == 0 else None
    step_no = [0]

    def emit(timed):
        if not gather_on:
            return
        if dry:  # rehearsal of the same exchange on CPU tensors (gloo), no streams
            gm.gather_outputs_device(pack_h.ctypes.data)
            gm.wait()
            if dist is not None:
                dist.gather(torch.from_numpy(pack_h), gather_list=gathered_h, dst=0)
            return
        slot = step_no[0] & 1
        step_no[0] += 1
        main_s.wait_event(ev_moved[slot])
        gm.gather_outputs_device(pack[slot].data_ptr())
        ev_packed[slot].record(main_s)
        side_s.wait_event(ev_packed[slot])
        with torch.cuda.stream(side_s):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side_s)
            if world > 1:
                dist.gather(pack[slot], gather_list=[recv[slot][r] for r in range(world)] if rank == 0 else None, dst=0)
            if rank == 0:
                pinned[slot].copy_(recv[slot].view(world * B, row) if world > 1 else pack[slot], non_blocking=True)
            if timed:
                e1.record(side_s)
                g_events.append((e0, e1))
            ev_moved[slot].record(side_s)

    gather_error = None
    for _ in range(args.warmup):
        step()
        if gather_timed:
            try:
                emit(False)
            except Exception as exc:  # (a collective this build / node cannot run fails on every rank alike, at the first warm-up step)
                if world > 1:
                    # N > 1: the return-set exchange is part of the measured step ("xGMI used only to gather outputs").  A run that
                    # cannot execute it must not report a throughput without it: fail, loudly, on every rank.
                    sys.stderr.write("bench.py: rank %d: the return-set exchange failed at the first warm-up step: %r\n" % (rank, exc))
                    sys.stderr.flush()
                    if dist is not None:
                        try:
                            dist.destroy_process_group()
                        except Exception:
                            pass
                    sys.exit(3)
                gather_error = repr(exc)
                gather_on = gather_timed = False
    profile = not args.no_profile and not dry
    if profile:
        gm.set_profiling(True)
        gm.reset_kernel_times()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if gather_timed:
            emit(True)
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    kt = gm.kernel_times() if profile else {}
    dt_with_gather = None
    if gather_on and not gather_timed:
        # N = 1: the same closed loop with the return set moved to the pinned buffer every step, measured beside the timed region
        if profile:
            gm.set_profiling(False)
        for _ in range(3):
            step()
            emit(False)
        sync()
        tg = time.perf_counter()
        ng = max(10, min(40, args.steps))
        for _ in range(ng):
            step()
            emit(True)
        sync()
        dt_with_gather = (time.perf_counter() - tg) / ng
    gather_ms = sum(a.elapsed_time(b) for a, b in g_events) / max(1, len(g_events)) if g_events else None
    if gather_on and not dry and rank == 0:
        # the last step's rows are in the pinned buffer, every rank's block in place
        last = pinned[(step_no[0] - 1) & 1]
        gather_ok = bool(torch.isfinite(last).all()) and bool((last.view(world, B, row)[:, :, 3:7].norm(dim=2) - 1).abs().max() < 1e-9)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    info = gm.info
    ok = bool(np.all(np.isfinite(info)))

    if rank == 0:
        value = world * B * args.steps / dt
        H, ndx, nu, nc = gm.H, gm.ndx, gm.nu, gm.nc
        out = {
            "metric": "MPC control-steps/sec at fixed ProxDDP iters, Go2 kinodyn H=50",
            "value": value,
            "unit": "control-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "Go2 kinodynamics (go2_like table), H=%d, %d ProxDDP iters/step, batch=%d per GPU, trot 10/30/10/30, "
                "closed loop x_meas = xs[1] + N(0,1e-3^2)" % (H, args.iters, B),
                "batch_per_gpu": B,
                "global_batch": world * B,
                "parallelism": "instance-sharded x%d, no collective on the solve path" % world,
                "streams": args.streams,
                "finite": ok,
            },
        }
        if gather_error is not None:
            out["gather"] = {"in_timed_region": False, "error": gather_error}
        if gather_on:
            out["gather"] = {"what": "rows [x1 | u0 | K0] of every instance, every control step, into one pinned host buffer on rank 0 (on a side stream: "
                             "overlaps the next step)", "in_timed_region": bool(gather_timed), "row_doubles": row, "bytes_per_step": world * B * row * 8,
                             "mode": "rehearsal (gloo, host tensors)" if dry else ("device pack + RCCL gather to rank 0 + D2H" if world > 1 else "device pack + D2H")}
            if gather_ms is not None:
                step_ms = 1e3 * (dt_with_gather if dt_with_gather is not None else dt / args.steps)
                out["gather"].update({"side_stream_ms_per_step": gather_ms, "share_of_step": gather_ms / step_ms, "rows_ok": gather_ok})
                if dt_with_gather is not None:
                    out["gather"].update({"ms_per_step_with_gather": step_ms, "value_with_gather": B / dt_with_gather,
                                          "note": "N = 1: no exchange between devices; the device-to-host copy is PCIe traffic and stays outside `value` -- the same "
                                                  "loop with the copy every step, measured right after the timed region (--gather puts it inside)"})
        if dry and gather_on and dist is not None and gathered_h is not None:
            # the rehearsal checks the same thing the device path does: every rank's block arrived (finite rows, unit quaternions of x1)
            gl = torch.stack(gathered_h)
            out["gather"]["rows_ok"] = bool(torch.isfinite(gl).all()) and bool((gl[:, :, 3:7].norm(dim=2) - 1).abs().max() < 1e-9)
        if dry:
            out["data"] = "synthetic (DRY RUN on CPU: emulated kernel bodies + gloo, launch-path rehearsal, not a measurement)"
        if kt:
            out["kernel_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}
            out["kernel_share"] = {k: round(v[0] / max(1e-9, sum(x[0] for x in kt.values())), 3) for k, v in kt.items()}
            rl = rooflines(kt, B, H, ndx, nu, nc, gm.nx, B == 4096 and args.iters == 3)
            share = {k: kt[k][0] + (kt.get("tree", (0.0, 0))[0] if k == "deriv" else 0.0) for k in rl}
            dom = max(rl, key=lambda k: share[k])  # dominant kernel (pass) = largest share of the timed region
            if args.streams == 1:
                out["roofline"] = rl[dom]
                out["roofline_other"] = {k: v for k, v in rl.items() if k != dom}
            # whole step against both bounds: Riccati + derivative FLOPs of k iterations / compulsory I/O of a control step
            fc = flop_counts().get("kinodynamics", {})
            step_fl = args.iters * H * (f_ric(ndx, nu, nc) + fc.get("deriv_flops_per_stage", 0.0))
            step_io = 8 * (2 * (H + 1) * gm.nx + 2 * H * nu + nu * ndx + (H + 1) * ndx + H * nc)
            out["step_roofline"] = both_bounds(B * step_fl, B * step_io, dt / args.steps, "mfma")
            # HBM bytes of a whole control step by the PMC counters: per-launch bytes of the newest committed summary x launches per step
            # (the materialised LQ problem: knots and gains go through HBM between the kernels of an iteration)
            per = {k: pmc_traffic(k, B == 4096 and args.iters == 3) for k in ("lane_tree_body", "deriv2_body", "riccati_kino_body", "forward_kino_body", "apply_body", "trial_rows_body")}
            if all(v[0] is not None for v in per.values()):
                it = args.iters
                tls = pmc_traffic("lane_tree_ls_body", True)[0]  # (line-search launch of the tree kernel; older summaries: one key for both modes)
                tot = it * (per["lane_tree_body"][0] + per["deriv2_body"][0] + per["riccati_kino_body"][0] + per["forward_kino_body"][0] + per["apply_body"][0]) \
                    + (tls if tls is not None else per["lane_tree_body"][0]) + per["trial_rows_body"][0]
                out["step_roofline"].update({"traffic": tot, "traffic_source": per["deriv2_body"][1], "traffic_over_compulsory": tot / (B * step_io),
                                             "traffic_GBps": tot / (dt / args.steps) / 1e9})
            if args.streams > 1:
                # launches of different streams overlap: event-to-event durations of single launches include the time they share the
                # GPU with other launches, so only the whole-step figure is meaningful in this mode
                out["roofline"] = dict(out["step_roofline"], kernel="whole control step (--streams %d: per-kernel durations overlap)" % args.streams, traffic=None)
                del out["kernel_share"]
        if world == 1 and not args.no_cpu_baseline and not dry:
            out["cpu_baseline"] = cpu_baseline(args.iters)
        if world == 1 and profile and not args.headline_only:
            # the other single-GPU BASELINE configurations, measured briefly beside the headline (not part of `value`)
            other = {"fulldynamics_forward_dynamics": constraint_dynamics_line(gm, mh, B, gm.H)}
            del gm
            other["centroidal"] = centroidal_line(B, args.iters, 40, 5, local_rank, not args.no_cpu_baseline)
            other["fulldynamics_go2"] = fulldynamics_line(min(B, 4096), args.iters, 20, 3, local_rank, not args.no_cpu_baseline)
            other["fulldynamics_talos"] = fulldynamics_line(1024, args.iters, 10, 2, local_rank, not args.no_cpu_baseline, robot="talos")
            other["talos_kinodynamics_6d"] = talos_flat_feet_line("talos_kinodynamics", 1024, args.iters, 8, 2, local_rank, not args.no_cpu_baseline)
            other["talos_centroidal_6d"] = talos_flat_feet_line("talos_centroidal", 1024, args.iters, 20, 3, local_rank, not args.no_cpu_baseline)
            other["inverse_dynamics_qp"] = inverse_dynamics_line(B, local_rank, not args.no_cpu_baseline)
            other["inverse_dynamics_qp_flat_feet"] = inverse_dynamics_quad_line(B, local_rank, not args.no_cpu_baseline)
            other["control_stack"] = control_stack_line(B, local_rank)
            other["control_stack_talos"] = control_stack_talos_line(1024, local_rank)
            other["single_robot_latency"] = single_robot_latency(args.iters, local_rank)
            out["other_workloads"] = other
            # the other BASELINE configurations as flat keys (configs[1], configs[3]; configs[0] = cpu_baseline.cfg1_k1_b1; configs[4] = --gpus 8)
            out["cfg2_centroidal_steps_per_s"] = other["centroidal"]["value"]
            out["cfg4_talos_fulldynamics_steps_per_s"] = other["fulldynamics_talos"]["value"]
            out["go2_fulldynamics_steps_per_s"] = other["fulldynamics_go2"]["value"]
            out["inverse_dynamics_qps_per_s"] = other["inverse_dynamics_qp"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
