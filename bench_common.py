"""Shared pieces of bench.py (the headline line) and bench_side.py (the other single-GPU workloads): peaks, roofline entries, the PMC
summary reader, the oracle import of the cpu_baseline legs, the handle factory."""
import glob
import json
import os
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

