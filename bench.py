#!/usr/bin/env python3
"""bench.py -- MPC control-steps/sec at fixed ProxDDP iterations, Go2 kinodynamics, H=50 (BASELINE.json).

A "step" is one batched MPC::iterate (reference src/mpc.cpp:189-218) of B instances per GPU with exactly
k ProxDDP iterations each; value = (instances x steps) / wall time, whole job.  Inputs (measured states)
are resident in HBM when the timed region starts: the closed loop feeds back xs[1] + N(0, 1e-3^2) noise
generated on the device.  Multi-GPU: one process per GPU, batch sharded by instance, no
collective on the solve path (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 4096] [--iters 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))

FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix dense peak (public spec)


def f_ric(ndx, nu, nc):
    """Algorithmic FLOPs of the proximal Riccati backward+forward per (instance, stage, iteration): SURVEY 8(d)."""
    return (4 * ndx**3 + 4 * ndx**2 * nu + 2 * ndx * nu**2 + (nu + nc) ** 3 / 3 + 2 * (nu + nc) ** 2 * (ndx + 1)
            + 2 * ndx**2 * (nu + nc) + 2 * ndx * (nu + nc))


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6300 GB/s achievable)


def rooflines(kt, B, H, ndx, nu, nc, nx):
    """Roofline entries of the two kernels that carry the step: average launch duration from the
    HIP events the engine records on ITS stream around every launch inside the timed region."""
    out = {}
    # HBM traffic per launch: PMC counters cannot be collected from inside this process; the figure comes from the
    # committed rocprofv3 --pmc summary of this same command (profiles/, newest round), collected and corrected as
    # MI355X_MICROARCH.md prescribes.  None if no summary is present.
    import glob
    pmc = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")))
    if files and B == 4096:
        with open(files[-1]) as f:
            pmc = json.load(f).get("kernels", {})
    traffic = lambda k: (pmc.get(k, {}).get("hbm_bytes_per_launch_corrected"), os.path.basename(files[-1]) if pmc else None)
    if "deriv" in kt and kt["deriv"][1]:
        # algorithmic bytes per (instance, stage): the state-dependent part of the LQ knot, written per iteration --
        # the upper tiles of Q, the force columns of S, the force block of R + the regularised diagonal, the contact rows of C + the box
        # selectors, the 12 dense rows of [A|B], q r f d lx lu lpd vpd (integrator rows, zero blocks and constant
        # weight entries are written once at start-up) -- plus the iterate read (x, u, nu, lam, lam+, centres)
        nfc = 12                 # 3 * nf force components (Go2: 4 point feet)
        na = nu - nfc            # actuated joints = box rows
        tl = [min(16, ndx - 16 * i) for i in range((ndx + 15) // 16)]
        q_upper = sum(tl[i] * tl[j] for i in range(len(tl)) for j in range(i, len(tl)))  # upper 16x16 tiles of Q (912 of 1296)
        per_stage = 8 * (q_upper + ndx * nfc + nfc * nfc + na + (nc - na) * ndx + na + 12 * (ndx + nu) + 4 * ndx + 2 * nu + 2 * nc
                         + nx + nu + 2 * nc + 4 * ndx)
        avg = kt["deriv"][0] / kt["deriv"][1] * 1e-3
        ach = B * H * per_stage / avg / 1e9
        out["deriv"] = {"bound": "hbm", "kernel": "deriv_body (stage evaluation + derivatives + LQ knot)", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic("deriv_body")[0],
                        "traffic_source": traffic("deriv_body")[1], "algorithmic_bytes": B * H * per_stage, "avg_launch_ms": avg * 1e3,
                        "note": "algorithmic bytes = B*H*%d per launch; the kernel is FP64-VALU issue bound (one wave per (instance, stage) "
                                "keeps its SIMD's issue slot busy; co-resident waves do not add throughput), not bandwidth bound" % per_stage}
    if "riccati" in kt and kt["riccati"][1]:
        avg = kt["riccati"][0] / kt["riccati"][1] * 1e-3
        ach = B * H * f_ric(ndx, nu, nc) / avg / 1e12
        out["riccati"] = {"bound": "mfma", "kernel": "riccati_kino_body (proximal Riccati backward sweep)", "achieved": ach,
                          "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic("riccati_kino_body")[0],
                          "traffic_source": traffic("riccati_kino_body")[1], "avg_launch_ms": avg * 1e3,
                          "note": "FP64 dense peak (vector = matrix on MI355X); algorithmic FLOPs = B*H*F_ric(36,24,24) of the "
                                  "unstructured recursion (SURVEY 8d) per launch"}
    return out


def cpu_baseline(iters, seconds_budget=20.0):
    """Oracle (CPU restatement, not Aligator) on the host cores, bounded sample of the same workload."""
    import numpy as np
    import mpc_setup as S
    import oracle_lib as O

    threads = O.lib().orc_num_threads()
    B = max(threads * 2, 8)
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
        if time.time() - t0 > seconds_budget or n >= 50:
            break
    dt = time.time() - t0
    return {
        "value": B * n / dt,
        "unit": "control-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": "CPU restatement (oracle/, not Aligator): %d instances x %d steps, k=%d, OpenMP over instances" % (B, n, iters),
    }


def constraint_dynamics_line(gm, rb, batch, horizon):
    """First device block of the full-dynamics model (BASELINE config 4 is not on the device yet): the constrained forward
    dynamics kernel on batch x horizon states, all feet in contact -- one launch, timed around the launch itself."""
    import numpy as np
    import mpc_setup as S

    n = batch * horizon
    X = np.tile(S.random_states(rb, 512, seed=3), ((n + 511) // 512, 1))[:n]
    tau, mask = np.zeros((n, rb.nv - 6)), np.full(n, (1 << rb.nf) - 1, np.uint32)
    ms = [gm.constraintDynamics(X, tau, mask)["kernel_ms"] for _ in range(3)]
    return {"metric": "constrained forward dynamics (full-dynamics model), states/sec", "value": n / (min(ms) * 1e-3), "unit": "states/s",
            "kernel_ms": min(ms), "states": n, "dtype": "f64",
            "note": "smpc_full_forward_dynamics: forward dynamics only; derivatives / stage / solver of the full-dynamics OCP "
                    "are not on the device yet (DESIGN.md 0, 3.8)"}


def centroidal_line(batch, iters, steps, warmup, device_id, with_cpu=True):
    """BASELINE config "Go2 centroidal (9-dim state), H=50, batch=4096": same step definition on the centroidal OCP
    (one fused kernel per control step).  Measured states: x_ref (+) N(0, sigma^2), resident in HBM, re-drawn on the
    device every step (the centroidal solution has no multibody state to feed back)."""
    import numpy as np
    import torch
    import mpc_setup as S
    import oracle_lib as O

    gm, rb, _, _ = S.make_cent_product(batch, max_iters=iters, device_id=device_id)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    dev = torch.device("cuda", device_id)
    X0 = torch.from_numpy(S.random_states(rb, batch)).to(dev)
    X = X0.clone()
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)

    def step():
        gm.iterate_device(X.data_ptr())
        gm.wait()
        X.copy_(X0)
        X[:, :3].add_(torch.randn((batch, 3), generator=gen, device=dev, dtype=torch.float64) * 1e-3)
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    gm.set_profiling(True)
    gm.reset_kernel_times()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = gm.kernel_times()
    avg = kt["step"][0] / max(kt["step"][1], 1) * 1e-3
    flops = batch * gm.H * iters * f_ric(9, gm.nu, gm.nc)
    import glob
    traffic, src = None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")))
    if files and batch == 4096 and iters == 3:
        with open(files[-1]) as f:
            traffic = json.load(f).get("kernels", {}).get("cent_step_body", {}).get("hbm_bytes_per_launch_corrected")
        src = os.path.basename(files[-1])
    out = {
        "metric": "MPC control-steps/sec at fixed ProxDDP iters, Go2 centroidal H=50",
        "value": batch * steps / dt, "unit": "control-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": "f64",
        "config": {"workload": "Go2 centroidal (go2_like table), H=%d, %d ProxDDP iters/step, batch=%d, trot 10/30/10/30, "
                   "x_meas = x_ref (+) N(0, sigma^2)" % (gm.H, iters, batch), "finite": bool(np.all(np.isfinite(gm.info)))},
        "kernel_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items() if k != "-"},
        "roofline": {"bound": "mfma", "kernel": "cent_step_body (whole control step: recede + %d ProxDDP iterations)" % iters,
                     "achieved": flops / avg / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / avg / 1e12 / FP64_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_source": src, "avg_launch_ms": avg * 1e3,
                     "note": "algorithmic FLOPs = B*H*k*F_ric(9,12,8) (SURVEY 8d); the 21 x 21 stage systems leave the matrix cores "
                             "mostly idle: the kernel is bound by the instruction issue of its index / assembly code (DESIGN.md 3.4)"},
    }
    if with_cpu:
        threads = O.lib().orc_num_threads()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--iters", type=int, default=3, help="ProxDDP iterations per control step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--workload", default="kinodynamics", choices=["kinodynamics", "centroidal"],
                    help="kinodynamics = the headline metric (with the centroidal configuration measured briefly beside it at 1 GPU); "
                    "centroidal = only the centroidal line")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MPC engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g

    if rank == 0:
        # rank 0 (re)builds what is stale, the others wait: no concurrent compiler / make runs on the shared tree
        g.build_hip()
        g.build_oracle()
    if dist is not None:
        dist.barrier()
    import mpc_setup as S
    import oracle_lib as O

    B = args.batch
    if args.workload == "centroidal":
        if world > 1:
            raise SystemExit("--workload centroidal is a single-GPU line")
        line = centroidal_line(B, args.iters, args.steps, args.warmup, local_rank, not args.no_cpu_baseline)
        line.update({"n_gpus": 1, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
        print(json.dumps(line))
        return
    gm, rb, _, _ = S.make_product(B, max_iters=args.iters, device_id=local_rank)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))

    dev = torch.device("cuda", local_rank)
    X0 = S.random_states(rb, B, seed=20240529 + rank)
    X = torch.from_numpy(X0).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20240529 + rank)

    def step():
        gm.iterate_device(X.data_ptr())
        gm.get_x_device(1, X.data_ptr())  # x_meas <- xs[1] (same stream, ordered after the solve)
        gm.wait()
        noise = torch.randn(X.shape, generator=gen, device=dev, dtype=torch.float64) * 1e-3
        X.add_(noise)
        q = X[:, 3:7]
        q.div_(q.norm(dim=1, keepdim=True))
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        gm.set_profiling(True)
        gm.reset_kernel_times()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kt = gm.kernel_times() if not args.no_profile else {}
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
                "global_batch": world * B,
                "parallelism": "instance-sharded x%d, no collective on the solve path" % world,
                "finite": ok,
            },
        }
        if kt:
            out["kernel_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}
            out["kernel_share"] = {k: round(v[0] / max(1e-9, sum(x[0] for x in kt.values())), 3) for k, v in kt.items()}
            rl = rooflines(kt, B, H, ndx, nu, nc, gm.nx)
            dom = max(rl, key=lambda k: kt[k][0])  # dominant kernel = largest share of the timed region
            out["roofline"] = rl[dom]
            out["roofline_other"] = {k: v for k, v in rl.items() if k != dom}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.iters)
        if world == 1 and not args.no_profile:
            # the other single-GPU BASELINE configuration, measured briefly beside the headline (not part of `value`)
            fd = constraint_dynamics_line(gm, rb, B, gm.H)
            del gm
            out["other_workloads"] = {"centroidal": centroidal_line(B, args.iters, 10, 3, local_rank, not args.no_cpu_baseline),
                                      "fulldynamics_forward_dynamics": fd}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
