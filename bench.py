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

import bench_common as C  # noqa: E402
from bench_common import both_bounds, cpu_baseline, f_ric, flop_counts, loop_stream, make_mpc, measure_fp64_peak, pmc_traffic, rooflines, step_sync  # noqa: E402


LINE_LIMIT = 3072  # bytes of the last stdout line (round 5's 24 KB line was not parsed by the driver)


def _r(v, n=5):
    """Floats to n significant digits (the full-precision values are in bench_full.json / stderr)."""
    return float("%.*g" % (n, v)) if isinstance(v, float) else v


def compact_line(out):
    """The one stdout line: headline fields, the dominant kernel's roofline (flat), the cpu_baseline, per-kernel ms, the flat cfg* values --
    nothing nested twice, no notes.  Optional keys are dropped, last first, until the line is under LINE_LIMIT bytes."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "global_batch", "parallelism", "streams", "finite") if k in cfg}
    rl = out.get("roofline")
    if rl:
        line["roofline"] = {k: _r(rl[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "kernel", "traffic", "traffic_source", "peak_measured")
                            if k in rl}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "b1_latency_ms") if k in cb}
        if "cfg1_k1_b1" in cb:
            line["cpu_baseline"]["cfg1_k1_b1_ms"] = _r(cb["cfg1_k1_b1"]["ms_per_step"])
    optional = []
    if "kernel_ms" in out:
        line["kernel_ms"] = out["kernel_ms"]
        optional.append("kernel_ms")
    for k in ("cfg2_centroidal_steps_per_s", "cfg4_talos_fulldynamics_steps_per_s", "go2_fulldynamics_steps_per_s", "inverse_dynamics_qps_per_s"):
        if k in out:
            line[k] = _r(out[k], 6)
    sr = out.get("step_roofline")
    if sr:
        line["step_roofline"] = {k: _r(sr[k]) for k in ("bound", "frac", "traffic", "traffic_over_compulsory", "traffic_GBps") if k in sr}
        optional.append("step_roofline")
    ro = out.get("roofline_other")
    if ro:
        line["roofline_other"] = {n: {k: _r(e[k]) for k in ("bound", "frac", "avg_launch_ms", "traffic") if k in e} for n, e in ro.items()}
        optional.append("roofline_other")
    g = out.get("gather")
    if g:
        line["gather"] = {k: _r(g[k]) for k in ("in_timed_region", "bytes_per_step", "mode", "side_stream_ms_per_step", "rows_ok", "value_with_gather", "error") if k in g}
        optional.append("gather")
    text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_LIMIT and optional:
        del line[optional.pop()]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


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
    C.SYNC_STEPS = bool(args.sync_steps)
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
        import bench_side

        measure_fp64_peak()
        if args.workload == "centroidal":
            line = bench_side.centroidal_line(B, args.iters, args.steps, args.warmup, local_rank, not args.no_cpu_baseline)
        else:
            line = bench_side.fulldynamics_line(B, args.iters, args.steps, args.warmup, local_rank, not args.no_cpu_baseline,
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
            if C.SYNC_STEPS:
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
            # the other single-GPU BASELINE configurations, measured briefly beside the headline (not part of `value`): bench_side.py.  Their full
            # entries go to stderr / bench_side.json; this line keeps the flat values (configs[1], configs[3]; configs[0] = cpu_baseline.cfg1_k1_b1_ms;
            # configs[4] = --gpus 8)
            import bench_side

            other = {"fulldynamics_forward_dynamics": bench_side.constraint_dynamics_line(gm, mh, B, gm.H)}
            del gm
            other.update(bench_side.run_all(B, args.iters, local_rank, not args.no_cpu_baseline))
            bench_side.report(other)
            out["cfg2_centroidal_steps_per_s"] = other["centroidal"]["value"]
            out["cfg4_talos_fulldynamics_steps_per_s"] = other["fulldynamics_talos"]["value"]
            out["go2_fulldynamics_steps_per_s"] = other["fulldynamics_go2"]["value"]
            out["inverse_dynamics_qps_per_s"] = other["inverse_dynamics_qp"]["value"]
        # everything measured, in full, to stderr and bench_full.json; the LAST stdout line is the compact object the driver parses
        full = json.dumps(out)
        sys.stderr.write("bench.py full line: " + full + "\n")
        sys.stderr.flush()
        try:
            with open(os.path.join(ROOT, "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        print(compact_line(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
