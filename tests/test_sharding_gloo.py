"""N > 1 path on CPU: the batch shards by instance across ranks with no collective on the solve path
(SURVEY 8e).  Two gloo ranks each solve their shard (emulated kernel bodies); the gathered outputs must be
bit-identical to a single process solving the whole batch, and the timing reduction used by bench.py
(MAX over ranks) must work."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import mpc_setup as S, oracle_lib as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B = 2
gm, rb, _, _ = S.make_product(B, lib=S.emu_lib())
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))
Xall = S.random_states(rb, B * world)
X = Xall[rank * B:(rank + 1) * B]          # contiguous block partition of the batch
for _ in range(2):
    gm.iterate(X); X = gm.xs[:, 1, :].copy()
mine = torch.from_numpy(gm.xs.copy())
parts = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(parts, mine)               # output gather only; nothing on the solve path
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    np.save(sys.argv[1], torch.cat(parts).numpy()); assert t.item() == world
dist.barrier(); dist.destroy_process_group()
""" % HERE


def test_two_rank_sharding_matches_single_process(built, tmp_path):
    import mpc_setup as S
    import oracle_lib as O

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    subprocess.check_call(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script), out],
        env=env, timeout=600,
    )
    gathered = np.load(out)
    gm, rb, _, _ = S.make_product(4, lib=S.emu_lib())
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 4)
    for _ in range(2):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    assert np.array_equal(gathered, gm.xs)


def test_two_handles_gather_into_one_buffer():
    """The in-process form of the sharded batch: two handles ("devices"), each with its block of instances, launched without
    waiting, their return sets [x1 | u0 | K0] gathered into ONE host buffer -- identical to a single handle over the whole batch."""
    import mpc_setup as S
    import oracle_lib as O

    lib = S.emu_lib()
    parts = [S.make_product(n, max_iters=2, lib=lib, horizon=10)[0] for n in (2, 3)]
    whole, rb, _, _ = S.make_product(5, max_iters=2, lib=lib, horizon=10)
    for m in parts + [whole]:
        m.generateCycleHorizon(O.trot_cycle())
        m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 5, seed=9)
    row = whole.nx + whole.nu + whole.nu * whole.ndx
    out = np.full((5, row), np.nan)
    for _ in range(3):
        parts[0].iterateAsync(X[:2])
        parts[1].iterateAsync(X[2:])
        parts[0].gatherOutputs(out, 0)
        parts[1].gatherOutputs(out, 2)
        for p in parts:
            p.wait()
        whole.iterate(X)
        assert np.array_equal(out[:, : whole.nx], whole.xs[:, 1, :])
        assert np.array_equal(out[:, whole.nx : whole.nx + whole.nu], whole.us[:, 0, :])
        assert np.array_equal(out[:, whole.nx + whole.nu :].reshape(5, whole.nu, whole.ndx), whole.K0.reshape(5, whole.nu, whole.ndx))
        # the packed form ("device" memory is host memory in the CPU test build), with a row stride wider than the row
        packed = np.full((5, row + 3), np.nan)
        whole.gather_outputs_device(packed.ctypes.data, row + 3)
        whole.wait()
        assert np.array_equal(packed[:, :row], out) and np.isnan(packed[:, row:]).all()
        # the peer-copy form: every handle's rows into one buffer "on device 0"
        peer = np.full((5, row), np.nan)
        parts[0].gather_outputs_peer(peer.ctypes.data, 0)
        parts[1].gather_outputs_peer(peer.ctypes.data + 2 * row * 8, 0)
        for p in parts:
            p.wait()
        assert np.array_equal(peer, out)
        X = whole.xs[:, 1, :].copy()


@pytest.mark.gpu
def test_gpu_two_handles_gather_into_one_pinned_buffer():
    """The same on the HIP library: two handles (two streams; two devices when the box has them, else both on device 0), launched
    back to back from one host thread without waiting, return sets gathered into ONE pinned host buffer; bit-identical to one handle over
    the whole batch (instances are independent: the split cannot change a result), and the launches of the two handles overlap in time."""
    import time

    import torch

    import mpc_setup as S
    import oracle_lib as O

    ndev = torch.cuda.device_count()
    sizes = (96, 160)
    parts = [S.make_product(n, max_iters=2, device_id=(i % ndev))[0] for i, n in enumerate(sizes)]
    whole, rb, _, _ = S.make_product(sum(sizes), max_iters=2)
    for m in parts + [whole]:
        m.generateCycleHorizon(O.trot_cycle())
        m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, sum(sizes), seed=9)
    row = whole.nx + whole.nu + whole.nu * whole.ndx
    pinned = torch.full((sum(sizes), row), float("nan"), dtype=torch.float64).pin_memory()
    out = pinned.numpy()
    assert parts[0].stream() != parts[1].stream() and parts[0].stream() != 0
    for it in range(4):
        t0 = time.perf_counter()
        parts[0].iterateAsync(X[: sizes[0]])
        parts[1].iterateAsync(X[sizes[0] :])
        parts[0].gatherOutputs(out, 0)
        parts[1].gatherOutputs(out, sizes[0])
        t_launch = time.perf_counter() - t0
        for p in parts:
            p.wait()
        t_all = time.perf_counter() - t0
        whole.iterate(X)
        assert np.array_equal(out[:, : whole.nx], whole.xs[:, 1, :])
        assert np.array_equal(out[:, whole.nx : whole.nx + whole.nu], whole.us[:, 0, :])
        assert np.array_equal(out[:, whole.nx + whole.nu :].reshape(-1, whole.nu, whole.ndx), whole.K0.reshape(-1, whole.nu, whole.ndx))
        packed = torch.full((sum(sizes), row + 3), float("nan"), dtype=torch.float64, device="cuda:0")
        whole.gather_outputs_device(packed.data_ptr(), row + 3)
        whole.wait()
        ph = packed.cpu().numpy()
        assert np.array_equal(ph[:, :row], out) and np.isnan(ph[:, row:]).all()
        # peer-copy form (hipMemcpyPeerAsync on each handle's stream): all rows into one buffer on device 0
        peer = torch.full((sum(sizes), row), float("nan"), dtype=torch.float64, device="cuda:0")
        parts[0].gather_outputs_peer(peer.data_ptr(), 0)
        parts[1].gather_outputs_peer(peer.data_ptr() + sizes[0] * row * 8, 0)
        for p in parts:
            p.wait()
        assert np.array_equal(peer.cpu().numpy(), out)
        # the host thread came back from the launches before the device work was done (nothing in iterateAsync / gatherOutputs waits)
        if it > 0:
            assert t_launch < 0.5 * t_all, (t_launch, t_all)
        X = whole.xs[:, 1, :] + 1e-3 * np.random.default_rng(it).standard_normal(whole.xs[:, 1, :].shape)
        X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
    # the accessors of a part agree with the gathered rows (same ring slots)
    for p, r0 in zip(parts, (0, sizes[0])):
        assert np.array_equal(p.xs[:, 1, :], out[r0 : r0 + p.B, : whole.nx])
