"""smpc_get_stream: a caller that produces the measured states on the device (here torch, through torch.cuda.ExternalStream) enqueues on the
handle's own stream; the closed loop is then one in-order queue and the host may run control steps ahead of the device.  The per-step host
tables (the shared stage descriptors, which change at every step of a walking cycle) go through a ring of pinned staging buffers guarded by
events (UploadRing, smpc_backend.h): the loop without host-side waits must give bit for bit what the loop with a wait after every step gives."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def test_stream_handle_of_the_cpu_test_build_is_null(built):
    gm, _, _, _ = S.make_product(1, 1, lib=S.emu_lib(), horizon=10)
    assert gm.stream() == 0


def _loop(kind, on_stream, steps=40, B=64):
    import torch

    dev = torch.device("cuda", 0)
    if kind == "kino":
        gm, rb, _, _ = S.make_product(B, 2)
    elif kind == "cent":
        gm, rb, _, _ = S.make_cent_product(B, 2)
    else:
        gm, rb, _, _ = S.make_full_product(B, 1)
    gm.generateCycleHorizon(O.trot_cycle(2, 6))  # a quick trot: the stage table changes at every control step
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.1]))
    X0 = torch.from_numpy(S.random_states(rb, B)).to(dev)
    X = X0.clone()
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    assert gm.stream() != 0
    ext = torch.cuda.ExternalStream(gm.stream(), device=dev)
    torch.cuda.synchronize()

    def feedback():
        if kind == "cent":  # (xs of a centroidal handle are centroidal states: perturb the measured multibody state instead)
            X.copy_(X0)
            X[:, :3].add_(torch.randn((B, 3), generator=gen, device=dev, dtype=torch.float64) * 1e-3)
        else:
            X.add_(torch.randn(X.shape, generator=gen, device=dev, dtype=torch.float64) * 1e-3)
            q = X[:, 3:7]
            q.div_(q.norm(dim=1, keepdim=True))

    for _ in range(steps):
        gm.iterate_device(X.data_ptr())
        if kind != "cent":
            gm.get_x_device(1, X.data_ptr())
        if on_stream:
            with torch.cuda.stream(ext):
                feedback()
        else:
            gm.wait()
            feedback()
            torch.cuda.synchronize()
    gm.wait()
    torch.cuda.synchronize()
    return X.cpu().numpy(), gm.xs.copy(), gm.us.copy()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kino", "cent", "full"])
def test_closed_loop_without_host_waits_is_bit_identical(built, kind):
    a = _loop(kind, False)
    b = _loop(kind, True)
    for u, v in zip(a, b):
        assert np.all(np.isfinite(u)) and np.array_equal(u, v)
