"""FrictionCompensation: the reference's own test (tests/friction.cpp:13-39: dry 0.5, viscuous 0.05,
random velocity / torque, expected = torque + dry * sign(v) + viscuous * v) on the oracle, the CPU build of the kernel
bodies and -- GPU tier -- the HIP library; plus the reference's size errors."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc

NU = 12  # solo12 / go2: nv - 6


def _case(batch, rng):
    v = rng.uniform(-1, 1, (batch, NU))
    v[0, 3] = 0.0  # sign(0) = 0
    tau = rng.uniform(-1, 1, (batch, NU))
    dry, vis = np.full(NU, 0.5), np.full(NU, 0.05)
    expected = tau + dry * np.sign(v) + vis * v  # reference tests/friction.cpp:29-33
    return dry, vis, v, tau, expected


def test_oracle_reproduces_the_reference_test():
    dry, vis, v, tau, expected = _case(5, np.random.default_rng(0))
    out = tau.copy()
    O.lib().orc_friction(dry, vis, NU, np.ascontiguousarray(v), out, 5)
    assert np.allclose(out, expected, rtol=0, atol=1e-15)


def _check(lib):
    rng = np.random.default_rng(1)
    for batch in (1, 7, 1000):
        dry, vis, v, tau, expected = _case(batch, rng)
        fc = simple_mpc.FrictionCompensation(dry, vis, lib=lib)
        assert fc.nu_ == NU and np.array_equal(fc.dry_friction_, dry) and np.array_equal(fc.viscuous_friction_, vis)
        out = tau.copy()
        fc.computeFriction(v, out)
        ref = tau.copy()
        O.lib().orc_friction(dry, vis, NU, np.ascontiguousarray(v), ref, batch)
        assert np.allclose(out, ref, rtol=0, atol=1e-15)  # (fused multiply-add contraction may differ by one ulp)
        assert np.allclose(out, expected, rtol=0, atol=1e-15)
    fc = simple_mpc.FrictionCompensation(np.full(NU, 0.5), np.full(NU, 0.05), lib=lib)
    one_v, one_t = np.linspace(-1, 1, NU), np.zeros(NU)
    fc.computeFriction(one_v, one_t)
    assert np.allclose(one_t, 0.05 * one_v + 0.5 * np.sign(one_v))
    with pytest.raises(RuntimeError, match="Velocity has wrong size"):
        fc.computeFriction(np.zeros(NU + 1), np.zeros(NU))
    with pytest.raises(RuntimeError, match="Torque has wrong size"):
        fc.computeFriction(np.zeros(NU), np.zeros(NU - 1))


def test_kernel_body_on_cpu(built):
    _check(S.emu_lib())


@pytest.mark.gpu
def test_hip_library(built):
    _check(None)
