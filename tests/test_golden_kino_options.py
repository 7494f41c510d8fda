"""Committed closed loops of the optional constraint blocks of the kinodynamics OCP (tests/golden/go2_kino_options_golden.npz, made by
tests/golden/make_golden_kino_options.py): the oracle must still reproduce them, and the kernels -- CPU build and HIP library -- must
follow them (terminal constraint / land rows to 1e-7, eliminated friction-cone rows to 1e-5)."""
import os
import sys

import numpy as np
import pytest

import mpc_setup as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden_kino_options as MK  # noqa: E402  (the scenario table of the generator: data, not oracle code)

G = np.load(os.path.join(GOLDEN, "go2_kino_options_golden.npz"))


def _replay(m, tag):
    c = MK.CASES[tag]
    m.generateCycleHorizon(MK.cycle_of(c["cycle"]))
    m.switchToWalk(np.array(c["walk"], float))
    X = G[tag + "_X0"].copy()
    for _ in range(c["steps"]):
        m.iterate(X)
        X = m.xs[:, 1, :].copy()
    return m


@pytest.mark.parametrize("tag", ["tc", "cone", "land"])
def test_oracle_reproduces_the_golden_loops(tag):
    c = MK.CASES[tag]
    om, _, _ = S.make_oracle(2, 2, 20, settings_override=c["so"], mpc_override=c["mo"])
    _replay(om, tag)
    assert S.rel_err(G[tag + "_xs"], om.xs) < 1e-9 and S.rel_err(G[tag + "_us"], om.us) < 1e-8
    assert np.array_equal(G[tag + "_alpha"], om.info[:, 2])
    rows = {"tc": 0, "cone": 10, "land": 4}[tag]
    assert int((G[tag + "_vs"][:, :, 24:] != 0).sum()) >= rows  # the fixture exercises the rows it is named after


def _product(tag, lib):
    c = MK.CASES[tag]
    gm, _, _, _ = S.make_product(2, 2, lib, 20, settings_override=c["so"], mpc_override=c["mo"])
    _replay(gm, tag)
    tol = 1e-5 if tag == "cone" else 1e-7
    assert S.rel_err(G[tag + "_xs"], gm.xs) < tol and S.rel_err(G[tag + "_us"], gm.us) < 10 * tol
    assert np.allclose(G[tag + "_alpha"], gm.info[:, 2])


@pytest.mark.parametrize("tag", ["tc", "cone", "land"])
def test_emulated_kernels_follow_the_golden_loops(built, tag):
    _product(tag, S.emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tc", "cone", "land"])
def test_hip_follows_the_golden_loops(built, tag):
    _product(tag, None)
