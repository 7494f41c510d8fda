"""Shared construction of (oracle MPC, product MPC) pairs on the Go2 kinodynamics settings of record
(reference examples/go2_kinodynamics.py:30-139).  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))

import oracle_lib as O  # noqa: E402
import simple_mpc  # noqa: E402
from simple_mpc._capi import SmpcLib  # noqa: E402

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libsmpc_emu.so")
FEET = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
MPC_KEYS = ["support_force", "TOL", "mu_init", "max_iters", "num_threads", "swing_apex", "T_fly", "T_contact", "timestep"]
SIGMA = np.concatenate([np.ones(3) * 0.02, np.ones(3) * 0.05, np.ones(12) * 0.1, np.ones(3) * 0.1, np.ones(3) * 0.2, np.ones(12) * 0.5])

_emu = None


def emu_lib():
    """The test-only sequential-lane build of the kernel bodies (never used by the product path)."""
    global _emu
    if _emu is None:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g

        g.build_emu()
        _emu = SmpcLib(EMU_LIB)
    return _emu


_xcheck = None


def xcheck_lib():
    """The -DSMPC_CROSSCHECK HIP build (alternative engine paths + their environment switches; test infrastructure of the GPU tier)."""
    global _xcheck
    if _xcheck is None:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g

        g.build_hip(with_xcheck=True)
        _xcheck = SmpcLib(g.XCHECK_LIB)
    return _xcheck


def make_product(batch, max_iters=1, lib=None, horizon=50, settings_override=None, mpc_override=None, device_id=0):
    rb = O.Robot("go2_like")
    s = O.go2_kino_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in FEET:
        mh.addPointFoot(n, "root_joint")
    ocp = simple_mpc.KinodynamicsOCP(s, mh)
    ocp.createProblem(mh.getReferenceState(), horizon, 3, -9.81, bool(ms.get("terminal_constraint", False)))
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_oracle(batch, max_iters=1, horizon=50, settings_override=None, mpc_override=None):
    rb = O.Robot("go2_like")
    s = O.go2_kino_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    K = O.Kino(rb, s)
    om = O.OracleMPC(K, ms, batch)
    return om, rb, K


def make_pair(batch, max_iters=1, lib=None, horizon=50, walk=(0.2, 0, 0, 0, 0, 0), **kw):
    om, rb, _ = make_oracle(batch, max_iters, horizon, **kw)
    gm, _, _, _ = make_product(batch, max_iters, lib, horizon, **kw)
    cs = O.trot_cycle()
    om.generateCycleHorizon(cs)
    gm.generateCycleHorizon(cs)
    v = np.array(walk, float)
    om.switchToWalk(v)
    gm.switchToWalk(v)
    return om, gm, rb


def random_states(rb, batch, seed=20240529, scale=1.0):
    """Synthetic initial states of SURVEY 8(d): x_ref (+) N(0, diag(sigma^2))."""
    rng = np.random.default_rng(seed)
    return np.stack([rb.integrate(rb.x_ref, rng.normal(size=rb.ndx) * SIGMA * scale) for _ in range(batch)])


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(a).max()))


def make_cent_product(batch, max_iters=1, lib=None, horizon=50, settings_override=None, mpc_override=None, device_id=0):
    """simple_mpc.BatchedMPC over a CentroidalOCP with the centroidal settings of record (oracle_lib.go2_centroidal_settings)."""
    rb = O.Robot("go2_like")
    s = O.go2_centroidal_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in FEET:
        mh.addPointFoot(n, "root_joint")
    ocp = simple_mpc.CentroidalOCP(s, mh)
    ocp.createProblem(np.zeros(9), horizon, 3, -9.81, False)
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_cent_oracle(batch, max_iters=1, horizon=50, settings_override=None, mpc_override=None):
    rb = O.Robot("go2_like")
    s = O.go2_centroidal_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    Cn = O.Cent(rb, s)
    return O.OracleCentMPC(Cn, ms, batch), rb, Cn


def make_cent_pair(batch, max_iters=1, lib=None, horizon=50, walk=(0.2, 0, 0, 0, 0, 0), **kw):
    om, rb, _ = make_cent_oracle(batch, max_iters, horizon, **kw)
    gm, _, _, _ = make_cent_product(batch, max_iters, lib, horizon, **kw)
    cs = O.trot_cycle()
    om.generateCycleHorizon(cs)
    gm.generateCycleHorizon(cs)
    v = np.array(walk, float)
    om.switchToWalk(v)
    gm.switchToWalk(v)
    return om, gm, rb


def make_full_product(batch, max_iters=1, lib=None, horizon=50, settings_override=None, mpc_override=None, device_id=0):
    """simple_mpc.BatchedMPC over a FullDynamicsOCP with the Go2 settings of record (oracle_lib.go2_full_settings)."""
    rb = O.Robot("go2_like")
    s = O.go2_full_settings(rb)
    s.update(dict(force_size=3, mu=0.8, Lfoot=0.01, Wfoot=0.01, force_cone=False, land_cstr=False))
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in FEET:
        mh.addPointFoot(n, "root_joint")
    ocp = simple_mpc.FullDynamicsOCP(s, mh)
    ocp.createProblem(mh.getReferenceState(), horizon, 3, -9.81, bool(ms.get("terminal_constraint", False)))
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_full_oracle(batch, max_iters=1, horizon=50, settings_override=None, mpc_override=None, walk=(0.2, 0, 0, 0, 0, 0)):
    rb = O.Robot("go2_like")
    s = O.go2_full_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.go2_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    F = O.Full(rb, s)
    om = O.OracleFullMPC(F, ms, batch)
    if walk is not None:
        om.generateCycleHorizon(O.trot_cycle())
        om.switchToWalk(np.array(walk, float))
    return om, rb


def make_full_pair(batch, max_iters=1, lib=None, horizon=50, walk=(0.2, 0, 0, 0, 0, 0), **kw):
    om, rb = make_full_oracle(batch, max_iters, horizon, walk=walk, **kw)
    gm, _, _, _ = make_full_product(batch, max_iters, lib, horizon, **kw)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array(walk, float))
    return om, gm, rb


TALOS_FEET = ["left_sole_link", "right_sole_link"]
TALOS_QUAD = np.array([[0.1, 0.075, 0], [-0.1, 0.075, 0], [-0.1, -0.075, 0], [0.1, -0.075, 0]])


def make_talos_product(batch, max_iters=1, lib=None, horizon=100, settings_override=None, mpc_override=None, device_id=0):
    """simple_mpc.BatchedMPC over the Talos full-dynamics OCP (6-D feet, wrench cones; oracle_lib.talos_full_settings)."""
    rb = O.Robot("talos_like")
    s = O.talos_full_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("talos_like", lib), "standing", "root_joint")
    for n in TALOS_FEET:
        mh.addQuadFoot(n, "root_joint", TALOS_QUAD)
    ocp = simple_mpc.FullDynamicsOCP(s, mh)
    ocp.createProblem(mh.getReferenceState(), horizon, 6, -9.81, bool(ms.get("terminal_constraint", False)))
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_talos_pair(batch, max_iters=1, lib=None, horizon=100, walk=(0.1, 0, 0, 0, 0, 0), cycle=None, **kw):
    rb = O.Robot("talos_like")
    s = O.talos_full_settings(rb)
    if kw.get("settings_override"):
        s.update(kw["settings_override"])
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if kw.get("mpc_override"):
        ms.update(kw["mpc_override"])
    om = O.OracleFullMPC(O.Full(rb, s), ms, batch)
    gm, _, _, _ = make_talos_product(batch, max_iters, lib, horizon, **kw)
    cs = O.walk_cycle() if cycle is None else cycle
    for m in (om, gm):
        m.generateCycleHorizon(cs)
        m.switchToWalk(np.array(walk, float))
    return om, gm, rb


def talos_random_states(rb, batch, seed=20240529, scale=1.0):
    rng = np.random.default_rng(seed)
    sg = np.concatenate([np.ones(3) * 0.02, np.ones(3) * 0.05, np.ones(rb.nv - 6) * 0.1, np.ones(3) * 0.1, np.ones(3) * 0.2, np.ones(rb.nv - 6) * 0.5])
    return np.stack([rb.integrate(rb.x_ref, rng.normal(size=rb.ndx) * sg * scale) for _ in range(batch)])


def alphas_agree(om, gm, rtol=1e-8):
    """Line-search decisions of oracle and product: identical, except where the Armijo test itself is decided by rounding -- then
    the side that accepted the larger step must have done so with a slack below rtol * |phi0| (info: phi0, dphi0, alpha, phi_new)."""
    io, ig = om.info, gm.info
    for b in range(io.shape[0]):
        ao, ag = io[b, 2], ig[b, 2]
        if ao == ag:
            continue
        big = io[b] if ao > ag else ig[b]
        slack = big[0] + 1e-4 * big[2] * big[1] - big[3]
        if not (0.0 <= slack <= rtol * max(1.0, abs(big[0]))):
            return False
    return True


def make_talos_kino_product(batch, max_iters=1, lib=None, horizon=100, settings_override=None, mpc_override=None, device_id=0):
    """simple_mpc.BatchedMPC over the Talos KINODYNAMICS OCP with 6-D feet (oracle_lib.talos_kino_settings: the weights of the reference's
    examples/talos_kinodynamics.py, force_cone as in its tests/test_utils.cpp)."""
    rb = O.Robot("talos_like")
    s = O.talos_kino_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("talos_like", lib), "standing", "root_joint")
    for n in TALOS_FEET:
        mh.addQuadFoot(n, "root_joint", TALOS_QUAD)
    ocp = simple_mpc.KinodynamicsOCP(s, mh)
    ocp.createProblem(mh.getReferenceState(), horizon, 6, -9.81, bool(ms.get("terminal_constraint", False)))
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_talos_kino_pair(batch, max_iters=1, lib=None, horizon=100, walk=(0.1, 0, 0, 0, 0, 0), cycle=None, **kw):
    rb = O.Robot("talos_like")
    s = O.talos_kino_settings(rb)
    if kw.get("settings_override"):
        s.update(kw["settings_override"])
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if kw.get("mpc_override"):
        ms.update(kw["mpc_override"])
    if ms.get("terminal_constraint", False):
        O.lib().orc_set_terminal_constraint(1)
    om = O.OracleMPC(O.Kino(rb, s), ms, batch)
    O.lib().orc_set_terminal_constraint(0)
    gm, _, _, _ = make_talos_kino_product(batch, max_iters, lib, horizon, **kw)
    cs = O.walk_cycle() if cycle is None else cycle
    for m in (om, gm):
        m.generateCycleHorizon(cs)
        m.switchToWalk(np.array(walk, float))
    return om, gm, rb


def make_talos_cent_product(batch, max_iters=1, lib=None, horizon=100, settings_override=None, mpc_override=None, device_id=0):
    """simple_mpc.BatchedMPC over the Talos CENTROIDAL OCP with 6-D feet (oracle_lib.talos_centroidal_settings: examples/talos_centroidal.py)."""
    rb = O.Robot("talos_like")
    s = O.talos_centroidal_settings(rb)
    if settings_override:
        s.update(settings_override)
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if mpc_override:
        ms.update(mpc_override)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("talos_like", lib), "standing", "root_joint")
    for n in TALOS_FEET:
        mh.addQuadFoot(n, "root_joint", TALOS_QUAD)
    ocp = simple_mpc.CentroidalOCP(s, mh)
    ocp.createProblem(np.zeros(9), horizon, 6, -9.81, False)
    conf = {k: ms[k] for k in MPC_KEYS}
    gm = simple_mpc.BatchedMPC(conf, ocp, batch, device_id=device_id, lib=lib)
    return gm, rb, s, ms


def make_talos_cent_pair(batch, max_iters=1, lib=None, horizon=100, walk=(0.1, 0, 0, 0, 0, 0), cycle=None, **kw):
    rb = O.Robot("talos_like")
    s = O.talos_centroidal_settings(rb)
    if kw.get("settings_override"):
        s.update(kw["settings_override"])
    ms = O.talos_mpc_settings(rb, max_iters=max_iters)
    ms["T"] = horizon
    if kw.get("mpc_override"):
        ms.update(kw["mpc_override"])
    om = O.OracleCentMPC(O.Cent(rb, s), ms, batch)
    gm, _, _, _ = make_talos_cent_product(batch, max_iters, lib, horizon, **kw)
    cs = O.walk_cycle() if cycle is None else cycle
    for m in (om, gm):
        m.generateCycleHorizon(cs)
        m.switchToWalk(np.array(walk, float))
    return om, gm, rb
