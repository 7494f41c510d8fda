"""CPU tier: the drop-in boundary.  libsmpc_hip.so loads and exports every symbol include/smpc.h declares
(no compute call without a GPU), the product fails loudly without a device / without the library, and the
Python surface mirrors the reference's names, dict keys and error behaviour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc
from simple_mpc import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "smpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smpc_[A-Za-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol(built):
    lib = C.CDLL(_capi.DEFAULT_LIB)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_capi.SYMBOLS) == syms, "python binding list out of sync with include/smpc.h"


def test_product_fails_loudly_without_gpu(built):
    lib = _capi.SmpcLib(_capi.DEFAULT_LIB)
    if lib.L.smpc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        S.make_product(1)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.SmpcLib(str(tmp_path / "libsmpc_hip.so"))


def test_product_sources_do_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "simple-mpc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".h", ".cpp", ".py", ".hip")):
                txt = open(os.path.join(dp, f)).read()
                for needle in ("liborc", "oracle_lib", "orc_capi", "orc_mpc", "oracle/", "libsmpc_emu", "SMPC_CPU_EMU"):
                    assert needle not in txt, (f, needle)


def test_builtin_robot_table_via_abi(built):
    lib = S.emu_lib()
    m = simple_mpc.load_robot("go2_like", lib).contents
    assert (m.njoints, m.nq, m.nv, m.nfeet) == (13, 19, 18, 4)
    assert [m.foot_name[i].value.decode() for i in range(4)] == S.FEET
    with pytest.raises(RuntimeError):
        simple_mpc.load_robot("does_not_exist", lib)


def test_settings_dict_keys_and_errors(built):
    """reference bindings/expose-kinodynamics.cpp:12-31 and expose-mpc.cpp:28-45: a missing key is a KeyError;
    size / order violations are RuntimeError (reference throws std::runtime_error)."""
    lib = S.emu_lib()
    rb = O.Robot("go2_like")
    s = O.go2_kino_settings(rb)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    assert mh.getFeetNb() == 4 and mh.getFootFrameName(2) == "RL_foot" and abs(mh.getMass() - rb.mass) < 1e-12
    assert np.array_equal(mh.getReferenceState(), rb.x_ref)
    bad = dict(s)
    del bad["w_centder"]
    with pytest.raises(KeyError):
        simple_mpc.KinodynamicsOCP(bad, mh)
    ocp = simple_mpc.KinodynamicsOCP(s, mh)
    assert ocp.getNu() == 24  # nv - 6 + force_size * nfeet (reference src/kinodynamics.cpp:34)
    ms = O.go2_mpc_settings(rb)
    conf = {k: ms[k] for k in S.MPC_KEYS}
    with pytest.raises(RuntimeError, match="Create problem first"):
        simple_mpc.BatchedMPC(conf, ocp, 1, lib=lib)
    with pytest.raises(RuntimeError, match="force size"):
        ocp.createProblem(rb.x_ref, 50, 6, -9.81, False)
    ocp.createProblem(rb.x_ref, 50, 3, -9.81, False)
    assert ocp.getSize() == 50
    c2 = dict(conf)
    del c2["T_fly"]
    with pytest.raises(KeyError):
        simple_mpc.BatchedMPC(c2, ocp, 1, lib=lib)
    s_bad = dict(s, w_x=np.eye(10))
    ocp_bad = simple_mpc.KinodynamicsOCP(s_bad, mh)
    ocp_bad.createProblem(rb.x_ref, 50, 3, -9.81, False)
    with pytest.raises(RuntimeError, match="w_x"):
        simple_mpc.BatchedMPC(conf, ocp_bad, 1, lib=lib)
    # every option of the kinodynamics OCP with 3-D feet is built (tests/test_kino_force_cone.py, test_kino_land_cstr.py,
    # test_terminal_constraint.py); 6-D feet are not: the reference's own size check fires first, as there
    s_all = dict(s, land_cstr=True, force_cone=True)
    ocp_all = simple_mpc.KinodynamicsOCP(s_all, mh)
    ocp_all.createProblem(rb.x_ref, 50, 3, -9.81, True)
    assert simple_mpc.BatchedMPC(conf, ocp_all, 1, lib=lib).nc == 24
    with pytest.raises(RuntimeError, match="force size"):
        ocp_all.createProblem(rb.x_ref, 50, 6, -9.81, False)


def test_mpc_single_instance_surface(built):
    """MPC(dict, ocp).iterate(x); xs/us/Ks as lists (reference bindings/expose-mpc.cpp:73-106)."""
    lib = S.emu_lib()
    rb = O.Robot("go2_like")
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    ocp = simple_mpc.KinodynamicsOCP(O.go2_kino_settings(rb), mh)
    ocp.createProblem(rb.x_ref, 50, 3, -9.81, False)
    ms = O.go2_mpc_settings(rb)
    mpc = simple_mpc.MPC({k: ms[k] for k in S.MPC_KEYS}, ocp, lib=lib)
    with pytest.raises(RuntimeError, match="generateCycleHorizon"):
        mpc.iterate(rb.x_ref)
    cs = O.trot_cycle()
    mpc.generateCycleHorizon([{n: bool(row[i]) for i, n in enumerate(S.FEET)} for row in cs])
    mpc.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    with pytest.raises(RuntimeError):
        mpc.iterate(np.zeros(5))
    mpc.iterate(rb.x_ref)
    assert len(mpc.xs) == 51 and len(mpc.us) == 50 and len(mpc.Ks) == 50
    assert mpc.xs[0].shape == (37,) and mpc.us[0].shape == (24,) and mpc.Ks[0].shape == (24, 36)
    assert np.allclose(mpc.xs[0], rb.x_ref)
    assert mpc.getStateDerivative(0).shape == (36,)
    assert mpc.getFootTakeoffCycle("FL_foot") == [59]
    mpc.x_reference = rb.x_ref
    with pytest.raises(RuntimeError):
        mpc.x_reference = np.zeros(3)
    assert set(S.MPC_KEYS) <= set(mpc.getSettings())


def test_c_abi_argument_validation(built):
    lib = S.emu_lib()
    L = lib.L
    h = C.c_void_p()
    assert L.smpc_create(None, None, None, 1, -9.81, 0, C.byref(h)) < 0
    assert b"null" in L.smpc_last_error()
    gm, rb, _, _ = S.make_product(1, lib=lib)
    out = np.zeros(gm._lib.L.smpc_lq_size(gm._h))
    assert L.smpc_debug_get_lq(gm._h, 0, 50, out) < 0  # reference: "Stage index exceeds stage vector size"
    assert b"Stage index" in L.smpc_last_error()
    buf = np.zeros(8, np.int32)
    assert L.smpc_get_foot_timing(gm._h, 0, 0, buf, 8) < 0  # cycle horizon not generated yet


def test_status_word_flags_failed_instances(built):
    """smpc_get_status: healthy instances report 0; an instance fed a non-finite measured state is flagged (and only that one)."""
    lib = S.emu_lib()
    gm, rb, _, _ = S.make_product(3, max_iters=1, lib=lib, horizon=20)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 3)
    gm.iterate(X)
    assert np.array_equal(gm.status, [0, 0, 0])
    X[1, 9] = np.nan
    gm.iterate(X)
    st = gm.status
    assert st[1] & 1 and st[0] == 0 and st[2] == 0
    assert np.all(np.isfinite(gm.xs[0])) and np.all(np.isfinite(gm.xs[2]))  # the others are untouched
