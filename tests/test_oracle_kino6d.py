"""Oracle checks of the kinodynamics stage with 6-D (flat) feet on the Talos-class robot: what reference src/kinodynamics.cpp:66-72
(FramePlacementResidual pose cost), :105-123 (6-row LOCAL frame velocity, CentroidalWrenchConeResidual) and the force_size == 6 branches
of Aligator's kinodynamics dynamics / momentum-derivative residual add.  Every Jacobian against central finite differences on the
manifold; the structural facts the reference's own test pins (tests/problem.cpp:139-140: 6 cost components, 3 constraint blocks with
one foot in contact)."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def rb():
    return O.Robot("talos_like")


@pytest.fixture(scope="module")
def kino(rb):
    return O.Kino(rb, O.talos_kino_settings(rb, force_cone=True))


def _randx(rb, rng, s=0.2):
    return rb.integrate(rb.x_ref, rng.normal(size=rb.ndx) * s)


def _randu(rb, rng):
    w = np.concatenate([rng.normal(size=3) * 30 + [0, 0, 400], rng.normal(size=3) * 5])
    w2 = np.concatenate([rng.normal(size=3) * 30 + [0, 0, 400], rng.normal(size=3) * 5])
    return np.concatenate([w, w2, rng.normal(size=rb.nv - 6) * 2])


def test_dimensions(rb, kino):
    nv = rb.nv
    assert kino.nu == nv - 6 + 12          # reference src/kinodynamics.cpp:34
    assert kino.nc == (nv - 6) + 12 + 34   # joint box | 6 frame-velocity rows per foot | 17 wrench-cone rows per foot


def test_momentum_balance_with_contact_torques(rb, kino):
    """d/dt hg along xdot = [m g + sum f ; sum (p - c) x f + tau] for the solved base acceleration."""
    rng = np.random.default_rng(2)
    x = _randx(rb, rng)
    u = _randu(rb, rng)
    for mask in (0b11, 0b01, 0b10):
        e = kino.eval(mask, np.zeros(kino.nu), rb.x_ref, np.zeros((2, 3)), x, u)
        a = e["xdot"][rb.nv:]
        c = rb.centroidal(x)
        hdot = c["Ag"] @ a + c["dAgv"]
        w = u[:12].reshape(2, 6)
        act = [(mask >> i) & 1 for i in range(2)]
        lin = rb.mass * np.array([0, 0, -9.81]) + sum(w[i, :3] for i in range(2) if act[i])
        ang = sum(np.cross(c["feet"][i] - c["com"], w[i, :3]) + w[i, 3:] for i in range(2) if act[i])
        assert np.abs(hdot[:3] - lin).max() < 1e-8
        assert np.abs(hdot[3:] - ang).max() < 1e-8
        assert np.abs(a[6:] - u[12:]).max() == 0.0


@pytest.mark.parametrize("mask", [0b11, 0b01, 0b10])
def test_stage_derivatives_vs_finite_differences(rb, kino, mask):
    rng = np.random.default_rng(20 + mask)
    x = _randx(rb, rng)
    u = _randu(rb, rng)
    u_ref = np.concatenate([[0, 0, 450.0, 0, 0, 0] * 2, np.zeros(rb.nv - 6)])
    x_tgt = _randx(rb, rng, 0.3)
    foot_ref = rng.normal(size=(2, 3)) * 0.2
    d = kino.deriv(mask, u_ref, x_tgt, foot_ref, x, u)
    e0 = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u)
    n, m = kino.ndx, kino.nu
    A, B = np.zeros((n, n)), np.zeros((n, m))
    lx, lu = np.zeros(n), np.zeros(m)
    Cx, Cu = np.zeros((kino.nc, n)), np.zeros((kino.nc, m))
    h = 1e-6
    for k in range(n):
        dd = np.zeros(n)
        dd[k] = h
        ep = kino.eval(mask, u_ref, x_tgt, foot_ref, rb.integrate(x, dd), u)
        em = kino.eval(mask, u_ref, x_tgt, foot_ref, rb.integrate(x, -dd), u)
        A[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lx[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cx[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for k in range(m):
        dd = np.zeros(m)
        dd[k] = h
        ep = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u + dd)
        em = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u - dd)
        B[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lu[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cu[:, k] = (ep["c"] - em["c"]) / (2 * h)
    assert np.abs(A - d["A"]).max() < 2e-7
    assert np.abs(B - d["B"]).max() < 1e-8
    assert np.abs(lx - d["lx"]).max() < 2e-6 * max(1, np.abs(d["lx"]).max())
    assert np.abs(lu - d["lu"]).max() < 2e-6 * max(1, np.abs(d["lu"]).max())
    assert np.abs(Cx - d["Cx"]).max() < 1e-6
    assert np.abs(Cu - d["Cu"]).max() < 2e-7  # (linear rows of size 1e2: the differences carry the rounding of c / h)
    na = rb.nv - 6
    for f in range(2):
        vel = slice(na + 6 * f, na + 6 * f + 6)
        cone = slice(na + 12 + 17 * f, na + 12 + 17 * f + 17)
        if (mask >> f) & 1:
            assert np.abs(d["Cx"][vel]).max() > 0       # 6 frame-velocity rows: state only
            assert np.all(d["Cu"][vel] == 0)
            assert np.all(d["Cx"][cone] == 0)           # wrench-cone rows: constant, control only
            assert np.abs(d["Cu"][cone]).max() > 0
        else:                                           # a swinging foot has no rows
            assert np.all(d["Cx"][vel] == 0) and np.all(d["Cu"][cone] == 0)
            assert np.all(e0["c"][vel] == 0) and np.all(e0["c"][cone] == 0)


def test_wrench_cone_rows_are_the_cone_of_a_rectangular_sole(rb, kino):
    """A wrench is inside the cone iff unilateral, inside the friction pyramid, centre of pressure inside the 2L x 2W sole and the
    yaw torque within its bound (Caron et al. 2015): checked on wrenches built from four corner forces."""
    rng = np.random.default_rng(7)
    s = kino.s
    mu, L, W = s["mu"], s["Lfoot"], s["Wfoot"]
    corners = np.array([[L, W, 0], [L, -W, 0], [-L, W, 0], [-L, -W, 0]])
    x = rb.x_ref
    for trial in range(50):
        fz = rng.uniform(0.1, 100, size=4)
        ft = rng.uniform(-1, 1, size=(4, 2)) * (mu / np.sqrt(2)) * fz[:, None] * (0.99 if trial % 2 == 0 else 3.0)
        fc = np.concatenate([ft, fz[:, None]], axis=1)
        wrench = np.concatenate([fc.sum(0), sum(np.cross(corners[i], fc[i]) for i in range(4))])
        u = np.concatenate([wrench, np.zeros(6), np.zeros(rb.nv - 6)])
        c = kino.eval(0b01, np.zeros(kino.nu), x, np.zeros((2, 3)), x, u)["c"]
        rows = c[rb.nv - 6 + 12: rb.nv - 6 + 12 + 17]
        if trial % 2 == 0:
            assert rows.max() <= 1e-9          # forces inside their friction pyramids at the corners: inside the wrench cone
    # clearly outside: pulling force, sliding force, centre of pressure beyond the edge
    for wrench in ([0, 0, -10, 0, 0, 0], [100, 0, 10, 0, 0, 0], [0, 0, 10, 0, 10 * L * 1.5, 0], [0, 0, 10, 10 * W * 1.5, 0, 0]):
        u = np.concatenate([wrench, np.zeros(6), np.zeros(rb.nv - 6)])
        c = kino.eval(0b01, np.zeros(kino.nu), x, np.zeros((2, 3)), x, u)["c"]
        assert c[rb.nv - 6 + 12: rb.nv - 6 + 12 + 17].max() > 0


def test_pose_residual_is_a_placement_error(rb, kino):
    """log6(M_ref^-1 oMf) with M_ref = (identity, reference translation): zero translation error for a flat foot standing on its
    reference, and the angular part follows a tilt of the base."""
    c = rb.centroidal(rb.x_ref)
    feet = c["feet"]
    e = kino.eval(0b11, np.zeros(kino.nu), rb.x_ref, feet, rb.x_ref, np.zeros(kino.nu))
    d = kino.deriv(0b11, np.zeros(kino.nu), rb.x_ref, feet, rb.x_ref, np.zeros(kino.nu))
    # at the reference state the soles are flat: the only cost left is the momentum / control part, and the pose gradient vanishes
    dx = np.zeros(rb.ndx)
    dx[3] = 0.05  # roll of the base tilts both feet
    xt = rb.integrate(rb.x_ref, dx)
    et = kino.eval(0b11, np.zeros(kino.nu), rb.x_ref, feet, xt, np.zeros(kino.nu))
    assert et["cost"] > e["cost"] + 0.5 * 100000.0 * 2 * 0.05 ** 2 * 0.9  # two feet, w_frame 1e5 on the angular error alone
    assert np.abs(d["lx"]).max() < 1e-6 * 100000.0


def test_gauss_newton_hessian_is_psd_and_symmetric(rb, kino):
    rng = np.random.default_rng(5)
    x = _randx(rb, rng)
    u = _randu(rb, rng)
    d = kino.deriv(3, np.zeros(kino.nu), rb.x_ref, np.zeros((2, 3)), x, u)
    Hm = np.block([[d["Lxx"], d["Lxu"]], [d["Lxu"].T, d["Luu"]]])
    assert np.abs(Hm - Hm.T).max() < 1e-6
    assert np.linalg.eigvalsh(Hm).min() > -1e-6 * np.abs(Hm).max()


def test_constraint_blocks_of_the_reference_test():
    """reference tests/problem.cpp:114-140: left foot in contact, right foot swinging, settings of tests/test_utils.cpp (force_cone on):
    cost stack of 6 components, 3 constraint blocks (joint box, wrench cone, frame velocity)."""
    rb = O.Robot("talos_like")
    kino = O.Kino(rb, O.talos_kino_settings(rb, force_cone=True))
    na = rb.nv - 6
    c = kino.eval(0b01, np.zeros(kino.nu), rb.x_ref, np.zeros((2, 3)), rb.x_ref, np.ones(kino.nu))["c"]
    d = kino.deriv(0b01, np.zeros(kino.nu), rb.x_ref, np.zeros((2, 3)), rb.x_ref, np.ones(kino.nu))
    present = (np.abs(d["Cx"]).sum(1) + np.abs(d["Cu"]).sum(1)) > 0
    blocks = [present[:na].all(), present[na:na + 6].all(), present[na + 12:na + 12 + 17].all()]
    absent = [not present[na + 6:na + 12].any(), not present[na + 12 + 17:].any()]
    assert all(blocks) and all(absent)
    assert sum(blocks) == 3
    components = 4 + rb.nf  # state, control, centroidal, centroidal_derivative, one pose cost per foot
    assert components == 6
    assert c.shape == (kino.nc,)
