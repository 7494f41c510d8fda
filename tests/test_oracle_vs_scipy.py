"""The oracle's CONVERGED answer against an independent NLP solver (round-5 hardening of SURVEY 8c; the row stays "parity unpinned":
Aligator is not in this image).

Every other oracle test checks pieces -- Jacobians against finite differences, the Riccati sweep against a dense KKT solve, identities of
the dynamics.  Here a small optimal-control problem of each family is solved twice: by the oracle's own ProxDDP (augmented Lagrangian,
proximal Riccati, line search -- orc_proxddp.hpp) run to convergence, and by scipy.optimize.minimize (SLSQP, an SQP method that shares no
code and no solver constant with it) on the SAME costs, dynamics and constraint rows, taken from the oracle's stage evaluation only
(Jacobians by scipy's own finite differences: the oracle's derivative code is not used on the scipy side).  Both must find the same
optimum.  That pins, independently of the solver: the sign and typing conventions of the constraint rows (equality / box / <= 0), the way
dynamics defects and multipliers enter the Lagrangian, the terminal cost, and that the proximal / AL iteration converges to a KKT point of
the problem the stage model states -- not to the solution of a subtly different one."""
import numpy as np
import pytest
from scipy.optimize import minimize

import oracle_lib as O


@pytest.fixture(scope="module")
def rb():
    return O.Robot("go2_like")


def _solve_slsqp(fun, cons, z0, maxiter=400, ftol=1e-12):
    r = minimize(fun, z0, method="SLSQP", constraints=cons, options=dict(maxiter=maxiter, ftol=ftol))
    # (status 9 = iteration limit: on finite-difference gradients SLSQP may never meet `ftol` although it has long stopped moving; what it
    #  found is judged by the comparisons of the caller)
    assert r.success or r.status == 9, r.message
    return r


# ------------------------------------------------------------------------------------------------ centroidal OCP
def _cent_problem(rb, mu_fric):
    s = O.go2_centroidal_settings(rb)
    s["mu"] = mu_fric
    s["w_com"] = np.diag([2e3, 2e3, 2e3])  # (a stiff CoM target makes the problem pull sideways: with mu = 0.25 the friction cones become active)
    cent = O.Cent(rb, s)
    H, nf = 6, rb.nf
    masks = [0b1111, 0b1111, 0b1001, 0b1001, 0b1111, 0b1111]
    com0 = rb.centroidal(rb.x_ref)["com"]
    pos = np.asarray(rb.centroidal(rb.x_ref)["feet"], float).reshape(nf, 3)
    u_ref = np.zeros((H, 3 * nf))
    for t, m in enumerate(masks):
        on = [(m >> f) & 1 for f in range(nf)]
        for f in range(nf):
            if on[f]:
                u_ref[t, 3 * f + 2] = rb.mass * 9.81 / sum(on)
    x_tgt = np.zeros((H, 9))
    x_tgt[:, :3] = com0 + np.array([0.10, 0.30, 0.0])  # sideways pull
    x0 = np.r_[com0 + np.array([0.0, -0.01, 0.005]), 0.3, -0.2, 0.0, 0.02, 0.0, 0.01]
    return cent, masks, u_ref, x_tgt, np.tile(pos.ravel(), (H, 1)), x0


def _cent_nlp(cent, masks, u_ref, x_tgt, pos, x0):
    """z = [u_0, x_1, u_1, ..., x_H]; objective, equality rows (dynamics), inequality rows (>= 0 in scipy's convention)."""
    H, nu = len(masks), cent.nu
    kinds = [cent.row_kinds(m)[0] for m in masks]

    def split(z):
        z = z.reshape(H, nu + 9)
        xs = np.vstack([x0[None, :], z[:, nu:]])
        return xs, z[:, :nu]

    def evals(z):
        xs, us = split(z)
        return xs, us, [cent.eval(masks[t], u_ref[t], x_tgt[t], pos[t], xs[t], us[t]) for t in range(H)]

    def cost(z):
        xs, us, ev = evals(z)
        return sum(e["cost"] for e in ev) + cent.term(xs[H])[0]

    def defects(z):
        xs, us, ev = evals(z)
        return np.concatenate([ev[t]["xnext"] - xs[t + 1] for t in range(H)])

    def ineq(z):
        xs, us, ev = evals(z)
        out = []
        for t in range(H):
            assert set(kinds[t]) <= {0, 3}  # centroidal OCP: friction-cone rows of the feet in contact only, all of type c <= 0
            out.append(-ev[t]["c"][kinds[t] == 3])
        return np.concatenate(out)

    return split, cost, [dict(type="eq", fun=defects), dict(type="ineq", fun=ineq)], kinds


@pytest.mark.parametrize("mu_fric", [0.8, 0.25])
def test_centroidal_optimum_is_the_sqp_optimum(rb, mu_fric):
    cent, masks, u_ref, x_tgt, pos, x0 = _cent_problem(rb, mu_fric)
    H = len(masks)
    sol = cent.solve(masks, u_ref, x_tgt, pos, x0, u_ref[0], max_iter=300, tol=1e-10)
    assert sol["trace"][-1, 0] < 1e-7 and sol["trace"][-1, 1] < 1e-5, sol["trace"][-3:]  # (dual residual: gradient units, weights up to 2e4)
    split, cost, cons, kinds = _cent_nlp(cent, masks, u_ref, x_tgt, pos, x0)
    z_or = np.hstack([np.hstack([sol["us"][t], sol["xs"][t + 1]]) for t in range(H)])
    # the oracle's point is feasible for the NLP as scipy sees it ...
    assert np.abs(cons[0]["fun"](z_or)).max() < 1e-7 and cons[1]["fun"](z_or).min() > -1e-6
    # ... and SLSQP, started from the constant guess the oracle started from, arrives at the same optimum
    z0 = np.hstack([np.hstack([u_ref[0], x0]) for _ in range(H)])
    r = _solve_slsqp(cost, cons, z0)
    xs_s, us_s = split(r.x)
    assert abs(r.fun - cost(z_or)) < 1e-6 * max(1.0, abs(r.fun)), (r.fun, cost(z_or))
    # (agreement to the accuracy of an SQP method on finite-difference gradients: the momentum weights of record are 1e-2, so the optimum is
    #  flat in those directions -- the costs above agree to 1e-6, the arguments to 1e-3 of their size)
    assert np.abs(xs_s - sol["xs"]).max() < 1e-3 * max(1.0, np.abs(xs_s).max()) and np.abs(us_s - sol["us"]).max() < 1e-2 * max(1.0, np.abs(us_s).max()), (
        np.abs(xs_s - sol["xs"]).max(), np.abs(us_s - sol["us"]).max())
    if mu_fric < 0.5:
        # the tight cone is active at the optimum (otherwise this case would not test the constraint rows), on both sides alike, and the
        # oracle's multipliers of the active rows are positive (rows are c <= 0 with nu >= 0)
        act = cons[1]["fun"](r.x) < 1e-6
        assert act.sum() >= 2
        assert (cons[1]["fun"](z_or)[act] < 1e-5).all()
        vs = np.concatenate([sol["vs"][t][kinds[t] == 3] for t in range(H)])
        assert (vs[act] > 1e-6).all() and (vs > -1e-9).all()


# ------------------------------------------------------------------------------------------------ kinodynamics OCP
def _kino_problem(rb):
    s = O.go2_kino_settings(rb)
    kino = O.Kino(rb, s)
    H = 3
    masks = [0b1111, 0b1001, 0b1111]
    nf = rb.nf
    feet = np.asarray(rb.centroidal(rb.x_ref)["feet"], float)  # the frame-translation cost pulls every foot to its place in the reference posture
    u_ref = np.zeros((H, kino.nu))
    for t, m in enumerate(masks):
        on = [(m >> f) & 1 for f in range(nf)]
        for f in range(nf):
            if on[f]:
                u_ref[t, 3 * f + 2] = rb.mass * 9.81 / sum(on)
    x_tgt = np.tile(rb.x_ref, (H, 1))
    dx0 = np.zeros(kino.ndx)
    dx0[[0, 1, 2]] = [0.01, -0.01, -0.01]
    dx0[6:18] = 0.03 * np.sin(np.arange(12))
    # (velocities stay zero: x_0 is fixed, and the frame-velocity rows of the feet in contact at t = 0 depend on it alone)
    x0 = rb.integrate(rb.x_ref, dx0)
    return kino, masks, u_ref, x_tgt, feet, x0


def test_kinodynamics_optimum_is_the_sqp_optimum(rb):
    kino, masks, u_ref, x_tgt, feet, x0 = _kino_problem(rb)
    H, nu, ndx = len(masks), kino.nu, kino.ndx
    foot_ref = np.tile(np.asarray(feet, float).ravel(), (H, 1))
    sol = kino.solve(masks, u_ref, x_tgt, foot_ref, rb.x_ref, x0, u_ref[0], max_iter=300, tol=1e-9)
    assert max(sol["trace"][-1, 0], sol["trace"][-1, 1]) < 1e-6, sol["trace"][-3:]
    kinds = [kino.row_kinds(m) for m in masks]

    # z = [u_0, dx_1, u_1, ..., dx_H] with x_t = x_ref (+) dx_t: the state lives on a manifold, scipy sees its tangent coordinates
    def split(z):
        z = z.reshape(H, nu + ndx)
        xs = [x0] + [rb.integrate(rb.x_ref, z[t, nu:]) for t in range(H)]
        return xs, z[:, :nu]

    def evals(z):
        xs, us = split(z)
        return xs, us, [kino.eval(masks[t], u_ref[t], x_tgt[t], foot_ref[t], xs[t], us[t]) for t in range(H)]

    def cost(z):
        xs, us, ev = evals(z)
        return sum(e["cost"] for e in ev) + kino.term(rb.x_ref, xs[H])[0]

    def eqs(z):
        xs, us, ev = evals(z)
        out = [rb.difference(xs[t + 1], ev[t]["xnext"]) for t in range(H)]
        # (the rows of this OCP depend on the state alone -- frame velocities, joint box: at t = 0 they are constants of the fixed x_0,
        #  satisfied by construction, and a constant equality row would make the SQP subproblem singular)
        out += [ev[t]["c"][kinds[t][0] == 1] for t in range(1, H)]
        return np.concatenate(out)

    def ineq(z):
        xs, us, ev = evals(z)
        out = []
        for t in range(1, H):
            k, lo, hi = kinds[t]
            c = ev[t]["c"]
            out += [-c[k == 3], c[k == 2] - lo[k == 2], hi[k == 2] - c[k == 2]]
        return np.concatenate(out)

    cons = [dict(type="eq", fun=eqs), dict(type="ineq", fun=ineq)]
    z_or = np.hstack([np.hstack([sol["us"][t], rb.difference(rb.x_ref, sol["xs"][t + 1])]) for t in range(H)])
    assert np.abs(eqs(z_or)).max() < 1e-6 and ineq(z_or).min() > -1e-6
    # SQP from the oracle's own start would need hundreds of finite-difference Jacobians of a 180-variable problem; started NEAR the
    # oracle's answer (perturbed by 1e-2 in every coordinate) it must come back to it: the point is a strict local optimum of the NLP
    rng = np.random.default_rng(0)
    r = _solve_slsqp(cost, cons, z_or + 1e-2 * rng.normal(size=z_or.size) * np.maximum(1e-1, np.abs(z_or)), maxiter=120, ftol=1e-10)
    assert abs(r.fun - cost(z_or)) < 1e-6 * max(1.0, abs(r.fun)), (r.fun, cost(z_or))
    xs_s, us_s = split(r.x)
    dxs = max(np.abs(rb.difference(np.asarray(a), np.asarray(b))).max() for a, b in zip(xs_s, sol["xs"]))
    # (the cost above agrees to 1e-6; the force distribution over the feet is weakly determined -- w_u = 1e-4 .. 1e-3 -- so the arguments agree
    #  to what 120 SQP steps on finite-difference gradients resolve in those flat directions)
    assert dxs < 5e-3 and np.abs(us_s - sol["us"]).max() < 2e-2 * max(1.0, np.abs(us_s).max()), (dxs, np.abs(us_s - sol["us"]).max())


# ------------------------------------------------------------------------------------------------ full-dynamics OCP (round 6)
def test_fulldynamics_optimum_is_the_sqp_optimum(rb):
    """Go2, H = 3, two feet leave the ground in the middle stage (a contact switch: the constrained forward dynamics changes its constraint set
    between stages), joint torques limited to 85 % of the largest one the unconstrained optimum takes -- box rows on u active at the optimum.
    The oracle's ProxDDP answer (orc_full_solve: constrained dynamics with Baumgarte terms, semi-implicit Euler, state / control / centroidal /
    force costs, terminal cost) against SLSQP on the stage evaluations alone.  The stage references of orc_full_solve are the same at every stage."""
    s = O.go2_full_settings(rb)
    full0 = O.Full(rb, s)
    H, masks = 3, [0b1111, 0b1001, 0b1111]
    nu, ndx = full0.nu, full0.ndx
    feet = np.asarray(rb.centroidal(rb.x_ref)["feet"], float).ravel()
    u_ref = np.zeros(nu + 3 * rb.nf)  # [torque reference ; force reference per foot]
    for f in range(rb.nf):
        u_ref[nu + 3 * f + 2] = rb.mass * 9.81 / rb.nf
    dx0 = np.zeros(ndx)
    dx0[[0, 1, 2]] = [0.01, -0.01, -0.01]
    dx0[6:18] = 0.03 * np.sin(np.arange(12))
    x0 = rb.integrate(rb.x_ref, dx0)
    u0 = np.zeros(nu)
    free = full0.solve(masks, u_ref, rb.x_ref, feet, x0, u0, max_iter=300, tol=1e-8)
    assert free["trace"][-1, 0] < 1e-6 and free["trace"][-1, 1] < 1e-4, free["trace"][-3:]  # (stops on the merit stall rule; dual residual in gradient units)
    lim = 0.85 * np.abs(free["us"]).max()
    s["umin"], s["umax"] = -np.full(nu, lim), np.full(nu, lim)
    full = O.Full(rb, s)
    sol = full.solve(masks, u_ref, rb.x_ref, feet, x0, u0, max_iter=400, tol=1e-8)
    assert sol["trace"][-1, 0] < 1e-6 and sol["trace"][-1, 1] < 1e-4, sol["trace"][-3:]
    kinds = [full.row_kinds(m) for m in masks]

    def split(z):
        z = z.reshape(H, nu + ndx)
        xs = [x0] + [rb.integrate(rb.x_ref, z[t, nu:]) for t in range(H)]
        return xs, z[:, :nu]

    def evals(z):
        xs, us = split(z)
        return xs, us, [full.eval(masks[t], u_ref, rb.x_ref, feet, xs[t], us[t]) for t in range(H)]

    def cost(z):
        xs, us, ev = evals(z)
        return sum(e["cost"] for e in ev) + full.term(rb.x_ref, xs[H])

    def eqs(z):
        xs, us, ev = evals(z)
        return np.concatenate([rb.difference(xs[t + 1], ev[t]["xnext"]) for t in range(H)])

    def ineq(z):
        xs, us, ev = evals(z)
        out = []
        for t in range(H):
            k, lo, hi = kinds[t]
            assert set(k) <= {0, 2}  # (this configuration: box rows only -- torque limits on u, joint limits on q)
            c = ev[t]["c"]
            # (the joint-limit rows of t = 0 are constants of the fixed x_0: harmless as inequalities, left in)
            out += [c[k == 2] - lo[k == 2], hi[k == 2] - c[k == 2]]
        return np.concatenate(out)

    cons = [dict(type="eq", fun=eqs), dict(type="ineq", fun=ineq)]
    z_or = np.hstack([np.hstack([sol["us"][t], rb.difference(rb.x_ref, sol["xs"][t + 1])]) for t in range(H)])
    assert np.abs(eqs(z_or)).max() < 1e-6 and ineq(z_or).min() > -1e-6
    assert (np.abs(sol["us"]) > lim - 1e-6).sum() >= 1, "the scenario must end on a torque limit"
    assert cost(z_or) > free["trace"][-1, 2] + 1e-9  # (the limit costs something: it is active, not decorative)
    # The oracle stops on its merit stall rule with dynamics defects of 8e-7; the dynamics multipliers of this problem are O(1e2) (frame weights
    # of 1e3), so that residual is worth 2.5e-4 of cost -- SLSQP, which closes the defects to 1e-12, must be compared with the oracle's controls
    # ROLLED OUT through the dynamics (an exactly feasible point next to the oracle's)
    xr = [x0]
    for t in range(H):
        xr.append(full.eval(masks[t], u_ref, rb.x_ref, feet, xr[t], sol["us"][t])["xnext"])
    z_roll = np.hstack([np.hstack([sol["us"][t], rb.difference(rb.x_ref, xr[t + 1])]) for t in range(H)])
    assert np.abs(eqs(z_roll)).max() < 1e-10 and ineq(z_roll).min() > -1e-6
    rng = np.random.default_rng(1)
    r = _solve_slsqp(cost, cons, z_or + 1e-2 * rng.normal(size=z_or.size) * np.maximum(1e-1, np.abs(z_or)), maxiter=150, ftol=1e-10)
    assert abs(r.fun - cost(z_roll)) < 2e-6 * max(1.0, abs(r.fun)), (r.fun, cost(z_roll), cost(z_or))
    xs_s, us_s = split(r.x)
    dxs = max(np.abs(rb.difference(np.asarray(a), np.asarray(b))).max() for a, b in zip(xs_s, sol["xs"]))
    assert dxs < 5e-3 and np.abs(us_s - sol["us"]).max() < 2e-2 * max(1.0, np.abs(us_s).max()), (dxs, np.abs(us_s - sol["us"]).max())
    # the same rows are active on both sides
    act_s = np.abs(us_s) > lim - 1e-5
    assert (act_s == (np.abs(sol["us"]) > lim - 1e-5)).all()


# ------------------------------------------------------------------------------------------------ centroidal OCP with 6-D feet (round 6)
def test_centroidal_6d_optimum_is_the_sqp_optimum():
    """Biped with flat feet (talos_like table, 6-D contact wrenches, the 17-row wrench cone of oracle/orc_kino.hpp per foot): H = 6 with a
    single-support phase, a sideways pull on the CoM and a low friction coefficient -- wrench-cone rows active at the optimum.  Pins the sign /
    typing of those 17 rows and the 6-D wrench's entry into the momentum dynamics against SLSQP."""
    rb6 = O.Robot("talos_like")
    s = O.talos_centroidal_settings(rb6)
    s["mu"] = 0.05  # (a 91 kg biped: the pull below needs more tangential force than mu = 0.05 lets the soles transmit)
    s["w_com"] = np.diag([2e4, 2e4, 2e4])
    cent = O.Cent(rb6, s)
    assert cent.fs == 6 and cent.nc == 17 * rb6.nf
    H, nf = 6, rb6.nf
    masks = [0b11, 0b11, 0b01, 0b01, 0b11, 0b11]
    com0 = rb6.centroidal(rb6.x_ref)["com"]
    pos = np.asarray(rb6.centroidal(rb6.x_ref)["feet"], float).reshape(nf, 3)
    u_ref = np.zeros((H, 6 * nf))
    for t, m in enumerate(masks):
        on = [(m >> f) & 1 for f in range(nf)]
        for f in range(nf):
            if on[f]:
                u_ref[t, 6 * f + 2] = rb6.mass * 9.81 / sum(on)
    x_tgt = np.zeros((H, 9))
    x_tgt[:, :3] = com0 + np.array([0.20, 0.60, 0.0])
    x0 = np.r_[com0 + np.array([0.0, -0.01, 0.005]), 3.0, -2.0, 0.0, 0.2, 0.0, 0.1]
    posr = np.tile(pos.ravel(), (H, 1))
    sol = cent.solve(masks, u_ref, x_tgt, posr, x0, u_ref[0], max_iter=400, tol=1e-10)
    assert sol["trace"][-1, 0] < 1e-7 and sol["trace"][-1, 1] < 1e-4, sol["trace"][-3:]
    split, cost, cons, kinds = _cent_nlp(cent, masks, u_ref, x_tgt, posr, x0)
    z_or = np.hstack([np.hstack([sol["us"][t], sol["xs"][t + 1]]) for t in range(H)])
    assert np.abs(cons[0]["fun"](z_or)).max() < 1e-7 and cons[1]["fun"](z_or).min() > -1e-6
    z0 = np.hstack([np.hstack([u_ref[0], x0]) for _ in range(H)])
    r = _solve_slsqp(cost, cons, z0, maxiter=600)
    xs_s, us_s = split(r.x)
    assert abs(r.fun - cost(z_or)) < 1e-6 * max(1.0, abs(r.fun)), (r.fun, cost(z_or))
    assert np.abs(xs_s - sol["xs"]).max() < 1e-3 * max(1.0, np.abs(xs_s).max()) and np.abs(us_s - sol["us"]).max() < 1e-2 * max(1.0, np.abs(us_s).max()), (
        np.abs(xs_s - sol["xs"]).max(), np.abs(us_s - sol["us"]).max())
    act = cons[1]["fun"](r.x) < 1e-6
    assert act.sum() >= 2, "the scenario must end on the wrench cone"
    assert (cons[1]["fun"](z_or)[act] < 1e-5).all()
    vs = np.concatenate([sol["vs"][t][kinds[t] == 3] for t in range(H)])
    assert (vs[act] > 1e-6).all() and (vs > -1e-9).all()
