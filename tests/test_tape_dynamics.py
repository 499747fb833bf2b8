"""General dynamics: the expression system (include/cascade/expression.hpp), its lowering to a
tape, and the CPU oracle's tape interpreter (oracle/orc.hpp::tape_compute_tcs).

The reference hands kw::dyn to heyoka's JIT (src/sim_basic.cpp:405-469, 553-577); heyoka is not
available, so nothing here can be compared with the reference bit for bit ("parity unpinned").
What pins the tape path instead:
  * the SAME right-hand sides written independently (numpy / sympy-derived gradients of the
    reference's potentials, cascade.py/dynamics/_simple_earth.py:136-409, benchmark/haumea.cpp:
    83-139) and integrated with scipy's DOP853 must give the same trajectories;
  * Kepler / J2 / drag through the tape must agree with the oracle's hand-written recurrences,
    which are pinned by the reference's known-answer tests (tests/test_oracle_kat.py).
"""
import numpy as np
import pytest

GM, RE = 3.986004407799724e14, 6378136.3
DEG = np.pi / 180


@pytest.fixture(scope="module")
def orc():
    from oracle import orc_py

    orc_py.build()
    return orc_py


def test_expression_algebra_and_errors():
    from cascade_amd import dynamics as D

    x, y = D.make_vars("x", "y")
    assert repr(x + 0.0) == "x" and repr(1.0 * x) == "x" and repr(x * 0.0) == "0"
    assert repr(D.diff(x * x, x)) == "(x + x)"
    assert repr(D.diff(D.sin(x * y), y)) == "(cos((x * y)) * x)"
    assert repr(D.diff(D.pow(x, -1.5), x)) == "(-1.5 * x**-2.5)"
    with pytest.raises(ValueError, match="constant exponents"):
        x**y
    dyn = D.kepler(2.0)
    assert len(dyn) == 6 and dyn.kind == 0
    lhs, rhs = dyn[3]
    dyn[3] = (lhs, rhs + D.par[1] * D.time)
    assert dyn.kind == 0x100 and dyn.tape["npars"] == 2
    # src/sim_basic.cpp:411-451: number of equations, LHS names, RHS variables.
    with pytest.raises(ValueError, match="6 dynamical equations are expected, but 2 were provided instead"):
        D.lower_dynamics([(x, y), (y, x)])
    eqs = [dyn[k] for k in range(6)]
    eqs[1] = (D.make_vars("q"), eqs[1][1])
    with pytest.raises(ValueError, match='The LHS of the dynamics at index 1 must be a variable named "y"'):
        D.lower_dynamics(eqs)
    eqs = [dyn[k] for k in range(6)]
    eqs[4] = (eqs[4][0], eqs[4][1] + D.make_vars("foo"))
    with pytest.raises(ValueError, match='for the variable "vy" contains the invalid variables \\["foo"\\]'):
        D.lower_dynamics(eqs)


def test_tape_is_well_formed():
    """Every operation defines a fresh slot from earlier slots (what both interpreters rely on)."""
    from cascade_amd import dynamics as D

    t = D.simple_earth(J2=True, J3=True, J4=True, C22S22=True, sun=True, moon=True, SRP=True, drag=True).tape
    ops = t["ops"]
    nxt = 7
    for o in ops:
        assert o["dst"] == nxt and 0 <= o["a"] < o["dst"] and o["b"] < o["dst"]
        nxt += 2 if o["op"] == 11 else 1
    assert nxt == t["nslots"] and t["npars"] == 2 and all(0 <= r < nxt for r in t["rhs"])


def test_tape_matches_builtin_recurrences(orc):
    """Kepler, J2 and J2 + drag lowered from expressions give the trajectories of the oracle's
    hand-written recurrences (same step-size rule, independent operation order)."""
    from cascade_amd import dynamics as D
    from cascade_amd.workloads import leo_si_population, random_leo_pop_kep

    st = random_leo_pop_kep(300, seed=3)
    a = orc.Sim(st.copy(), 0.23, n_par_ct=3, conj_thresh=1e-3)
    b = orc.Sim(st.copy(), 0.23, n_par_ct=3, conj_thresh=1e-3, tape=D.kepler(1.0).tape)
    for _ in range(4):
        assert a.step() == b.step()
        assert np.abs(a.state - b.state).max() < 1e-13
    assert len(a.conjunctions) == len(b.conjunctions)

    st, pars = leo_si_population(200, seed=3)
    for drag in (False, True):
        dyn = D.simple_earth(J2=True, C22S22=False, drag=drag)
        assert dyn.kind == (3 if drag else 1)  # untouched: the hand-unrolled kernels
        a = orc.Sim(st.copy(), 185.5663, n_par_ct=4, dyn_kind=dyn.kind, mu=dyn.mu, J2=dyn.J2, Re=dyn.Re, pars=pars if drag else None)
        b = orc.Sim(st.copy(), 185.5663, n_par_ct=4, tape=dyn.tape, pars=pars if drag else None)
        for _ in range(3):
            assert a.step() == b.step()
            # Both are valid tol = eps integrations in SI units (errors scale with |r| ~ 7e6).
            assert np.abs(a.state[:, :3] - b.state[:, :3]).max() < 1e-6
            assert np.abs(a.state[:, 3:6] - b.state[:, 3:6]).max() < 1e-8


def _simple_earth_numpy(flags):
    """simple_earth written independently: sympy gradients of the reference's potentials for the
    zonal terms, plain numpy for the rest (cascade.py/dynamics/_simple_earth.py:136-409)."""
    import sympy as sp

    X, Y, Z = sp.symbols("x y z", real=True)
    r2 = X**2 + Y**2 + Z**2
    r = sp.sqrt(r2)
    pots = 0
    if flags.get("J2"):
        pots += (0.1082635854e-2 * GM * RE**2) / (2 * r**5) * (3 * Z**2 - r2)
    if flags.get("J3"):
        pots += (-0.2532435346e-5 * GM * RE**3) * Z / (2 * r**7) * (5 * Z**2 - 3 * r2)
    if flags.get("J4"):
        pots += (-0.1619331205e-5 * GM * RE**4) / 8 * (35 * Z**4 - 30 * r2 * Z**2 + 3 * r2**2) / r**9
    grad = sp.lambdify((X, Y, Z), [-sp.diff(pots, q) for q in (X, Y, Z)], "numpy") if pots != 0 else None
    best = np.array([1.01709935e-06, 7.86443375e-01, 7.50341883e-09, 8.63934252e-14, 4.63822910e-02, 1.86080048e-01, 2.48667176e-02, 4.81080852e-03, 5.01594516e00, 2.28429809e01, 4.27957829e00, 1.56291673e-01])
    al, be, ga = best[:4], best[4:8] / 1000, best[8:] * 1000
    eps = DEG * 23.4392911

    def sun_pos(t):
        lo = DEG * 357.5256 + DEG * 1.1407410259335311e-5 * t
        lam = DEG * 282.94 + lo + DEG * ((6892 / 3600) * np.sin(lo) + (72 / 3600) * np.sin(2 * lo))
        ro = (149.619 - 2.499 * np.cos(lo) - 0.021 * np.cos(2 * lo)) * 1e9
        return ro * np.array([np.cos(lam), np.sin(lam) * np.cos(eps), np.sin(lam) * np.sin(eps)])

    def moon_pos(t):
        pm, pma, pmp, pms = (DEG * c * t for c in (1.1407410259335311e-5, 1.512151961904581e-4, 1.2893925235125941e-6, 6.128913003523574e-7))
        L0 = pmp + pma + DEG * 218.31617
        lm = pma + DEG * 134.96292
        llm = pm + DEG * 357.5256
        Fm = pmp + pma + pms + DEG * 93.27283
        Dm = pmp + pma - pm + DEG * 297.85027
        rm = (385000 - 20905 * np.cos(lm) - 3699 * np.cos(2 * Dm - lm) - 2956 * np.cos(2 * Dm) - 570 * np.cos(2 * lm) + 246 * np.cos(2 * lm - 2 * Dm) - 205 * np.cos(llm - 2 * Dm) - 171 * np.cos(lm + 2 * Dm) - 152 * np.cos(lm + llm - 2 * Dm)) * 1000
        lam = L0 + DEG * ((22640 / 3600) * np.sin(lm) + (769 / 3600) * np.sin(2 * lm) - (4856 / 3600) * np.sin(lm - 2 * Dm) + (2370 / 3600) * np.sin(2 * Dm) - (668 / 3600) * np.sin(llm) - (412 / 3600) * np.sin(2 * Fm) - (212 / 3600) * np.sin(2 * lm - 2 * Dm) - (206 / 3600) * np.sin(lm + llm - 2 * Dm) + (192 / 3600) * np.sin(lm + 2 * Dm) - (165 / 3600) * np.sin(llm - 2 * Dm) + (148 / 3600) * np.sin(lm - llm) - (125 / 3600) * np.sin(Dm) - (110 / 3600) * np.sin(lm + llm) - (55 / 3600) * np.sin(2 * Fm - 2 * Dm))
        Bm = DEG * ((18520 / 3600) * np.sin(Fm + lam - L0 + DEG * ((412 / 3600) * np.sin(2 * Fm) + (541 / 3600) * np.sin(llm))) - (526 / 3600) * np.sin(Fm - 2 * Dm) + (44 / 3600) * np.sin(lm + Fm - 2 * Dm) - (31 / 3600) * np.sin(-lm + Fm - 2 * Dm) - (25 / 3600) * np.sin(-2 * lm + Fm) - (23 / 3600) * np.sin(llm + Fm - 2 * Dm) + (21 / 3600) * np.sin(-lm + Fm) + (11 / 3600) * np.sin(-llm + Fm - 2 * Dm))
        return rm * np.array([np.cos(Bm) * np.cos(lam), -np.sin(eps) * np.sin(Bm) + np.cos(eps) * np.cos(Bm) * np.sin(lam), np.cos(eps) * np.sin(Bm) + np.cos(Bm) * np.sin(eps) * np.sin(lam)])

    def third(GMb, p, R):
        d = p - R
        return -GMb * (d / np.linalg.norm(d) ** 3 + R / np.linalg.norm(R) ** 3)

    def rhs(t, s, pars):
        p, v = s[:3], s[3:6]
        rr = np.linalg.norm(p)
        a = -GM * p / rr**3
        if grad is not None:
            a = a + np.array(grad(*p), dtype=float)
        if flags.get("C22S22"):
            C22, S22 = 2.43914352398e-6, -1.40016683654e-6
            th = DEG * 280.4606 + DEG * 4.178074622024230e-3 * t
            c, sn = np.cos(th), np.sin(th)
            Xf, Yf, Zf = p[0] * c + p[1] * sn, -p[0] * sn + p[1] * c, p[2]
            k15 = GM * RE**2 * np.sqrt(15)
            c1, c2 = 5 * k15 * C22 / (2 * rr**7), k15 * C22 / rr**5
            s1, s2 = 5 * k15 * S22 / rr**7, k15 * S22 / rr**5
            FX = c1 * Xf * (Yf**2 - Xf**2) + c2 * Xf - s1 * Xf**2 * Yf + s2 * Yf
            FY = c1 * Yf * (Yf**2 - Xf**2) - c2 * Yf - s1 * Xf * Yf**2 + s2 * Xf
            FZ = c1 * Zf * (Yf**2 - Xf**2) - s1 * Xf * Yf * Zf
            a = a + np.array([FX * c - FY * sn, FX * sn + FY * c, FZ])
        if flags.get("sun"):
            a = a + third(1.32712440018e11 * 1e9, p, sun_pos(t))
        if flags.get("moon"):
            a = a + third(4.9028e3 * 1e9, p, moon_pos(t))
        ip = 0
        if flags.get("drag"):
            dens = np.sum(al * np.exp(-((rr - RE) - ga) * be))
            a = a - dens / (0.1570 / RE) * pars[0] * np.linalg.norm(v) * v
            ip = 1
        if flags.get("SRP"):
            d = p - sun_pos(t)
            a = a + pars[ip] * (4.56e-3 / 1000) * (1.49619e8 * 1000) ** 2 / np.linalg.norm(d) ** 3 * d
        return np.concatenate([v, a])

    return rhs


@pytest.mark.parametrize(
    "flags",
    [
        dict(J2=True, J3=True, J4=True, C22S22=False, drag=False),
        dict(J2=True, C22S22=True, drag=True),  # the reference's defaults
        dict(J2=False, C22S22=False, sun=True, moon=True, drag=False),
        dict(J2=True, J3=True, J4=True, C22S22=True, sun=True, moon=True, SRP=True, drag=True),
    ],
)
def test_simple_earth_tape_vs_independent_integration(orc, flags):
    """Every term of simple_earth through expression -> diff -> tape -> oracle Taylor integrator
    against the independently written right-hand side under DOP853: positions to 1e-10 relative
    after 8 collisional timesteps, started at a non-zero epoch (time-dependent terms)."""
    from scipy.integrate import solve_ivp

    from cascade_amd import dynamics as D
    from cascade_amd.workloads import leo_si_population

    dyn = D.simple_earth(**flags)
    npars = dyn.tape["npars"]
    assert npars == int(bool(flags.get("drag"))) + int(bool(flags.get("SRP")))
    st, bstar = leo_si_population(4, seed=8)
    pars = np.concatenate([bstar, np.full((4, 1), 0.02)], axis=1)[:, :npars] if npars else None
    if flags.get("SRP") and not flags.get("drag"):
        pars = np.full((4, 1), 0.02)
    t0 = 3.0e8
    s = orc.Sim(st.copy(), 185.5663, n_par_ct=4, tape=dyn.tape, pars=pars)
    s.time = t0
    assert s.step() == "success" and s.step() == "success"
    rhs = _simple_earth_numpy(flags)
    for i in range(len(st)):
        pr = pars[i] if pars is not None else ()
        sol = solve_ivp(rhs, (t0, s.time), st[i, :6], args=(pr,), method="DOP853", rtol=1e-13, atol=1e-6)
        ref = sol.y[:3, -1]
        assert np.max(np.abs(s.state[i, :3] - ref)) <= 1e-10 * np.linalg.norm(ref), (flags, i)


def test_haumea_tape_vs_independent_integration(orc):
    """benchmark/haumea.cpp:83-139 (moon on a time-dependent orbit + triaxial body) through the
    tape against the same forces written in numpy under DOP853."""
    from scipy.integrate import solve_ivp

    from cascade_amd.workloads import haumea_dynamics, haumea_ring

    dyn, k = haumea_dynamics()
    st = haumea_ring(6)

    def rhs(t, s):
        p, v = s[:3], s[3:6]
        M = np.sqrt((k["mu_c"] + k["mu_s"]) / k["a_s"] ** 3) * t
        xo = k["a_s"] * (np.cos(M) - k["e_s"] * (1 + np.sin(M) ** 2))
        yo = k["a_s"] * (np.sin(M) + k["e_s"] * np.sin(M) * np.cos(M))
        R = np.array([xo, yo * np.cos(k["inc_s"]), yo * np.sin(k["inc_s"])])
        a = -k["mu_c"] * p / np.linalg.norm(p) ** 3
        a = a + k["mu_s"] * ((R - p) / np.linalg.norm(R - p) ** 3 - R / np.linalg.norm(R) ** 3)
        # + grad of V = G/2 (A + B + C - 3 I) / r^3, I = (A x^2 + B y^2 + C z^2) / r^2
        ABC = np.array([k["Ac"], k["Bc"], k["Cc"]])
        r2 = p @ p
        I = (ABC * p * p).sum() / r2
        dI = 2 * ABC * p / r2 - 2 * I * p / r2
        a = a + k["G"] / 2 * (-3 * dI * r2**-1.5 - 3 * (ABC.sum() - 3 * I) * p * r2**-2.5)
        return np.concatenate([v, a])

    s = orc.Sim(st.copy(), 5000.0, n_par_ct=2, tape=dyn.tape, reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
    s.time = 1.0e5
    for _ in range(3):
        assert s.step() in ("success", "collision")
    for i in range(len(st)):
        sol = solve_ivp(rhs, (1.0e5, s.time), st[i, :6], method="DOP853", rtol=1e-13, atol=1e-6)
        ref = sol.y[:3, -1]
        assert np.max(np.abs(s.state[i, :3] - ref)) <= 1e-10 * np.linalg.norm(ref), i


def _mascon_case(nm=12, seed=5):
    """A lumpy body of nm point masses (unit total mass, G = 1) spinning about a tilted axis."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-0.35, 0.35, (nm, 3)) * np.array([1.0, 0.7, 0.5])
    ms = rng.uniform(0.5, 1.5, nm)
    ms /= ms.sum()
    pts -= (ms[:, None] * pts).sum(axis=0)
    om = np.array([0.02, -0.03, 0.25])
    return 1.0, pts, ms, om


def _mascon_rhs(G, pts, ms, om):
    def f(_t, s):
        r, v = s[:3], s[3:]
        d = r - pts
        a = -G * (ms[:, None] * d / np.linalg.norm(d, axis=1)[:, None] ** 3).sum(axis=0)
        a = a - 2.0 * np.cross(om, v) - np.cross(om, np.cross(om, r))
        return np.concatenate([v, a])

    return f


def test_mascon_asteroid_signature_and_errors():
    from cascade_amd import dynamics as D

    G, pts, ms, om = _mascon_case()
    dyn = D.mascon_asteroid(G, pts, ms, om)
    assert len(dyn) == 6 and dyn.kind == 0x100 and dyn.tape["npars"] == 0
    assert [repr(dyn[k][0]) for k in range(6)] == ["x", "y", "z", "vx", "vy", "vz"]
    with pytest.raises(ValueError, match="shape"):
        D.mascon_asteroid(G, pts[:, :2], ms, om)
    with pytest.raises(ValueError, match="number of masses"):
        D.mascon_asteroid(G, pts, ms[:-1], om)
    with pytest.raises(ValueError, match="3 components"):
        D.mascon_asteroid(G, pts, ms, om[:2])
    # The reference's own energy function, term by term (cascade.py/dynamics/_mascon_asteroid.py:58-68).
    s = np.array([1.3, -0.4, 0.6, 0.1, 0.7, -0.2])
    pe, qe, re = om
    want = (s[3] ** 2 + s[4] ** 2 + s[5] ** 2) / 2 - sum(G * m / np.linalg.norm(s[:3] - p) for p, m in zip(pts, ms))
    want += -0.5 * (pe * pe + qe * qe + re * re) * (s[:3] @ s[:3]) + 0.5 * (s[0] * pe + s[1] * qe + s[2] * re) ** 2
    assert abs(D.mascon_asteroid_energy(s, G, pts, ms, om) - want) < 1e-15


def test_mascon_asteroid_tape_vs_dop853_and_energy(orc):
    """cascade.dynamics.mascon_asteroid (heyoka.model.mascon in the reference) through the
    oracle's tape interpreter: trajectories of an independent numpy right-hand side integrated by
    DOP853, and the Jacobi integral conserved to rounding."""
    from scipy.integrate import solve_ivp

    from cascade_amd import dynamics as D

    G, pts, ms, om = _mascon_case()
    dyn = D.mascon_asteroid(G, pts, ms, om)
    rng = np.random.default_rng(1)
    n = 24
    a = rng.uniform(1.2, 2.5, n)
    th, ph = rng.uniform(0, 2 * np.pi, n), rng.uniform(-0.4, 0.4, n)
    pos = a[:, None] * np.stack([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)], axis=1)
    # Roughly circular in the inertial frame, expressed in the rotating one: v_rot = v_in - w x r.
    vin = np.sqrt(G / a)[:, None] * np.stack([-np.sin(th), np.cos(th), np.zeros(n)], axis=1)
    vel = vin - np.cross(om, pos)
    st = np.concatenate([pos, vel, np.full((n, 1), 1e-4)], axis=1)
    e0 = D.mascon_asteroid_energy(st, G, pts, ms, om)
    o = orc.Sim(st.copy(), 0.5, n_par_ct=2, tape=dyn.tape, min_coll_radius=float("inf"))
    for _ in range(20):
        assert o.step() == "success"
    e1 = D.mascon_asteroid_energy(o.state, G, pts, ms, om)
    assert np.abs(e1 - e0).max() < 2e-14 * np.abs(e0).max()
    f = _mascon_rhs(G, pts, ms, om)
    for k in range(0, n, 5):
        sol = solve_ivp(f, (0.0, o.time), st[k, :6], method="DOP853", rtol=1e-13, atol=1e-15)
        assert np.abs(sol.y[:, -1] - o.state[k, :6]).max() < 5e-11
