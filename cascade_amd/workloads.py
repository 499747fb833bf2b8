"""Synthetic populations of the reference's benchmark drivers (seeded)."""
import numpy as np


def _mt19937_raw(seed, count):
    """Raw 32-bit outputs of std::mt19937(seed) (init_genrand seeding)."""
    bg = np.random.MT19937()
    try:
        bg._legacy_seeding(seed)
        return bg.random_raw(count).astype(np.float64)
    except AttributeError:  # pragma: no cover
        rs = np.random.RandomState(seed)
        return rs.randint(0, 2**32, size=count, dtype=np.uint64).astype(np.float64)


def _canonical(raw_pairs):
    """libstdc++ std::generate_canonical<double, 53>(mt19937): two 32-bit draws."""
    s = raw_pairs[:, 0] + raw_pairs[:, 1] * 4294967296.0
    r = s / 18446744073709551616.0
    return np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)


def kep_to_cart_bench(a, e, inc, Om, om, nu, mu=1.0):
    """benchmark/random_leo_pop_kep.cpp:31-59 (element order {a, e, i, Om, om, nu})."""
    E = 2 * np.arctan(np.sqrt((1 - e) / (1 + e)) * np.tan(nu / 2))
    n = np.sqrt(mu / (a * a * a))
    q = np.stack([a * (np.cos(E) - e), a * np.sqrt(1 - e * e) * np.sin(E), np.zeros_like(a)], axis=1)
    vq = np.stack([-n * a * np.sin(E) / (1 - e * np.cos(E)), n * a * np.sqrt(1 - e * e) * np.cos(E) / (1 - e * np.cos(E)), np.zeros_like(a)], axis=1)
    r1 = np.stack([np.cos(Om) * np.cos(om) - np.sin(Om) * np.cos(inc) * np.sin(om), -np.cos(Om) * np.sin(om) - np.sin(Om) * np.cos(inc) * np.cos(om), np.sin(Om) * np.sin(inc)], axis=1)
    r2 = np.stack([np.sin(Om) * np.cos(om) + np.cos(Om) * np.cos(inc) * np.sin(om), -np.sin(Om) * np.sin(om) + np.cos(Om) * np.cos(inc) * np.cos(om), -np.cos(Om) * np.sin(inc)], axis=1)
    r3 = np.stack([np.sin(inc) * np.sin(om), np.sin(inc) * np.cos(om), np.cos(inc)], axis=1)
    x = np.stack([(r1 * q).sum(1), (r2 * q).sum(1), (r3 * q).sum(1)], axis=1)
    v = np.stack([(r1 * vq).sum(1), (r2 * vq).sum(1), (r3 * vq).sum(1)], axis=1)
    return x, v


def random_leo_pop_kep(nparts=17378, seed=42):
    """The population of benchmark/random_leo_pop_kep.cpp:75-101 (a ~ U(1.02, 1.3),
    e ~ U(0, 0.02), i ~ U(0, 0.05), angles ~ U(0, 2 pi), size ~ U(1.57e-8, 1.57e-7); draw
    order per object a, e, i, om, Om, nu, size; std::mt19937). The reference seeds from
    random_device (:71); the seed is fixed here. Returns state [nparts, 7]."""
    raw = _mt19937_raw(seed, nparts * 14).reshape(nparts * 7, 2)
    u = _canonical(raw).reshape(nparts, 7)

    def uni(col, lo, hi):
        return (hi - lo) * u[:, col] + lo

    a = uni(0, 1.02, 1.3)
    e = uni(1, 0.0, 0.02)
    inc = uni(2, 0.0, 0.05)
    om = uni(3, 0.0, 2 * np.pi)
    Om = uni(4, 0.0, 2 * np.pi)
    nu = uni(5, 0.0, 2 * np.pi)
    size = uni(6, 1.57e-8, 1.57e-7)
    x, v = kep_to_cart_bench(a, e, inc, Om, om, nu, 1.0)
    st = np.empty((nparts, 7))
    st[:, :3] = x
    st[:, 3:6] = v
    st[:, 6] = size
    return st


# Quantiles (0, 1, 5, 10, 25, 50, 75, 90, 95, 99, 100 %) of the reference's 19 647-object catalogue
# benchmark/data/test_ic_19647.txt / test_par_19647.txt (semi-major axis altitude, eccentricity,
# inclination, BSTAR in the SI form of simple_earth, radius): summary statistics only, the
# catalogue itself is not shipped.
_Q = np.array([0, 1, 5, 10, 25, 50, 75, 90, 95, 99, 100]) / 100.0
_Q_ALT_KM = np.array([152.96, 339.20, 446.75, 492.87, 564.23, 790.30, 980.99, 1414.44, 1532.13, 2172.11, 3621.33])
_Q_ECC = np.array([8.44e-06, 3.22e-04, 6.54e-04, 9.22e-04, 1.62e-03, 4.17e-03, 1.13e-02, 2.62e-02, 5.35e-02, 1.70e-01, 3.42e-01])
_Q_INC_DEG = np.array([0.05, 31.0, 52.02, 53.06, 73.88, 83.21, 98.62, 99.23, 100.15, 102.39, 144.64])
_Q_BSTAR = np.array([0.0, 0.0, 0.0, 9.79e-13, 1.29e-11, 4.65e-11, 1.40e-10, 3.10e-10, 5.58e-10, 3.04e-09, 1.56e-07])
_Q_SIZE = np.array([5.64e-03, 2.39e-02, 4.26e-02, 5.29e-02, 6.91e-02, 0.15, 0.689, 2.0, 2.0, 2.0, 11.27])


def leo_si_population(n, seed=42, with_bstar=True, min_perigee_km=200.0):
    """Synthetic LEO catalogue in SI units shaped like the reference's 2022 population
    (benchmark/data/test_ic_19647.txt, BASELINE configs[2]): semi-major axis, eccentricity,
    inclination, BSTAR and radius are resampled independently from the catalogue's quantiles
    (inverse-CDF interpolation), angles are uniform; eccentricities are capped so that every
    perigee stays above min_perigee_km (the reference's driver removes objects below the reentry
    radius Re + 150 km before it starts). Returns (state [n, 7], pars [n, 1])."""
    rng = np.random.default_rng(seed)
    GM, Re = 3.986004407799724e14, 6378136.3

    def draw(table):
        return np.interp(rng.uniform(0.0, 1.0, n), _Q, table)

    a = Re + 1e3 * np.maximum(draw(_Q_ALT_KM), min_perigee_km + 20.0)
    e = np.minimum(draw(_Q_ECC), 1.0 - (Re + 1e3 * min_perigee_km) / a)
    inc = np.deg2rad(draw(_Q_INC_DEG))
    Om, om, nu = (rng.uniform(0, 2 * np.pi, n) for _ in range(3))
    p = a * (1 - e * e)
    r = p / (1 + e * np.cos(nu))
    h = np.sqrt(GM * p)
    cO, sO, ci, si = np.cos(Om), np.sin(Om), np.cos(inc), np.sin(inc)
    cu, su = np.cos(om + nu), np.sin(om + nu)
    x = r * (cO * cu - sO * su * ci)
    y = r * (sO * cu + cO * su * ci)
    z = r * (si * su)
    vr = h * e * np.sin(nu) / p
    vt = h / r
    vx = vr * x / r - vt * (cO * su + sO * cu * ci)
    vy = vr * y / r - vt * (sO * su - cO * cu * ci)
    vz = vr * z / r + vt * si * cu
    size = draw(_Q_SIZE)
    st = np.stack([x, y, z, vx, vy, vz, size], axis=1)
    pars = draw(_Q_BSTAR).reshape(n, 1) if with_bstar else np.zeros((n, 0))
    return np.ascontiguousarray(st), np.ascontiguousarray(pars)


def haumea_ring(nparts=10000, seed=5489):
    """The ring population of benchmark/haumea.cpp:142-167: a ~ U(5e6, 1e7) m, e ~ U(0, 1e-3),
    i ~ U(0, 1 deg), angles ~ U(0, 2 pi), radius 10 km / 3; draw order a, e, i, om, Om, nu with
    std::mt19937 left at its default seed 5489 (:73). Returns state [nparts, 7] (SI units)."""
    G, m_c = 6.674e-11, 4.006e21
    raw = _mt19937_raw(seed, nparts * 12).reshape(nparts * 6, 2)
    u = _canonical(raw).reshape(nparts, 6)
    a = (10000e3 - 5000e3) * u[:, 0] + 5000e3
    e = 1e-3 * u[:, 1]
    inc = (2 * np.pi / 360) * u[:, 2]
    om, Om, nu = (2 * np.pi * u[:, k] for k in (3, 4, 5))
    x, v = kep_to_cart_bench(a, e, inc, Om, om, nu, m_c * G)
    st = np.empty((nparts, 7))
    st[:, :3] = x
    st[:, 3:6] = v
    st[:, 6] = 10e3 / 3
    return st


def haumea_dynamics():
    """The dynamics of benchmark/haumea.cpp:83-139 written with cascade_amd.dynamics expressions:
    Kepler around Haumea, the tidal perturbation of the moon Hi'iaka on its (time-dependent)
    Keplerian orbit in the non-inertial frame, and the MacCullagh term of Haumea's triaxial
    ellipsoid. Returns (dyn, constants)."""
    from . import dynamics as D

    G = 6.674e-11
    m_c = 4.006e21
    mu_c = m_c * G
    ra = rb = 2000e3
    rc = ra / 2
    Ac, Bc, Cc = (m_c / 5.0 * (p * p + q * q) for p, q in ((rb, rc), (ra, rc), (ra, rb)))
    mu_s = m_c / 223 * G
    a_s, e_s, inc_s = 20000e3, 0.05, 55.0 * 2 * np.pi / 360
    x, y, z = D.make_vars("x", "y", "z")
    M = float(np.sqrt((mu_c + mu_s) / a_s**3)) * D.time
    sM, cM = D.sin(M), D.cos(M)
    # Position of the moon to first order in its eccentricity.
    xo = a_s * (cM - e_s * (1.0 + sM * sM))
    yo = a_s * (sM + e_s * sM * cM)
    xs, ys, zs = xo, yo * float(np.cos(inc_s)), yo * float(np.sin(inc_s))
    dps = D.pow((xs - x) * (xs - x) + (ys - y) * (ys - y) + (zs - z) * (zs - z), -1.5)
    dcs = D.pow(xs * xs + ys * ys + zs * zs, -1.5)
    r2 = x * x + y * y + z * z
    inertia = (Ac * x * x + Bc * y * y + Cc * z * z) / r2
    vell = (G / 2.0) * ((Ac + Bc + Cc) - 3.0 * inertia) * D.pow(r2, -1.5)
    dyn = D.kepler(mu_c)
    for k, (q, qs) in enumerate(((x, xs), (y, ys), (z, zs))):
        lhs, rhs = dyn[3 + k]
        dyn[3 + k] = (lhs, rhs + mu_s * ((qs - q) * dps - qs * dcs) + D.diff(vell, q))
    consts = dict(G=G, m_c=m_c, mu_c=mu_c, ra=ra, rb=rb, rc=rc, Ac=Ac, Bc=Bc, Cc=Cc, mu_s=mu_s, a_s=a_s, e_s=e_s, inc_s=inc_s)
    return dyn, consts


def leo_shells(nparts, seed=42):
    """BASELINE configs[3]: objects in Walker-like shells: semi-major axes
    {1.08, 1.09, 1.17, 1.19} +- 0.002 Earth radii, inclinations {53, 70, 97.6} +- 0.5 degrees,
    e < 1e-3, uniform angles, size 1e-7; normalised Kepler units (mu = 1). Meant to be run with
    conj_thresh = 7.8e-4 (5 km) and min_coll_radius = inf. Returns state [nparts, 7]."""
    rng = np.random.default_rng(seed)
    n = nparts
    a = rng.choice([1.08, 1.09, 1.17, 1.19], n) + rng.uniform(-0.002, 0.002, n)
    inc = np.deg2rad(rng.choice([53.0, 70.0, 97.6], n) + rng.uniform(-0.5, 0.5, n))
    x, v = kep_to_cart_bench(a, rng.uniform(0, 1e-3, n), inc, rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n))
    return np.ascontiguousarray(np.concatenate([x, v, np.full((n, 1), 1e-7)], axis=1))


def load_leo_2022(path):
    """The 19 647-object catalogue of benchmark/2022_leo_pop.cpp:197-209 from a .npz made by
    tests/golden/make_fixtures.py (the data files themselves are the reference's:
    benchmark/data/test_ic_19647.txt, test_par_19647.txt, best_fit_density.txt). Returns
    (state [n, 7], bstar [n, 1], best_fit_density [12])."""
    d = np.load(path)
    return np.ascontiguousarray(d["state"]), np.ascontiguousarray(d["bstar"]), np.array(d["best_fit_density"])


def leo_2022_dynamics(best_x):
    """The dynamics of benchmark/2022_leo_pop.cpp:218-287 in its own form, written with
    cascade_amd.dynamics expressions (runs from the tape): Kepler + J2 from C20 (:228-238), C22/S22
    in the rotating frame (:240-272), drag with the 4-term density fit best_x (:80-91, :274-284);
    like the driver, the C22 force enters a second time together with the drag term (:285-287).
    Returns (dyn, constants); par[0] = BSTAR, reentry at Re + 150 km (:223, :297)."""
    from . import dynamics as D

    GMe, C20, C22, S22, Re = 398600441800000.0, -4.84165371736e-4, 2.43914352398e-6, -1.40016683654e-6, 6378137.0
    pi = float(np.pi)
    x, y, z, vx, vy, vz = D.make_vars("x", "y", "z", "vx", "vy", "vz")
    dyn = D.kepler(GMe)
    magr2 = x * x + y * y + z * z
    J2term1 = GMe * (Re * Re) * float(np.sqrt(5.0)) * C20 / (2.0 * D.sqrt(magr2))
    J2term2 = 3.0 / (magr2 * magr2)
    J2term3 = 15.0 * (z * z) / (magr2 * magr2 * magr2)
    fJ2 = (J2term1 * x * (J2term2 - J2term3), J2term1 * y * (J2term2 - J2term3), J2term1 * z * (3.0 * J2term2 - J2term3))

    theta_g = (pi / 180) * 280.4606
    nu_e = (pi / 180) * 4.178074622024230e-3
    ang = theta_g + nu_e * D.time
    ca, sa = D.cos(ang), D.sin(ang)
    X, Y, Z = x * ca + y * sa, -x * sa + y * ca, z
    s15 = float(np.sqrt(15.0))
    C22term1 = 5.0 * GMe * (Re * Re) * s15 * C22 / (2.0 * D.pow(magr2, 7.0 / 2))
    C22term2 = GMe * (Re * Re) * s15 * C22 / D.pow(magr2, 5.0 / 2)
    fC22X = C22term1 * X * (Y * Y - X * X) + C22term2 * X
    fC22Y = C22term1 * Y * (Y * Y - X * X) - C22term2 * Y
    fC22Z = C22term1 * Z * (Y * Y - X * X)
    S22term1 = 5 * GMe * (Re * Re) * s15 * S22 / D.pow(magr2, 7.0 / 2)
    S22term2 = GMe * (Re * Re) * s15 * S22 / D.pow(magr2, 5.0 / 2)
    fS22X = -S22term1 * (X * X) * Y + S22term2 * Y
    fS22Y = -S22term1 * X * (Y * Y) + S22term2 * X
    fS22Z = -S22term1 * X * Y * Z
    fC22 = (fC22X * ca - fC22Y * sa, fC22X * sa + fC22Y * ca, fC22Z)
    fS22 = (fS22X * ca - fS22Y * sa, fS22X * sa + fS22Y * ca, fS22Z)

    magv = D.sqrt(vx * vx + vy * vy + vz * vz)
    altitude = D.sqrt(magr2) - Re
    density = None
    for i in range(4):
        alpha, beta, gamma = float(best_x[i]), float(best_x[i + 4]) / 1000, float(best_x[i + 8]) * 1000
        term = alpha * D.exp(-(altitude - gamma) * beta)
        density = term if density is None else density + term
    fdrag = density / (0.1570 / Re) * D.par[0] * magv
    for k, v in enumerate((vx, vy, vz)):
        lhs, rhs = dyn[3 + k]
        dyn[3 + k] = (lhs, rhs + fJ2[k] + (fC22[k] + fS22[k]) + (fC22[k] + (-fdrag * v)))
    return dyn, dict(GMe=GMe, Re=Re, reentry_radius=Re + 150000.0, ct=185.5663, n_par_ct=30)
