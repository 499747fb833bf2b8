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


def leo_si_population(n, seed=42, with_bstar=True):
    """Synthetic LEO population in SI units for the perturbed dynamics (J2 / drag): near-circular
    orbits, altitude 250-1200 km, all inclinations, sizes 0.1-5 m, BSTAR log-uniform in
    [1e-5, 1e-3] (SI form of simple_earth: acceleration = rho / rho_ref * BSTAR * |v| v).
    Returns (state [n, 7], pars [n, 1])."""
    rng = np.random.default_rng(seed)
    GM, Re = 3.986004407799724e14, 6378136.3
    a = Re + rng.uniform(250e3, 1200e3, n)
    e = rng.uniform(0.0, 0.01, n)
    inc = np.arccos(rng.uniform(-1.0, 1.0, n))
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
    size = rng.uniform(0.1, 5.0, n)
    st = np.stack([x, y, z, vx, vy, vz, size], axis=1)
    pars = np.exp(rng.uniform(np.log(1e-5), np.log(1e-3), n)).reshape(n, 1) if with_bstar else np.zeros((n, 0))
    return np.ascontiguousarray(st), np.ascontiguousarray(pars)
