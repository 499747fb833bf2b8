"""Pins the CPU oracle against the reference's own known-answer tests.

Each test restates one of the reference's Catch2 / unittest cases (file:line in
the docstring) against oracle/liborc.so. The oracle is the checker for every
GPU parity test, so these must hold before any GPU result is trusted.
"""
import math

import numpy as np
import pytest

PSIZE = 1.57e-8
PI = math.pi


def k2c(orc, kep):
    x, v = orc.kep_to_cart_test(kep, 1.0)
    return list(x) + list(v)


@pytest.mark.parametrize("n_par_ct", [1])
def test_python_conjunction_kat(orc, n_par_ct):
    """cascade.py/test.py:155-187: literal ICs -> 0 / 1 (t<=2) / 6 (t<=20) conjunctions."""
    r1 = (0.3342377271241684, 0.942488801930755, 0.0)
    v1 = (0.009424730938644767, -0.003342321565239028, 0.9999500004176654)
    r2 = (-0.15179985849820377, -0.988411251938142, 0.0)
    v2 = (6.052273379648475e-17, -9.295060541060909e-18, 1.000000000001)
    s = orc.Sim([list(r1) + list(v1) + [PSIZE], list(r2) + list(v2) + [PSIZE]], 0.23, conj_thresh=PSIZE * 100000000, n_par_ct=n_par_ct)
    assert len(s.conjunctions) == 0
    assert s.propagate_until(2.0) == "time_limit"
    assert len(s.conjunctions) == 1
    assert s.propagate_until(20.0) == "time_limit"
    assert len(s.conjunctions) == 6
    s.reset_conjunctions()
    assert len(s.conjunctions) == 0


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_polar_conj(orc, n_par_ct):
    """test/conj_tracking.cpp:37-118 (without the heyoka cross-integration)."""
    p1 = k2c(orc, (1.0, 0.000005, PI / 2 + 1e-6, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    s = orc.Sim([p1 + [PSIZE], p2 + [PSIZE]], 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while len(s.conjunctions) == 0:
        assert s.step() == "success"
    c = s.conjunctions
    assert len(c) == 1
    assert c[0]["i"] == 0 and c[0]["j"] == 1
    assert abs(c[0]["time"] - PI / 2) < 1e-4
    assert c[0]["dist"] < PSIZE * 100
    # The recorded states must be consistent with the recorded distance.
    d = np.linalg.norm(c[0]["state_i"][:3] - c[0]["state_j"][:3])
    assert abs(d - c[0]["dist"]) < 1e-9  # tolerance of test/conj_tracking.cpp:94-97
    # At a conjunction the relative position is orthogonal to the relative velocity.
    dr = c[0]["state_i"][:3] - c[0]["state_j"][:3]
    dv = c[0]["state_i"][3:] - c[0]["state_j"][3:]
    assert abs(np.dot(dr, dv)) < 1e-12


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_polar_conj_near_miss_and_barely(orc, n_par_ct):
    """test/conj_tracking.cpp:122-173."""
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p1 = k2c(orc, (1.0, 0.000005, PI / 2 + 1.5775e-6, 0, 1.23, 0))
    s = orc.Sim([p1 + [PSIZE], p2 + [PSIZE]], 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while True:
        assert s.step() == "success"
        assert len(s.conjunctions) == 0
        if s.time > 3.14:
            break
    p1 = k2c(orc, (1.0, 0.000005, PI / 2 + 1.5765e-6, 0, 1.23, 0))
    s = orc.Sim([p1 + [PSIZE], p2 + [PSIZE]], 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while len(s.conjunctions) == 0:
        assert s.step() == "success"
    c = s.conjunctions
    assert len(c) == 1 and c[0]["i"] == 0 and c[0]["j"] == 1
    assert abs(c[0]["time"] - PI / 2) < 1e-4
    assert c[0]["dist"] < PSIZE * 100


def _four(orc, x4_shift):
    p1 = k2c(orc, (1.0, 0.000005, PI / 2 + 1e-6, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p3 = k2c(orc, (1.0 + 1e-7, 0.0, 0.0, 0.0, 0.0, 0))
    p4 = list(p3)
    p4[0] = -p4[0] + x4_shift
    return [p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]]


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_polar_conj_discard(orc, n_par_ct):
    """test/conj_tracking.cpp:228-262: conjunctions after the collision time are dropped."""
    s = orc.Sim(_four(orc, 0.0), 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while True:
        oc = s.step()
        if oc != "success":
            assert oc == "collision"
            break
    c = s.conjunctions
    assert len(c) == 1 and c[0]["i"] == 0 and c[0]["j"] == 1
    assert c[0]["time"] < s.time


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_multiple_conjs(orc, n_par_ct):
    """test/conj_tracking.cpp:267-297: 8 conjunctions by t > 26, chronological."""
    s = orc.Sim(_four(orc, 1e-6), 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while True:
        assert s.step() == "success"
        if s.time > 26.0:
            break
    c = s.conjunctions
    assert len(c) == 8
    assert np.all(np.diff(c["time"]) >= 0)
    assert c["time"][-1] < s.time


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_coll_filtering(orc, n_par_ct):
    """test/coll_conj_filter.cpp:28-92."""
    p1 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p3 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    p4 = list(p3)
    p4[0] = -p4[0]
    ic = np.array([p1 + [PSIZE], p2 + [PSIZE]])
    s = orc.Sim(ic, 0.23, min_coll_radius=PSIZE * 2, n_par_ct=n_par_ct)
    assert s.propagate_until(10.0) == "time_limit"

    s.time = 0
    ic2 = ic.copy()
    ic2[0, 6] = PSIZE * 3
    s.set_new_state(ic2)
    assert s.propagate_until(10.0) == "collision"

    s.time = 0
    s.set_new_state(ic2)
    s.set_whitelists(coll=[0])
    assert s.propagate_until(10.0) == "collision"

    ic4 = np.array([p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]])
    ic4[2, 6] = PSIZE * 3  # get_state_data()[20]
    s.time = 0
    s.set_new_state(ic4)
    s.set_whitelists()
    assert s.propagate_until(10.0) == "collision"
    assert s.interrupt_info == (2, 3)

    s.time = 0
    s.set_new_state(ic4)
    s.set_whitelists(coll=[0])
    assert s.propagate_until(10.0) == "time_limit"


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_conj_filtering(orc, n_par_ct):
    """test/coll_conj_filter.cpp:95-179: conjunction counts 0, 3, 3, 6, 3, 3."""
    p1 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p3 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    p4 = list(p3)
    p4[0] = -p4[0]
    ic = np.array([p1 + [PSIZE], p2 + [PSIZE]])
    s = orc.Sim(ic, 0.23, min_coll_radius=float("inf"), n_par_ct=n_par_ct)
    assert s.propagate_until(10.0) == "time_limit"
    assert len(s.conjunctions) == 0

    s.time = 0
    s.set_new_state(ic)
    s.set_conj_thresh(PSIZE / 2)
    assert s.propagate_until(10.0) == "time_limit"
    assert len(s.conjunctions) == 3

    s.time = 0
    s.set_new_state(ic)
    s.set_whitelists(conj=[0])
    s.reset_conjunctions()
    assert s.propagate_until(10.0) == "time_limit"
    assert len(s.conjunctions) == 3

    ic4 = np.array([p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]])
    s.time = 0
    s.set_new_state(ic4)
    s.set_whitelists()
    s.reset_conjunctions()
    assert s.propagate_until(10.0) == "time_limit"
    assert len(s.conjunctions) == 6

    s.time = 0
    s.set_new_state(ic4)
    s.set_whitelists(conj=[0])
    s.reset_conjunctions()
    assert s.propagate_until(10.0) == "time_limit"
    c = s.conjunctions
    assert len(c) == 3 and np.all(c["i"] == 0) and np.all(c["j"] == 1)

    s.time = 0
    s.set_new_state(ic4)
    s.set_whitelists(conj=[2, 3])
    s.reset_conjunctions()
    assert s.propagate_until(10.0) == "time_limit"
    c = s.conjunctions
    assert len(c) == 3 and np.all(c["i"] == 2) and np.all(c["j"] == 3)


def _d2(st, a, b):
    return float(np.sum((st[a, :3] - st[b, :3]) ** 2))


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_interrupt_order(orc, n_par_ct):
    """test/interrupt_order.cpp:25-130."""
    s = orc.Sim(
        [[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [4.9, 0, 0, 1, 0, 0, 1e-6]],
        0.23,
        exit_radius=5.0,
        reentry_radius=0.75,
        n_par_ct=n_par_ct,
    )
    assert s.propagate_until(1000.0) == "collision"
    assert s.interrupt_info == (0, 1)
    assert abs(_d2(s.state, 0, 1) - 4e-12) < 1e-14

    def rerun(st):
        s.poke_state(st)
        s.time = 0.0
        return s.propagate_until(1000.0)

    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [4.9, 0, 0, 1000, 0, 0, 1e-6]]) == "exit"
    assert s.interrupt_info == 2
    assert abs(float(np.sum(s.state[2, :3] ** 2)) - 25.0) < 1e-14

    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [0.76, 0, 0, 0, 0, 0, 1e-6]]) == "collision"
    assert s.interrupt_info == (0, 1)
    assert abs(_d2(s.state, 0, 1) - 4e-12) < 1e-14

    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [0.76, 0, 0, -1, 0, 0, 1e-6]]) == "reentry"
    assert s.interrupt_info == 2
    assert abs(float(np.sum(s.state[2, :3] ** 2)) - 0.75 * 0.75) < 1e-14

    assert rerun([[0.76, 0, 0, -1, 0, 0, 1e-6], [1.0, 0, 0, 0, 1, 0, 1e-6], [2, 0, 0, 1, 0, 0, 1e-6]]) == "reentry"
    assert s.interrupt_info == 0
    assert abs(float(np.sum(s.state[0, :3] ** 2)) - 0.75 * 0.75) < 1e-14

    assert rerun([[0.76, 0, 0, 0, 0, 0, 1e-6], [1.0, 0, 0, 0, 1, 0, 1e-6], [4.9, 0, 0, 1, 0, 0, 1e-6]]) == "exit"
    assert s.interrupt_info == 2
    assert abs(float(np.sum(s.state[2, :3] ** 2)) - 25.0) < 1e-14


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_single_collision(orc, n_par_ct):
    """test/single_collision.cpp:35-204."""
    eps = np.finfo(float).eps
    p1 = k2c(orc, (1.05, 0.05, PI / 2, 0, 1.23, 0))
    p2 = k2c(orc, (1.05, 0.05, PI / 2, 0, 4.56, 0))
    s0 = np.array([p1 + [PSIZE], p2 + [PSIZE]])
    s = orc.Sim(s0, 0.23, n_par_ct=n_par_ct)
    while s.step() != "collision":
        pass
    assert s.interrupt_info == (0, 1)
    assert _d2(s.state, 0, 1) - 4 * PSIZE * PSIZE < eps * 10

    s = orc.Sim(s0, 0.23, n_par_ct=n_par_ct)
    assert s.propagate_until(10) == "collision"
    assert s.time < 9
    assert _d2(s.state, 0, 1) - 4 * PSIZE * PSIZE < eps * 10
    s.poke_state(s0)
    s.time = 0
    assert s.propagate_until(0.5) == "time_limit"
    assert s.interrupt_info is None
    assert s.time == 0.5
    s.poke_state(s0)
    s.time = 0
    assert s.propagate_until(1.57) == "time_limit"
    assert abs(s.time - 1.57) < 1e-15
    assert s.interrupt_info is None

    # equatorial: a zero-thickness dimension of the global AABB
    e1 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    e2 = list(e1)
    e2[0] = -e2[0]
    s = orc.Sim([e1 + [PSIZE], e2 + [PSIZE]], 0.23, n_par_ct=n_par_ct)
    while s.step() != "collision":
        pass
    assert _d2(s.state, 0, 1) - 4 * PSIZE * PSIZE < eps * 10
    s = orc.Sim([e1 + [PSIZE], e2 + [PSIZE]], 0.23, n_par_ct=n_par_ct)
    assert s.propagate_until(0.5) == "time_limit"
    assert s.time == 0.5

    # only incoming collisions
    o2 = list(e1)
    o2[4] = -o2[4]
    s = orc.Sim([e1 + [PSIZE], o2 + [PSIZE]], 0.23, n_par_ct=n_par_ct)
    assert s.step() == "success"
    assert s.propagate_until(100) == "collision"


def _rnorm(st, i):
    return float(np.sqrt(np.sum(st[i, :3] ** 2)))


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_reentry(orc, n_par_ct):
    """test/reentry.cpp:27-215 (sphere and ellipsoid, scalar and 'batch' populations)."""
    s = orc.Sim([[1.5, 0, 0, 0, 0.01, 0.0, 0]], 0.23, reentry_radius=1.0, n_par_ct=n_par_ct)
    assert s.propagate_until(1000.0) == "reentry"
    assert abs(_rnorm(s.state, 0) - 1.0) < 1e-14
    s.poke_state([[0.0, 1.5, 0, 0, 0.0, 0.0, 0.0]])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert abs(_rnorm(s.state, 0) - 1.0) < 1e-14

    a = [1.1, 0.0, 0.0, 0.0, 0.953, 0.0, 0.0]
    s.set_new_state([a, [1.5, 0.0, 0.0, 0.0, 0.01, 0.0, 0.0], a, a, a, a])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert s.interrupt_info == 1
    assert abs(_rnorm(s.state, 1) - 1.0) < 1e-14

    s = orc.Sim([[1.5, 0, 0, 0, 0.0, 0.0, 0]], 0.23, reentry_radius=[1.0, 1.1, 1.2], n_par_ct=n_par_ct)
    assert s.propagate_until(1000.0) == "reentry"
    assert abs(_rnorm(s.state, 0) - 1.0) < 1e-14
    s.poke_state([[0.0, 1.5, 0, 0, 0.0, 0.0, 0]])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert abs(_rnorm(s.state, 0) - 1.1) < 1e-14
    s.poke_state([[0.0, 0, 1.5, 0, 0.0, 0.0, 0]])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert abs(_rnorm(s.state, 0) - 1.2) < 1e-14

    b = [0.0, 0.0, 2.1, 0.69, 0.0, 0.0, 0.0]
    s.set_new_state([b, [0.0, 0.0, 1.5, 0.0, 0.0, 0.0, 0.0], b, b, b, b])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert s.interrupt_info == 1
    assert abs(_rnorm(s.state, 1) - 1.2) < 1e-14
    c = [0.0, 2.1, 0.0, 0.69, 0.0, 0.0, 0.0]
    s.set_new_state([c, c, c, [0.0, 1.5, 0.0, 0.0, 0.0, 0.0, 0.0], c, c])
    s.time = 0.0
    assert s.propagate_until(1000) == "reentry"
    assert s.interrupt_info == 3
    assert abs(_rnorm(s.state, 3) - 1.1) < 1e-14


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_domain_exit(orc, n_par_ct):
    """test/domain_exit.cpp."""
    s = orc.Sim([[1.0, 0, 0, 150.0, 0.0, 0.0, 0]], 0.23, exit_radius=10.0, n_par_ct=n_par_ct)
    assert s.propagate_until(1000.0) == "exit"
    assert abs(_rnorm(s.state, 0) - 10.0) < 1e-14
    s.poke_state([[1.0, 0, 0, 150, 150.0, 0.0, 0.0]])
    s.time = 0.0
    assert s.propagate_until(1000.0) == "exit"
    assert abs(_rnorm(s.state, 0) - 10.0) < 1e-14
    a = [1.1, 0, 0, 0, 0.953, 0, 0]
    s.set_new_state([a, [1.5, 0, 0, 0, 0.953, 0, 0], a, [1.1, 0, 0, 0, 150, 0, 0], a, a])
    s.time = 0.0
    assert s.propagate_until(1000) == "exit"
    assert s.interrupt_info == 3
    assert abs(_rnorm(s.state, 3) - 10.0) < 1e-14


def test_invalid_state(orc):
    """test/invalid_state.cpp:27-64."""
    e1 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    e2 = list(e1)
    e2[0] = -e2[0]
    s0 = np.array([e1 + [PSIZE], e2 + [PSIZE]])
    s = orc.Sim(s0, 0.23)
    bad = s0.copy()
    bad[0, 6] = -1
    s.poke_state(bad)
    with pytest.raises(ValueError, match="An invalid particle size of -1"):
        s.step()
    s.poke_state(s0)
    assert np.array_equal(s.state, s0) and s.time == 0.0
    bad = s0.copy()
    bad[0, 0] = np.inf
    s.poke_state(bad)
    assert s.step() == "err_nf_state"
    assert s.interrupt_info == (0, 0.0)
    s.poke_state(s0)
    assert np.array_equal(s.state, s0) and s.time == 0.0


def _kepler_closed_form(st, t):
    """Two-body propagation (mu = 1) of [x, y, z, vx, vy, vz] by time t via f and g
    functions with a Newton solve of Kepler's equation in extended precision."""
    r0 = np.array(st[:3], dtype=np.longdouble)
    v0 = np.array(st[3:6], dtype=np.longdouble)
    t = np.longdouble(t)
    rn = np.sqrt(np.dot(r0, r0))
    v2 = np.dot(v0, v0)
    a = 1 / (2 / rn - v2)
    n = np.sqrt(1 / a**3)
    sig0 = np.dot(r0, v0)
    # solve dE - (1 - rn/a) sin dE + sig0/sqrt(a) (1 - cos dE) = n t
    dE = n * t
    for _ in range(60):
        f = dE - (1 - rn / a) * np.sin(dE) + sig0 / np.sqrt(a) * (1 - np.cos(dE)) - n * t
        fp = 1 - (1 - rn / a) * np.cos(dE) + sig0 / np.sqrt(a) * np.sin(dE)
        dE = dE - f / fp
    F = 1 - a / rn * (1 - np.cos(dE))
    G = t + (np.sin(dE) - dE) / n
    r = F * r0 + G * v0
    rr = np.sqrt(np.dot(r, r))
    Fd = -np.sqrt(a) / (rr * rn) * np.sin(dE)
    Gd = 1 - a / rr * (1 - np.cos(dE))
    v = Fd * r0 + Gd * v0
    return np.concatenate([r, v]).astype(np.float64)


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_kepler_closed_form(orc, n_par_ct):
    """Stands in for test/heyoka_comparison.cpp:87-107 (|delta| <= 1e-13 after 10 steps);
    heyoka is absent, the two-body solution is known in closed form."""
    rng = np.random.default_rng(42)
    st = []
    for _ in range(10):
        kep = (rng.uniform(1.02, 1.3), rng.uniform(0, 0.02), rng.uniform(0, 0.05), rng.uniform(0, 2 * PI), rng.uniform(0, 2 * PI), rng.uniform(0, 2 * PI))
        st.append(k2c(orc, kep) + [0.0])
    st = np.array(st)
    s = orc.Sim(st, 0.23, n_par_ct=n_par_ct)
    for _ in range(10):
        assert s.step() == "success"
    out = s.state
    for i in range(10):
        ref = _kepler_closed_form(st[i], s.time)
        assert np.max(np.abs(ref - out[i, :6])) < 1e-13


def test_same_morton_code_and_largish(orc):
    """test/same_morton_code.cpp, test/largish.cpp: multi-particle leaves, BVH and
    broad-phase verifiers (src/sim_bvh.cpp:498-656, src/sim_broad_phase.cpp:221-321)."""
    rng = np.random.default_rng(7)
    n = 2000
    st = []
    for _ in range(n):
        kep = (rng.uniform(1.02, 1.3), rng.uniform(0, 0.02), rng.uniform(0, 0.05), rng.uniform(0, 2 * PI), rng.uniform(0, 2 * PI), rng.uniform(0, 2 * PI))
        st.append(k2c(orc, kep) + [1e-6])
    eps = np.finfo(float).eps
    for idx in (1, 123, 456):
        p = list(st[idx])
        p[0] += eps * 1000
        p[1] += eps * 1000
        p[2] += eps * 1000
        p[6] = 0.1
        st.append(p)
    st = np.array(st)
    L = orc.lib()
    for n_par_ct in (1, 3):
        s = orc.Sim(st, 0.23, n_par_ct=n_par_ct)
        assert s.step() in ("success", "collision")
        ca, cj = s.active_flags()
        for c in range(s.nchunks):
            codes, slb, sub = s.sorted_data(c)
            assert np.all(np.diff(codes.astype(np.uint64)) >= 0) or np.all(codes[1:] >= codes[:-1])
            tree = s.bvh(c)
            rc = L.orc_bvh_verify(len(codes), orc._p(codes), orc._p(slb), orc._p(sub), orc._p(tree), len(tree))
            assert rc == 0, L.orc_last_error()
            # multi-particle leaves exist (identical codes)
            leaves = tree[tree["left"] == -1]
            assert np.any(leaves["end"] - leaves["begin"] > 1)
            # broad phase is a superset of the exact O(N^2) overlaps, unique and i < j
            bp = s.bp_pairs(c)
            assert np.all(bp[:, 0] < bp[:, 1])
            bpset = set(map(tuple, bp.tolist()))
            assert len(bpset) == len(bp)
            lb, ub = s.aabbs(c)
            cap = 4 * len(bp) + 1024
            brute = np.empty((cap, 2), dtype=np.uint32)
            nb = L.orc_broad_phase_brute(len(lb), orc._p(lb), orc._p(ub), orc._p(ca), orc._p(cj), orc._p(brute), cap)
            assert 0 <= nb <= cap
            assert set(map(tuple, brute[:nb].tolist())) <= bpset
            # global AABB == serial min/max (verify_global_aabbs, src/sim_propagate.cpp:1348-1380),
            # modulo the in-place nextafter bump of the upper bound (:448-449)
            glb, gub = s.global_aabb(c)
            assert np.array_equal(glb, lb.min(axis=0))
            assert np.array_equal(gub, np.nextafter(ub.max(axis=0), np.float32(np.inf)))


def test_oracle_j2_drag_vs_independent_integration():
    """Pins the oracle's J2 and drag Taylor recurrences (simple_earth, cascade.py/dynamics/
    _simple_earth.py:169-178, 365-387 with the density fit :16-46): the same right-hand side
    written independently in numpy and integrated with scipy's DOP853 must give the same
    positions after 8 collisional timesteps to 1e-10 relative."""
    import numpy as np
    from scipy.integrate import solve_ivp

    from cascade_amd.workloads import leo_si_population
    from oracle import orc_py as orc

    GM, Re, J2 = 3.986004407799724e14, 6378136.3, 0.1082635854e-2
    best = np.array([1.01709935e-06, 7.86443375e-01, 7.50341883e-09, 8.63934252e-14, 4.63822910e-02, 1.86080048e-01, 2.48667176e-02, 4.81080852e-03, 5.01594516e00, 2.28429809e01, 4.27957829e00, 1.56291673e-01])
    al, be, ga = best[:4], best[4:8] / 1000, best[8:] * 1000

    def rhs(t, s, bstar, j2, drag):
        x, y, z, vx, vy, vz = s
        r2 = x * x + y * y + z * z
        r = np.sqrt(r2)
        a = -GM * np.array([x, y, z]) / r**3
        if j2:
            k = 1.5 * J2 * GM * Re**2
            a = a + k / r**5 * np.array([x * (5 * z * z / r2 - 1), y * (5 * z * z / r2 - 1), z * (5 * z * z / r2 - 3)])
        if drag:
            dens = np.sum(al * np.exp(-((r - Re) - ga) * be))
            v = np.sqrt(vx * vx + vy * vy + vz * vz)
            a = a - dens / (0.1570 / Re) * bstar * v * np.array([vx, vy, vz])
        return [vx, vy, vz, *a]

    st, pars = leo_si_population(5, seed=8)
    ct = 185.5663
    for kind in (1, 3):
        s = orc.Sim(st.copy(), ct, n_par_ct=4, dyn_kind=kind, mu=GM, J2=J2, Re=Re, pars=pars)
        s.step()
        s.step()
        for i in range(len(st)):
            sol = solve_ivp(rhs, (0, s.time), st[i, :6], args=(pars[i, 0], True, kind == 3), method="DOP853", rtol=1e-13, atol=1e-6)
            ref = sol.y[:3, -1]
            assert np.max(np.abs(s.state[i, :3] - ref)) <= 1e-10 * np.linalg.norm(ref), (kind, i)
