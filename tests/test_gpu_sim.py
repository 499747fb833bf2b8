"""GPU: the reference's known-answer tests run against cascade_amd.sim (the drop-in for
cascade.sim, cascade.py/core.cpp:87-330), and whole-step parity against the CPU oracle on
random populations (event sets equal, times and states within 1e-9 relative)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PSIZE = 1.57e-8
PI = math.pi
def _ii(info):
    """interrupt_info as the reference's Python class gives it: [i, j] | i | (i, t); lists are
    compared as tuples."""
    return tuple(info) if isinstance(info, list) else info


RTOL = 1e-9


def k2c(orc, kep):
    x, v = orc.kep_to_cart_test(kep, 1.0)
    return list(x) + list(v)


def test_python_conjunction_kat():
    """cascade.py/test.py:155-208."""
    from cascade_amd import sim

    r1 = (0.3342377271241684, 0.942488801930755, 0.0)
    v1 = (0.009424730938644767, -0.003342321565239028, 0.9999500004176654)
    r2 = (-0.15179985849820377, -0.988411251938142, 0.0)
    v2 = (6.052273379648475e-17, -9.295060541060909e-18, 1.000000000001)
    s = sim([list(r1) + list(v1) + [PSIZE], list(r2) + list(v2) + [PSIZE]], 0.23, conj_thresh=PSIZE * 100000000)
    cv0 = s.conjunctions
    assert len(cv0) == 0
    s.propagate_until(2.0)
    cv1 = s.conjunctions
    assert len(cv1) == 1
    s.propagate_until(20.0)
    cv2 = s.conjunctions
    assert len(cv2) == 6
    s.reset_conjunctions()
    assert len(s.conjunctions) == 0
    del s
    assert len(cv0) == 0 and len(cv1) == 1 and len(cv2) == 6
    assert np.all(cv2["dist"] == cv2["dist"])
    with pytest.raises(ValueError):
        cv2["dist"][0] = 0


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_multiple_conjs_and_discard(orc, n_par_ct):
    """test/conj_tracking.cpp:228-297."""
    from cascade_amd import outcome, sim

    p1 = k2c(orc, (1.0, 0.000005, PI / 2 + 1e-6, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p3 = k2c(orc, (1.0 + 1e-7, 0.0, 0.0, 0.0, 0.0, 0))
    p4 = list(p3)
    p4[0] = -p4[0] + 1e-6
    s = sim([p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]], 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while True:
        assert s.step() == outcome.success
        if s.time > 26.0:
            break
    c = s.conjunctions
    assert len(c) == 8 and np.all(np.diff(c["time"]) >= 0) and c["time"][-1] < s.time
    p4[0] = -p3[0]
    s = sim([p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]], 0.23, conj_thresh=PSIZE * 100, n_par_ct=n_par_ct)
    while True:
        oc = s.step()
        if oc != outcome.success:
            assert oc == outcome.collision
            break
    c = s.conjunctions
    assert len(c) == 1 and c[0]["i"] == 0 and c[0]["j"] == 1 and c[0]["time"] < s.time


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_interrupt_order(n_par_ct):
    """test/interrupt_order.cpp:25-130."""
    from cascade_amd import outcome, sim

    s = sim([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [4.9, 0, 0, 1, 0, 0, 1e-6]], 0.23, exit_radius=5.0, reentry_radius=0.75, n_par_ct=n_par_ct)
    d2 = lambda a, b: float(np.sum((s.state[a, :3] - s.state[b, :3]) ** 2))
    r2 = lambda a: float(np.sum(s.state[a, :3] ** 2))
    assert s.propagate_until(1000.0) == outcome.collision
    assert _ii(s.interrupt_info) == (0, 1) and abs(d2(0, 1) - 4e-12) < 1e-14

    def rerun(st):
        s.state[:] = np.array(st)
        s.time = 0.0
        return s.propagate_until(1000.0)

    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [4.9, 0, 0, 1000, 0, 0, 1e-6]]) == outcome.exit
    assert _ii(s.interrupt_info) == 2 and abs(r2(2) - 25.0) < 1e-14
    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [0.76, 0, 0, 0, 0, 0, 1e-6]]) == outcome.collision
    assert _ii(s.interrupt_info) == (0, 1) and abs(d2(0, 1) - 4e-12) < 1e-14
    assert rerun([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [0.76, 0, 0, -1, 0, 0, 1e-6]]) == outcome.reentry
    assert _ii(s.interrupt_info) == 2 and abs(r2(2) - 0.75 * 0.75) < 1e-14
    assert rerun([[0.76, 0, 0, -1, 0, 0, 1e-6], [1.0, 0, 0, 0, 1, 0, 1e-6], [2, 0, 0, 1, 0, 0, 1e-6]]) == outcome.reentry
    assert _ii(s.interrupt_info) == 0 and abs(r2(0) - 0.75 * 0.75) < 1e-14
    assert rerun([[0.76, 0, 0, 0, 0, 0, 1e-6], [1.0, 0, 0, 0, 1, 0, 1e-6], [4.9, 0, 0, 1, 0, 0, 1e-6]]) == outcome.exit
    assert _ii(s.interrupt_info) == 2 and abs(r2(2) - 25.0) < 1e-14


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_single_collision_and_reentry(orc, n_par_ct):
    """test/single_collision.cpp:35-117, test/reentry.cpp:98-215, test/invalid_state.cpp."""
    from cascade_amd import outcome, sim

    eps = np.finfo(float).eps
    p1 = k2c(orc, (1.05, 0.05, PI / 2, 0, 1.23, 0))
    p2 = k2c(orc, (1.05, 0.05, PI / 2, 0, 4.56, 0))
    s0 = np.array([p1 + [PSIZE], p2 + [PSIZE]])
    s = sim(s0, 0.23, n_par_ct=n_par_ct)
    while s.step() != outcome.collision:
        pass
    assert _ii(s.interrupt_info) == (0, 1)
    assert float(np.sum((s.state[0, :3] - s.state[1, :3]) ** 2)) - 4 * PSIZE * PSIZE < eps * 10
    s.state[:] = s0
    s.time = 0
    assert s.propagate_until(0.5) == outcome.time_limit and s.interrupt_info is None and s.time == 0.5
    s.state[:] = s0
    s.time = 0
    assert s.propagate_until(1.57) == outcome.time_limit and abs(s.time - 1.57) < 1e-15

    s = sim([[1.5, 0, 0, 0, 0.0, 0.0, 0]], 0.23, reentry_radius=[1.0, 1.1, 1.2], n_par_ct=n_par_ct)
    rn = lambda i: float(np.sqrt(np.sum(s.state[i, :3] ** 2)))
    assert s.propagate_until(1000.0) == outcome.reentry and abs(rn(0) - 1.0) < 1e-14
    s.state[:] = [[0.0, 1.5, 0, 0, 0.0, 0.0, 0]]
    s.time = 0.0
    assert s.propagate_until(1000) == outcome.reentry and abs(rn(0) - 1.1) < 1e-14
    c = [0.0, 2.1, 0.0, 0.69, 0.0, 0.0, 0.0]
    s.set_new_state_pars([c, c, c, [0.0, 1.5, 0.0, 0.0, 0.0, 0.0, 0.0], c, c])
    s.time = 0.0
    assert s.propagate_until(1000) == outcome.reentry and _ii(s.interrupt_info) == 3 and abs(rn(3) - 1.1) < 1e-14

    # invalid states: nothing user-visible changes (src/sim_propagate.cpp:513-515)
    e1 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    e2 = list(e1)
    e2[0] = -e2[0]
    s0 = np.array([e1 + [PSIZE], e2 + [PSIZE]])
    s = sim(s0, 0.23)
    s.state[0, 6] = -1
    with pytest.raises(ValueError, match="An invalid particle size of -1 was detected for the particle at index 0"):
        s.step()
    s.state[0, 6] = PSIZE
    assert np.array_equal(s.state, s0) and s.time == 0.0
    s.state[0, 0] = np.inf
    assert s.step() == outcome.err_nf_state and _ii(s.interrupt_info) == (0, 0.0)
    s.state[0, 0] = s0[0, 0]
    assert np.array_equal(s.state, s0) and s.time == 0.0


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_conj_filtering_counts(orc, n_par_ct):
    """test/coll_conj_filter.cpp:95-179: conjunction counts 0, 3, 3, 6, 3, 3."""
    from cascade_amd import outcome, sim

    p1 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 1.23, 0))
    p2 = k2c(orc, (1.0, 0.000005, PI / 2, 0, 4.56, 0))
    p3 = k2c(orc, (1.05, 0.0, 0.0, 0.0, 0.0, 0))
    p4 = list(p3)
    p4[0] = -p4[0]
    ic = np.array([p1 + [PSIZE], p2 + [PSIZE]])
    ic4 = np.array([p1 + [PSIZE], p2 + [PSIZE], p3 + [PSIZE], p4 + [PSIZE]])
    s = sim(ic, 0.23, min_coll_radius=float("inf"), n_par_ct=n_par_ct)
    assert s.propagate_until(10.0) == outcome.time_limit and len(s.conjunctions) == 0

    def run(st, wl, expect):
        s.time = 0
        s.set_new_state_pars(st)
        s.conj_whitelist = wl
        s.reset_conjunctions()
        assert s.propagate_until(10.0) == outcome.time_limit
        assert len(s.conjunctions) == expect
        return s.conjunctions

    s.conj_thresh = PSIZE / 2
    run(ic, set(), 3)
    run(ic, {0}, 3)
    run(ic4, set(), 6)
    c = run(ic4, {0}, 3)
    assert np.all(c["i"] == 0) and np.all(c["j"] == 1)
    c = run(ic4, {2, 3}, 3)
    assert np.all(c["i"] == 2) and np.all(c["j"] == 3)


@pytest.mark.parametrize("n,n_par_ct,conj,scale", [(3000, 1, 0.0, 1.0), (3000, 3, 1e-3, 1.0), (2000, 4, 0.0, 3e4)])
def test_steps_vs_oracle(orc, n, n_par_ct, conj, scale):
    """Whole-step parity with the CPU oracle over several supersteps: same outcomes and
    interrupt info, same conjunction sets, times and states within 1e-9 relative."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(n, seed=5)
    st[:, 6] *= scale
    g = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
    o = orc.Sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
    n_int = 0
    for _ in range(6):
        og, oo = g.step(), o.step()
        assert og.name == oo
        assert _ii(g.interrupt_info) == o.interrupt_info
        n_int += og.name != "success"
        assert abs(g.time - o.time) <= RTOL * max(1.0, abs(o.time))
        assert np.all(np.abs(g.state - o.state) <= RTOL * np.maximum(1.0, np.abs(o.state)))
        assert g.last_bp_pairs == int(o.stage_times()["n_bp_pairs"])
    cg, co = g.conjunctions, o.conjunctions
    assert len(cg) == len(co)
    if len(co):
        assert np.array_equal(cg["i"], co["i"]) and np.array_equal(cg["j"], co["j"])
        assert np.all(np.abs(cg["time"] - co["time"]) <= RTOL * np.maximum(1.0, np.abs(co["time"])))
        assert np.all(np.abs(cg["dist"] - co["dist"]) <= 1e-9)
    if scale > 1:
        assert n_int > 0, "the scaled population should produce collisions"


@pytest.mark.parametrize("n,n_par_ct,conj,scale", [(20000, 1, 0.0, 1.0), (8000, 1, 2e-3, 3e3), (6000, 4, 1e-3, 3e4), (4000, 2, 0.0, 1e5)])
def test_prefilter_changes_nothing(n, n_par_ct, conj, scale):
    """The geometric pre-filter of the narrow phase (k_np_prefilter, no counterpart in the
    reference) must not change a single event: same collisions and conjunctions, bit for bit,
    with and without it, while most AABB pairs never reach the polynomial test."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(n, seed=11)
    st[:, 6] *= scale
    a = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
    b = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
    b.np_prefilter = False
    n_events = 0
    for _ in range(4):
        oa, ob = a.step(), b.step()
        assert oa == ob and a.interrupt_info == b.interrupt_info
        assert a.last_bp_pairs == b.last_bp_pairs
        assert b.last_np_candidates == b.last_bp_pairs
        assert a.last_np_candidates <= a.last_bp_pairs
        assert a.last_collisions.tobytes() == b.last_collisions.tobytes()
        assert a.last_step_conjunctions.tobytes() == b.last_step_conjunctions.tobytes()
        assert np.array_equal(a.state, b.state)
        n_events += len(a.last_collisions) + len(a.last_step_conjunctions)
        if a.last_bp_pairs > 1000:
            assert a.last_np_candidates < 0.5 * a.last_bp_pairs, (a.last_np_candidates, a.last_bp_pairs)
    if scale > 1 or conj > 0:
        assert n_events > 0, "the test population should produce events"


@pytest.mark.parametrize("kind,n,n_par_ct,conj", [("j2", 3000, 3, 0.0), ("j2drag", 3000, 4, 5000.0), ("j2drag", 1500, 1, 0.0)])
@pytest.mark.gpu
def test_perturbed_dynamics_vs_oracle(orc, kind, n, n_par_ct, conj):
    """J2 and J2 + drag (simple_earth, SI units): whole-step parity with the CPU oracle -- same
    outcomes, pair counts and conjunction sets, states within 1e-9 relative."""
    from cascade_amd import dynamics, sim
    from cascade_amd.workloads import leo_si_population

    st, pars = leo_si_population(n, seed=3)
    st[:, 6] *= 50.0  # bigger objects: a few collisions
    drag = kind == "j2drag"
    dyn = dynamics.simple_earth(J2=True, C22S22=False, drag=drag)
    ct = 185.5663
    rr = 6378136.3 + 150e3
    g = sim(st.copy(), ct, dyn=dyn, pars=pars if drag else None, n_par_ct=n_par_ct, conj_thresh=conj, reentry_radius=rr)
    o = orc.Sim(st.copy(), ct, n_par_ct=n_par_ct, conj_thresh=conj, dyn_kind=3 if drag else 1, mu=dyn.mu, J2=dyn.J2, Re=dyn.Re, reentry_radius=rr, pars=pars if drag else None)
    assert g.npars == (1 if drag else 0)
    for _ in range(4):
        og, oo = g.step(), o.step()
        assert og.name == oo
        assert _ii(g.interrupt_info) == o.interrupt_info
        assert abs(g.time - o.time) <= RTOL * max(1.0, abs(o.time))
        assert np.all(np.abs(g.state - o.state) <= RTOL * np.maximum(1.0, np.abs(o.state)))
        assert g.last_bp_pairs == int(o.stage_times()["n_bp_pairs"])
        if og.name in ("reentry", "collision"):
            # Resolve the interrupt the same way in both simulations (the benchmark removes the
            # objects, benchmark/2022_leo_pop.cpp:311-337; here they are lifted to a higher
            # orbit, which keeps the indices): a re-step without that re-triggers the event.
            ii = g.interrupt_info
            hit = [ii] if og.name == "reentry" else [ii[1]]
            so = o.state
            for k in hit:
                g.state[k, :3] *= 1.3
                so[k, :3] *= 1.3
            o.poke_state(so)
    cg, co = g.conjunctions, o.conjunctions
    assert len(cg) == len(co)
    if len(co):
        assert np.array_equal(cg["i"], co["i"]) and np.array_equal(cg["j"], co["j"])
        assert np.all(np.abs(cg["time"] - co["time"]) <= RTOL * np.maximum(1.0, np.abs(co["time"])))


@pytest.mark.gpu
def test_drag_decays_and_reenters():
    """A low object with a large BSTAR loses altitude under drag and triggers the reentry event."""
    from cascade_amd import dynamics, outcome, sim

    Re, GM = 6378136.3, 3.986004407799724e14
    r0 = Re + 180e3
    st = np.array([[r0, 0.0, 0.0, 0.0, np.sqrt(GM / r0), 0.0, 1.0]])
    s = sim(st, 185.5663, dyn=dynamics.simple_earth(C22S22=False), pars=np.array([[5e-2]]), n_par_ct=8, reentry_radius=Re + 150e3)
    oc = s.propagate_until(86400.0 * 2)
    assert oc == outcome.reentry and s.interrupt_info == 0
    assert abs(np.linalg.norm(s.state[0, :3]) - (Re + 150e3)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind,wave", [("j2drag", 0), ("tape", 0), ("tape", 1)])
def test_single_pass_rows_change_nothing(kind, wave, monkeypatch):
    """Dynamics other than plain Kepler write their dense-output rows during the one integration
    (csc_propagate_rows) instead of re-integrating in a second pass. With the one-lane-per-object
    kernels on both sides (CASCADE_WAVE=0) every event and state must be bit-identical with the
    two-pass path, including a superstep whose row capacity overflows. With the several-lanes-per-
    object kernel (k_propagate_wave, the default) the sums are taken in another order: the same
    events, times and states to 1e-9 relative."""
    from cascade_amd import _capi, dynamics, sim
    from cascade_amd.workloads import leo_si_population

    st, pars = leo_si_population(6000, seed=12)
    st[:, 6] *= 200.0
    dyn = dynamics.simple_earth(C22S22=(kind == "tape"))
    kw = dict(dyn=dyn, pars=pars, n_par_ct=4, conj_thresh=20e3, reentry_radius=6378136.3 + 150e3)
    monkeypatch.setenv("CASCADE_WAVE", str(wave))
    monkeypatch.setenv("CASCADE_SINGLE_PASS", "0")
    a = sim(st.copy(), 185.5663, **kw)
    monkeypatch.setenv("CASCADE_SINGLE_PASS", "1")
    b = sim(st.copy(), 185.5663, **kw)
    ctx = _capi.Context.from_handle(b._hip_context)
    ctx.profile(True)
    n_events = 0

    def same(x, y):
        if not wave:
            return x.tobytes() == y.tobytes()
        if len(x) != len(y):
            return False
        for f in x.dtype.names:
            u, v = x[f], y[f]
            if u.dtype.kind in "ui":
                if not np.array_equal(u, v):
                    return False
            else:
                # (a vector field is compared at the scale of its largest component: 1e-9 of |r|
                # for a position, not of a coordinate that happens to be near zero)
                # (the miss distance is a difference of positions: 1e-9 of |r| as well)
                w = y["state_i"] if f == "dist" else v
                scale = np.abs(w).reshape(len(w), -1).max(axis=1).reshape((-1,) + (1,) * (v.ndim - 1)) if len(v) else 1.0
                if not np.all(np.abs(u - v) <= RTOL * np.maximum(1.0, scale)):
                    print("field", f, "max diff", np.max(np.abs(u - v)))
                    return False
        return True

    for k in range(5):
        if k == 3:
            # Longer chunks at the same n_par_ct would keep the hint; a new n_par_ct drops it,
            # and the larger superstep after that exercises hint -> capacity again.
            a.n_par_ct = b.n_par_ct = 9
        oa, ob = a.step(), b.step()
        assert oa == ob and a.interrupt_info == b.interrupt_info
        assert same(a.last_collisions, b.last_collisions)
        assert same(a.last_step_conjunctions, b.last_step_conjunctions)
        if wave:
            scale = np.maximum(1.0, np.abs(b.state[:, :6]).max(axis=1))[:, None]
            assert np.all(np.abs(a.state[:, :6] - b.state[:, :6]) <= RTOL * scale) and abs(a.time - b.time) <= 1e-9 * a.time
        else:
            assert np.array_equal(a.state, b.state) and a.time == b.time
        n_events += len(a.last_collisions) + len(a.last_step_conjunctions)
        if oa.name in ("reentry", "collision"):
            ii = a.interrupt_info
            for s in (a, b):
                s.remove_particles([ii] if isinstance(ii, int) else list(ii))
    prof = ctx.profile_get()
    assert prof["k_propagate_wave" if wave else "k_propagate_rows"][0] >= 2, prof.keys()
    assert n_events > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kepler", "j2drag", "tape_wave"])
def test_row_overflow_falls_back(kind, monkeypatch):
    """A particle that takes more substeps than it has row slots: its rows, and the boxes computed
    from them (k_chunk_boxes), are incomplete; the superstep is integrated again without rows and
    gives the same events and states as a run whose slots suffice. (tape_wave: the several-lanes-
    per-object kernel counts the substeps in full and is run again with exactly enough slots.)"""
    from cascade_amd import _capi, dynamics, sim
    from cascade_amd.workloads import leo_si_population, random_leo_pop_kep

    if kind == "kepler":
        st = random_leo_pop_kep(20000, seed=5)
        st[:, 6] *= 3e3
        args, kw = (st, 0.23), dict(n_par_ct=8, conj_thresh=2e-3)
    else:
        st, pars = leo_si_population(5000, seed=3)
        st[:, 6] *= 200.0
        args, kw = (st, 185.5663), dict(dyn=dynamics.simple_earth(C22S22=(kind == "tape_wave")), pars=pars, n_par_ct=6, conj_thresh=20e3)
    monkeypatch.setenv("CASCADE_ROWS_ON_DEMAND", "0")  # (Kepler: rows for every particle, written in pass 1)
    monkeypatch.setenv("CASCADE_WAVE", "1" if kind == "tape_wave" else "0")
    a = sim(args[0].copy(), args[1], **kw)
    monkeypatch.setenv("CASCADE_ROW_CAP", "1")
    b = sim(args[0].copy(), args[1], **kw)
    ctx = _capi.Context.from_handle(b._hip_context)
    ctx.profile(True)
    n_events = 0
    for _ in range(3):
        oa, ob = a.step(), b.step()
        assert oa == ob and a.interrupt_info == b.interrupt_info
        assert a.last_collisions.tobytes() == b.last_collisions.tobytes()
        assert a.last_step_conjunctions.tobytes() == b.last_step_conjunctions.tobytes()
        assert np.array_equal(a.state, b.state) and a.time == b.time
        n_events += len(a.last_collisions) + len(a.last_step_conjunctions)
        if oa.name in ("reentry", "collision"):
            ii = a.interrupt_info
            for s in (a, b):
                s.remove_particles([ii] if isinstance(ii, int) else list(ii))
    prof = ctx.profile_get()
    if kind == "tape_wave":
        # supersteps 2 and 3 had one slot each, overflowed and ran again with the slots they needed
        assert prof["k_propagate_wave"][0] >= 5 and "k_propagate" not in prof, {k: v[0] for k, v in prof.items() if "propagate" in k}
        return
    # supersteps 2 and 3 tried rows (one slot each), overflowed and were integrated again without
    assert prof["k_propagate_rows"][0] >= 2 and prof["k_chunk_boxes"][0] >= 2 and prof["k_propagate"][0] >= 3, {k: v[0] for k, v in prof.items() if "propagate" in k or "boxes" in k}
    assert n_events > 0


@pytest.mark.gpu
def test_broad_phase_formulations_agree(monkeypatch):
    """The dual traversal (default) and the per-particle cooperative walks (fallback for trees
    beyond the dual traversal's item encoding, CASCADE_BP_MODE=walk) give the same pairs, events
    and states; so does a superstep with more chunks than the encoding holds (n_par_ct > 256)."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(30000, seed=21)
    st[:, 6] *= 2e3
    def run(mode):
        # The mode is read when the simulation's HIP context runs its first broad phase.
        if mode:
            monkeypatch.setenv("CASCADE_BP_MODE", mode)
        else:
            monkeypatch.delenv("CASCADE_BP_MODE", raising=False)
        s = sim(st.copy(), 0.23, n_par_ct=3, conj_thresh=2e-3)
        out = []
        for _ in range(3):
            oc = s.step()
            out.append((oc, s.interrupt_info, s.last_bp_pairs, s.last_bp_node_pairs, s.last_collisions.tobytes(), s.last_step_conjunctions.tobytes(), s.state.copy()))
        return out

    ra, rb = run("walk"), run(None)
    for x, y in zip(ra, rb):
        assert x[0] == y[0] and x[1] == y[1]
        assert x[2] == y[2] and x[2] > 0
        assert x[3] == 0 and y[3] > 0
        assert x[4] == y[4] and x[5] == y[5]
        assert np.array_equal(x[6], y[6])
    c = sim(st[:300].copy(), 0.001, n_par_ct=300)
    d = sim(st[:300].copy(), 0.001 * 300, n_par_ct=1)
    assert c.step() == d.step()
    assert c.last_bp_node_pairs == 0 and d.last_bp_node_pairs > 0
    assert np.allclose(c.state, d.state, rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_conj_register_search_same_bits(n_par_ct, monkeypatch):
    """The in-register conjunction search of k_np_roots (np_conj_fast: one sign change at the first
    node of the isolation) against the general root isolation on the same encounters
    (CASCADE_NP_CONJ_FAST=0): the conjunction arrays are the same bytes."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(30000, 11)
    out = []
    for fast in ("0", "1"):
        monkeypatch.setenv("CASCADE_NP_CONJ_FAST", fast)
        s = sim(st, 0.23, n_par_ct=n_par_ct, conj_thresh=4e-3, min_coll_radius=float("inf"))
        for _ in range(2):
            s.step()
        out.append(np.array(s.conjunctions))
    assert len(out[0]) > 1000
    assert out[0].tobytes() == out[1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n_par_ct,conj", [(1, 0.0), (3, 2e-3)])
def test_rows_on_demand_change_nothing(n_par_ct, conj, monkeypatch):
    """Plain Kepler writes no Taylor rows while it integrates: the particles of the candidate pairs
    are integrated again after the pre-filter (csc_narrow_phase_prefilter / csc_candidate_objects /
    csc_propagate_store on the list / csc_narrow_phase_finish), every particle only when
    dense_propagate needs them (interrupts, the last superstep of propagate_until). Against the
    rows written during the integration (CASCADE_ROWS_ON_DEMAND=0): same outcomes, states,
    conjunctions, bit for bit."""
    from cascade_amd import outcome, sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(20000, 5)
    st[:, 6] *= 5e3  # collisions within a few supersteps
    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("CASCADE_ROWS_ON_DEMAND", mode)
        s = sim(st, 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
        log = []
        for _ in range(6):
            oc = s.step()
            info = s.interrupt_info
            log.append((oc, tuple(info) if isinstance(info, list) else info, s.time))
            if oc == outcome.collision:
                s.remove_particles(list(info))
        oc = s.propagate_until(s.time + 0.23 * n_par_ct * 2.4)
        log.append((oc, s.time))
        res.append((log, np.array(s.state), np.array(s.conjunctions)))
    assert any(oc == outcome.collision for oc, *_ in res[0][0][:-1])
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][2].tobytes() == res[1][2].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("tol,order,kind", [(1e-12, 15, "kepler"), (1e-9, 12, "j2drag"), (1e-14, 18, "kepler_events"), (1e-17, 21, "kepler")])
def test_tolerance_selects_the_taylor_order(orc, tol, order, kind):
    """kw::tol (src/sim_basic.cpp:553-577; test/sim_api.cpp:54-83 constructs with tol = 1e-12): the
    Taylor order follows heyoka's rule p = ceil(-ln(tol) / 2 + 1) and the order-p kernels
    (cascade_order.hip) run the whole step: parity with the oracle at the same order."""
    from cascade_amd import dynamics, sim
    from cascade_amd.workloads import leo_si_population, random_leo_pop_kep

    if kind == "j2drag":
        st, pars = leo_si_population(2500, seed=5)
        st[:, 6] *= 50.0
        dyn = dynamics.simple_earth(J2=True, C22S22=False, drag=True)
        rr = 6378136.3 + 150e3
        g = sim(st.copy(), 185.5663, dyn=dyn, pars=pars, n_par_ct=4, conj_thresh=5000.0, reentry_radius=rr, tol=tol)
        o = orc.Sim(st.copy(), 185.5663, n_par_ct=4, conj_thresh=5000.0, dyn_kind=3, mu=dyn.mu, J2=dyn.J2, Re=dyn.Re, reentry_radius=rr, pars=pars, tol=tol)
    else:
        st = random_leo_pop_kep(3000, seed=13)
        st[:, 6] *= 3e4 if kind == "kepler" else 1.0
        kw = dict(n_par_ct=3, conj_thresh=1e-3)
        if kind == "kepler_events":
            kw.update(exit_radius=1.299, reentry_radius=1.0201)
            keep = (np.linalg.norm(st[:, :3], axis=1) < 1.298) & (np.linalg.norm(st[:, :3], axis=1) > 1.021)
            st = st[keep]
        g = sim(st.copy(), 0.23, tol=tol, **kw)
        o = orc.Sim(st.copy(), 0.23, tol=tol, **kw)
    assert g.tol == tol and o.order == order
    n_int = 0
    for _ in range(5):
        og, oo = g.step(), o.step()
        assert og.name == oo
        assert _ii(g.interrupt_info) == o.interrupt_info
        assert abs(g.time - o.time) <= RTOL * max(1.0, abs(o.time))
        assert np.all(np.abs(g.state - o.state) <= RTOL * np.maximum(1.0, np.abs(o.state)))
        assert g.last_bp_pairs == int(o.stage_times()["n_bp_pairs"])
        if oo != "success":
            n_int += 1
            ii = g.interrupt_info
            idx = [ii] if isinstance(ii, int) else list(ii)
            g.remove_particles(idx)
            o.remove_particles(idx)
    cg, co = g.conjunctions, o.conjunctions
    assert len(cg) == len(co)
    if len(co):
        assert np.array_equal(cg["i"], co["i"]) and np.array_equal(cg["j"], co["j"])
        assert np.all(np.abs(cg["time"] - co["time"]) <= RTOL * np.maximum(1.0, np.abs(co["time"])))
    if kind != "j2drag":
        assert n_int > 0 or kind == "kepler"


def test_tolerance_out_of_range_is_refused():
    from cascade_amd import sim

    with pytest.raises(ValueError, match="needs Taylor order"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, tol=1e-1)


@pytest.mark.gpu
def test_host_edits_between_steps_are_honoured_without_any_flag():
    """The state / pars arrays are the caller's (cascade.py/core.cpp:199-230): default mode, no
    switch touched. A step on unedited arrays runs on the device copy (checked by hash, no upload)
    and gives the bits of the resident mode; an in-place edit of the state view or of the pars view
    between two steps is picked up by the next step; an invalid edit raises the reference's error
    and leaves everything as it was."""
    from cascade_amd import dynamics, sim
    from cascade_amd.workloads import leo_si_population

    n = 120000  # 6.7 MB of state: the pipelined transfer path
    st, pars = leo_si_population(n, seed=31)
    kw = dict(dyn=dynamics.simple_earth(J2=True, C22S22=False, drag=True), n_par_ct=2, reentry_radius=6378136.3 + 150e3)
    a = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)  # default: host arrays authoritative
    b = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
    b.host_sync = False
    view, pview = a.state, a.pars
    for k in range(3):
        assert a.step() == b.step()
    b.download_state()
    assert np.array_equal(view, b.state), "unedited steps: same bits as the resident mode"
    # Edit through views taken BEFORE the steps (zero-copy aliases of the simulation's vectors).
    view[17, :3] *= 1.3
    pview[5, 0] = 3e-9
    c = sim(np.array(view), 185.5663, pars=np.array(pview), **kw)
    c.time = a.time
    assert a.step() == c.step()
    assert np.array_equal(a.state, c.state), "the edited state and parameter were used"
    assert not np.array_equal(a.state[17], b.state[17])
    # An invalid edit: the reference's message, nothing changes (src/sim_propagate.cpp:250-253, 513-515).
    before, t0 = np.array(a.state), a.time
    view[3, 6] = -1.0
    with pytest.raises(ValueError, match="An invalid particle size of -1 was detected for the particle at index 3"):
        a.step()
    view[3, 6] = before[3, 6]
    assert np.array_equal(a.state, before) and a.time == t0
    assert a.step() == c.step() and np.array_equal(a.state, c.state)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 1000, 120000])
def test_structured_host_edits_are_honoured(n):
    """Edits that a linear checksum cannot see (round 2's missed all four): (a) time reversal
    `state[:, 3:6] *= -1` with 3n even, (b) two rows swapped, (c) +d / -d on two words, (d) sign
    flips of two parameters. Default mode, no flag: after the next step the state must be, bit for
    bit, that of a fresh simulation started from the edited arrays
    (cascade.py/core.cpp:199-230, benchmark/haumea.cpp:203-212)."""
    from cascade_amd import dynamics, sim
    from cascade_amd.workloads import leo_si_population

    st, pars = leo_si_population(n, seed=33)
    kw = dict(dyn=dynamics.simple_earth(J2=True, C22S22=False, drag=True), n_par_ct=2)
    a = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
    view, pview = a.state, a.pars
    a.step()

    def edit_a():
        view[:, 3:6] *= -1

    def edit_b():
        view[[0, n - 1]] = view[[n - 1, 0]]
        pview[[0, n - 1]] = pview[[n - 1, 0]]

    def edit_c():
        u = view.view(np.uint64)
        u[0, 0] += 1 << 30
        u[1, 0] -= 1 << 30

    def edit_d():
        pview[0, 0] *= -1
        pview[n - 1, 0] *= -1

    for edit in (edit_a, edit_b, edit_c, edit_d):
        edit()
        c = sim(np.array(view), 185.5663, pars=np.array(pview), **kw)
        c.time = a.time
        assert a.step() == c.step(), edit.__name__
        assert np.array_equal(a.state, c.state), edit.__name__
        assert a.step() == c.step() and np.array_equal(a.state, c.state), "and an unedited step after it"


@pytest.mark.gpu
def test_trace_log_lines(capfd):
    """At trace level step() prints the reference's per-phase lines (sample in
    cascade.py/docstrings.cpp:581-596), including the tuning signal "Average number of AABB
    collisions per particle per chunk" (src/sim_broad_phase.cpp:211-213) with the right value."""
    import re

    import cascade
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(5000, seed=2)
    s = cascade.sim(st, 0.23, n_par_ct=2, conj_thresh=1e-3)
    cascade.set_logger_level_trace()
    try:
        s.step()
    finally:
        cascade.set_logger_level_info()
    out = capfd.readouterr().out
    for what in (
        "---- STEP BEGIN ---",
        "Number of chunks: 2",
        "Propagation + AABB computation time: ",
        "Morton encoding and sorting time: ",
        "BVH construction time: ",
        "Broad phase collision detection time: ",
        "Narrow phase collision detection time: ",
        "Total number of collisions detected: 0",
        "Total number of conjunctions detected: ",
        "Total propagation time: ",
        "---- STEP END ---",
    ):
        assert "[cascade] [trace] " + what in out, what
    m = re.search(r"Average number of AABB collisions per particle per chunk: ([0-9.e+-]+)", out)
    assert m and abs(float(m.group(1)) - s.last_bp_pairs / 2 / 5000) < 1e-9
    s.step()
    assert "[cascade]" not in capfd.readouterr().out, "silent at the default level"


@pytest.mark.gpu
def test_verify_mode_runs_the_reference_verifiers(monkeypatch):
    """CASCADE_VERIFY=1: the reference's debug verifiers (src/sim_propagate.cpp:1348-1380,
    src/sim_bvh.cpp:498-656, src/sim_broad_phase.cpp:221-321) on the device outputs of every stage
    of step(): silent on correct runs (same results as without), for chunked and unchunked
    supersteps, equal Morton codes (multi-particle leaves) and conjunction tracking."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(3000, seed=6)
    st[:, 6] *= 3e3
    st[7] = st[3]  # two particles with the same trajectory: equal Morton codes in every chunk
    st[7, 6] = 0.0
    st[3, 6] = 0.0
    for n_par_ct in (1, 3):
        kw = dict(n_par_ct=n_par_ct, conj_thresh=2e-3)
        monkeypatch.delenv("CASCADE_VERIFY", raising=False)
        a = sim(st.copy(), 0.23, **kw)
        monkeypatch.setenv("CASCADE_VERIFY", "1")
        b = sim(st.copy(), 0.23, **kw)
        for _ in range(3):
            assert a.step() == b.step()
            assert np.array_equal(a.state, b.state) and a.last_bp_pairs == b.last_bp_pairs
            assert a.conjunctions.tobytes() == b.conjunctions.tobytes()
    # A non-finite box is the reference's ValueError before any verifier looks.
    bad = st.copy()
    bad[0, :3] = 1e39
    s = sim(bad, 0.23)
    with pytest.raises(ValueError):
        s.step()
