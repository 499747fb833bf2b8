"""GPU: the reference's regression fixtures, its benchmark catalogue and BASELINE configs[2]/[3] at
(or near) their real sizes, HIP path against the CPU oracle.

  * test/last_bit_flip.cpp (tests/golden/ref_lbf.npz, 19 114 objects): outcome as the reference
    asserts it, whole-step parity with the oracle, and -- on the oracle's AABBs -- Morton codes,
    permutation, BVH node bytes and pair sets bit-identical.
  * test/haumea_bug_00.cpp (tests/golden/ref_haumea_00.npz, 9 983 objects, tape dynamics).
  * BASELINE configs[2]: the real 19 647-object catalogue (tests/golden/ref_leo_19647.npz) with
    J2 + drag and reentry at n_par_ct = 30, two supersteps with the benchmark's removal loop; and
    the benchmark's own dynamics (benchmark/2022_leo_pop.cpp:218-287) through the tape.
  * BASELINE configs[3]: Walker-like shells with conjunction tracking -- oracle parity at 20 k
    objects, size-independent properties at 2 M objects.
"""
import os

import numpy as np
import pytest

from test_ref_fixtures import GME_LBF, GOLD, HAUMEA_00_TIME, haumea_00_state, lbf_state

pytestmark = pytest.mark.gpu
RTOL = 1e-9  # BASELINE.json north_star: event times and final states within 1e-9 relative


def _ii(info):
    return tuple(info) if isinstance(info, list) else info


def _same_step(g, o, og, oo):
    assert og.name == oo
    assert _ii(g.interrupt_info) == o.interrupt_info
    assert abs(g.time - o.time) <= RTOL * max(1.0, abs(o.time))
    assert np.all(np.abs(g.state - o.state) <= RTOL * np.maximum(1.0, np.abs(o.state)))
    assert g.last_bp_pairs == int(o.stage_times()["n_bp_pairs"])


def _same_conjunctions(cg, co):
    assert len(cg) == len(co)
    if len(co):
        assert np.array_equal(cg["i"], co["i"]) and np.array_equal(cg["j"], co["j"])
        assert np.all(np.abs(cg["time"] - co["time"]) <= RTOL * np.maximum(1.0, np.abs(co["time"])))
        assert np.all(np.abs(cg["dist"] - co["dist"]) <= RTOL * np.maximum(1.0, np.abs(co["dist"])))
        for f in ("state_i", "state_j"):
            assert np.all(np.abs(cg[f] - co[f]) <= RTOL * np.maximum(1.0, np.abs(co[f])))


@pytest.mark.parametrize("n_par_ct", [1, 3])
def test_last_bit_flip(orc, n_par_ct):
    """test/last_bit_flip.cpp:25-66 on the HIP class + stage parity on the oracle's AABBs."""
    from _stage_parity import check_aabb_to_pairs
    from cascade_amd import _capi, dynamics, sim

    st = lbf_state()
    ct = 0.23 * 806.81
    g = sim(st.copy(), ct, dyn=dynamics.kepler(GME_LBF), n_par_ct=n_par_ct)
    o = orc.Sim(st.copy(), ct, n_par_ct=n_par_ct, mu=GME_LBF)
    og, oo = g.step(), o.step()
    assert og.name in ("success", "collision")
    _same_step(g, o, og, oo)
    ctx = _capi.Context(0)
    try:
        assert check_aabb_to_pairs(orc, ctx, o, len(st)) == g.last_bp_pairs
    finally:
        ctx.close()


def test_haumea_bug_00(orc):
    """test/haumea_bug_00.cpp:33-131: ends, with neither err_nf_state nor time_limit; and agrees
    with the oracle (same event, same particles, states to 1e-9)."""
    from cascade_amd import sim
    from cascade_amd.workloads import haumea_dynamics

    dyn, k = haumea_dynamics()
    st = haumea_00_state()
    kw = dict(n_par_ct=2, reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
    g = sim(st.copy(), 5000.0, dyn=dyn, **kw)
    o = orc.Sim(st.copy(), 5000.0, tape=dyn.tape, **kw)
    g.time = HAUMEA_00_TIME
    o.time = HAUMEA_00_TIME
    og, oo = g.step(), o.step()
    assert og.name not in ("err_nf_state", "time_limit")
    _same_step(g, o, og, oo)


def _resolve(g, o, og):
    """benchmark/2022_leo_pop.cpp:311-337: the particles of a collision / the reentered one are removed."""
    if og.name in ("reentry", "collision", "exit"):
        ii = g.interrupt_info
        idx = [ii] if isinstance(ii, int) else list(ii)
        g.remove_particles(idx)
        o.remove_particles(idx)


def test_leo_2022_catalogue_j2_drag_vs_oracle(orc):
    """BASELINE configs[2] at its real size: the 19 647-object catalogue, J2 + drag (hand-unrolled
    kernels), reentry at Re + 150 km, ct = 185.5663 s, n_par_ct = 30 (2022_leo_pop.cpp:126,165,297)."""
    from cascade_amd import dynamics, sim
    from cascade_amd.workloads import load_leo_2022

    st, bstar, best = load_leo_2022(os.path.join(GOLD, "ref_leo_19647.npz"))
    dyn = dynamics.simple_earth(J2=True, C22S22=False, drag=True)
    ct, C_, rr = 185.5663, 30, 6378137.0 + 150e3
    g = sim(st.copy(), ct, dyn=dyn, pars=bstar.copy(), n_par_ct=C_, reentry_radius=rr)
    o = orc.Sim(st.copy(), ct, n_par_ct=C_, dyn_kind=3, mu=dyn.mu, J2=dyn.J2, Re=dyn.Re, reentry_radius=rr, pars=bstar.copy())
    outcomes = []
    for _ in range(3):
        og, oo = g.step(), o.step()
        _same_step(g, o, og, oo)
        outcomes.append(oo)
        _resolve(g, o, og)
        assert g.nparts == o.nparts
    assert len(outcomes) == 3


def test_leo_2022_benchmark_dynamics_vs_oracle(orc):
    """the benchmark's own J2 / C22S22 / drag form (2022_leo_pop.cpp:228-287, the C22
    force entering twice as there) through the expression tape, on 4 000 catalogue objects."""
    from cascade_amd import sim
    from cascade_amd.workloads import leo_2022_dynamics, load_leo_2022

    st, bstar, best = load_leo_2022(os.path.join(GOLD, "ref_leo_19647.npz"))
    n = 4000
    st, bstar = st[:n].copy(), bstar[:n].copy()
    dyn, k = leo_2022_dynamics(best)
    g = sim(st.copy(), k["ct"], dyn=dyn, pars=bstar.copy(), n_par_ct=10, reentry_radius=k["reentry_radius"], conj_thresh=5000.0)
    o = orc.Sim(st.copy(), k["ct"], n_par_ct=10, tape=dyn.tape, pars=bstar.copy(), reentry_radius=k["reentry_radius"], conj_thresh=5000.0)
    for _ in range(2):
        og, oo = g.step(), o.step()
        _same_step(g, o, og, oo)
        _resolve(g, o, og)
    _same_conjunctions(g.conjunctions, o.conjunctions)


def test_shells_vs_oracle(orc):
    """BASELINE configs[3] at oracle size: 20 000 objects in Walker-like shells, conj_thresh = 5 km
    (7.8e-4 Earth radii), collisions off: the conjunction records of three supersteps agree with
    the oracle's (same pairs in the same order, times / distances / states to 1e-9)."""
    from cascade_amd import sim
    from cascade_amd.workloads import leo_shells

    st = leo_shells(20000, seed=42)
    kw = dict(n_par_ct=4, conj_thresh=7.8e-4, min_coll_radius=float("inf"))
    g = sim(st.copy(), 0.02, **kw)
    o = orc.Sim(st.copy(), 0.02, **kw)
    for _ in range(3):
        og, oo = g.step(), o.step()
        assert oo == "success"
        _same_step(g, o, og, oo)
    cg, co = g.conjunctions, o.conjunctions
    assert len(co) > 20, "the shells should produce conjunctions"
    _same_conjunctions(cg, co)


def test_shells_2m_properties():
    """BASELINE configs[3] beyond oracle size (2 000 000 objects, n_par_ct = 2): size-independent
    properties of the conjunction list -- sorted by time, inside the superstep, i < j, unique,
    dist < conj_thresh and equal to the distance of the two recorded states -- the same records
    bit for bit with the geometric pre-filter on and off, and the first superstep, whose list
    capacities are guesses, goes through the grow-and-redo path."""
    from cascade_amd import sim
    from cascade_amd.workloads import leo_shells

    n, thr = 2_000_000, 7.8e-4
    st = leo_shells(n, seed=42)
    kw = dict(n_par_ct=2, conj_thresh=thr, min_coll_radius=float("inf"))
    a = sim(st, 0.02, **kw)
    a.host_sync = False
    b = sim(st, 0.02, **kw)
    b.host_sync = False
    b.np_prefilter = False
    attempts = []
    for k in range(2):
        t0 = a.time
        assert a.step().name == "success" and b.step().name == "success"
        attempts.append(a.last_np_attempts)
        ca, cb = a.last_step_conjunctions, b.last_step_conjunctions
        assert len(ca) > 100_000
        assert ca.tobytes() == cb.tobytes(), "the pre-filter must not change a single record"
        assert a.last_bp_pairs == b.last_bp_pairs and a.last_np_candidates < b.last_np_candidates == b.last_bp_pairs
        assert np.all(np.diff(ca["time"]) >= 0)
        assert np.all(ca["time"] >= t0) and np.all(ca["time"] < a.time)
        assert np.all(ca["i"] < ca["j"]) and ca["j"].max() < n
        key = ca["i"] * np.uint64(n) + ca["j"]
        assert np.all(ca["dist"] < thr)
        d = np.linalg.norm(ca["state_i"][:, :3] - ca["state_j"][:, :3], axis=1)
        assert np.all(np.abs(d - ca["dist"]) <= 1e-12)
        # A pair can have several conjunctions in a superstep, but never two at the same time.
        kt = np.stack([key.astype(np.float64), ca["time"]], axis=1)
        assert len(np.unique(kt, axis=0)) == len(kt)
        # The recorded states lie on the two Keplerian orbits: energy of each object as at t = 0.
        for f, idx in (("state_i", ca["i"]), ("state_j", ca["j"])):
            s0 = st[idx.astype(np.int64)]
            e0 = 0.5 * np.sum(s0[:, 3:6] ** 2, axis=1) - 1.0 / np.linalg.norm(s0[:, :3], axis=1)
            e1 = 0.5 * np.sum(ca[f][:, 3:6] ** 2, axis=1) - 1.0 / np.linalg.norm(ca[f][:, :3], axis=1)
            assert np.max(np.abs(e1 - e0)) < 1e-12
    assert attempts[0] > 1, "the first superstep should overflow its guessed capacities"
    assert len(a.conjunctions) == len(b.conjunctions)


def test_shells_10m_properties():
    """BASELINE configs[3] at its full size (10 000 000 objects, n_par_ct = 2, conj_thresh = 5 km):
    two supersteps with the records left on the device, then the size-independent properties of
    the ~2.8e7 conjunction records -- per superstep sorted by time and inside the superstep, i < j,
    dist < conj_thresh and equal to the distance of the two recorded states, recorded states on the
    two Keplerian orbits -- and the device store's count equal to what the list then holds."""
    from cascade_amd import _capi, sim
    from cascade_amd.workloads import leo_shells

    n, thr = 10_000_000, 7.8e-4
    st = leo_shells(n, seed=42)
    s = sim(st, 0.02, n_par_ct=2, conj_thresh=thr, min_coll_radius=float("inf"))
    s.host_sync = False
    cx = _capi.Context.from_handle(s._hip_context)
    times = [s.time]
    for _ in range(2):
        assert s.step().name == "success"
        times.append(s.time)
    pending = int(cx.L.csc_conj_store_size(cx.h))
    assert pending > 10_000_000, "the records of both supersteps wait on the device"
    conj = s.conjunctions
    assert len(conj) == pending and cx.L.csc_conj_store_size(cx.h) == 0
    seg = np.asarray(s._conj_segments, dtype=np.int64)[:, 1]  # (superstep, records appended by it)
    assert seg.sum() == len(conj) and len(seg) == 2
    lo = 0
    for k, cnt in enumerate(seg):
        c = conj[lo : lo + cnt]
        lo += cnt
        assert cnt > 5_000_000
        assert np.all(np.diff(c["time"]) >= 0)
        assert c["time"][0] >= times[k] and c["time"][-1] < times[k + 1]
    assert np.all(conj["i"] < conj["j"]) and conj["j"].max() < n
    assert np.all(conj["dist"] < thr)
    # (the rest on every 16th record: 1.8e6 of them)
    c = conj[::16]
    d = np.linalg.norm(c["state_i"][:, :3] - c["state_j"][:, :3], axis=1)
    assert np.all(np.abs(d - c["dist"]) <= 1e-12)
    for f, idx in (("state_i", c["i"]), ("state_j", c["j"])):
        s0 = st[idx.astype(np.int64)]
        e0 = 0.5 * np.sum(s0[:, 3:6] ** 2, axis=1) - 1.0 / np.linalg.norm(s0[:, :3], axis=1)
        e1 = 0.5 * np.sum(c[f][:, 3:6] ** 2, axis=1) - 1.0 / np.linalg.norm(c[f][:, :3], axis=1)
        assert np.max(np.abs(e1 - e0)) < 1e-12
