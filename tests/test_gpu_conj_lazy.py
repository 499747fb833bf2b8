"""Conjunction-heavy supersteps keep their records on the device (csc_narrow_phase_lazy,
csc_conj_store_*) and `sim.conjunctions` is materialised on access. What a caller sees must be
what the eager path (CASCADE_CONJ_LAZY=0: records downloaded and appended at every step, as
src/sim_propagate.cpp:1644-1701 does) gives, byte for byte: the list, the arrays handed out
earlier (cascade.py/test.py:189-208: they outlive later steps), reset_conjunctions(), copies, and
a superstep interrupted by a collision (records at or after the collision dropped, :1329-1336)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shells(n, seed=5):
    from cascade_amd.workloads import kep_to_cart_bench

    rng = np.random.default_rng(seed)
    a = rng.choice([1.08, 1.09], n) + rng.uniform(-0.002, 0.002, n)
    inc = np.deg2rad(rng.choice([53.0, 70.0, 97.6], n) + rng.uniform(-0.5, 0.5, n))
    x, v = kep_to_cart_bench(a, rng.uniform(0, 1e-3, n), inc, rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n))
    return np.concatenate([x, v, np.full((n, 1), 1e-7)], axis=1)


def test_lazy_conjunction_records_are_the_eager_ones(monkeypatch):
    from cascade_amd import _capi, sim

    st = _shells(150000)
    kw = dict(n_par_ct=3, conj_thresh=4e-3, min_coll_radius=float("inf"))
    monkeypatch.setenv("CASCADE_CONJ_LAZY", "0")
    e = sim(st.copy(), 0.02, **kw)
    monkeypatch.setenv("CASCADE_CONJ_LAZY", "1")
    z = sim(st.copy(), 0.02, **kw)
    assert e.step() == z.step()
    zx = _capi.Context.from_handle(z._hip_context)
    assert len(e.last_step_conjunctions) > 32768, "large enough to be sorted, and left, on the device"
    assert zx.L.csc_conj_store_size(zx.h) == len(e.conjunctions), "the lazy simulation has its records in the device store"
    view1 = z.conjunctions  # materialises
    assert zx.L.csc_conj_store_size(zx.h) == 0
    assert view1.tobytes() == e.conjunctions.tobytes()
    snap1 = view1.copy()
    for _ in range(2):
        assert e.step() == z.step()
    assert z.last_step_conjunctions.tobytes() == e.last_step_conjunctions.tobytes()
    assert np.array_equal(z._conj_segments, e._conj_segments)
    c = copy.copy(z)  # (a copy of a simulation with pending records takes them along)
    assert c.conjunctions.tobytes() == e.conjunctions.tobytes()
    assert z.conjunctions.tobytes() == e.conjunctions.tobytes()
    assert view1.tobytes() == snap1.tobytes(), "an array handed out earlier is untouched by later steps"
    z.step()
    e.step()
    z.reset_conjunctions()
    e.reset_conjunctions()
    assert len(z.conjunctions) == 0 and zx.L.csc_conj_store_size(zx.h) == 0
    assert e.step() == z.step()
    assert z.conjunctions.tobytes() == e.conjunctions.tobytes() and len(z.conjunctions) > 0


def test_lazy_records_of_an_interrupted_superstep(monkeypatch):
    from cascade_amd import outcome, sim

    st = _shells(150000, seed=9)
    st[:, 6] = 4e-5  # (large enough for a collision within a few supersteps; conjunctions stay plentiful)
    kw = dict(n_par_ct=3, conj_thresh=4e-3)
    monkeypatch.setenv("CASCADE_CONJ_LAZY", "0")
    e = sim(st.copy(), 0.02, **kw)
    monkeypatch.setenv("CASCADE_CONJ_LAZY", "1")
    z = sim(st.copy(), 0.02, **kw)
    hit = False
    for _ in range(6):
        oe, oz = e.step(), z.step()
        assert oe == oz and e.interrupt_info == z.interrupt_info and e.time == z.time
        assert z.conjunctions.tobytes() == e.conjunctions.tobytes()
        if oz == outcome.collision:
            hit = True
            assert len(z.conjunctions) == 0 or z.conjunctions["time"].max() < z.time
            ii = list(z.interrupt_info)
            e.remove_particles(ii)
            z.remove_particles(ii)
    assert hit, "the population should collide within six supersteps"
