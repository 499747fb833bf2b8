"""GPU: removal of particles on the device copies (remove_particles,
src/sim_basic.cpp:280-338) -- csc_remove_particles through the C ABI against numpy.delete, and the
reference drivers' remove-and-continue loop (benchmark/2022_leo_pop.cpp:311-337) with the state
resident in HBM against the same loop with the host vectors authoritative: bit-identical."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cascade_amd import _capi

    c = _capi.Context(0)
    yield c
    c.close()


def _remove(ctx, state, pars, idx):
    n, npars = state.shape[0], (pars.shape[1] if pars is not None else 0)
    d_state = ctx.array(state, np.float64)
    d_pars = ctx.array(pars, np.float64) if npars else None
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    n_out = C.c_uint32(0xFFFFFFFF)
    rc = ctx.L.csc_remove_particles(ctx.h, n, idx.ctypes.data_as(C.c_void_p), len(idx), d_state.ptr, npars,
                                    d_pars.ptr if npars else None, C.byref(n_out))
    return rc, n_out.value, d_state.download(), (d_pars.download() if npars else None)


@pytest.mark.parametrize("n,npars,nrm", [(1, 0, 1), (7, 1, 3), (1000, 2, 1), (1000, 0, 999), (100003, 1, 5000), (100003, 9, 100003), (64, 1, 0)])
def test_remove_rows_equals_numpy_delete(ctx, n, npars, nrm):
    rng = np.random.default_rng(n + npars)
    state = rng.standard_normal((n, 7))
    pars = rng.standard_normal((n, npars)) if npars else None
    idx = np.sort(rng.choice(n, size=nrm, replace=False))
    rc, n_out, st, ps = _remove(ctx, state, pars, idx)
    assert rc == 0 and n_out == n - nrm
    assert np.array_equal(st[: n - nrm], np.delete(state, idx, axis=0))
    assert np.array_equal(st[n - nrm :], state[n - nrm :])  # rows beyond the new end are left alone
    if npars:
        assert np.array_equal(ps[: n - nrm], np.delete(pars, idx, axis=0))


def test_remove_rejects_bad_indices(ctx):
    state = np.zeros((10, 7))
    for bad in ([3, 2], [2, 2], [10], [0, 11]):
        rc, _, st, _ = _remove(ctx, state, None, bad)
        assert rc == -2  # CSC_ERR_INVALID
        assert b"sorted, unique" in ctx.L.csc_last_error(ctx.h)
        assert np.array_equal(st, state)


def _decaying_population():
    from cascade_amd.workloads import leo_si_population

    Re, GM = 6378136.3, 3.986004407799724e14
    st, pars = leo_si_population(400, seed=7)
    rng = np.random.default_rng(3)
    low = rng.choice(400, size=12, replace=False)
    for k, i in enumerate(low):
        r0 = Re + 158e3 + 2.5e3 * k
        ang = rng.uniform(0, 2 * np.pi)
        v = np.sqrt(GM / r0)
        st[i, :6] = [r0 * np.cos(ang), r0 * np.sin(ang), 0.0, -v * np.sin(ang), v * np.cos(ang), 0.0]
        pars[i, 0] = 5e-2
    return st, pars, Re


def _run_loop(host_sync, nsteps=60):
    from cascade_amd import dynamics, outcome, sim

    st, pars, Re = _decaying_population()
    s = sim(st, 185.5663, dyn=dynamics.simple_earth(C22S22=False), pars=pars, n_par_ct=8, reentry_radius=Re + 150e3)
    s.host_sync = host_sync
    log = []
    for _ in range(nsteps):
        oc = s.step()
        info = s.interrupt_info
        log.append((oc, tuple(info) if isinstance(info, list) else info, s.nparts))
        if oc in (outcome.reentry, outcome.exit):
            s.remove_particles([info])
        elif oc == outcome.collision:
            s.remove_particles(list(info))
        else:
            assert oc == outcome.success
    if not host_sync:
        s.download_state()
    return log, s.time, np.array(s.state), np.array(s.pars)


def test_resident_removal_loop_matches_host_loop():
    from cascade_amd import outcome

    log_h, t_h, st_h, ps_h = _run_loop(True)
    log_d, t_d, st_d, ps_d = _run_loop(False)
    assert sum(oc == outcome.reentry for oc, _, _ in log_h) >= 5  # the loop did remove particles
    assert log_h == log_d
    assert t_h == t_d
    assert st_h.shape == st_d.shape and np.array_equal(st_h, st_d)
    assert np.array_equal(ps_h, ps_d)


def test_new_state_after_resident_steps_is_uploaded():
    """set_new_state_pars while the device copy is the authoritative one: the next step() takes the
    new host vectors (src/sim_basic.cpp:340-357), also when the particle count is unchanged."""
    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(500, 1)
    st2 = random_leo_pop_kep(500, 2)
    a = sim(st, 0.23)
    a.host_sync = False
    a.step()
    a.set_new_state_pars(st2)
    a.step()
    a.download_state()
    b = sim(st, 0.23)
    b.step()
    b.set_new_state_pars(st2)
    b.step()
    assert np.array_equal(np.array(a.state), np.array(b.state))
    # switching host_sync back on refreshes the host coordinates by itself
    c = sim(st, 0.23)
    c.host_sync = False
    c.step()
    c.set_new_state_pars(st2)
    c.step()
    c.host_sync = True
    assert np.array_equal(np.array(c.state), np.array(b.state))


def test_copy_of_a_resident_simulation_takes_the_device_state():
    """A copy made while the state is resident on the device (host_sync off, host vector stale)
    must hold the current state, not the one of the last download."""
    import copy

    from cascade_amd import sim
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(3000, seed=3)
    a = sim(st.copy(), 0.23)
    a.host_sync = False
    for _ in range(3):
        a.step()
    b = copy.copy(a)
    a.download_state()
    assert np.array_equal(np.array(a.state), np.array(b.state)) and a.time == b.time
    assert a.step() == b.step()
    a.download_state()
    b.download_state()
    assert np.array_equal(np.array(a.state), np.array(b.state))
