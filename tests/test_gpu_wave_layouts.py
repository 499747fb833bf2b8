"""Every layout of k_propagate_wave (groups per object x lanes per operation, with and without
terminal events: 14 kernel instances of which the library's own choice exercises four) against the
one-lane-per-object interpreter on the same dynamics: same outcomes, interrupt info, substep
counts, broad-phase pair counts and conjunctions; states within 1e-9 of the vector's scale."""
import numpy as np
import pytest

from _tol import states_close

pytestmark = pytest.mark.gpu
LAYOUTS = [(1, 8), (2, 8), (4, 8), (8, 8), (2, 4), (4, 4), (8, 4)]


def _run(monkeypatch, wave, ng, lpg, events, which):
    from cascade_amd import dynamics as D
    from cascade_amd import sim
    from cascade_amd.workloads import haumea_dynamics, haumea_ring, leo_si_population

    monkeypatch.setenv("CASCADE_WAVE", "1" if wave else "0")
    if wave:
        monkeypatch.setenv("CASCADE_WAVE_NG", str(ng))
        monkeypatch.setenv("CASCADE_WAVE_LPG", str(lpg))
    if which == "haumea":
        dyn, k = haumea_dynamics()
        kw = dict(n_par_ct=2)
        if events:
            kw.update(reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
        s = sim(haumea_ring(1500), 5000.0, dyn=dyn, **kw)
    else:
        dyn = D.simple_earth()
        n = 1203  # (not a multiple of the objects of a workgroup: the last wave has lanes without an object)
        st, bstar = leo_si_population(n, seed=7)
        st[:, 6] *= 50.0
        s = sim(st, 185.5663, dyn=dyn, pars=bstar.copy(), n_par_ct=5, conj_thresh=5000.0,
                reentry_radius=(6378136.3 + 150e3) if events else 0.0)
        s.time = 2.5e8
    from cascade_amd import _capi

    ctx = _capi.Context.from_handle(s._hip_context)
    ctx.profile(True)
    ctx.profile_reset()
    out = []
    for _ in range(3):
        oc = s.step()
        ii = s.interrupt_info
        out.append((oc.name, ii, s.time, s.state.copy(), s.last_bp_pairs, s.last_total_substeps))
        if oc.name in ("reentry", "exit"):
            s.remove_particles([ii])
        elif oc.name == "collision":
            s.remove_particles(list(ii))
    conj = s.conjunctions.copy()
    # (a layout that does not fit falls back on the interpreter silently: the comparison would be with itself)
    kernels = set(ctx.profile_get())
    assert ("k_propagate_wave" in kernels) == bool(wave), kernels
    return out, conj


@pytest.fixture(scope="module")
def interp():
    return {}


@pytest.mark.parametrize("which", ["default", "haumea"])
@pytest.mark.parametrize("events", [True, False])
@pytest.mark.parametrize("ng,lpg", LAYOUTS)
def test_layout_matches_the_interpreter(ng, lpg, events, which, interp, monkeypatch):
    key = (which, events)
    if key not in interp:
        interp[key] = _run(monkeypatch, False, 0, 0, events, which)
    want, wconj = interp[key]
    got, gconj = _run(monkeypatch, True, ng, lpg, events, which)
    for (gn, gi, gt, gs, gp, gsub), (wn, wi, wt, ws, wp, wsub) in zip(got, want):
        assert gn == wn
        if isinstance(wi, (tuple, list)) and len(wi) == 2 and isinstance(wi[1], float):
            assert gi[0] == wi[0] and abs(gi[1] - wi[1]) <= 1e-9 * max(1.0, abs(wi[1]))
        else:
            assert gi == wi
        assert abs(gt - wt) <= 1e-9 * max(1.0, abs(wt))
        assert gs.shape == ws.shape and states_close(gs, ws, 1e-9)
        assert gp == wp and gsub == wsub
    assert len(gconj) == len(wconj)
    if len(wconj):
        assert np.array_equal(gconj["i"], wconj["i"]) and np.array_equal(gconj["j"], wconj["j"])


@pytest.mark.parametrize("seed", range(8))
def test_random_dynamics_match_the_interpreter(seed, monkeypatch):
    """The random right-hand sides of tests/test_wave_plan_fuzz.py (every operation kind, constants
    and runtime parameters on either side, sums larger than an operation holds, time dependence) on
    the GPU: the fused kernel against the one-lane interpreter of the plain tape."""
    from cascade_amd import _capi, sim
    from cascade_amd.workloads import random_leo_pop_kep
    from test_wave_plan_fuzz import _random_dynamics

    dyn = _random_dynamics(seed)
    npars = dyn.tape["npars"]
    n = 777
    st = random_leo_pop_kep(n, seed=seed)
    pars = np.random.default_rng(seed).uniform(0.5, 1.5, (n, max(1, npars)))[:, :npars].copy()
    res = []
    for wave in (0, 1):
        monkeypatch.setenv("CASCADE_WAVE", str(wave))
        kw = dict(dyn=dyn, n_par_ct=3, conj_thresh=1e-2, min_coll_radius=float("inf"))
        if npars:
            kw["pars"] = pars
        s = sim(st.copy(), 0.23, **kw)
        s.time = 0.37
        ctx = _capi.Context.from_handle(s._hip_context)
        ctx.profile(True)
        ctx.profile_reset()
        for _ in range(3):
            assert s.step().name == "success"
        assert ("k_propagate_wave" in set(ctx.profile_get())) == bool(wave)
        res.append((s.state.copy(), s.time, s.last_bp_pairs, s.last_total_substeps, s.conjunctions.copy()))
    (s0, t0, p0, n0, c0), (s1, t1, p1, n1, c1) = res
    assert t0 == t1 and n0 == n1 and p0 == p1
    assert states_close(s1, s0, 1e-9)
    assert len(c0) == len(c1) and np.array_equal(c0["i"], c1["i"]) and np.array_equal(c0["j"], c1["j"])


@pytest.mark.parametrize("which", ["default", "j2drag", "haumea"])
def test_compiled_tapes_match_the_interpreted_rows(which, monkeypatch):
    """The dynamics known when the library is built run from straight-line code
    (cascade_tapes.hip; the tagged J2 + drag below 49 152 objects): same outcomes and counts as the
    interpreted schedule (CASCADE_WAVE_COMPILED=0), states within 1e-9 of the vector's scale."""
    from cascade_amd import _capi
    from cascade_amd import dynamics as D
    from cascade_amd import sim
    from cascade_amd.workloads import haumea_dynamics, haumea_ring, leo_si_population

    res = []
    for compiled in (0, 1):
        monkeypatch.setenv("CASCADE_WAVE_COMPILED", str(compiled))
        if which == "haumea":
            dyn, k = haumea_dynamics()
            s = sim(haumea_ring(3001), 5000.0, dyn=dyn, n_par_ct=2, reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
        else:
            dyn = D.simple_earth(C22S22=(which == "default"))
            st, bstar = leo_si_population(2501, seed=3)
            st[:, 6] *= 50.0
            s = sim(st, 185.5663, dyn=dyn, pars=bstar.copy(), n_par_ct=5, conj_thresh=5000.0, reentry_radius=6378136.3 + 150e3)
            s.time = 2.5e8
        ctx = _capi.Context.from_handle(s._hip_context)
        out = []
        for k in range(4):
            if k == 1:
                ctx.profile(True)
                ctx.profile_reset()
            oc = s.step()
            ii = s.interrupt_info
            out.append((oc.name, ii, s.time, s.state.copy(), s.last_bp_pairs, s.last_total_substeps))
            if oc.name in ("reentry", "exit"):
                s.remove_particles([ii])
            elif oc.name == "collision":
                s.remove_particles(list(ii))
        kernels = set(ctx.profile_get())
        # (the tagged J2 + drag runs on its unrolled kernel when the compiled tapes are off)
        assert ("k_propagate_wave" in kernels) == (compiled == 1 or which != "j2drag"), kernels
        res.append(out)
    for (gn, gi, gt, gs, gp, gsub), (wn, wi, wt, ws, wp, wsub) in zip(res[1], res[0]):
        assert gn == wn
        if isinstance(wi, (tuple, list)) and len(wi) == 2 and isinstance(wi[1], float):
            assert gi[0] == wi[0] and abs(gi[1] - wi[1]) <= 1e-9 * max(1.0, abs(wi[1]))
        else:
            assert gi == wi
        assert abs(gt - wt) <= 1e-9 * max(1.0, abs(wt))
        assert gs.shape == ws.shape and states_close(gs, ws, 1e-9)
        assert gp == wp and gsub == wsub


@pytest.mark.timeout(400)
def test_run_time_compiled_tape_matches_the_interpreted_rows(monkeypatch, capfd):
    """CASCADE_WAVE_JIT=1: dynamics without an ahead-of-time tape are compiled by hipRTC at first
    use (cascade_tapes.hip, half a minute per kernel variant) and run from straight-line code: same
    counts and pairs as the interpreted schedule, states within 1e-9 of the vector's scale."""
    from cascade_amd import _capi, sim
    from cascade_amd.workloads import random_leo_pop_kep
    from test_wave_plan_fuzz import _random_dynamics

    seed = 12  # (not one of test_random_dynamics_match_the_interpreter: a compiled kernel lives as long as the process)
    dyn = _random_dynamics(seed)
    npars = dyn.tape["npars"]
    n = 1501
    st = random_leo_pop_kep(n, seed=seed)
    pars = np.random.default_rng(seed).uniform(0.5, 1.5, (n, max(1, npars)))[:, :npars].copy()
    monkeypatch.setenv("CASCADE_WAVE_DEBUG", "1")
    res = []
    for jit in (0, 1):
        monkeypatch.setenv("CASCADE_WAVE_JIT", str(jit))
        kw = dict(dyn=dyn, n_par_ct=3, conj_thresh=1e-2, min_coll_radius=float("inf"))
        if npars:
            kw["pars"] = pars
        s = sim(st.copy(), 0.23, **kw)
        s.time = 0.37
        ctx = _capi.Context.from_handle(s._hip_context)
        ctx.profile(True)
        ctx.profile_reset()
        capfd.readouterr()
        for _ in range(3):
            assert s.step().name == "success"
        err = capfd.readouterr().err
        assert "k_propagate_wave" in set(ctx.profile_get())
        assert ("[wave] run-time compilation: ok" in err) == bool(jit), err
        assert "interpreted schedule runs instead" not in err, err
        res.append((s.state.copy(), s.time, s.last_bp_pairs, s.last_total_substeps, s.conjunctions.copy()))
    (s0, t0, p0, n0, c0), (s1, t1, p1, n1, c1) = res
    assert t0 == t1 and n0 == n1 and p0 == p1
    assert states_close(s1, s0, 1e-9)
    assert len(c0) == len(c1) and np.array_equal(c0["i"], c1["i"]) and np.array_equal(c0["j"], c1["j"])


@pytest.mark.timeout(400)
def test_failed_run_time_compilation_falls_back_to_the_interpreted_rows(monkeypatch, capfd, tmp_path):
    """When the tape cannot be compiled (here: the kernel headers are not where the library is told
    to look) the library says so once and the interpreted schedule runs: same results as with
    run-time compilation switched off."""
    from cascade_amd import _capi, sim
    from cascade_amd.workloads import random_leo_pop_kep
    from test_wave_plan_fuzz import _random_dynamics

    seed = 13  # (a schedule no other test compiles: a failure is remembered per schedule too)
    dyn = _random_dynamics(seed)
    npars = dyn.tape["npars"]
    n = 901
    st = random_leo_pop_kep(n, seed=seed)
    pars = np.random.default_rng(seed).uniform(0.5, 1.5, (n, max(1, npars)))[:, :npars].copy()
    res = []
    for broken in (0, 1):
        if broken:
            monkeypatch.setenv("CASCADE_WAVE_JIT", "1")
            monkeypatch.setenv("CASCADE_JIT_INCLUDE_DIR", str(tmp_path))
        else:
            monkeypatch.setenv("CASCADE_WAVE_JIT", "0")
        kw = dict(dyn=dyn, n_par_ct=3, conj_thresh=1e-2, min_coll_radius=float("inf"))
        if npars:
            kw["pars"] = pars
        s = sim(st.copy(), 0.23, **kw)
        s.time = 0.37
        ctx = _capi.Context.from_handle(s._hip_context)
        ctx.profile(True)
        ctx.profile_reset()
        capfd.readouterr()
        for _ in range(3):
            assert s.step().name == "success"
        err = capfd.readouterr().err
        assert "k_propagate_wave" in set(ctx.profile_get())
        assert (err.count("interpreted schedule runs instead") == 1) == bool(broken), err
        res.append((s.state.copy(), s.time, s.last_bp_pairs, s.last_total_substeps, s.conjunctions.copy()))
    (s0, t0, p0, n0, c0), (s1, t1, p1, n1, c1) = res
    assert t0 == t1 and n0 == n1 and p0 == p1
    assert np.array_equal(s0, s1)
    assert len(c0) == len(c1) and np.array_equal(c0["i"], c1["i"]) and np.array_equal(c0["j"], c1["j"])
