"""The fusion and the schedule behind k_propagate_wave (cascade_amd/csrc/cascade_wave.hip), without
a GPU: csc_wave_plan_eval builds the plan for a dynamics tape and executes it on the host the way
the kernel does; its Taylor coefficients must be those the oracle computes from the PLAIN tape
(oracle/orc.hpp::tape_compute_tcs: the first substep of a superstep) -- to rounding, since the
fused operations associate sums differently."""
import ctypes as C

import numpy as np
import pytest

ORDER = 20


def _dynamics(which):
    from cascade_amd import dynamics as D

    if which == "default":
        return D.simple_earth(), 2.5e8
    if which == "full":
        return D.simple_earth(J2=True, J3=True, J4=True, C22S22=True, sun=True, moon=True, SRP=True, drag=True), 2.5e8
    if which == "j2drag":
        d = D.simple_earth(J2=True, C22S22=False, drag=True)
        return d, 0.0
    from cascade_amd.workloads import haumea_dynamics

    return haumea_dynamics()[0], 0.0


def _plan_eval(lib, tape, state7, pars, t, ng):
    ops = np.ascontiguousarray(tape["ops"])
    rhs = np.ascontiguousarray(np.asarray(tape["rhs"], dtype=np.int32))
    out = np.empty((7, ORDER + 1))
    stats = np.zeros(8, dtype=np.uint32)
    st = np.ascontiguousarray(state7, dtype=np.float64)
    # (dynamics without runtime parameters come with a null array, as in cascade::sim: no place of
    # the schedule may then ask for one -- an unused place of a row did, and faulted the GPU)
    pp = np.ascontiguousarray(pars, dtype=np.float64) if len(pars) else None
    rc = lib.csc_wave_plan_eval(
        ops.ctypes.data, len(ops), int(tape["nslots"]), rhs.ctypes.data, ng, ORDER, st.ctypes.data, pp.ctypes.data if pp is not None else None, C.c_double(t),
        out.ctypes.data, stats.ctypes.data,
    )
    assert rc in (0, 2)
    return (out, stats) if rc == 0 else (None, None)


@pytest.mark.parametrize("which", ["default", "full", "j2drag", "haumea"])
def test_fused_plan_gives_the_coefficients_of_the_plain_tape(which):
    from cascade_amd import _capi
    from cascade_amd.workloads import haumea_ring, leo_si_population
    from oracle import orc_py

    orc_py.build()
    lib = _capi.lib()
    dyn, t0 = _dynamics(which)
    tape = dyn.tape
    n = 6
    if which == "haumea":
        st = haumea_ring(64)[:n].copy()
        pars = np.zeros((n, 0))
        ct = 100.0
    else:
        st, bstar = leo_si_population(n, seed=11)
        pars = np.concatenate([bstar, np.full((n, 1), 0.02)], axis=1)[:, : tape["npars"]].copy()
        ct = 50.0
    o = orc_py.Sim(st.copy(), ct, n_par_ct=1, tape=tape, pars=pars if pars.shape[1] else None)
    o.time = t0
    o.step()
    nsub, tcs, _ = o.substeps()
    first = np.concatenate([[0], np.cumsum(nsub.astype(np.int64))[:-1]]).astype(np.int64)
    plain = None
    for ng in (0, 1, 2, 4, 8):
        for i in range(n):
            s7 = np.concatenate([st[i, :6], [np.linalg.norm(st[i, :3])]])
            got, stats = _plan_eval(lib, tape, s7, pars[i], t0, ng)
            if got is None:
                # (a forced layout whose series do not fit the LDS of a compute unit)
                assert ng != 0
                continue
            want = tcs[first[i]]
            # At the scale of each order across the variables of one kind (a coefficient that passes
            # through zero has no scale of its own). Orders 0 to 3 agree to 1e-15; the error grows
            # with the order (1e24-sized gravity coefficients cancelling) to ~1e-12 at order 20,
            # where a term weighs 1e-16 of the step. r' = (r . v) / r cancels on near-circular orbits.
            for rows, tol in (((0, 1, 2), 1e-11), ((3, 4, 5), 1e-11), ((6,), 1e-9)):
                sc = np.abs(want[list(rows)]).max(axis=0)
                assert np.all(np.abs(got[list(rows)] - want[list(rows)]) <= tol * sc + 1e-300), (which, ng, i, rows)
                assert np.all(np.abs(got[list(rows)][:, :4] - want[list(rows)][:, :4]) <= 1e-3 * tol * sc[:4] + 1e-300) or rows == (6,)
        if ng == 0:
            plain = stats.copy()
    # the fusion is what the kernel's speed rests on: say what it achieved
    assert plain[0] < len(tape["ops"]) and plain[1] < tape["nslots"]
    if which == "default":
        assert plain[0] <= 50 and plain[1] <= 60 and plain[2] <= 8


def test_plan_rejects_a_tape_that_assigns_twice():
    from cascade_amd import _capi
    from cascade_amd import dynamics as D

    lib = _capi.lib()
    tape = D.simple_earth().tape
    ops = np.ascontiguousarray(tape["ops"]).copy()
    ops[5]["dst"] = ops[4]["dst"]
    rhs = np.ascontiguousarray(np.asarray(tape["rhs"], dtype=np.int32))
    out = np.empty((7, ORDER + 1))
    st = np.ones(7)
    rc = lib.csc_wave_plan_eval(ops.ctypes.data, len(ops), int(tape["nslots"]), rhs.ctypes.data, 0, ORDER, st.ctypes.data, st.ctypes.data,
                                C.c_double(0.0), out.ctypes.data, None)
    assert rc == 2


def test_compiled_tapes_header_is_current():
    """cascade_amd/csrc/gen/compiled_tapes.hpp (generated, committed) holds the straight-line code of
    the schedules the fusion produces TODAY for the built-in dynamics: same hashes in its table, same
    text. (A stale entry would never run -- schedules are found by hash -- but silently fall back on
    the interpreted rows.)"""
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_compiled_tapes", os.path.join(root, "profiles", "gen_compiled_tapes.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from cascade_amd import _capi

    lib = _capi.lib()
    text = open(os.path.join(root, "cascade_amd", "csrc", "gen", "compiled_tapes.hpp")).read()
    table = {int(i): int(h, 16) for h, i in re.findall(r"\{0x([0-9a-f]{16})ull, (\d+)\}", text)}
    seen = 0
    for ident, _, dyn in gen.tapes():
        src, h = gen.emit(lib, dyn, ident)
        assert src is not None
        assert table.get(ident) == h, "profiles/gen_compiled_tapes.py has to be run again (tape %d)" % ident
        assert src in text, "profiles/gen_compiled_tapes.py has to be run again (text of tape %d)" % ident
        seen += 1
    assert seen == len(table) >= 3
