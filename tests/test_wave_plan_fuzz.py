"""Random right-hand sides through the fusion and the scheduler of k_propagate_wave
(csc_wave_plan_eval, no GPU) against the oracle integrating the PLAIN tape: every operation kind,
constants and runtime parameters on either side of every operator, shared subexpressions, sums
with more products and linear terms than an operation holds, time dependence."""
import ctypes as C

import numpy as np
import pytest

ORDER = 20


def _random_dynamics(seed):
    from cascade_amd import dynamics as D

    rng = np.random.default_rng(seed)
    x, y, z, vx, vy, vz = D.make_vars("x", "y", "z", "vx", "vy", "vz")
    r2 = x * x + y * y + z * z
    pool = [x, y, z, vx, vy, vz, r2, D.par[0], D.par[1], D.time * 0.01]

    def pick():
        return pool[int(rng.integers(len(pool)))]

    def const():
        return float(rng.uniform(0.2, 1.7) * rng.choice([-1.0, 1.0]))

    for _ in range(int(rng.integers(10, 28))):
        k = int(rng.integers(14))
        a, b = pick(), pick()
        if k == 0:
            e = a + b
        elif k == 1:
            e = a - b
        elif k == 2:
            e = a * b
        elif k == 3:
            e = a * const()
        elif k == 4:
            e = const() - a
        elif k == 5:
            e = a + const()
        elif k == 6:
            e = a / (2.5 + b * b)  # (a denominator that stays away from zero)
        elif k == 7:
            e = const() / (3.0 + a * a)
        elif k == 8:
            e = D.pow(1.5 + a * a, float(rng.choice([-1.5, -0.5, 0.5, 1.0 / 3.0, 2.5])))
        elif k == 9:
            e = D.sqrt(2.0 + a * a + b * b)
        elif k == 10:
            e = D.exp(0.1 * a)
        elif k == 11:
            e = D.sin(a) if rng.integers(2) else D.cos(a)
        elif k == 12:
            e = a / D.par[1] if rng.integers(2) else D.par[0] / (2.0 + a * a)
        else:
            e = a * a * b + a * b - b * b * const() + pick() * pick() + pick() * pick() + pick() * pick()  # (> 4 products)
        pool.append(e)
    dyn = D.kepler(1.0)
    for k in range(3):
        lhs, rhs = dyn[3 + k]
        extra = sum((const() * 1e-3) * pool[int(rng.integers(10, len(pool)))] for _ in range(3))
        dyn[3 + k] = (lhs, rhs + extra)
    return dyn


@pytest.mark.parametrize("seed", range(80))
def test_random_tapes(seed):
    from cascade_amd import _capi
    from cascade_amd.workloads import random_leo_pop_kep
    from oracle import orc_py

    orc_py.build()
    lib = _capi.lib()
    dyn = _random_dynamics(seed)
    tape = dyn.tape
    assert tape["npars"] == 2 or tape["npars"] < 2
    n = 4
    st = random_leo_pop_kep(64, seed=seed)[:n].copy()
    rng = np.random.default_rng(1000 + seed)
    pars = rng.uniform(0.5, 1.5, (n, max(1, tape["npars"])))
    t0 = 0.37
    o = orc_py.Sim(st.copy(), 0.05, n_par_ct=1, tape=tape, pars=pars[:, : tape["npars"]].copy() if tape["npars"] else None)
    o.time = t0
    o.step()
    nsub, tcs, _ = o.substeps()
    first = np.concatenate([[0], np.cumsum(nsub.astype(np.int64))[:-1]]).astype(np.int64)
    ops = np.ascontiguousarray(tape["ops"])
    rhs = np.ascontiguousarray(np.asarray(tape["rhs"], dtype=np.int32))
    for ng in (0, 1, 2, 8):
        for i in range(n):
            s7 = np.ascontiguousarray(np.concatenate([st[i, :6], [np.linalg.norm(st[i, :3])]]))
            pp = np.ascontiguousarray(pars[i])
            out = np.empty((7, ORDER + 1))
            stats = np.zeros(8, dtype=np.uint32)
            rc = lib.csc_wave_plan_eval(ops.ctypes.data, len(ops), int(tape["nslots"]), rhs.ctypes.data, ng, ORDER, s7.ctypes.data,
                                        pp.ctypes.data, C.c_double(t0), out.ctypes.data, stats.ctypes.data)
            assert rc == 0, (seed, ng)
            want = tcs[first[i]]
            for rows, tol in (((0, 1, 2), 1e-10), ((3, 4, 5), 1e-10), ((6,), 1e-8)):
                sc = np.abs(want[list(rows)]).max(axis=0)
                err = np.abs(out[list(rows)] - want[list(rows)])
                assert np.all(err <= tol * sc + 1e-300), (seed, ng, i, rows, (err / (sc + 1e-300)).max())
                # (orders 0-2 to rounding; r'' = ... / r cancels on near-circular orbits)
                assert np.all(err[:, :3] <= (1e-13 if rows != (6,) else 1e-9) * sc[:3] + 1e-300), (seed, ng, i, rows)
        assert stats[0] <= len(ops) + 2
