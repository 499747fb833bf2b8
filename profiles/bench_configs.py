"""Secondary workloads of BASELINE.json (configs 3-5) on one GPU: per-superstep wall time,
object-days/s and the per-kernel HIP-event times. Not the bench.py line (that is configs[1]);
results are quoted in DESIGN.md.

  python profiles/bench_configs.py leo_j2drag|leo_default|leo_full|haumea|shells [n] [n_par_ct] [steps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from cascade_amd import _capi, dynamics, sim  # noqa: E402
from cascade_amd.workloads import haumea_dynamics, haumea_ring, leo_si_population, random_leo_pop_kep  # noqa: E402

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 19647
C = int(sys.argv[3]) if len(sys.argv) > 3 else 30
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
unit = 1.0
if which.startswith("leo_"):
    flags = {"leo_j2drag": dict(C22S22=False), "leo_default": dict(), "leo_full": dict(J3=True, J4=True, sun=True, moon=True, SRP=True)}[which]
    dyn = dynamics.simple_earth(**flags)
    if os.environ.get("UNTAG"):
        dyn[0] = dyn[0]  # (the tagged built-ins then run from their tape like any other dynamics)
    st, bstar = leo_si_population(n, seed=42)
    npars = dyn.tape["npars"]
    pars = np.concatenate([bstar, np.full((n, 1), 0.02)], axis=1)[:, :npars].copy()
    # benchmark/2022_leo_pop.cpp:126,165,223,297: ct = 185.5663 s, reentry at Re + 150 km
    s = sim(st, 185.5663, dyn=dyn, pars=pars, n_par_ct=C, reentry_radius=0.0 if os.environ.get("NOEV") else 6378136.3 + 150e3, conj_thresh=float(os.environ.get("CONJ", "0")))
elif which == "haumea":
    dyn, k = haumea_dynamics()
    s = sim(haumea_ring(n), 5000.0, dyn=dyn, n_par_ct=C, reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
elif which == "shells":
    # BASELINE configs[3]: Walker-like shells, conjunction tracking, collisions off.
    rng = np.random.default_rng(42)
    from cascade_amd.workloads import kep_to_cart_bench

    a = rng.choice([1.08, 1.09, 1.17, 1.19], n) + rng.uniform(-0.002, 0.002, n)
    inc = np.deg2rad(rng.choice([53.0, 70.0, 97.6], n) + rng.uniform(-0.5, 0.5, n))
    x, v = kep_to_cart_bench(a, rng.uniform(0, 1e-3, n), inc, rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n))
    st = np.concatenate([x, v, np.full((n, 1), 1e-7)], axis=1)
    s = sim(st, float(os.environ.get("CT", "0.02")), n_par_ct=C, conj_thresh=7.8e-4, min_coll_radius=float("inf"))
    unit = 806.81
else:
    raise SystemExit("unknown workload")

ctx = _capi.Context.from_handle(s._hip_context)
outcomes = {}


def run(k):
    for _ in range(k):
        oc = s.step()
        outcomes[oc.name] = outcomes.get(oc.name, 0) + 1
        if oc.name in ("reentry", "collision", "exit"):
            ii = s.interrupt_info
            s.remove_particles([ii] if isinstance(ii, int) else list(ii))


run(1)
ctx.profile(True)
ctx.profile_reset()
t_sim0, t0 = s.time, time.perf_counter()
run(steps)
ctx.sync()
wall = time.perf_counter() - t0
prof = ctx.profile_get()
od = s.nparts * (s.time - t_sim0) * unit / 86400.0
print(json.dumps({"workload": which, "n": s.nparts, "n_par_ct": C, "steps": steps, "ms_per_step": round(wall / steps * 1e3, 3), "object_days_per_s": round(od / wall, 1), "outcomes": outcomes, "bp_pairs": s.last_bp_pairs, "np_candidates": s.last_np_candidates, "substeps": s.last_total_substeps, "conjunctions": len(s.conjunctions), "kernel_ms_per_step": {k: round(v[1] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}}))
