"""Propagation kernel timing for the built-in J2 + drag dynamics under variants (diagnostic)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cascade_amd import _capi, dynamics, sim
from cascade_amd.workloads import leo_si_population
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
st, bstar = leo_si_population(n, seed=42)
for kind, C, recs, rr in [("j2drag", 8, True, True), ("j2drag", 8, False, True), ("j2drag", 8, True, False), ("j2drag", 1, True, True), ("j2", 8, True, True), ("j2", 8, True, False)]:
    drag = kind == "j2drag"
    dyn = dynamics.simple_earth(C22S22=False, drag=drag)
    s = sim(st.copy(), 185.5663, dyn=dyn, pars=bstar if drag else None, n_par_ct=C, reentry_radius=(6378136.3 + 150e3) if rr else None)
    s.np_prefilter = recs
    ctx = _capi.Context.from_handle(s._hip_context)
    oc = []

    def step():
        o = s.step()
        oc.append(o.name)
        if o.name in ("reentry", "collision"):
            ii = s.interrupt_info
            s.remove_particles([ii] if isinstance(ii, int) else list(ii))

    step()
    ctx.profile(True); ctx.profile_reset()
    step()
    step()
    prof = ctx.profile_get()
    print(json.dumps(dict(kind=kind, C=C, recs=recs, reentry=rr, outcomes=oc, substeps=s.last_total_substeps, k_propagate=round(prof.get("k_propagate", (0, 0))[1] / 2, 3), k_rows=round(prof.get("k_propagate_rows", (0, 0))[1] / 2, 3), k_store=round(prof.get("k_propagate_store", (0, 0))[1] / 2, 3))), flush=True)
    del s
