"""Two supersteps of J2 dynamics with / without the reentry event, two-pass mode (diagnostic for
rocprofv3 --pmc: instruction mix of the k_propagate variants)."""
import os, sys
os.environ["CASCADE_SINGLE_PASS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascade_amd import dynamics, sim
from cascade_amd.workloads import leo_si_population
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
st, bstar = leo_si_population(n, seed=42)
for rr in (True, False):
    s = sim(st.copy(), 185.5663, dyn=dynamics.simple_earth(C22S22=False, drag=False), n_par_ct=8, reentry_radius=(6378136.3 + 150e3) if rr else None)
    print(rr, s.step(), s.step(), s.last_total_substeps)
