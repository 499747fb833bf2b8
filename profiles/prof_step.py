"""Three supersteps of the bench workload, for PMC passes (rocprofv3 --pmc ... -- python3 profiles/prof_step.py [n])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascade_amd import sim
from cascade_amd.workloads import random_leo_pop_kep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
s = sim(random_leo_pop_kep(n, 42), 0.23, n_par_ct=1)
s.host_sync = False
s.upload_state()
for _ in range(3):
    s.step()
print("pairs", s.last_bp_pairs)
