"""Per-superstep wall time of the headline workload (1M Kepler objects, ct = 0.23) with the host
vectors authoritative (host_sync = True, the drop-in default: 56 B up and down per object and
step over PCIe) against the resident mode bench.py measures. Quoted in DESIGN.md section 2.

  python profiles/host_sync_cost.py [n] [steps]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascade_amd import sim  # noqa: E402
from cascade_amd.workloads import random_leo_pop_kep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st = random_leo_pop_kep(n, 42)
out = {"n": n, "steps": steps}
for mode in (True, False):
    s = sim(st, 0.23)
    s.host_sync = mode
    for _ in range(3):
        s.step()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    if not mode:
        s.download_state()
    out["host_sync" if mode else "resident"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
# propagate_until() with the host vectors authoritative: one upload and one download per call
s = sim(st, 0.23)
s.propagate_until(3 * 0.23)
t0 = time.perf_counter()
s.propagate_until((3 + steps) * 0.23)
out["host_sync_propagate_until"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
print(json.dumps(out))
