import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from cascade_amd import sim, _capi
from cascade_amd.workloads import random_leo_pop_kep
n=int(sys.argv[1]) if len(sys.argv)>1 else 1000000
st = random_leo_pop_kep(n, 42)
s = sim(st, 0.23, n_par_ct=1)
s.step()
ctx=s._ctx
# read counters: d_flags + 64 bytes -> need pointer; use csc debug: not exported. Use bp_nodes_visited path instead
print("pairs", s.last_bp_pairs, "ref nodes visited", s.bp_nodes_visited())
s.step()
buf=np.zeros(8,np.uint64)
p=C.c_void_p()
ctx.L.csc_debug_counters.restype=C.c_int
ctx.L.csc_debug_counters(C.c_void_p(ctx.h), buf.ctypes.data_as(C.c_void_p))
print("counters", buf)
print("per group: nodes", buf[2]/(n/64), "leaves", buf[3]/(n/64), "rounds", buf[4]/(n/64))
# box sizes
