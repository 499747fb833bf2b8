"""Experiment (round 3): how selective are cheap first-stage tests in front of the chord test of
k_chord.hpp? 1M-object LEO population, one chunk; a sample of the broad-phase pairs."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascade_amd import _capi  # noqa: E402
from cascade_amd.workloads import random_leo_pop_kep  # noqa: E402
from oracle import orc_py as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pop = random_leo_pop_kep(n, seed=42)
ctx = _capi.Context(0)
L, h, chk = ctx.L, ctx.h, ctx._chk
cfg = _capi.PropCfg()
orc.build()
orc.lib().orc_rhofac.restype = C.c_double
orc.lib().orc_rhofac.argtypes = [C.c_uint32]
for k, v in dict(dyn_kind=0, order=20, mu=1.0, rhofac=orc.lib().orc_rhofac(20), delta_t=0.23, ct=0.23, nchunks=1, reentry_c0=0.0).items():
    setattr(cfg, k, v)
d_state = ctx.array(pop)
d = dict(nsub=ctx.empty(n, np.uint32), status=ctx.empty(n, np.int32), ev_idx=ctx.empty(n, np.uint32), ev_time=ctx.empty(n, np.float64), nf=ctx.empty(n, np.float64),
         final=ctx.empty((n, 6), np.float64), lbs=ctx.empty((1, n, 4), np.float32), ubs=ctx.empty((1, n, 4), np.float32), recs=ctx.empty((1, n, 16), np.float32))
chk(L.csc_propagate(h, C.byref(cfg), n, d_state.ptr, d["nsub"].ptr, d["status"].ptr, d["ev_idx"].ptr, d["ev_time"].ptr, d["nf"].ptr, d["final"].ptr, d["lbs"].ptr, d["ubs"].ptr, d["recs"].ptr))
d_glb, d_gub = ctx.empty((1, 4), np.float32), ctx.empty((1, 4), np.float32)
chk(L.csc_global_aabb(h, n, 1, d["lbs"].ptr, d["ubs"].ptr, d_glb.ptr, d_gub.ptr))
d_codes = ctx.empty((1, n), np.uint64)
chk(L.csc_morton_encode(h, n, 1, d["lbs"].ptr, d["ubs"].ptr, d_glb.ptr, d_gub.ptr, d_codes.ptr))
d_vidx, d_sc = ctx.empty((1, n), np.uint32), ctx.empty((1, n), np.uint64)
d_slb, d_sub = ctx.empty((1, n, 4), np.float32), ctx.empty((1, n, 4), np.float32)
chk(L.csc_sort_gather(h, n, 1, d_codes.ptr, d["lbs"].ptr, d["ubs"].ptr, d_vidx.ptr, d_sc.ptr, d_slb.ptr, d_sub.ptr, None, None))
chk(L.csc_bvh_build(h, n, 1, d_sc.ptr, d_slb.ptr, d_sub.ptr))
d_act = ctx.array(np.ones(n, dtype=np.uint8))
npairs = C.c_uint64()
chk(L.csc_broad_phase(h, n, 1, d_slb.ptr, d_sub.ptr, d_vidx.ptr, d_act.ptr, C.byref(npairs)))
got = np.empty((npairs.value, 2), dtype=np.uint32)
cnt = C.c_uint64()
chk(L.csc_broad_phase_get(h, 0, got.ctypes.data_as(C.c_void_p), len(got), C.byref(cnt)))
recs = d["recs"].download()[0]
rng = np.random.default_rng(0)
sel = got[rng.choice(len(got), min(len(got), 2_000_000), replace=False)]
a, b = recs[sel[:, 0]], recs[sel[:, 1]]
dd = (a[:, :15] - b[:, :15]).reshape(-1, 5, 3)  # sample differences
thr = (a[:, 15] + b[:, 15]) * 1.000002
# full chord test
keep = ~np.isfinite(thr)
for s in range(4):
    e = dd[:, s + 1] - dd[:, s]
    ee = np.sum(e * e, axis=1)
    de = np.sum(dd[:, s] * e, axis=1)
    u = np.clip(np.where(ee > 0, -de / np.where(ee > 0, ee, 1), 0), 0, 1)
    m = dd[:, s] + u[:, None] * e
    keep |= ~(np.sum(m * m, axis=1) > thr * thr * 1.000002)
print("pairs", len(got), "sample", len(sel), "full test keeps", keep.mean())
# T1: some axis where all five differences have one sign and |d| > thr throughout
mn, mx = dd.min(axis=1), dd.max(axis=1)
sep = ((mn > thr[:, None]) | (mx < -thr[:, None])).any(axis=1)
print("T1 per-axis separation excludes", sep.mean(), "; passes", (~sep).mean(), "; full-keep among passes", keep[~sep].mean(), "; wrongly excluded", (sep & keep).sum())
# T2: sphere bound at the middle sample: |d_2| - max_s |d_s - d_2| > thr
r2 = np.linalg.norm(dd[:, 2], axis=1)
dev = np.max(np.linalg.norm(dd - dd[:, 2:3], axis=2), axis=1)
t2 = (r2 - dev) > thr
print("T2 mid-sample sphere excludes", t2.mean(), "wrongly", (t2 & keep).sum(), "; T1|T2 excludes", (sep | t2).mean())
# T3: per axis, min |d| over the 5 samples with same sign; combine the axes' lower bounds in quadrature
lo = np.where((mn > 0), mn, np.where(mx < 0, -mx, 0.0))
t3 = np.sum(lo * lo, axis=1) > thr * thr * 1.00001
print("T3 quadrature of per-axis bounds excludes", t3.mean(), "wrongly", (t3 & keep).sum())
