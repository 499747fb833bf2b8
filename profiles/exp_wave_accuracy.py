"""Experiment (round 3): k_propagate_wave against the CPU oracle on the Haumea ring and on the
default simple_earth: largest deviation of position / velocity relative to |r| / |v|."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cascade_amd import sim  # noqa: E402
from cascade_amd.workloads import haumea_dynamics, haumea_ring  # noqa: E402
from oracle import orc_py as orc  # noqa: E402

orc.build()
dyn, k = haumea_dynamics()
st = haumea_ring(4000)
kw = dict(n_par_ct=2, reentry_radius=[k["ra"], k["rb"], k["rc"]], exit_radius=k["a_s"] * 10)
for wave in ("0", "1"):
    os.environ["CASCADE_WAVE"] = wave
    g = sim(st.copy(), 5000.0, dyn=dyn, **kw)
    o = orc.Sim(st.copy(), 5000.0, tape=dyn.tape, **kw)
    for step in range(12):
        og, oo = g.step(), o.step()
        a, b = g.state, o.state
        rp = np.linalg.norm(a[:, :3] - b[:, :3], axis=1) / np.linalg.norm(b[:, :3], axis=1)
        rv = np.linalg.norm(a[:, 3:6] - b[:, 3:6], axis=1) / np.linalg.norm(b[:, 3:6], axis=1)
        comp = np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))
        print("wave", wave, "step", step, og.name, oo, "pos", rp.max(), "vel", rv.max(), "per-component", comp, "time", g.time - o.time)
        if og.name == "collision":
            i, j = g.interrupt_info
            for s_, poke in ((g.state, None), (o.state, o.poke_state)):
                ri, rj, vi, vj = s_[i, :3], s_[j, :3], s_[i, 3:6].copy(), s_[j, 3:6].copy()
                u = (rj - ri) / np.linalg.norm(rj - ri)
                a_, b_ = vi @ u, vj @ u
                s_[i, 3:6] = vi - a_ * u + (0.3 * (b_ - a_) + a_ + b_) / 2 * u
                s_[j, 3:6] = vj - b_ * u + (0.3 * (a_ - b_) + a_ + b_) / 2 * u
                if poke:
                    poke(s_)
    from cascade_amd import _capi
    print(_capi.Context.from_handle(g._hip_context).profile_get() if False else "")
