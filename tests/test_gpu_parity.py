"""GPU parity tests: every stage of the HIP pipeline, called through the C ABI of
libcascade_hip.so (include/cascade_hip.h), against the CPU oracle on the same inputs.

Contract (BASELINE.json north_star):
  * bit-exact: global AABBs, Morton codes, sort permutation, BVH node arrays (canonical
    level order, reference node layout), broad-phase pair sets -- all given identical AABBs
    (the oracle's AABBs are injected);
  * tolerance: substep times / Taylor coefficients / final states within 1e-9 relative;
    collision and conjunction event SETS equal, event times and states within 1e-9 relative.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9  # the tolerance of the north star


@pytest.fixture(scope="module")
def ctx():
    from cascade_amd import _capi

    c = _capi.Context(0)
    yield c
    c.close()


def _pop(n, seed=42, same_code=False):
    from cascade_amd.workloads import random_leo_pop_kep

    st = random_leo_pop_kep(n, seed)
    if same_code:
        # test/same_morton_code.cpp:65-76: particles (almost) on top of others -> equal codes
        eps = np.finfo(float).eps
        extra = []
        for idx in (1, 123, 456):
            p = st[idx].copy()
            p[:3] += eps * 1000
            p[6] = 0.1
            extra.append(p)
        st = np.concatenate([st, np.array(extra)])
        st[:-3, 6] = 1e-6
    return st


def _oracle_step(orc, st, n_par_ct, **kw):
    s = orc.Sim(st, 0.23, n_par_ct=n_par_ct, **kw)
    oc = s.step()
    return s, oc


@pytest.mark.parametrize("n,n_par_ct,same_code", [(2, 1, False), (6, 3, False), (64, 1, False), (2003, 1, False), (2003, 3, False), (2003, 3, True), (20000, 2, False), (4, 2, False), (3, 1, False), (70, 3, False)])
def test_aabb_to_pairs_bit_exact(orc, ctx, n, n_par_ct, same_code):
    """K2b-K6 on injected (oracle) AABBs: global AABB, Morton codes, stable permutation,
    sorted arrays, BVH nodes and broad-phase pairs are bit-identical."""
    from _stage_parity import check_aabb_to_pairs

    st = _pop(n, same_code=same_code)
    s, oc = _oracle_step(orc, st, n_par_ct)
    check_aabb_to_pairs(orc, ctx, s, len(st))


@pytest.mark.parametrize("n_par_ct", [1, 2])
def test_pairs_with_multi_particle_leaves(orc, ctx, n_par_ct):
    """Groups of 2..20 particles with identical states (identical boxes -> equal Morton codes ->
    multi-particle leaves, test/same_morton_code.cpp) among ordinary ones: leaves within the
    tile limit of k_bp_tiles (leaf boxes instead of particle boxes), leaves beyond it (the
    traversal's one-by-one path), a third of the particles without activity flags; the pair set
    stays bit-identical to the oracle's."""
    from _stage_parity import check_aabb_to_pairs

    rng = np.random.default_rng(5)
    st = _pop(3000, seed=17)
    k = 0
    for g in range(60):
        size = 2 + g % 19
        src = int(rng.integers(1500, 3000))
        st[k : k + size] = st[src]
        k += size
    assert k < 1500
    st[:, 6] = np.where(rng.random(len(st)) < 0.33, 0.0, 1e-7)
    st = st[rng.permutation(len(st))]
    s = orc.Sim(st, 0.23, n_par_ct=n_par_ct, min_coll_radius=1e-9)
    s.step()
    ca, cj = s.active_flags()
    assert 0 < int(ca.sum()) < len(st)
    codes, _ = s.codes(0)
    _, cnts = np.unique(codes, return_counts=True)
    assert cnts.max() >= 20 and ((cnts > 1) & (cnts <= 8)).any()  # leaves beyond and within the tile limit
    check_aabb_to_pairs(orc, ctx, s, len(st))


def test_tree_order_changes_nothing(orc, monkeypatch):
    """The broad phase on the tree as the build numbers it (straight from the Karras ids) and on
    the tree in the reference's node order (CASCADE_BVH_CANONICAL=1): same pair set; the export is
    byte-identical to the oracle's tree either way (check_aabb_to_pairs)."""
    from cascade_amd import _capi

    from _stage_parity import check_aabb_to_pairs

    for n in (5003, 40001):
        st = _pop(n, seed=11, same_code=True)
        s, _ = _oracle_step(orc, st, 2)
        counts, work = [], []
        # the reference's node order (one refit kernel from the leaves); Karras numbering with the
        # workgroup-local refit in front of the cross-workgroup one (the default)
        for canon in ("1", None):
            for var, val in (("CASCADE_BVH_CANONICAL", canon),):
                if val is None:
                    monkeypatch.delenv(var, raising=False)
                else:
                    monkeypatch.setenv(var, val)
            c2 = _capi.Context(0)
            try:
                counts.append(check_aabb_to_pairs(orc, c2, s, len(st)))
                # The traversal's work depends on every internal box: equal counts = equal boxes
                # as far as any pair of nodes can tell.
                work.append((int(c2.L.csc_broad_phase_node_pairs(c2.h)), int(c2.L.csc_broad_phase_tiles(c2.h))))
            finally:
                c2.close()
        assert counts[0] == counts[1] == sum(len(s.bp_pairs(c)) for c in range(2))
        assert work[0] == work[1] and work[0][0] > 0, work


def test_tiles_change_nothing(orc, ctx, monkeypatch):
    """The pair set with the tile kernel (default) and without tiles."""
    import ctypes as C

    from cascade_amd import _capi

    st = _pop(20000, seed=3)
    s, _ = _oracle_step(orc, st, 2)
    ref = [s.bp_pairs(c) for c in range(2)]
    from _stage_parity import check_aabb_to_pairs

    for tile in ("0", None):
        if tile is None:
            monkeypatch.delenv("CASCADE_DT_TILE", raising=False)
        else:
            monkeypatch.setenv("CASCADE_DT_TILE", tile)
        c2 = _capi.Context(0)
        try:
            assert check_aabb_to_pairs(orc, c2, s, len(st)) == sum(len(r) for r in ref)
        finally:
            c2.close()


def _split_rows(tcs, nsub, order):
    """Oracle coefficients [tot][7][21] (particle order) -> the library's CSR layout with the
    rows of particle order[k] stored k-th: (tc_row [n], xyz [tot][64], vr [tot][84], row_perm)
    where row_perm[new_row] = old_row."""
    n = len(nsub)
    off = np.concatenate([[0], np.cumsum(nsub)]).astype(np.int64)
    ns_ord = nsub[order].astype(np.int64)
    base = np.concatenate([[0], np.cumsum(ns_ord)])[:-1]
    tc_row = np.zeros(n, np.uint64)
    tc_row[order] = base.astype(np.uint64)
    row_perm = np.concatenate([np.arange(off[i], off[i + 1]) for i in order]) if n else np.zeros(0, np.int64)
    t = tcs[row_perm]
    xyz = np.zeros((len(t), 64))
    xyz[:, :63] = t[:, :3, :].reshape(len(t), 63)
    vr = np.ascontiguousarray(t[:, 3:, :].reshape(len(t), 84))
    return tc_row, xyz, vr, row_perm


def _gpu_propagate(ctx, st, cfg_kw, order=None):
    from cascade_amd import _capi

    L, h, chk = ctx.L, ctx.h, ctx._chk
    n = len(st)
    cfg = _capi.PropCfg()
    for k, v in cfg_kw.items():
        if k == "reentry_k":
            for i in range(3):
                cfg.reentry_k[i] = v[i]
        else:
            setattr(cfg, k, v)
    Cn = cfg.nchunks
    d_state = ctx.array(st)
    d = dict(
        nsub=ctx.empty(n, np.uint32),
        status=ctx.empty(n, np.int32),
        ev_idx=ctx.empty(n, np.uint32),
        ev_time=ctx.empty(n, np.float64),
        nf=ctx.empty(n, np.float64),
        final=ctx.empty((n, 6), np.float64),
        lbs=ctx.empty((Cn, n, 4), np.float32),
        ubs=ctx.empty((Cn, n, 4), np.float32),
    )
    chk(L.csc_propagate(h, C.byref(cfg), n, d_state.ptr, d["nsub"].ptr, d["status"].ptr, d["ev_idx"].ptr, d["ev_time"].ptr, d["nf"].ptr, d["final"].ptr, d["lbs"].ptr, d["ubs"].ptr, None))
    summ = _capi.PropSummary()
    chk(L.csc_propagate_summary(h, n, d["nsub"].ptr, d["status"].ptr, d["ev_idx"].ptr, d["ev_time"].ptr, d["nf"].ptr, C.byref(summ)))
    tot = int(summ.total_substeps)
    xs, vs = C.c_uint32(), C.c_uint32()
    chk(L.csc_tc_strides(20, C.byref(xs), C.byref(vs)))
    assert (xs.value, vs.value) == (64, 84)
    d_row = ctx.empty(n, np.uint64)
    d_xyz = ctx.empty((max(tot, 1), 64), np.float64)
    d_vr = ctx.empty((max(tot, 1), 84), np.float64)
    d_tco = ctx.empty((max(tot, 1), 2), np.float64)
    d_order = ctx.array(np.asarray(order, dtype=np.uint32)) if order is not None else None
    chk(L.csc_propagate_store(h, C.byref(cfg), n, d_state.ptr, d_order.ptr if d_order else None, d["nsub"].ptr, d_row.ptr, d_xyz.ptr, d_vr.ptr, d_tco.ptr))
    out = {k: v.download() for k, v in d.items()}
    # Back to the oracle's layout: [tot][7][21] with the rows in particle order.
    row, xyz, vr, tco = d_row.download(), d_xyz.download()[:tot], d_vr.download()[:tot], d_tco.download()[:tot]
    nsub = out["nsub"]
    rows = np.concatenate([np.arange(row[i], row[i] + nsub[i]) for i in range(n)]).astype(np.int64) if tot else np.zeros(0, np.int64)
    assert len(rows) == tot and np.array_equal(np.sort(rows), np.arange(tot)), "rows are a partition"
    assert np.all(xyz[:, 63] == 0.0)
    tcs = np.concatenate([xyz[rows, :63].reshape(tot, 3, 21), vr[rows].reshape(tot, 4, 21)], axis=1)
    out.update(summ=summ, row=row, tcs=tcs, tco=tco[rows])
    out["dev"] = dict(state=d_state, row=d_row, nsub=d["nsub"], xyz=d_xyz, vr=d_vr, tco=d_tco)
    return out


def _gpu_propagate_rows(ctx, st, cfg_kw, row_cap):
    """csc_propagate (boxes fused into the integration, no rows) and csc_propagate_rows (boxes
    from the rows by k_chunk_boxes) on the same input: (fused, rows) dictionaries of host arrays."""
    from cascade_amd import _capi

    L, h, chk = ctx.L, ctx.h, ctx._chk
    n = len(st)
    cfg = _capi.PropCfg()
    for k, v in cfg_kw.items():
        if k == "reentry_k":
            for i in range(3):
                cfg.reentry_k[i] = v[i]
        else:
            setattr(cfg, k, v)
    Cn = cfg.nchunks
    d_state = ctx.array(st)
    outs = []
    for rows in (False, True):
        d = dict(
            nsub=ctx.empty(n, np.uint32),
            status=ctx.empty(n, np.int32),
            ev_idx=ctx.empty(n, np.uint32),
            ev_time=ctx.empty(n, np.float64),
            nf=ctx.empty(n, np.float64),
            final=ctx.empty((n, 6), np.float64),
            lbs=ctx.array(np.full((Cn, n, 4), 7.0, np.float32)),
            ubs=ctx.array(np.full((Cn, n, 4), 7.0, np.float32)),
            recs=ctx.array(np.full((Cn, n, 16), 7.0, np.float32)),
        )
        args = (h, C.byref(cfg), n, d_state.ptr, d["nsub"].ptr, d["status"].ptr, d["ev_idx"].ptr, d["ev_time"].ptr, d["nf"].ptr, d["final"].ptr, d["lbs"].ptr, d["ubs"].ptr, d["recs"].ptr)
        if rows:
            d_row = ctx.empty(n, np.uint64)
            d_xyz = ctx.empty((n * row_cap, 64), np.float64)
            d_vr = ctx.empty((n * row_cap, 84), np.float64)
            d_tco = ctx.empty((n * row_cap, 2), np.float64)
            chk(L.csc_propagate_rows(*args, row_cap, d_row.ptr, d_xyz.ptr, d_vr.ptr, d_tco.ptr))
        else:
            chk(L.csc_propagate(*args))
        chk(L.csc_sync(h))
        outs.append({k: v.download() for k, v in d.items()})
    return outs


@pytest.mark.parametrize("n,ct,n_par_ct,conj,events", [(2003, 0.23, 1, 0.0, False), (2003, 0.23, 8, 0.05, False), (5000, 0.23, 3, 0.02, True), (3, 0.5, 5, 0.0, False), (1500, 0.01, 7, 0.1, True)])
def test_chunk_boxes_from_rows(orc, ctx, n, ct, n_par_ct, conj, events):
    """Pass 1 with rows leaves the per-chunk boxes and trajectory samples to k_chunk_boxes, which
    computes them from the rows: the same bits as the code fused into the integration (which the
    other tests of this file hold against the oracle), chunks never reached and chunks cut short
    by a terminal event included."""
    st = _pop(n)
    kw = _cfg_kw(orc, ct, n_par_ct, conj_thresh=conj, exit_radius=1.25 if events else 0.0, reentry=1.03 if events else 0.0)
    fused, _ = _gpu_propagate_rows(ctx, st, kw, 1)
    cap = int(fused["nsub"].max()) + 1
    fused, rows = _gpu_propagate_rows(ctx, st, kw, cap)
    if events:
        assert np.any((fused["status"] & 0xff) != 0), "some particle stops at an event"
    for k in ("nsub", "status", "ev_idx", "lbs", "ubs"):
        assert np.array_equal(fused[k].view(np.uint8), rows[k].view(np.uint8)), k
    assert np.array_equal(fused["final"], rows["final"])
    fd, rd = fused["recs"][..., 15], rows["recs"][..., 15]
    assert np.array_equal(fd.view(np.uint32), rd.view(np.uint32)), "record radii"
    ok = np.isfinite(fd)
    assert ok.any()
    assert np.array_equal(fused["recs"][ok].view(np.uint32), rows["recs"][ok].view(np.uint32)), "samples of the valid records"
    # Too few row slots: the flag is raised, and the particles that fit are unaffected.
    if cap > 2:
        _, short = _gpu_propagate_rows(ctx, st, kw, cap - 2)
        over = fused["nsub"] > cap - 2
        assert over.any() and np.all((short["status"][over] & 0x1000) != 0) and not np.any(short["status"][~over] & 0x1000)
        assert np.array_equal(fused["lbs"][:, ~over].view(np.uint32), short["lbs"][:, ~over].view(np.uint32))
        assert np.array_equal(fused["ubs"][:, ~over].view(np.uint32), short["ubs"][:, ~over].view(np.uint32))


def _cfg_kw(orc, ct, n_par_ct, conj_thresh=0.0, t0=(0.0, 0.0), exit_radius=0.0, reentry=0.0):
    kw = dict(dyn_kind=0, order=20, mu=1.0, rhofac=orc.lib().orc_rhofac(20), init_time_hi=t0[0], init_time_lo=t0[1], delta_t=ct * n_par_ct, ct=ct, nchunks=n_par_ct, max_substeps=0, conj_thresh=conj_thresh)
    kw.update(has_exit=1 if exit_radius > 0 else 0, exit_c0=exit_radius * exit_radius)
    if np.ndim(reentry) == 0:
        kw.update(has_reentry=1 if reentry > 0 else 0, reentry_k=(1.0, 1.0, 1.0), reentry_c0=reentry * reentry)
    else:
        a, b, c = reentry
        kw.update(has_reentry=1, reentry_k=((b * c) * (b * c), (a * c) * (a * c), (a * b) * (a * b)), reentry_c0=a * a * b * b * c * c)
    return kw


def _rel_close(a, b, rtol=RTOL, scale=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    sc = np.maximum(np.abs(b), 1.0 if scale is None else scale)
    return np.all(np.abs(a - b) <= rtol * sc)


@pytest.mark.parametrize("n,ct,n_par_ct,conj", [(64, 0.23, 1, 0.0), (2003, 0.23, 1, 0.0), (2003, 0.23, 3, 1e-4), (500, 0.23, 8, 0.0), (300, 1.7, 4, 0.0)])
def test_propagate_parity(orc, ctx, n, ct, n_par_ct, conj):
    """K1/K2: substep counts equal; substep times, Taylor coefficients and final states within
    1e-9 relative (they are in fact expected bit-identical: same operation order); AABBs
    bit-identical when the coefficients are."""
    orc.lib().orc_rhofac.restype = C.c_double
    orc.lib().orc_rhofac.argtypes = [C.c_uint32]
    st = _pop(n, seed=7)
    s = orc.Sim(st, ct, n_par_ct=n_par_ct, conj_thresh=conj)
    s.time = 3.25
    assert s.step() in ("success", "collision")
    nsub, tcs, tco = s.substeps()
    order = np.random.default_rng(n).permutation(n) if n_par_ct > 1 else None
    g = _gpu_propagate(ctx, st, _cfg_kw(orc, ct, n_par_ct, conj, t0=(3.25, 0.0)), order=order)
    assert np.array_equal(g["nsub"], nsub)
    assert np.all((g["status"] & 0xFF) == 0) and g["summ"].flags == 0
    assert _rel_close(g["tco"][:, 0], tco[:, 0])
    # Taylor coefficients: relative to the magnitude of the order-0 coefficient of the same variable
    scale = np.abs(tcs[:, :, :1]) + 1.0
    assert np.all(np.abs(g["tcs"] - tcs) <= RTOL * scale)
    assert _rel_close(g["final"], s.final_state())
    bit_exact = np.array_equal(g["tcs"], tcs) and np.array_equal(g["tco"], tco)
    print("bit-exact TCs:", bit_exact)
    if bit_exact:
        for c in range(s.nchunks):
            lb, ub = s.aabbs(c)
            assert np.array_equal(g["lbs"][c], lb) and np.array_equal(g["ubs"][c], ub)
    else:
        for c in range(s.nchunks):
            lb, ub = s.aabbs(c)
            assert np.allclose(g["lbs"][c], lb, rtol=1e-6, atol=1e-6) and np.allclose(g["ubs"][c], ub, rtol=1e-6, atol=1e-6)


def test_propagate_events_and_nf(orc, ctx):
    """Terminal events (src/sim_basic.cpp:509-578) and non-finite states in K1."""
    orc.lib().orc_rhofac.restype = C.c_double
    orc.lib().orc_rhofac.argtypes = [C.c_uint32]
    a = [1.1, 0.0, 0.0, 0.0, 0.953, 0.0, 0.0]
    st = np.array([a, [1.5, 0.0, 0.0, 0.0, 0.01, 0.0, 0.0], a, [1.1, 0, 0, 0, 150.0, 0, 0], a, a])
    s = orc.Sim(st, 0.23, n_par_ct=3, reentry_radius=1.0, exit_radius=10.0)
    oc = s.step()
    g = _gpu_propagate(ctx, st, _cfg_kw(orc, 0.23, 3, exit_radius=10.0, reentry=1.0))
    nsub, tcs, tco = s.substeps()
    assert np.array_equal(g["nsub"], nsub)
    assert (g["status"][3] & 0xFF) == 1 and g["ev_idx"][3] == 0  # exit
    assert oc == "exit" and s.interrupt_info == 3
    assert g["summ"].n_ste == 1 and g["summ"].ste_idx == 3
    assert _rel_close(g["tco"][:, 0], tco[:, 0])
    # non-finite state
    st2 = st.copy()
    st2[2, 0] = np.inf
    g = _gpu_propagate(ctx, st2, _cfg_kw(orc, 0.23, 1))
    assert (g["status"][2] & 0xFF) == 2 and g["summ"].n_nf == 1 and g["summ"].nf_idx == 2 and g["summ"].nf_time == 0.0


@pytest.mark.parametrize("n,n_par_ct,conj", [(2003, 1, 0.0), (2003, 3, 2e-3), (3000, 8, 1e-3)])
def test_narrow_phase_injected(orc, ctx, n, n_par_ct, conj):
    """K7 on the oracle's Taylor coefficients and the oracle's broad-phase pairs: collision and
    conjunction sets equal, times/distances/states within 1e-9 relative."""
    st = _pop(n, seed=11)
    st[:, 6] *= 3e4  # bigger objects -> some collisions in one superstep
    s = orc.Sim(st, 0.23, n_par_ct=n_par_ct, conj_thresh=conj)
    s.time = 1.5
    oc = s.step()
    nsub, tcs, tco = s.substeps()
    order = np.random.default_rng(5).permutation(n)
    tc_row, xyz, vr, row_perm = _split_rows(tcs, nsub, order)
    pairs = [s.bp_pairs(c) for c in range(s.nchunks)]
    poff = np.concatenate([[0], np.cumsum([len(p) for p in pairs])]).astype(np.uint64)
    allp = np.concatenate(pairs) if poff[-1] else np.zeros((0, 2), np.uint32)
    ca, cj = s.active_flags()
    L, h, chk = ctx.L, ctx.h, ctx._chk
    d_state, d_act = ctx.array(st), ctx.array((ca | (cj << 1)).astype(np.uint8))
    d_row, d_ns, d_xyz, d_vr = ctx.array(tc_row), ctx.array(nsub.astype(np.uint32)), ctx.array(xyz), ctx.array(vr)
    d_tco, d_pairs = ctx.array(tco[row_perm]), ctx.array(allp)
    ncoll, nconj, warn = C.c_uint64(), C.c_uint64(), C.c_uint32()
    chk(
        L.csc_narrow_phase_pairs(
            h, n, s.nchunks, 0.23, s.delta_t, 1.5, 0.0, conj, 20, d_state.ptr, d_act.ptr, d_row.ptr, d_ns.ptr, d_xyz.ptr, d_vr.ptr, d_tco.ptr, d_pairs.ptr, poff.ctypes.data_as(C.c_void_p), C.byref(ncoll), C.byref(nconj), C.byref(warn)
        )
    )
    from cascade_amd import _capi

    coll = np.empty(ncoll.value, dtype=_capi.COLL_DTYPE)
    cj_ = np.empty(nconj.value, dtype=_capi.CONJ_DTYPE)
    chk(L.csc_narrow_phase_get(h, coll.ctypes.data_as(C.c_void_p), len(coll), cj_.ctypes.data_as(C.c_void_p), len(cj_)))
    ocoll, oconj = s.collisions(), s.step_conjunctions()
    print("pairs", poff[-1], "collisions", len(ocoll), "conjunctions", len(oconj), "outcome", oc)
    assert len(coll) == len(ocoll) and len(cj_) == len(oconj)
    key = lambda a: np.lexsort((a["j"], a["i"], a["time"]))
    coll, ocoll = coll[key(coll)], ocoll[key(ocoll)]
    assert np.array_equal(coll["i"], ocoll["i"]) and np.array_equal(coll["j"], ocoll["j"])
    assert _rel_close(coll["time"], ocoll["time"])
    cj_, oconj = cj_[key(cj_)], oconj[key(oconj)]
    assert np.array_equal(cj_["i"], oconj["i"]) and np.array_equal(cj_["j"], oconj["j"])
    assert _rel_close(cj_["time"], oconj["time"]) and _rel_close(cj_["state_i"], oconj["state_i"]) and _rel_close(cj_["state_j"], oconj["state_j"])
    assert np.all(np.abs(cj_["dist"] - oconj["dist"]) <= 1e-9)
    if True:
        assert len(ocoll) > 0, "test population should produce collisions"


def test_dense_output(orc, ctx):
    """K8 vs the oracle's dense_propagate (via an interrupted step)."""
    orc.lib().orc_rhofac.restype = C.c_double
    orc.lib().orc_rhofac.argtypes = [C.c_uint32]
    st = _pop(500, seed=3)
    g = _gpu_propagate(ctx, st, _cfg_kw(orc, 0.23, 4), order=np.random.default_rng(1).permutation(500))
    L, h, chk = ctx.L, ctx.h, ctx._chk
    d_out = ctx.empty((500, 6), np.float64)
    # state at t = full superstep must reproduce the final state to rounding
    chk(L.csc_dense_propagate(h, 500, 20, 0.92, g["dev"]["row"].ptr, g["dev"]["nsub"].ptr, g["dev"]["xyz"].ptr, g["dev"]["vr"].ptr, g["dev"]["tco"].ptr, d_out.ptr))
    chk(L.csc_check_flags(h))
    assert np.allclose(d_out.download(), g["final"], rtol=0, atol=1e-14)
    # mid-superstep: compare with an oracle propagate_until (which uses dense output)
    s = orc.Sim(st, 0.23, n_par_ct=4)
    assert s.propagate_until(0.5) == "time_limit"
    chk(L.csc_dense_propagate(h, 500, 20, 0.5, g["dev"]["row"].ptr, g["dev"]["nsub"].ptr, g["dev"]["xyz"].ptr, g["dev"]["vr"].ptr, g["dev"]["tco"].ptr, d_out.ptr))
    assert _rel_close(d_out.download(), s.state[:, :6])


@pytest.mark.parametrize("order", [None, [2, 0, 1]])
def test_store_pass_next_to_terminal_events(orc, ctx, order):
    """Particles of one wave that take the terminal-event root search at different substeps (one
    leaves the domain in its first substep, one re-enters in its fourth, one meets no event): the
    store pass writes the same substeps as the first pass counted, and they are the oracle's.
    (The root search used to be an out-of-line call inside the integration loop; it reset the
    substep counter of the lane that was not part of the call.)"""
    orc.lib().orc_rhofac.restype = C.c_double
    orc.lib().orc_rhofac.argtypes = [C.c_uint32]
    st = np.array([[0.99, 0, 0, 1, 0, 0, 1e-6], [1.01, 0, 0, -1, 0, 0, 1e-6], [4.9, 0, 0, 1, 0, 0, 1e-6]])
    s = orc.Sim(st, 0.23, n_par_ct=3, exit_radius=5.0, reentry_radius=0.75)
    s.step()
    nsub, tcs, tco = s.substeps()
    g = _gpu_propagate(ctx, st, _cfg_kw(orc, 0.23, 3, exit_radius=5.0, reentry=0.75), order=order)
    assert np.array_equal(g["nsub"], nsub) and list(nsub) == [6, 4, 1]
    assert _rel_close(g["tco"], tco) and _rel_close(g["tcs"], tcs, scale=np.abs(tcs).max())
