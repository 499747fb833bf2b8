"""Full-size checks at BASELINE.json configs[1] (1,000,000 LEO objects, Kepler, ct = 0.23): the CPU
oracle cannot run this size in seconds, so the GPU path is checked through size-independent
properties and through parity on samples:

  * stage invariants on the 1M-object pipeline through the C ABI -- sorted codes, a valid
    permutation, gathered arrays, the BVH invariants of the reference's debug verifier
    (src/sim_bvh.cpp:498-656: ranges partition, children adjacent, parent box = exact union),
    the broad-phase properties of src/sim_broad_phase.cpp:221-321 (unique, i < j) plus EXACT
    equality with an O(N) brute-force overlap search for a sample of objects;
  * two independent broad-phase algorithms (dual traversal, per-particle walks) agree on the
    pair count and on the pair set;
  * whole supersteps: the states of a random sample agree with the CPU oracle run on that
    sample alone (objects only interact through collisions, and there are none) to 1e-9,
    Kepler energy and angular momentum are conserved, the run is deterministic.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1_000_000


@pytest.fixture(scope="module")
def pop():
    from cascade_amd.workloads import random_leo_pop_kep

    return random_leo_pop_kep(N, seed=42)


def test_pipeline_invariants_1m(orc, pop):
    from cascade_amd import _capi

    ctx = _capi.Context(0)
    L, h, chk = ctx.L, ctx.h, ctx._chk
    n = N
    cfg = _capi.PropCfg()
    for k, v in dict(dyn_kind=0, order=20, mu=1.0, rhofac=orc.lib().orc_rhofac(20), delta_t=0.23, ct=0.23, nchunks=1, reentry_c0=0.0).items():
        setattr(cfg, k, v)
    d_state = ctx.array(pop)
    d_nsub, d_status, d_ev_idx = ctx.empty(n, np.uint32), ctx.empty(n, np.int32), ctx.empty(n, np.uint32)
    d_ev_t, d_nf, d_final = ctx.empty(n, np.float64), ctx.empty(n, np.float64), ctx.empty((n, 6), np.float64)
    d_lbs, d_ubs = ctx.empty((1, n, 4), np.float32), ctx.empty((1, n, 4), np.float32)
    chk(L.csc_propagate(h, C.byref(cfg), n, d_state.ptr, d_nsub.ptr, d_status.ptr, d_ev_idx.ptr, d_ev_t.ptr, d_nf.ptr, d_final.ptr, d_lbs.ptr, d_ubs.ptr, None))
    summ = _capi.PropSummary()
    chk(L.csc_propagate_summary(h, n, d_nsub.ptr, d_status.ptr, d_ev_idx.ptr, d_ev_t.ptr, d_nf.ptr, C.byref(summ)))
    assert summ.n_nf == 0 and summ.n_ste == 0 and summ.flags == 0 and summ.total_substeps >= n
    lbs, ubs = d_lbs.download()[0], d_ubs.download()[0]
    assert np.all(np.isfinite(lbs)) and np.all(lbs <= ubs)
    # The box of a particle contains its initial and final position (+- its size).
    fin = d_final.download()
    for pos in (pop[:, :3], fin[:, :3]):
        assert np.all(lbs[:, :3] <= pos.astype(np.float32)) and np.all(pos.astype(np.float32) <= ubs[:, :3])

    d_glb, d_gub = ctx.empty((1, 4), np.float32), ctx.empty((1, 4), np.float32)
    chk(L.csc_global_aabb(h, n, 1, d_lbs.ptr, d_ubs.ptr, d_glb.ptr, d_gub.ptr))
    assert np.array_equal(d_glb.download()[0], lbs.min(axis=0)) and np.array_equal(d_gub.download()[0], ubs.max(axis=0))
    d_codes = ctx.empty((1, n), np.uint64)
    chk(L.csc_morton_encode(h, n, 1, d_lbs.ptr, d_ubs.ptr, d_glb.ptr, d_gub.ptr, d_codes.ptr))
    chk(L.csc_check_flags(h))
    d_vidx, d_sc = ctx.empty((1, n), np.uint32), ctx.empty((1, n), np.uint64)
    d_slb, d_sub = ctx.empty((1, n, 4), np.float32), ctx.empty((1, n, 4), np.float32)
    chk(L.csc_sort_gather(h, n, 1, d_codes.ptr, d_lbs.ptr, d_ubs.ptr, d_vidx.ptr, d_sc.ptr, d_slb.ptr, d_sub.ptr, None, None))
    codes, vidx, sc = d_codes.download()[0], d_vidx.download()[0], d_sc.download()[0]
    # Sortedness, permutation, stability, gather (src/sim_propagate.cpp:483-506).
    assert np.all(sc[1:] >= sc[:-1])
    assert np.array_equal(np.sort(vidx), np.arange(n, dtype=np.uint32))
    assert np.array_equal(sc, codes[vidx])
    ties = sc[1:] == sc[:-1]
    assert np.all(vidx[1:][ties] > vidx[:-1][ties]), "stable: ties in original order"
    assert np.array_equal(d_slb.download()[0], lbs[vidx]) and np.array_equal(d_sub.download()[0], ubs[vidx])
    assert np.array_equal(vidx, np.argsort(codes, kind="stable").astype(np.uint32))

    # BVH invariants (verify_bvh_trees_parallel, src/sim_bvh.cpp:498-656).
    chk(L.csc_bvh_build(h, n, 1, d_sc.ptr, d_slb.ptr, d_sub.ptr))
    d_nodes = ctx.empty((2 * n,), orc.BVH_NODE_DTYPE)
    nn = C.c_uint64()
    chk(L.csc_bvh_export(h, 0, d_nodes.ptr, 2 * n, C.byref(nn)))
    t = d_nodes.download()[: nn.value]
    leaf = t["left"] < 0
    internal = ~leaf
    assert t["begin"][0] == 0 and t["end"][0] == n and t["parent"][0] == -1
    assert np.all(t["right"][internal] == t["left"][internal] + 1), "children are adjacent"
    li, ri = t["left"][internal], t["right"][internal]
    me = np.flatnonzero(internal)
    assert np.all(t["parent"][li] == me) and np.all(t["parent"][ri] == me)
    assert np.all(t["begin"][li] == t["begin"][internal]) and np.all(t["end"][ri] == t["end"][internal])
    assert np.all(t["end"][li] == t["begin"][ri]), "children split the parent's range"
    assert np.all(li > me), "level order: children come after their parent"
    # leaves: ranges partition [0, n) in order, one distinct code per leaf
    lb_, le_ = t["begin"][leaf], t["end"][leaf]
    o = np.argsort(lb_)
    assert lb_[o][0] == 0 and le_[o][-1] == n and np.all(lb_[o][1:] == le_[o][:-1])
    assert np.all(sc[lb_] == sc[le_ - 1])
    assert nn.value == 2 * leaf.sum() - 1
    # boxes: a parent is the exact float union of its children; a single-particle leaf its particle
    assert np.array_equal(t["lb"][internal], np.minimum(t["lb"][li], t["lb"][ri]))
    assert np.array_equal(t["ub"][internal], np.maximum(t["ub"][li], t["ub"][ri]))
    single = leaf & (t["end"] - t["begin"] == 1)
    slb, sub = lbs[vidx], ubs[vidx]
    assert np.array_equal(t["lb"][single], slb[t["begin"][single]]) and np.array_equal(t["ub"][single], sub[t["begin"][single]])

    # Broad phase: pair list properties and brute force on a sample.
    act = np.ones(n, dtype=np.uint8)
    d_act = ctx.array(act)
    npairs = C.c_uint64()
    chk(L.csc_broad_phase(h, n, 1, d_slb.ptr, d_sub.ptr, d_vidx.ptr, d_act.ptr, C.byref(npairs)))
    assert L.csc_broad_phase_node_pairs(h) > 0, "the dual traversal ran"
    got = np.empty((npairs.value, 2), dtype=np.uint32)
    cnt = C.c_uint64()
    chk(L.csc_broad_phase_get(h, 0, got.ctypes.data_as(C.c_void_p), len(got), C.byref(cnt)))
    assert cnt.value == npairs.value > 20 * n
    assert np.all(got[:, 0] < got[:, 1])
    key = got[:, 0].astype(np.uint64) << np.uint64(32) | got[:, 1].astype(np.uint64)
    key.sort()
    assert np.all(key[1:] != key[:-1]), "no duplicates"
    rng = np.random.default_rng(0)
    multi = np.zeros(n, dtype=bool)  # particles in multi-particle leaves are tested against the leaf box
    for b, e in zip(t["begin"][leaf & ~single], t["end"][leaf & ~single]):
        multi[vidx[b:e]] = True
    sample = rng.choice(np.flatnonzero(~multi), 200, replace=False)
    ov_any = ~multi
    for p in sample:
        ov = np.all((ubs >= lbs[p]) & (lbs <= ubs[p]), axis=1) & ov_any
        ov[p] = False
        expect = np.flatnonzero(ov)
        mine = np.concatenate([got[got[:, 0] == p, 1], got[got[:, 1] == p, 0]])
        assert np.array_equal(np.sort(mine), expect), "pairs of a sampled object = brute force"
    # That was the tree in the reference's node order (the export rebuilt it that way). The tree as
    # csc_bvh_build numbers it (Karras ids, workgroup-local refit first): same pairs, and the same
    # traversal work, which depends on every internal box.
    work = (L.csc_broad_phase_node_pairs(h), L.csc_broad_phase_tiles(h))
    chk(L.csc_bvh_build(h, n, 1, d_sc.ptr, d_slb.ptr, d_sub.ptr))
    npairs2 = C.c_uint64()
    chk(L.csc_broad_phase(h, n, 1, d_slb.ptr, d_sub.ptr, d_vidx.ptr, d_act.ptr, C.byref(npairs2)))
    assert npairs2.value == npairs.value
    assert (L.csc_broad_phase_node_pairs(h), L.csc_broad_phase_tiles(h)) == work
    got2 = np.empty((npairs2.value, 2), dtype=np.uint32)
    chk(L.csc_broad_phase_get(h, 0, got2.ctypes.data_as(C.c_void_p), len(got2), C.byref(cnt)))
    key2 = got2[:, 0].astype(np.uint64) << np.uint64(32) | got2[:, 1].astype(np.uint64)
    key2.sort()
    assert np.array_equal(key, key2)
    ctx.close()


def test_supersteps_1m(orc, pop, monkeypatch):
    from cascade_amd import sim

    def run(mode, steps):
        if mode:
            monkeypatch.setenv("CASCADE_BP_MODE", mode)
        else:
            monkeypatch.delenv("CASCADE_BP_MODE", raising=False)
        s = sim(pop.copy(), 0.23)
        pairs = []
        for _ in range(steps):
            assert s.step().name == "success"
            pairs.append(s.last_bp_pairs)
        return s, pairs

    a, pa = run(None, 3)
    b, pb = run(None, 3)
    assert pa == pb and np.array_equal(a.state, b.state), "deterministic"
    w, pw = run("walk", 1)
    assert pw[0] == pa[0], "dual traversal and per-particle walks find the same number of pairs"

    # Kepler invariants after 3 supersteps, all objects.
    def invariants(st):
        r, v = st[:, :3], st[:, 3:6]
        e = 0.5 * np.sum(v * v, axis=1) - 1.0 / np.linalg.norm(r, axis=1)
        return e, np.cross(r, v)

    e0, h0 = invariants(pop)
    e1, h1 = invariants(a.state)
    assert np.max(np.abs(e1 - e0)) < 1e-13 and np.max(np.abs(h1 - h0)) < 1e-13
    assert abs(a.time - 0.69) < 1e-15 and np.array_equal(a.state[:, 6], pop[:, 6])

    # Parity on a sample: the oracle on 3000 of the objects alone.
    idx = np.sort(np.random.default_rng(1).choice(N, 3000, replace=False))
    o = orc.Sim(pop[idx].copy(), 0.23)
    for _ in range(3):
        assert o.step() == "success"
    assert np.all(np.abs(a.state[idx] - o.state) <= 1e-9 * np.maximum(1.0, np.abs(o.state)))


def test_bench_configuration_1m_8_chunks(orc, pop, monkeypatch):
    """bench.py's own configuration: 1M objects, ct = 0.23, n_par_ct = 8. At 8 chunks step() takes
    another path than at 1 (rows written during the integration, the boxes of all chunks from the
    rows by k_chunk_boxes, 8-chunk sort / tree / traversal / tile launches, store pass for
    everybody). Stage invariants per chunk through the C ABI, then whole supersteps."""
    from cascade_amd import _capi, sim

    Cn, n = 8, N
    ctx = _capi.Context(0)
    L, h, chk = ctx.L, ctx.h, ctx._chk
    cfg = _capi.PropCfg()
    for k, v in dict(dyn_kind=0, order=20, mu=1.0, rhofac=orc.lib().orc_rhofac(20), delta_t=0.23 * Cn, ct=0.23, nchunks=Cn, reentry_c0=0.0).items():
        setattr(cfg, k, v)
    d_state = ctx.array(pop)

    def outputs():
        return dict(nsub=ctx.empty(n, np.uint32), status=ctx.empty(n, np.int32), ev_idx=ctx.empty(n, np.uint32), ev_time=ctx.empty(n, np.float64),
                    nf=ctx.empty(n, np.float64), final=ctx.empty((n, 6), np.float64), lbs=ctx.empty((Cn, n, 4), np.float32),
                    ubs=ctx.empty((Cn, n, 4), np.float32), recs=ctx.empty((Cn, n, 16), np.float32))

    def args(d):
        return (h, C.byref(cfg), n, d_state.ptr, d["nsub"].ptr, d["status"].ptr, d["ev_idx"].ptr, d["ev_time"].ptr, d["nf"].ptr, d["final"].ptr, d["lbs"].ptr, d["ubs"].ptr, d["recs"].ptr)

    # Boxes fused into the integration (what the oracle-parity tests pin) against the boxes that
    # k_chunk_boxes computes from the rows of a pass that writes them: the same bits at full size.
    f = outputs()
    chk(L.csc_propagate(*args(f)))
    nsub = f["nsub"].download()
    cap = int(nsub.max()) + 1
    assert 2 <= cap <= 8
    r = outputs()
    d_row, d_xyz = ctx.empty(n, np.uint64), ctx.empty((n * cap, 64), np.float64)
    d_vr, d_tco = ctx.empty((n * cap, 84), np.float64), ctx.empty((n * cap, 2), np.float64)
    chk(L.csc_propagate_rows(*args(r), cap, d_row.ptr, d_xyz.ptr, d_vr.ptr, d_tco.ptr))
    summ = _capi.PropSummary()
    chk(L.csc_propagate_summary(h, n, r["nsub"].ptr, r["status"].ptr, r["ev_idx"].ptr, r["ev_time"].ptr, r["nf"].ptr, C.byref(summ)))
    assert summ.n_nf == 0 and summ.n_ste == 0 and summ.flags == 0 and summ.total_substeps == int(nsub.sum())
    lbs, ubs = r["lbs"].download(), r["ubs"].download()
    for k in ("nsub", "status", "final"):
        assert np.array_equal(f[k].download(), r[k].download()), k
    assert np.array_equal(f["lbs"].download().view(np.uint32), lbs.view(np.uint32)) and np.array_equal(f["ubs"].download().view(np.uint32), ubs.view(np.uint32))
    assert np.array_equal(f["recs"].download().view(np.uint32), r["recs"].download().view(np.uint32))
    assert np.all(np.isfinite(lbs)) and np.all(lbs <= ubs)
    # chunk 0 contains the initial position, chunk 7 the final one, consecutive chunks overlap
    fin = r["final"].download()
    p0, p1 = pop[:, :3].astype(np.float32), fin[:, :3].astype(np.float32)
    assert np.all(lbs[0, :, :3] <= p0) and np.all(p0 <= ubs[0, :, :3]) and np.all(lbs[7, :, :3] <= p1) and np.all(p1 <= ubs[7, :, :3])
    assert np.all(lbs[1:] <= ubs[:-1]) and np.all(lbs[:-1] <= ubs[1:])
    del f

    d_glb, d_gub = ctx.empty((Cn, 4), np.float32), ctx.empty((Cn, 4), np.float32)
    chk(L.csc_global_aabb(h, n, Cn, r["lbs"].ptr, r["ubs"].ptr, d_glb.ptr, d_gub.ptr))
    assert np.array_equal(d_glb.download(), lbs.min(axis=1)) and np.array_equal(d_gub.download(), ubs.max(axis=1))
    d_codes = ctx.empty((Cn, n), np.uint64)
    chk(L.csc_morton_encode(h, n, Cn, r["lbs"].ptr, r["ubs"].ptr, d_glb.ptr, d_gub.ptr, d_codes.ptr))
    chk(L.csc_check_flags(h))
    d_vidx, d_sc = ctx.empty((Cn, n), np.uint32), ctx.empty((Cn, n), np.uint64)
    d_slb, d_sub, d_srec = ctx.empty((Cn, n, 4), np.float32), ctx.empty((Cn, n, 4), np.float32), ctx.empty((Cn, n, 16), np.float32)
    chk(L.csc_sort_gather(h, n, Cn, d_codes.ptr, r["lbs"].ptr, r["ubs"].ptr, d_vidx.ptr, d_sc.ptr, d_slb.ptr, d_sub.ptr, r["recs"].ptr, d_srec.ptr))
    codes, vidx, sc = d_codes.download(), d_vidx.download(), d_sc.download()
    slb, sub, recs, srec = d_slb.download(), d_sub.download(), r["recs"].download(), d_srec.download()
    for c in range(Cn):
        assert np.all(sc[c, 1:] >= sc[c, :-1])
        assert np.array_equal(np.bincount(vidx[c], minlength=n), np.ones(n, dtype=np.int64)), "a permutation"
        assert np.array_equal(sc[c], codes[c][vidx[c]])
        ties = sc[c, 1:] == sc[c, :-1]
        assert np.all(vidx[c, 1:][ties] > vidx[c, :-1][ties]), "stable: ties in original order"
        assert np.array_equal(slb[c], lbs[c][vidx[c]]) and np.array_equal(sub[c], ubs[c][vidx[c]])
        assert np.array_equal(srec[c].view(np.uint32), recs[c][vidx[c]].view(np.uint32))
    assert len({codes[c].tobytes() for c in range(Cn)}) == Cn, "every chunk has its own codes"

    chk(L.csc_bvh_build(h, n, Cn, d_sc.ptr, d_slb.ptr, d_sub.ptr))
    act = np.ones(n, dtype=np.uint8)
    d_act = ctx.array(act)
    npairs = C.c_uint64()
    chk(L.csc_broad_phase(h, n, Cn, d_slb.ptr, d_sub.ptr, d_vidx.ptr, d_act.ptr, C.byref(npairs)))
    assert L.csc_broad_phase_node_pairs(h) > 0 and L.csc_broad_phase_tiles(h) > 0, "dual traversal ending in tiles"
    rng = np.random.default_rng(3)
    tot = 0
    for c in range(Cn):
        cnt = C.c_uint64()
        got = np.empty((npairs.value // Cn * 2, 2), dtype=np.uint32)
        chk(L.csc_broad_phase_get(h, c, got.ctypes.data_as(C.c_void_p), len(got), C.byref(cnt)))
        assert cnt.value <= len(got)
        got = got[: cnt.value]
        tot += cnt.value
        assert cnt.value > 20 * n and np.all(got[:, 0] < got[:, 1])
        if c in (0, 5):
            key = np.sort(got[:, 0].astype(np.uint64) << np.uint64(32) | got[:, 1].astype(np.uint64))
            assert np.all(key[1:] != key[:-1]), "no duplicates"
        # brute force for a sample of objects that do not share their Morton code (those are tested against the leaf's box)
        multi = np.zeros(n, dtype=bool)
        t = np.flatnonzero(sc[c, 1:] == sc[c, :-1])
        multi[vidx[c][t]] = True
        multi[vidx[c][t + 1]] = True
        for p in rng.choice(np.flatnonzero(~multi), 12, replace=False):
            ov = np.all((ubs[c] >= lbs[c][p]) & (lbs[c] <= ubs[c][p]), axis=1) & ~multi
            ov[p] = False
            mine = np.concatenate([got[got[:, 0] == p, 1], got[got[:, 1] == p, 0]])
            assert np.array_equal(np.sort(mine), np.flatnonzero(ov)), "pairs of a sampled object = brute force"
    assert tot == npairs.value
    del r, d_xyz, d_vr, d_tco
    ctx.close()

    # Whole supersteps: deterministic, the per-particle walks find as many pairs, Kepler invariants,
    # and the states of a sample agree with the oracle run on that sample alone.
    def run(mode, steps):
        if mode:
            monkeypatch.setenv("CASCADE_BP_MODE", mode)
        else:
            monkeypatch.delenv("CASCADE_BP_MODE", raising=False)
        s = sim(pop.copy(), 0.23, n_par_ct=Cn)
        s.host_sync = False
        pairs = []
        for _ in range(steps):
            assert s.step().name == "success"
            pairs.append(s.last_bp_pairs)
        s.download_state()
        return s, pairs

    a, pa = run(None, 3)
    assert pa[0] == npairs.value, "step() and the stage calls find the same pairs"
    b, pb = run(None, 3)
    assert pa == pb and np.array_equal(a.state, b.state), "deterministic"
    w, pw = run("walk", 1)
    assert pw[0] == pa[0], "dual traversal + tiles and per-particle walks find the same number of pairs"
    rv, vv = pop[:, :3], pop[:, 3:6]
    r1, v1 = a.state[:, :3], a.state[:, 3:6]
    e0 = 0.5 * np.sum(vv * vv, axis=1) - 1.0 / np.linalg.norm(rv, axis=1)
    e1 = 0.5 * np.sum(v1 * v1, axis=1) - 1.0 / np.linalg.norm(r1, axis=1)
    assert np.max(np.abs(e1 - e0)) < 1e-13 and np.max(np.abs(np.cross(r1, v1) - np.cross(rv, vv))) < 1e-13
    assert abs(a.time - 3 * 8 * 0.23) < 1e-14
    idx = np.sort(np.random.default_rng(1).choice(N, 3000, replace=False))
    o = orc.Sim(pop[idx].copy(), 0.23, n_par_ct=Cn)
    for _ in range(3):
        assert o.step() == "success"
    assert np.all(np.abs(a.state[idx] - o.state) <= 1e-9 * np.maximum(1.0, np.abs(o.state)))
