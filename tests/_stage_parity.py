"""Stage-by-stage parity on injected AABBs (shared by tests/test_gpu_parity.py and
tests/test_gpu_fixtures.py): the oracle's per-chunk AABBs go into the HIP pipeline through the
C ABI; global AABB, Morton codes, stable permutation, sorted arrays, BVH node bytes and the
broad-phase pair set must be bit-identical to the oracle's (BASELINE.json north_star)."""
import ctypes as C

import numpy as np


def check_aabb_to_pairs(orc, ctx, s, n):
    """s: an oracle simulation that has just run step(); n: its particle count."""
    C_ = s.nchunks
    L, h, chk = ctx.L, ctx.h, ctx._chk
    lbs = np.stack([s.aabbs(c)[0] for c in range(C_)])
    ubs = np.stack([s.aabbs(c)[1] for c in range(C_)])
    d_lbs, d_ubs = ctx.array(lbs), ctx.array(ubs)
    d_glb, d_gub = ctx.empty((C_, 4), np.float32), ctx.empty((C_, 4), np.float32)
    chk(L.csc_global_aabb(h, n, C_, d_lbs.ptr, d_ubs.ptr, d_glb.ptr, d_gub.ptr))
    glb, gub = d_glb.download(), d_gub.download()
    for c in range(C_):
        olb, oub = s.global_aabb(c)
        assert np.array_equal(glb[c], olb)
        assert np.array_equal(np.nextafter(gub[c], np.float32(np.inf)), oub)  # oracle holds the bumped value
    d_codes = ctx.empty((C_, n), np.uint64)
    chk(L.csc_morton_encode(h, n, C_, d_lbs.ptr, d_ubs.ptr, d_glb.ptr, d_gub.ptr, d_codes.ptr))
    chk(L.csc_check_flags(h))
    codes = d_codes.download()
    d_vidx, d_sc = ctx.empty((C_, n), np.uint32), ctx.empty((C_, n), np.uint64)
    d_slb, d_sub = ctx.empty((C_, n, 4), np.float32), ctx.empty((C_, n, 4), np.float32)
    chk(L.csc_sort_gather(h, n, C_, d_codes.ptr, d_lbs.ptr, d_ubs.ptr, d_vidx.ptr, d_sc.ptr, d_slb.ptr, d_sub.ptr, None, None))
    vidx, sc, slb, sub = d_vidx.download(), d_sc.download(), d_slb.download(), d_sub.download()
    chk(L.csc_bvh_build(h, n, C_, d_sc.ptr, d_slb.ptr, d_sub.ptr))
    ca, cj = s.active_flags()
    d_act = ctx.array((ca | (cj << 1)).astype(np.uint8))
    npairs = C.c_uint64()
    chk(L.csc_broad_phase(h, n, C_, d_slb.ptr, d_sub.ptr, d_vidx.ptr, d_act.ptr, C.byref(npairs)))
    tot = 0
    for c in range(C_):
        ocodes, ovidx = s.codes(c)
        assert np.array_equal(codes[c], ocodes), "Morton codes"
        assert np.array_equal(vidx[c], ovidx), "sort permutation"
        osc, oslb, osub = s.sorted_data(c)
        assert np.array_equal(sc[c], osc) and np.array_equal(slb[c], oslb) and np.array_equal(sub[c], osub)
        otree = s.bvh(c)
        d_nodes = ctx.empty((2 * n,), orc.BVH_NODE_DTYPE)
        nn = C.c_uint64()
        chk(L.csc_bvh_export(h, c, d_nodes.ptr, 2 * n, C.byref(nn)))
        tree = d_nodes.download()[: nn.value]
        assert nn.value == len(otree), "node count"
        for f in ("begin", "end", "parent", "left", "right", "nn_level", "split_idx", "lb", "ub"):
            assert np.array_equal(tree[f], otree[f]), f"BVH field {f}"
        assert tree.tobytes() == otree.tobytes()
        obp = s.bp_pairs(c)
        cnt = C.c_uint64()
        got = np.empty((max(len(obp), 1) * 2 + 16, 2), dtype=np.uint32)
        chk(L.csc_broad_phase_get(h, c, got.ctypes.data_as(C.c_void_p), len(got), C.byref(cnt)))
        assert cnt.value == len(obp), "pair count"
        g = got[: cnt.value]
        g = g[np.lexsort((g[:, 1], g[:, 0]))]
        assert np.array_equal(g, obp), "broad-phase pair set"
        tot += cnt.value
    assert tot == npairs.value
    return tot
