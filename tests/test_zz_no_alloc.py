"""GPU: csc_global_aabb_noalloc (the one entry point that allocates nothing: its partial boxes
live in a static device array inside libcascade_hip.so) against csc_global_aabb, through the C
ABI, bit for bit, on two calls in a row -- the second with a smaller input, so that stale
partials of the first call would show -- plus the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _boxes(rng, nchunks, n):
    c = rng.normal(0.0, 3.0, size=(nchunks, n, 4)).astype(np.float32)
    h = rng.uniform(0.0, 0.1, size=(nchunks, n, 4)).astype(np.float32)
    return c - h, c + h


def test_noalloc_global_aabb_matches_allocating_entry_point():
    from cascade_amd import _capi

    ctx = _capi.Context(0)
    try:
        L, h, chk = ctx.L, ctx.h, ctx._chk
        rng = np.random.default_rng(7)
        # Large first (fills many partial slots), then small, then one particle.
        for nchunks, n in [(3, 300_000), (2, 1000), (1, 1), (5, 70_001)]:
            lbs, ubs = _boxes(rng, nchunks, n)
            # The smaller input lies strictly inside the range of the first one, so any partial
            # left over from the first call would change the result.
            if n <= 1000:
                lbs *= 0.01
                ubs *= 0.01
            d_lbs, d_ubs = ctx.array(lbs), ctx.array(ubs)
            ref_lb, ref_ub = ctx.empty((nchunks, 4), np.float32), ctx.empty((nchunks, 4), np.float32)
            got_lb, got_ub = ctx.empty((nchunks, 4), np.float32), ctx.empty((nchunks, 4), np.float32)
            chk(L.csc_global_aabb(h, n, nchunks, d_lbs.ptr, d_ubs.ptr, ref_lb.ptr, ref_ub.ptr))
            chk(L.csc_global_aabb_noalloc(h, n, nchunks, d_lbs.ptr, d_ubs.ptr, got_lb.ptr, got_ub.ptr))
            chk(L.csc_sync(h))
            rl, ru, gl, gu = ref_lb.download(), ref_ub.download(), got_lb.download(), got_ub.download()
            assert rl.tobytes() == gl.tobytes() and ru.tobytes() == gu.tobytes(), (nchunks, n)
            assert np.array_equal(gl, lbs.min(axis=1)) and np.array_equal(gu, ubs.max(axis=1))
        # The property the entry point exists for: it allocates nothing. Free device memory (as
        # the HIP runtime reports it) is the same before and after a series of calls, while a first
        # call of the allocating entry point on a fresh size does take memory.
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

        def free_bytes():
            fr, tot = C.c_size_t(), C.c_size_t()
            assert hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
            return fr.value

        lbs, ubs = _boxes(rng, 4, 2_000_000)  # (larger than anything the context has seen so far)
        d_lbs, d_ubs = ctx.array(lbs), ctx.array(ubs)
        got_lb, got_ub = ctx.empty((4, 4), np.float32), ctx.empty((4, 4), np.float32)
        chk(L.csc_sync(h))
        before = free_bytes()
        for _ in range(3):
            chk(L.csc_global_aabb_noalloc(h, 2_000_000, 4, d_lbs.ptr, d_ubs.ptr, got_lb.ptr, got_ub.ptr))
        chk(L.csc_sync(h))
        assert free_bytes() == before, "csc_global_aabb_noalloc took device memory"
        assert np.array_equal(got_lb.download(), lbs.min(axis=1)) and np.array_equal(got_ub.download(), ubs.max(axis=1))
        # Beyond the maximum: a clear error, nothing launched.
        mx = int(L.csc_global_aabb_noalloc_max_chunks())
        assert mx == 32768
        rc = L.csc_global_aabb_noalloc(h, 1, mx + 1, d_lbs.ptr, d_ubs.ptr, got_lb.ptr, got_ub.ptr)
        assert rc == -2 and b"exceed the maximum" in L.csc_last_error(h)
    finally:
        ctx.close()
