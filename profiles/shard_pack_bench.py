"""Timing of the multi-GPU exchange's pack / unpack kernels on one GPU through the C ABI
(1M particles, C = 8 chunks, as rank 0 of R processes): python profiles/shard_pack_bench.py [R]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cascade_amd import _capi

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, nc = 1_000_000, 8
ctx = _capi.Context(0)
L, h = ctx.L, ctx.h
rng = np.random.default_rng(0)
arr = dict(nsub=ctx.array(rng.integers(1, 5, n, dtype=np.uint32)), status=ctx.array(np.zeros(n, np.int32)), ev=ctx.array(np.zeros(n, np.uint32)),
           evt=ctx.array(rng.random(n)), nft=ctx.array(rng.random(n)), fin=ctx.array(rng.random((n, 6))),
           lbs=ctx.array(rng.random((nc, n, 4), dtype=np.float32)), ubs=ctx.array(rng.random((nc, n, 4), dtype=np.float32)),
           recs=ctx.array(rng.random((nc, n, 16), dtype=np.float32)))
per = -(-n // R)
cl = -(-nc // R)
send = ctx.empty((per * (80 * R + 96 * nc),), np.uint8)
recv = ctx.empty((per * R * (80 + 96 * cl),), np.uint8)
ptrs = [arr[k].ptr for k in ("nsub", "status", "ev", "evt", "nft", "fin", "lbs", "ubs", "recs")]
ctx.profile(True)
ctx.profile_reset()
for _ in range(5):
    ctx._chk(L.csc_shard_pack_blocks(h, n, 0, per, per, nc, R, *ptrs, send.ptr))
    ctx._chk(L.csc_shard_unpack_block(h, n, nc, 0, cl, recv.ptr, *ptrs))
ctx.sync()
p = ctx.profile_get()
for k, v in p.items():
    print(k, "%.4f ms per call" % (v[1] / v[0]), "(pack: %d MB out; unpack: %d MB in)" % (send.nbytes // 2**20, recv.nbytes // 2**20))
