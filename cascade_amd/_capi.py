"""ctypes bindings of libcascade_hip.so (include/cascade_hip.h).

Thin by design: device memory is handled through csc_malloc / csc_memcpy_* so that
nothing here depends on torch. The library is required: importing this module does not
load it, but every call does, and a missing library or a machine without a HIP device
raises instead of falling back to anything on the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcascade_hip.so")
_LIB = None


class CascadeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcascade_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


class PropCfg(C.Structure):
    _fields_ = [
        ("dyn_kind", C.c_int32),
        ("order", C.c_uint32),
        ("mu", C.c_double),
        ("rhofac", C.c_double),
        ("init_time_hi", C.c_double),
        ("init_time_lo", C.c_double),
        ("delta_t", C.c_double),
        ("ct", C.c_double),
        ("nchunks", C.c_uint32),
        ("max_substeps", C.c_uint32),
        ("conj_thresh", C.c_double),
        ("has_exit", C.c_int32),
        ("exit_c0", C.c_double),
        ("has_reentry", C.c_int32),
        ("reentry_k", C.c_double * 3),
        ("reentry_c0", C.c_double),
        ("J2", C.c_double),
        ("Re", C.c_double),
        ("dens_alpha", C.c_double * 4),
        ("dens_beta", C.c_double * 4),
        ("dens_gamma", C.c_double * 4),
        ("d_pars", C.c_void_p),
        ("npars", C.c_uint32),
        ("tape", C.c_void_p),
        ("tape_nops", C.c_uint32),
        ("tape_nslots", C.c_uint32),
        ("tape_rhs", C.c_int32 * 7),
    ]


class PropSummary(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("n_nf", C.c_uint32),
        ("nf_idx", C.c_uint32),
        ("nf_time", C.c_double),
        ("n_ste", C.c_uint32),
        ("ste_idx", C.c_uint32),
        ("ste_ev_idx", C.c_uint32),
        ("ste_time", C.c_double),
        ("total_substeps", C.c_uint64),
        ("max_substeps", C.c_uint32),
    ]


BVH_NODE_DTYPE = np.dtype(
    [
        ("begin", "<u4"),
        ("end", "<u4"),
        ("parent", "<i4"),
        ("left", "<i4"),
        ("right", "<i4"),
        ("lb", "<f4", (4,)),
        ("ub", "<f4", (4,)),
        ("nn_level", "<u4"),
        ("split_idx", "<i4"),
    ]
)
CONJ_DTYPE = np.dtype([("i", "<u8"), ("j", "<u8"), ("time", "<f8"), ("dist", "<f8"), ("state_i", "<f8", (6,)), ("state_j", "<f8", (6,))])
COLL_DTYPE = np.dtype([("i", "<u8"), ("j", "<u8"), ("time", "<f8")])

# name -> (restype, argtypes); every symbol declared in include/cascade_hip.h.
_vp, _u64, _u32, _i32, _dbl, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_size_t
SYMBOLS = {
    "csc_ctx_create": (_i32, [_i32, C.POINTER(_vp)]),
    "csc_ctx_destroy": (None, [_vp]),
    "csc_last_error": (C.c_char_p, [_vp]),
    "csc_malloc": (_i32, [_vp, _sz, C.POINTER(_vp)]),
    "csc_free": (_i32, [_vp, _vp]),
    "csc_memcpy_h2d": (_i32, [_vp, _vp, _vp, _sz]),
    "csc_memcpy_d2h": (_i32, [_vp, _vp, _vp, _sz]),
    "csc_memcpy_d2d": (_i32, [_vp, _vp, _vp, _sz]),
    "csc_upload": (_i32, [_vp, _vp, _vp, _sz, C.POINTER(_u64)]),
    "csc_download": (_i32, [_vp, _vp, _vp, _sz, C.POINTER(_u64)]),
    "csc_host_hash": (None, [_vp, _sz, C.POINTER(_u64)]),
    "csc_memset": (_i32, [_vp, _vp, _i32, _sz]),
    "csc_sync": (_i32, [_vp]),
    "csc_timer_begin": (_i32, [_vp]),
    "csc_timer_end": (_i32, [_vp, C.POINTER(C.c_float)]),
    "csc_profile_enable": (_i32, [_vp, _i32]),
    "csc_profile_only": (_i32, [_vp, C.c_char_p]),
    "csc_profile_reset": (_i32, [_vp]),
    "csc_profile_get": (_i32, [_vp, C.c_char_p, _sz]),
    "csc_propagate": (_i32, [_vp, C.POINTER(PropCfg), _u32] + [_vp] * 10),
    "csc_broad_phase_node_pairs": (C.c_uint64, [_vp]),
    "csc_broad_phase_tiles": (C.c_uint64, [_vp]),
    "csc_narrow_phase_view": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "csc_propagate_rows": (_i32, [_vp, C.POINTER(PropCfg), _u32] + [_vp] * 10 + [_u32] + [_vp] * 4),
    "csc_propagate_summary": (_i32, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, C.POINTER(PropSummary)]),
    "csc_exclusive_scan": (_i32, [_vp, _vp, _u64, _vp]),
    "csc_tc_strides": (_i32, [_u32, C.POINTER(_u32), C.POINTER(_u32)]),
    "csc_propagate_store": (_i32, [_vp, C.POINTER(PropCfg), _u32] + [_vp] * 7),
    "csc_global_aabb": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "csc_morton_encode": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp]),
    "csc_check_flags": (_i32, [_vp]),
    "csc_sort_gather": (_i32, [_vp, _u32, _u32] + [_vp] * 9),
    "csc_bvh_build": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp]),
    "csc_bvh_export": (_i32, [_vp, _u32, _vp, _u64, C.POINTER(_u64)]),
    "csc_broad_phase": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "csc_narrow_phase_lazy": (_i32, [_vp, _i32]),
    "csc_narrow_phase_counts": (_i32, [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_i32)]),
    "csc_conj_store_append": (_i32, [_vp, _u64]),
    "csc_conj_store_size": (_u64, [_vp]),
    "csc_conj_store_download": (_i32, [_vp, _vp]),
    "csc_conj_store_reset": (_i32, [_vp, _i32]),
    "csc_broad_phase_filtered": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, C.c_double, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "csc_set_shard": (_i32, [_vp, _u32, _u32]),
    "csc_set_chunk_map": (_i32, [_vp, _u32, _u32]),
    "csc_broad_phase_stats": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "csc_broad_phase_get": (_i32, [_vp, _u32, _vp, _u64, C.POINTER(_u64)]),
    "csc_narrow_phase": (_i32, [_vp, _u32, _u32, _dbl, _dbl, _dbl, _dbl, _dbl, _u32] + [_vp] * 8 + [C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u32)]),
    "csc_propagate_range": (_i32, [_vp, C.POINTER(PropCfg), _u32, _u32, _u32] + [_vp] * 10),
    "csc_propagate_range_rows": (_i32, [_vp, C.POINTER(PropCfg), _u32, _u32, _u32] + [_vp] * 10 + [_u32]),
    "csc_wave_plan_eval": (_i32, [_vp, _u32, _u32, _vp, _i32, _u32, _vp, _vp, C.c_double, _vp, _vp]),
    "csc_wave_emit_source": (_u64, [_vp, _u32, _u32, _vp, _u32, _i32, _vp, _u64, _vp]),
    "csc_wave_would_run": (_i32, [_vp, C.POINTER(PropCfg), _u32]),
    "csc_shard_record_bytes": (_sz, [_u32]),
    "csc_shard_pack": (_i32, [_vp, _u32, _u32, _u32, _u32] + [_vp] * 10),
    "csc_shard_unpack": (_i32, [_vp, _u32, _u32] + [_vp] * 10),
    "csc_shard_block": (_i32, [_u32, _u32, _u32, _vp, _vp]),
    "csc_shard_pack_blocks": (_i32, [_vp, _u32, _u32, _u32, _u32, _u32, _u32] + [_vp] * 10),
    "csc_shard_unpack_block": (_i32, [_vp, _u32, _u32, _u32, _u32] + [_vp] * 10),
    "csc_narrow_phase_prefilter": (_i32, [_vp, _u32, _u32, _dbl, _vp, C.POINTER(_u64)]),
    "csc_candidate_objects": (_i32, [_vp, _u32, _vp, _vp, C.POINTER(_u32), C.POINTER(_u64)]),
    "csc_narrow_phase_finish": (_i32, [_vp, _u32, _u32, _dbl, _dbl, _dbl, _dbl, _dbl, _u32] + [_vp] * 7 + [C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u32)]),
    "csc_narrow_phase_candidates": (_u64, [_vp]),
    "csc_narrow_phase_attempts": (C.c_uint32, [_vp]),
    "csc_global_aabb_noalloc": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "csc_global_aabb_noalloc_max_chunks": (C.c_uint32, []),
    "csc_narrow_phase_pairs": (_i32, [_vp, _u32, _u32, _dbl, _dbl, _dbl, _dbl, _dbl, _u32] + [_vp] * 9 + [C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u32)]),
    "csc_narrow_phase_get": (_i32, [_vp, _vp, _u64, _vp, _u64]),
    "csc_dense_propagate": (_i32, [_vp, _u32, _u32, _dbl] + [_vp] * 6),
    "csc_copy_final_state": (_i32, [_vp, _u32, _vp, _vp]),
    "csc_activity_flags": (_i32, [_vp, _u32, _vp, _dbl, _i32, _vp, _u64, _vp, _u64, _vp]),
    "csc_remove_particles": (_i32, [_vp, _u32, _vp, _u64, _vp, _u32, _vp, _vp]),
}


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


class DeviceArray:
    """A typed device allocation owned by a Context."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        ctx._chk(ctx.L.csc_malloc(ctx.h, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            self.ctx.L.csc_free(self.ctx.h, self.ptr)
            self.ptr = None

    def upload(self, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if self.nbytes:
            self.ctx._chk(self.ctx.L.csc_memcpy_h2d(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes))
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            self.ctx._chk(self.ctx.L.csc_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out


class Context:
    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.csc_ctx_create(device, C.byref(h))
        if rc != 0:
            raise CascadeHipError(rc, self.L.csc_last_error(None).decode())
        self.h = h.value

    @classmethod
    def from_handle(cls, handle):
        """Non-owning wrapper of an existing csc_ctx * (e.g. ``sim._hip_context``)."""
        self = cls.__new__(cls)
        self.L = lib()
        self.h = int(handle)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.csc_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise CascadeHipError(rc, self.L.csc_last_error(self.h).decode())

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def array(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        return DeviceArray(self, a.shape, a.dtype).upload(a)

    def sync(self):
        self._chk(self.L.csc_sync(self.h))

    def profile(self, on=True):
        self._chk(self.L.csc_profile_enable(self.h, 1 if on else 0))

    def profile_only(self, name=None):
        """Time only the kernel `name` (None: all)."""
        self._chk(self.L.csc_profile_only(self.h, name.encode() if name else None))

    def profile_reset(self):
        self._chk(self.L.csc_profile_reset(self.h))

    def profile_get(self):
        """{kernel: (launches, total_ms)} since the last reset."""
        buf = C.create_string_buffer(16384)
        self._chk(self.L.csc_profile_get(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    def timer_begin(self):
        self._chk(self.L.csc_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float()
        self._chk(self.L.csc_timer_end(self.h, C.byref(ms)))
        return ms.value
