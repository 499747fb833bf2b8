"""ctypes front end of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT. Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module (see oracle/orc.hpp for what the oracle restates and what pins it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OUTCOMES = ["success", "time_limit", "collision", "reentry", "exit", "err_nf_state"]

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
assert BVH_NODE_DTYPE.itemsize == 60

# cascade.py/core.cpp:81 (numpy dtype of sim::conjunction)
CONJ_DTYPE = np.dtype(
    [
        ("i", "<u8"),
        ("j", "<u8"),
        ("time", "<f8"),
        ("dist", "<f8"),
        ("state_i", "<f8", (6,)),
        ("state_j", "<f8", (6,)),
    ]
)
assert CONJ_DTYPE.itemsize == 128

COLL_DTYPE = np.dtype([("i", "<u8"), ("j", "<u8"), ("time", "<f8")])


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc.hpp", "orc_sim.hpp", "orc_capi.cpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/detail"):
        ref = os.path.join(_HERE, "_ref", "libref_pin.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, u64, u32, dbl, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_double, C.c_int
        P = C.POINTER
        L.orc_last_error.restype = C.c_char_p
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        L.orc_morton4.restype = u64
        L.orc_morton4.argtypes = [u64] * 4
        L.orc_disc_single_coord.restype = u64
        L.orc_disc_single_coord.argtypes = [C.c_float] * 3
        L.orc_ival_horner.argtypes = [vp, u32, dbl, dbl, vp]
        L.orc_poly_translate_a.argtypes = [vp, vp, dbl, u32]
        L.orc_poly_ssdiff3.argtypes = [vp, vp, u32]
        L.orc_fex_check.restype = u32
        L.orc_fex_check.argtypes = [vp, dbl, u32]
        L.orc_rtscc.restype = u32
        L.orc_rtscc.argtypes = [vp, u32]
        L.orc_root_find.restype = i32
        L.orc_root_find.argtypes = [vp, u32, dbl, i32, vp, i32]
        L.orc_morton_encode.restype = i32
        L.orc_morton_encode.argtypes = [u64, vp, vp, vp, vp, vp]
        L.orc_sort_indices.argtypes = [u64, vp, vp]
        L.orc_bvh_build.restype = C.c_int64
        L.orc_bvh_build.argtypes = [u64, vp, vp, vp, vp, u64]
        L.orc_bvh_verify.restype = i32
        L.orc_bvh_verify.argtypes = [u64, vp, vp, vp, vp, u64]
        L.orc_broad_phase.restype = C.c_int64
        L.orc_broad_phase.argtypes = [u64, vp, vp, vp, vp, vp, vp, vp, u64]
        L.orc_broad_phase_brute.restype = C.c_int64
        L.orc_broad_phase_brute.argtypes = [u64, vp, vp, vp, vp, vp, u64]
        L.orc_sim_set_pars.restype = i32
        L.orc_sim_set_pars.argtypes = [vp, vp, u64, vp]
        L.orc_sim_set_tape.restype = i32
        L.orc_sim_set_tape.argtypes = [vp, vp, u64, C.c_uint32, vp, C.c_uint32]
        L.orc_sim_new.restype = vp
        L.orc_sim_new.argtypes = [vp, u64, dbl, u32, dbl, dbl, dbl, i32, dbl, dbl, dbl, dbl, i32, vp]
        L.orc_sim_free.argtypes = [vp]
        L.orc_sim_step.restype = i32
        L.orc_sim_step.argtypes = [vp]
        L.orc_sim_propagate_until.restype = i32
        L.orc_sim_propagate_until.argtypes = [vp, dbl]
        L.orc_sim_nparts.restype = u64
        L.orc_sim_nparts.argtypes = [vp]
        L.orc_sim_order.restype = u32
        L.orc_sim_order.argtypes = [vp]
        L.orc_sim_get_state.argtypes = [vp, vp]
        L.orc_sim_set_state.restype = i32
        L.orc_sim_set_state.argtypes = [vp, vp, u64]
        L.orc_sim_poke_state.argtypes = [vp, vp]
        L.orc_sim_get_time.restype = dbl
        L.orc_sim_get_time.argtypes = [vp]
        L.orc_sim_get_dtime.argtypes = [vp, vp]
        L.orc_sim_set_time.argtypes = [vp, dbl]
        L.orc_sim_set_ct.argtypes = [vp, dbl]
        L.orc_sim_set_n_par_ct.argtypes = [vp, u32]
        L.orc_sim_set_conj_thresh.argtypes = [vp, dbl]
        L.orc_sim_set_min_coll_radius.argtypes = [vp, dbl]
        L.orc_sim_set_whitelists.argtypes = [vp, vp, u64, vp, u64]
        L.orc_sim_nconj.restype = u64
        L.orc_sim_nconj.argtypes = [vp]
        L.orc_sim_get_conj.argtypes = [vp, vp]
        L.orc_sim_reset_conj.argtypes = [vp]
        L.orc_sim_interrupt_info.argtypes = [vp, vp, vp, vp, vp]
        L.orc_sim_nchunks.restype = u32
        L.orc_sim_nchunks.argtypes = [vp]
        L.orc_sim_delta_t.restype = dbl
        L.orc_sim_delta_t.argtypes = [vp]
        L.orc_sim_get_aabbs.argtypes = [vp, u32, vp, vp]
        L.orc_sim_get_global_aabb.argtypes = [vp, u32, vp, vp]
        L.orc_sim_get_codes.argtypes = [vp, u32, vp, vp]
        L.orc_sim_get_sorted.argtypes = [vp, u32, vp, vp, vp]
        L.orc_sim_bvh_size.restype = u64
        L.orc_sim_bvh_size.argtypes = [vp, u32]
        L.orc_sim_get_bvh.argtypes = [vp, u32, vp]
        L.orc_sim_bp_size.restype = u64
        L.orc_sim_bp_size.argtypes = [vp, u32]
        L.orc_sim_get_bp.argtypes = [vp, u32, vp]
        L.orc_sim_get_active.argtypes = [vp, vp, vp]
        L.orc_sim_get_nsub.restype = u64
        L.orc_sim_get_nsub.argtypes = [vp, vp]
        L.orc_sim_get_tcs.argtypes = [vp, vp, vp]
        L.orc_sim_get_final_state.argtypes = [vp, vp]
        L.orc_sim_ncoll.restype = u64
        L.orc_sim_ncoll.argtypes = [vp]
        L.orc_sim_get_coll.argtypes = [vp, vp]
        L.orc_sim_step_nconj.restype = u64
        L.orc_sim_step_nconj.argtypes = [vp]
        L.orc_sim_get_step_conj.argtypes = [vp, vp]
        L.orc_sim_stage_times.argtypes = [vp, vp]
        L.orc_rhofac.restype = dbl
        L.orc_rhofac.argtypes = [u32]
        L.orc_det_pow.restype = dbl
        L.orc_det_pow.argtypes = [dbl, dbl]
        _LIB = L
    return _LIB


def ref_lib():
    """oracle/_ref/libref_pin.so (the reference's own Morton LUT / ival headers), or None."""
    so = os.path.join(_HERE, "_ref", "libref_pin.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_morton_encode.restype = C.c_uint64
    L.ref_morton_encode.argtypes = [C.c_uint64] * 4
    L.ref_ival_horner.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
    L.ref_ival_mul.argtypes = [C.c_double] * 4 + [C.c_void_p]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


TAPE_DTYPE = np.dtype([("op", "<i4"), ("dst", "<i4"), ("a", "<i4"), ("b", "<i4"), ("c", "<f8")])


class OracleError(ValueError):
    pass


class Sim:
    """Oracle counterpart of cascade.sim (cascade.py/core.cpp:87-330)."""

    def __init__(
        self,
        state,
        ct,
        n_par_ct=1,
        conj_thresh=0.0,
        min_coll_radius=0.0,
        tol=0.0,
        dyn_kind=0,
        mu=1.0,
        J2=0.0,
        Re=1.0,
        exit_radius=0.0,
        reentry_radius=0.0,
        coll_whitelist=(),
        conj_whitelist=(),
        pars=None,
        dens=None,
        tape=None,
    ):
        L = lib()
        st = np.ascontiguousarray(np.asarray(state, dtype=np.float64).reshape(-1, 7))
        if np.isscalar(reentry_radius):
            rk = 1 if reentry_radius > 0 else 0
            re3 = np.array([reentry_radius, 0, 0], dtype=np.float64)
        else:
            rk = 2
            re3 = np.asarray(reentry_radius, dtype=np.float64)
        self._h = L.orc_sim_new(
            _p(st), st.shape[0], ct, n_par_ct, conj_thresh, min_coll_radius, tol, dyn_kind, mu, J2, Re, exit_radius, rk, _p(re3)
        )
        if not self._h:
            raise OracleError(L.orc_last_error().decode())
        self._L = L
        if tape is not None:
            # General dynamics: tape = {"ops": structured array (op, dst, a, b, c), "nslots", "rhs" [7],
            # "npars"} as produced by cascade_amd.core.lower_dynamics (the dynamics definition).
            ops = np.ascontiguousarray(tape["ops"], dtype=TAPE_DTYPE)
            rhs = np.ascontiguousarray(tape["rhs"], dtype=np.int32)
            if L.orc_sim_set_tape(self._h, _p(ops), len(ops), int(tape["nslots"]), _p(rhs), int(tape["npars"])) != 0:
                raise OracleError(L.orc_last_error().decode())
        self._pars = None
        if pars is not None:
            self._pars = np.array(pars, dtype=np.float64).reshape(st.shape[0], -1)
            # dyn_kind bit 1 (drag): pars[:, 0] = BSTAR; dens = the 12 density-fit constants or None
            pr = np.ascontiguousarray(np.asarray(pars, dtype=np.float64).reshape(-1))
            d12 = None if dens is None else np.ascontiguousarray(np.asarray(dens, dtype=np.float64).reshape(12))
            if L.orc_sim_set_pars(self._h, _p(pr), pr.size, None if d12 is None else _p(d12)) != 0:
                raise OracleError(L.orc_last_error().decode())
        self.set_whitelists(coll_whitelist, conj_whitelist)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_sim_free(self._h)
            self._h = None

    def _chk(self, rc):
        if rc == -1000:
            raise OracleError(self._L.orc_last_error().decode())
        return rc

    def step(self):
        return OUTCOMES[self._chk(self._L.orc_sim_step(self._h))]

    def propagate_until(self, t):
        return OUTCOMES[self._chk(self._L.orc_sim_propagate_until(self._h, float(t)))]

    @property
    def nparts(self):
        return int(self._L.orc_sim_nparts(self._h))

    @property
    def order(self):
        return int(self._L.orc_sim_order(self._h))

    @property
    def state(self):
        out = np.empty((self.nparts, 7), dtype=np.float64)
        self._L.orc_sim_get_state(self._h, _p(out))
        return out

    def set_new_state(self, st):
        st = np.ascontiguousarray(np.asarray(st, dtype=np.float64).reshape(-1, 7))
        self._chk(self._L.orc_sim_set_state(self._h, _p(st), st.shape[0]))

    def remove_particles(self, idx):
        """Drops the particles idx (state rows and their runtime parameters), keeping the order
        of the others: the caller pattern of benchmark/2022_leo_pop.cpp:311-337."""
        keep = np.ones(self.nparts, dtype=bool)
        keep[np.asarray(list(idx), dtype=np.int64)] = False
        st = self.state[keep]
        if self._pars is not None:
            self._pars = np.ascontiguousarray(self._pars[keep])
            pr = self._pars.reshape(-1)
            self._chk(self._L.orc_sim_set_pars(self._h, _p(pr), pr.size, None))
        self.set_new_state(st)

    def poke_state(self, st):
        st = np.ascontiguousarray(np.asarray(st, dtype=np.float64).reshape(-1, 7))
        assert st.shape[0] == self.nparts
        self._L.orc_sim_poke_state(self._h, _p(st))

    @property
    def time(self):
        return float(self._L.orc_sim_get_time(self._h))

    @time.setter
    def time(self, t):
        self._L.orc_sim_set_time(self._h, float(t))

    @property
    def dtime(self):
        out = np.empty(2)
        self._L.orc_sim_get_dtime(self._h, _p(out))
        return float(out[0]), float(out[1])

    def set_ct(self, ct):
        self._L.orc_sim_set_ct(self._h, float(ct))

    def set_n_par_ct(self, n):
        self._L.orc_sim_set_n_par_ct(self._h, int(n))

    def set_conj_thresh(self, v):
        self._L.orc_sim_set_conj_thresh(self._h, float(v))

    def set_min_coll_radius(self, v):
        self._L.orc_sim_set_min_coll_radius(self._h, float(v))

    def set_whitelists(self, coll=(), conj=()):
        a = np.asarray(sorted(coll), dtype=np.uint64)
        b = np.asarray(sorted(conj), dtype=np.uint64)
        self._L.orc_sim_set_whitelists(self._h, _p(a), len(a), _p(b), len(b))

    @property
    def conjunctions(self):
        n = int(self._L.orc_sim_nconj(self._h))
        out = np.empty(n, dtype=CONJ_DTYPE)
        if n:
            self._L.orc_sim_get_conj(self._h, _p(out))
        return out

    def reset_conjunctions(self):
        self._L.orc_sim_reset_conj(self._h)

    @property
    def interrupt_info(self):
        k, i, j, t = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_double()
        self._L.orc_sim_interrupt_info(self._h, C.byref(k), C.byref(i), C.byref(j), C.byref(t))
        if k.value == 0:
            return None
        if k.value == 1:
            return (int(i.value), int(j.value))
        if k.value == 2:
            return int(i.value)
        return (int(i.value), float(t.value))

    # ---- stage data of the last step ----
    @property
    def nchunks(self):
        return int(self._L.orc_sim_nchunks(self._h))

    @property
    def delta_t(self):
        return float(self._L.orc_sim_delta_t(self._h))

    def aabbs(self, chunk):
        n = self.nparts
        lb = np.empty((n, 4), dtype=np.float32)
        ub = np.empty((n, 4), dtype=np.float32)
        self._L.orc_sim_get_aabbs(self._h, chunk, _p(lb), _p(ub))
        return lb, ub

    def global_aabb(self, chunk):
        lb = np.empty(4, dtype=np.float32)
        ub = np.empty(4, dtype=np.float32)
        self._L.orc_sim_get_global_aabb(self._h, chunk, _p(lb), _p(ub))
        return lb, ub

    def codes(self, chunk):
        n = self.nparts
        c = np.empty(n, dtype=np.uint64)
        v = np.empty(n, dtype=np.uint32)
        self._L.orc_sim_get_codes(self._h, chunk, _p(c), _p(v))
        return c, v

    def sorted_data(self, chunk):
        n = self.nparts
        c = np.empty(n, dtype=np.uint64)
        lb = np.empty((n, 4), dtype=np.float32)
        ub = np.empty((n, 4), dtype=np.float32)
        self._L.orc_sim_get_sorted(self._h, chunk, _p(c), _p(lb), _p(ub))
        return c, lb, ub

    def bvh(self, chunk):
        n = int(self._L.orc_sim_bvh_size(self._h, chunk))
        out = np.empty(n, dtype=BVH_NODE_DTYPE)
        self._L.orc_sim_get_bvh(self._h, chunk, _p(out))
        return out

    def bp_pairs(self, chunk):
        n = int(self._L.orc_sim_bp_size(self._h, chunk))
        out = np.empty((n, 2), dtype=np.uint32)
        if n:
            self._L.orc_sim_get_bp(self._h, chunk, _p(out))
        return out

    def active_flags(self):
        n = self.nparts
        a = np.empty(n, dtype=np.int8)
        b = np.empty(n, dtype=np.int8)
        self._L.orc_sim_get_active(self._h, _p(a), _p(b))
        return a, b

    def substeps(self):
        """(nsub[N] u32, tcs[total, 7, order+1] f64, tcoords[total, 2] f64)."""
        n = self.nparts
        nsub = np.empty(n, dtype=np.uint32)
        tot = int(self._L.orc_sim_get_nsub(self._h, _p(nsub)))
        tcs = np.empty((tot, 7, self.order + 1), dtype=np.float64)
        tco = np.empty((tot, 2), dtype=np.float64)
        if tot:
            self._L.orc_sim_get_tcs(self._h, _p(tcs), _p(tco))
        return nsub, tcs, tco

    def final_state(self):
        out = np.empty((self.nparts, 6), dtype=np.float64)
        self._L.orc_sim_get_final_state(self._h, _p(out))
        return out

    def collisions(self):
        n = int(self._L.orc_sim_ncoll(self._h))
        out = np.empty(n, dtype=COLL_DTYPE)
        if n:
            self._L.orc_sim_get_coll(self._h, _p(out))
        return out

    def step_conjunctions(self):
        n = int(self._L.orc_sim_step_nconj(self._h))
        out = np.empty(n, dtype=CONJ_DTYPE)
        if n:
            self._L.orc_sim_get_step_conj(self._h, _p(out))
        return out

    def stage_times(self):
        out = np.empty(7)
        self._L.orc_sim_stage_times(self._h, _p(out))
        return dict(zip(["propagate", "morton_sort", "bvh", "broad", "narrow", "total", "n_bp_pairs"], out.tolist()))


def kep_to_cart_test(kep, mu=1.0):
    """test/keputils.hpp:26-54: element order {a, e, i, om, Om, nu}."""
    a, e, i, om, Om, nu = kep
    return _kep_to_cart(a, e, i, om, Om, nu, mu)


def kep_to_cart_bench(kep, mu=1.0):
    """benchmark/random_leo_pop_kep.cpp:31-59: element order {a, e, i, Om, om, nu}."""
    a, e, i, Om, om, nu = kep
    return _kep_to_cart(a, e, i, om, Om, nu, mu)


def _kep_to_cart(a, e, i, om, Om, nu, mu):
    from math import atan, cos, sin, sqrt, tan

    E = 2 * atan(sqrt((1 - e) / (1 + e)) * tan(nu / 2))
    n = sqrt(mu / (a * a * a))
    q = (a * (cos(E) - e), a * sqrt(1 - e * e) * sin(E), 0.0)
    vq = (-n * a * sin(E) / (1 - e * cos(E)), n * a * sqrt(1 - e * e) * cos(E) / (1 - e * cos(E)), 0.0)
    r1 = (cos(Om) * cos(om) - sin(Om) * cos(i) * sin(om), -cos(Om) * sin(om) - sin(Om) * cos(i) * cos(om), sin(Om) * sin(i))
    r2 = (sin(Om) * cos(om) + cos(Om) * cos(i) * sin(om), -sin(Om) * sin(om) + cos(Om) * cos(i) * cos(om), -cos(Om) * sin(i))
    r3 = (sin(i) * sin(om), sin(i) * cos(om), cos(i))
    dot = lambda u, v: u[0] * v[0] + u[1] * v[1] + u[2] * v[2]
    x = (dot(r1, q), dot(r2, q), dot(r3, q))
    v = (dot(r1, vq), dot(r2, vq), dot(r3, vq))
    return x, v
