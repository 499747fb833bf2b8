"""Host-side mirror of ``cascade.sim`` (cascade.py/core.cpp:87-330, include/cascade/sim.hpp:84-513)
driving the gfx950 kernels of libcascade_hip.so through its C ABI.

Same constructor keywords, properties, methods, outcome enum and error texts as the
reference's Python class; the per-superstep pipeline (``step()``, src/sim_propagate.cpp:516-1345)
runs on the GPU. The single intentional API deviation: ``dyn`` is a dynamics descriptor
(``cascade_amd.dynamics.kepler(mu)``) instead of a list of heyoka expressions.

There is no CPU fallback: constructing a ``sim`` needs libcascade_hip.so and a HIP device.
"""
import ctypes as C
import enum
import math

import numpy as np

from . import _capi
from .dynamics import Dynamics, kepler

_EPS = np.finfo(np.float64).eps


class outcome(enum.IntEnum):
    """include/cascade/sim.hpp:82"""

    success = 0
    time_limit = 1
    collision = 2
    reentry = 3
    exit = 4
    err_nf_state = 5


def _two_sum(a, b):
    x = a + b
    z = x - a
    return x, (a - (x - z)) + (b - z)


class _DFloat:
    """heyoka::detail::dfloat<double> (EXTERNAL): normalised hi + lo."""

    __slots__ = ("hi", "lo")

    def __init__(self, hi=0.0, lo=0.0):
        self.hi = float(hi)
        self.lo = float(lo)

    def add(self, other):
        xx, yy = _two_sum(self.hi, other.hi)
        yy += self.lo + other.lo
        h = xx + yy
        return _DFloat(h, yy - (h - xx))

    def add_d(self, d):
        return self.add(_DFloat(d))

    def sub(self, other):
        return self.add(_DFloat(-other.hi, -other.lo))

    def lt(self, other):
        return self.hi < other.hi or (self.hi == other.hi and self.lo < other.lo)

    def eq(self, other):
        return self.hi == other.hi and self.lo == other.lo


def _order_from_tol(tol):
    o = math.ceil(-math.log(tol) / 2 + 1)
    return max(2, int(o))


class sim:
    def __init__(
        self,
        state=None,
        ct=1.0,
        dyn=None,
        reentry_radius=None,
        exit_radius=None,
        pars=None,
        tol=None,
        high_accuracy=False,
        compact_mode=False,
        n_par_ct=1,
        conj_thresh=0.0,
        min_coll_radius=0.0,
        coll_whitelist=None,
        conj_whitelist=None,
        device=0,
    ):
        if state is None:
            state = np.zeros((0, 7))
        st = np.array(state, dtype=np.float64, order="C")
        # cascade.py/core.cpp:94-104
        if st.ndim != 2:
            raise ValueError("The input state must have 2 dimensions, but instead an array with {} dimension(s) was provided".format(st.ndim))
        if st.shape[1] != 7:
            raise ValueError("An input state with 7 columns is expected, but the number of columns is instead {}".format(st.shape[1]))
        self._state = st
        if not math.isfinite(ct) or ct <= 0:
            raise ValueError("The collisional timestep must be finite and positive, but it is {} instead".format(ct))
        self._ct = float(ct)
        self.n_par_ct = n_par_ct
        self.conj_thresh = conj_thresh
        self.min_coll_radius = min_coll_radius
        self._coll_whitelist = set(coll_whitelist or ())
        self._conj_whitelist = set(conj_whitelist or ())

        if dyn is None:
            dyn = kepler()
        if not isinstance(dyn, Dynamics):
            raise TypeError("The 'dyn' argument must be a dynamics descriptor (see cascade_amd.dynamics), heyoka expressions are not supported")
        self._dyn = dyn
        self._npars = dyn.npars
        self._pars = self._validate_pars(pars, st.shape[0])

        # src/sim_basic.cpp:471-501
        if reentry_radius is None:
            reentry_radius = 0.0
        if np.ndim(reentry_radius) == 0:
            rr = float(reentry_radius)
            if not math.isfinite(rr) or rr < 0:
                raise ValueError("The reentry radius must be finite and non-negative, but it is {} instead".format(rr))
            self._reentry_radius = rr
        else:
            rr = [float(v) for v in reentry_radius]
            if len(rr) != 3:
                raise ValueError(
                    "The reentry_radius argument must be either a scalar (for a spherical central body) "
                    "or a vector of 3 elements (for a triaxial ellipsoid), but instead it is a vector of {} element(s)".format(len(rr))
                )
            if any((not math.isfinite(v)) or v <= 0 for v in rr):
                raise ValueError("A non-finite or non-positive value was detected among the 3 semiaxes of the central body: {}".format(rr))
            self._reentry_radius = rr
        er = 0.0 if exit_radius is None else float(exit_radius)
        if not math.isfinite(er) or er < 0:
            raise ValueError("The exit radius must be finite and non-negative, but it is {} instead".format(er))
        self._exit_radius = er

        self._tol = float(_EPS if not tol else tol)
        if not math.isfinite(self._tol) or self._tol <= 0:
            raise ValueError("The tolerance must be finite and positive, but it is {} instead".format(self._tol))
        self._order = _order_from_tol(self._tol)
        self._high_accuracy = bool(high_accuracy)
        self._compact_mode = bool(compact_mode)

        self._verify_state_vector(self._state)

        self._time = _DFloat(0.0)
        self._int_info = None
        self._conj = np.empty(0, dtype=_capi.CONJ_DTYPE)
        self._conj.flags.writeable = False

        # Device side. Fails loudly without the library / a HIP device.
        self._ctx = _capi.Context(device)
        if self._order != 20:
            raise NotImplementedError("only tol = DBL_EPSILON (Taylor order 20) is built into the HIP kernels, tol={} needs order {}".format(self._tol, self._order))
        self._dev = {}
        self.stage_ms = {}
        self.last_bp_pairs = 0
        self.host_sync = True
        self.np_prefilter = True  # geometric pre-filter in front of the polynomial narrow phase

    # ---- validation helpers ---------------------------------------------------------
    def _validate_pars(self, pars, nparts):
        # src/sim_basic.cpp:252-278
        if self._npars == 0:
            if pars is not None and np.size(pars) != 0:
                raise ValueError("The input array of parameter values must be empty when the number of parameters in the dynamics is zero")
            return np.zeros((nparts, 0))
        if pars is None or np.size(pars) == 0:
            return np.zeros((nparts, self._npars))
        p = np.array(pars, dtype=np.float64, order="C")
        if p.size % self._npars != 0 or p.size // self._npars != nparts:
            raise ValueError("The input array of parameter values must have shape ({}, {}), but instead its flattened size is {}".format(nparts, self._npars, p.size))
        return p.reshape(nparts, self._npars)

    def _with_reentry(self):
        return isinstance(self._reentry_radius, list) or self._reentry_radius > 0

    def _with_exit(self):
        return self._exit_radius > 0

    def _verify_state_vector(self, st):
        # src/sim_basic.cpp:686-760
        if st.shape[0] == 0:
            return
        pos = st[:, :3]
        bad = ~np.all(np.isfinite(pos), axis=1)
        if bad.any():
            raise ValueError("An non-finite position was detected in the state vector of the particle at index {}".format(int(np.argmax(bad))))
        x, y, z = st[:, 0], st[:, 1], st[:, 2]
        if self._with_reentry():
            if isinstance(self._reentry_radius, list):
                a, b, c = self._reentry_radius
                inside = x * x / (a * a) + y * y / (b * b) + z * z / (c * c) < 1
                if inside.any():
                    raise ValueError("The particle at index {} is inside the ellipsoidal central body".format(int(np.argmax(inside))))
            else:
                inside = x * x + y * y + z * z < self._reentry_radius * self._reentry_radius
                if inside.any():
                    raise ValueError("The particle at index {} is inside the spherical central body".format(int(np.argmax(inside))))
        if self._with_exit():
            outside = x * x + y * y + z * z >= self._exit_radius * self._exit_radius
            if outside.any():
                raise ValueError("The particle at index {} is outside the exit radius {}".format(int(np.argmax(outside)), self._exit_radius))
        bad = ~np.all(np.isfinite(st[:, 3:6]), axis=1)
        if bad.any():
            raise ValueError("An non-finite velocity was detected in the state vector of the particle at index {}".format(int(np.argmax(bad))))
        bad = ~np.isfinite(st[:, 6]) | (st[:, 6] < 0)
        if bad.any():
            k = int(np.argmax(bad))
            raise ValueError("An invalid particle size of {:g} was detected for the particle at index {}".format(st[k, 6], k))

    # ---- properties (cascade.py/core.cpp:164-321) -----------------------------------
    @property
    def interrupt_info(self):
        return self._int_info

    @property
    def time(self):
        return self._time.hi

    @time.setter
    def time(self, t):
        if not math.isfinite(t):
            raise ValueError("Cannot set the simulation time to the non-finite value {}".format(t))
        self._time = _DFloat(float(t))

    @property
    def ct(self):
        return self._ct

    @ct.setter
    def ct(self, ct):
        if not math.isfinite(ct) or ct <= 0:
            raise ValueError("The collisional timestep must be finite and positive, but it is {} instead".format(ct))
        self._ct = float(ct)

    @property
    def n_par_ct(self):
        return self._n_par_ct

    @n_par_ct.setter
    def n_par_ct(self, n):
        if int(n) == 0:
            raise ValueError("The number of collisional timesteps to be processed in parallel cannot be zero")
        if int(n) < 0:
            raise ValueError("The number of collisional timesteps to be processed in parallel cannot be negative")
        self._n_par_ct = int(n)

    @property
    def conj_thresh(self):
        return self._conj_thresh

    @conj_thresh.setter
    def conj_thresh(self, v):
        if not math.isfinite(v) or v < 0:
            raise ValueError("The conjunction threshold value {} is invalid: it must be finite and non-negative".format(v))
        self._conj_thresh = float(v)

    @property
    def min_coll_radius(self):
        return self._min_coll_radius

    @min_coll_radius.setter
    def min_coll_radius(self, v):
        if math.isnan(v) or v < 0:
            raise ValueError("The minimum collisional radius cannot be NaN or negative, but the invalid value {} was provided".format(v))
        self._min_coll_radius = float(v)

    @property
    def coll_whitelist(self):
        return set(self._coll_whitelist)

    @coll_whitelist.setter
    def coll_whitelist(self, wl):
        self._coll_whitelist = set(wl)

    @property
    def conj_whitelist(self):
        return set(self._conj_whitelist)

    @conj_whitelist.setter
    def conj_whitelist(self, wl):
        self._conj_whitelist = set(wl)

    nparts = property(lambda self: self._state.shape[0])
    npars = property(lambda self: self._npars)
    tol = property(lambda self: self._tol)
    high_accuracy = property(lambda self: self._high_accuracy)
    compact_mode = property(lambda self: self._compact_mode)
    reentry_radius = property(lambda self: self._reentry_radius)
    exit_radius = property(lambda self: self._exit_radius)

    @property
    def state(self):
        """Writable view on the internal state (cascade.py/core.cpp:199-230): stays valid
        (detached) after set_new_state_pars() / remove_particles()."""
        return self._state

    @property
    def pars(self):
        return self._pars

    @property
    def conjunctions(self):
        return self._conj

    def reset_conjunctions(self):
        self._conj = np.empty(0, dtype=_capi.CONJ_DTYPE)
        self._conj.flags.writeable = False

    def set_new_state_pars(self, new_state, new_pars=None):
        st = np.array(new_state, dtype=np.float64, order="C")
        if st.ndim != 2:
            raise ValueError("The input state must have 2 dimensions, but instead an array with {} dimension(s) was provided".format(st.ndim))
        if st.shape[1] != 7:
            raise ValueError("An input state with 7 columns is expected, but the number of columns is instead {}".format(st.shape[1]))
        self._verify_state_vector(st)
        p = self._validate_pars(new_pars, st.shape[0])
        self._state = st
        self._pars = p

    def remove_particles(self, idxs):
        # src/sim_basic.cpp:280-338
        idx = sorted(set(int(i) for i in idxs))
        n = self.nparts
        if idx and (idx[0] < 0 or idx[-1] >= n):
            raise ValueError("An invalid vector of indices was passed to the function for particle removal: {}".format(sorted(int(i) for i in idxs)))
        keep = np.ones(n, dtype=bool)
        keep[idx] = False
        self._state = np.ascontiguousarray(self._state[keep])
        self._pars = np.ascontiguousarray(self._pars[keep])

    def __repr__(self):
        return "Total number of particles: {}\nCollisional timestep     : {}\n".format(self.nparts, self._ct)

    # ---- device plumbing ------------------------------------------------------------
    def _buf(self, name, shape, dtype):
        """Grow-only device buffer (like resize_if_needed, src/sim_propagate.cpp:561-569)."""
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        b = self._dev.get(name)
        if b is None or b.nbytes < nbytes:
            self._dev[name] = None
            b = self._ctx.empty((max(nbytes + nbytes // 4, 256),), np.uint8)
            self._dev[name] = b
        return b.ptr

    def _cfg(self, nchunks, delta_t):
        cfg = _capi.PropCfg()
        cfg.dyn_kind = self._dyn.kind
        cfg.order = self._order
        cfg.mu = self._dyn.mu
        cfg.rhofac = math.exp(-0.7 / (self._order - 1)) / (math.exp(1.0) * math.exp(1.0))
        cfg.init_time_hi = self._time.hi
        cfg.init_time_lo = self._time.lo
        cfg.delta_t = delta_t
        cfg.ct = self._ct
        cfg.nchunks = nchunks
        cfg.max_substeps = 0
        cfg.conj_thresh = self._conj_thresh
        cfg.has_exit = 1 if self._with_exit() else 0
        cfg.exit_c0 = self._exit_radius * self._exit_radius
        cfg.has_reentry = 1 if self._with_reentry() else 0
        if isinstance(self._reentry_radius, list):
            a, b, c = self._reentry_radius
            cfg.reentry_k[0] = (b * c) * (b * c)
            cfg.reentry_k[1] = (a * c) * (a * c)
            cfg.reentry_k[2] = (a * b) * (a * b)
            cfg.reentry_c0 = a * a * b * b * c * c
        else:
            cfg.reentry_k[0] = cfg.reentry_k[1] = cfg.reentry_k[2] = 1.0
            cfg.reentry_c0 = self._reentry_radius * self._reentry_radius
        return cfg

    def _tc_strides(self):
        xs, vs = C.c_uint32(), C.c_uint32()
        self._ctx._chk(self._ctx.L.csc_tc_strides(self._order, C.byref(xs), C.byref(vs)))
        return int(xs.value), int(vs.value)

    def upload_state(self):
        """Host -> device copy of the state (the host arrays are the source of truth at
        every step() entry, cascade.py/core.cpp:199-230)."""
        n = self.nparts
        L, h = self._ctx.L, self._ctx.h
        d_state = self._buf("state", (n, 7), np.float64)
        st = np.ascontiguousarray(self._state)
        self._ctx._chk(L.csc_memcpy_h2d(h, d_state, st.ctypes.data_as(C.c_void_p), st.nbytes))

    def download_state(self):
        n = self.nparts
        L, h = self._ctx.L, self._ctx.h
        d_state = self._buf("state", (n, 7), np.float64)
        tmp = np.empty((n, 7))
        self._ctx._chk(L.csc_memcpy_d2h(h, tmp.ctypes.data_as(C.c_void_p), d_state, tmp.nbytes))
        self._state[:, :6] = tmp[:, :6]

    # ---- step(): src/sim_propagate.cpp:516-1345 --------------------------------------
    def step(self):
        n = self.nparts
        if n == 0:
            raise ValueError("Cannot integrate a simulation with no particles")
        nchunks0 = self._n_par_ct
        delta_t0 = self._ct * nchunks0
        if not math.isfinite(delta_t0) or delta_t0 <= 0:
            raise ValueError("An invalid superstep size of {} was inferred".format(delta_t0))
        L, h, ctx = self._ctx.L, self._ctx.h, self._ctx
        chk = ctx._chk
        u8, u32, i32, u64, f32, f64 = np.uint8, np.uint32, np.int32, np.uint64, np.float32, np.float64
        timing = self.stage_ms is not None

        if self.host_sync:
            # src/sim_propagate.cpp:250-253 (the message of test/invalid_state.cpp:48-49)
            sz = self._state[:, 6]
            bad = ~np.isfinite(sz) | (sz < 0)
            if bad.any():
                k = int(np.argmax(bad))
                raise ValueError("An invalid particle size of {:g} was detected for the particle at index {}".format(sz[k], k))
            self.upload_state()

        d_state = self._buf("state", (n, 7), f64)
        d_active = self._buf("active", (n,), u8)
        d_nsub = self._buf("nsub", (n,), u32)
        d_status = self._buf("status", (n,), i32)
        d_ev_idx = self._buf("ev_idx", (n,), u32)
        d_ev_time = self._buf("ev_time", (n,), f64)
        d_nf_t = self._buf("nf_last_t", (n,), f64)
        d_final = self._buf("final_state", (n, 6), f64)
        d_lbs = self._buf("lbs", (nchunks0, n, 4), f32)
        d_ubs = self._buf("ubs", (nchunks0, n, 4), f32)
        d_glb = self._buf("glb", (nchunks0, 4), f32)
        d_gub = self._buf("gub", (nchunks0, 4), f32)
        d_codes = self._buf("codes", (nchunks0, n), u64)
        d_vidx = self._buf("vidx", (nchunks0, n), u32)
        d_scodes = self._buf("srt_codes", (nchunks0, n), u64)
        d_slbs = self._buf("srt_lbs", (nchunks0, n, 4), f32)
        d_subs = self._buf("srt_ubs", (nchunks0, n, 4), f32)
        d_tc_row = self._buf("tc_row", (n,), u64)
        # Sampled-trajectory records for the narrow phase's geometric pre-filter.
        d_recs = self._buf("recs", (nchunks0, n, 16), f32) if self.np_prefilter else None
        d_srecs = self._buf("srt_recs", (nchunks0, n, 16), f32) if self.np_prefilter else None

        ms = {}

        def lap(name):
            if timing:
                ms[name] = ms.get(name, 0.0) + ctx.timer_end()
                ctx.timer_begin()

        if timing:
            ctx.timer_begin()

        with_conj = self._conj_thresh != 0
        cw = np.array(sorted(self._coll_whitelist), dtype=u64)
        jw = np.array(sorted(self._conj_whitelist), dtype=u64)
        chk(L.csc_activity_flags(h, n, d_state, self._min_coll_radius, 1 if with_conj else 0, cw.ctypes.data_as(C.c_void_p), len(cw), jw.ctypes.data_as(C.c_void_p), len(jw), d_active))

        cfg = self._cfg(nchunks0, delta_t0)
        chk(L.csc_propagate(h, C.byref(cfg), n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs, d_recs))
        summ = _capi.PropSummary()
        chk(L.csc_propagate_summary(h, n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, C.byref(summ)))
        lap("propagate")

        if summ.n_nf:
            # src/sim_propagate.cpp:1127-1145: nothing user-visible changes.
            self._int_info = (int(summ.nf_idx), float(summ.nf_time))
            self.stage_ms = ms if timing else None
            return outcome.err_nf_state
        if summ.flags & 0x100:
            raise ValueError("The computation of the bounding box for a particle produced a non-finite bound")
        if summ.flags & 0x200:
            raise ValueError("A non-finite time coordinate was generated during the numerical integration of a particle")
        if summ.flags & 0x400:
            raise OverflowError("Overflow detected during the numerical integration of a particle (substep cap reached)")

        nchunks, delta_t = nchunks0, delta_t0
        if summ.n_ste:
            # :1148-1171
            delta_t = min(float(summ.ste_time), delta_t0)
            new_nchunks = int(math.ceil(delta_t / self._ct))
            if new_nchunks == 0:
                raise ValueError(
                    "The recomputed number of chunks after the triggering of a stopping terminal "
                    "event cannot be zero (this likely indicates that the simulation was restarted "
                    "without resolving a stopping terminal event)"
                )
            nchunks = new_nchunks

        chk(L.csc_global_aabb(h, n, nchunks, d_lbs, d_ubs, d_glb, d_gub))
        chk(L.csc_morton_encode(h, n, nchunks, d_lbs, d_ubs, d_glb, d_gub, d_codes))
        chk(L.csc_sort_gather(h, n, nchunks, d_codes, d_lbs, d_ubs, d_vidx, d_scodes, d_slbs, d_subs, d_recs, d_srecs))
        lap("morton_sort")

        # Taylor coefficients for dense output: CSR rows in the Morton order of chunk 0, so
        # that the rows the narrow phase gathers for neighbouring pairs are neighbours in HBM.
        tot = int(summ.total_substeps)
        self.last_total_substeps = tot
        xs, vs = self._tc_strides()
        d_txyz = self._buf("tcs_xyz", (max(tot, 1), xs), f64)
        d_tvr = self._buf("tcs_vr", (max(tot, 1), vs), f64)
        d_tco = self._buf("tcoords", (max(tot, 1), 2), f64)
        chk(L.csc_propagate_store(h, C.byref(cfg), n, d_state, d_vidx, d_nsub, d_tc_row, d_txyz, d_tvr, d_tco))
        lap("store_tcs")
        chk(L.csc_bvh_build(h, n, nchunks, d_scodes, d_slbs, d_subs))
        lap("bvh")
        npairs = C.c_uint64()
        chk(L.csc_broad_phase(h, n, nchunks, d_slbs, d_subs, d_vidx, d_active, C.byref(npairs)))
        rc = L.csc_check_flags(h)
        if rc == -4:
            raise ValueError("A global bounding box with non-finite boundaries and/or size was generated")
        chk(rc)
        self.last_bp_pairs = int(npairs.value)
        lap("broad")
        ncoll, nconj, warn = C.c_uint64(), C.c_uint64(), C.c_uint32()
        chk(
            L.csc_narrow_phase(
                h, n, nchunks, self._ct, delta_t, self._time.hi, self._time.lo, self._conj_thresh, self._order, d_state, d_active, d_tc_row, d_nsub, d_txyz, d_tvr, d_tco, d_srecs, C.byref(ncoll), C.byref(nconj), C.byref(warn)
            )
        )
        coll = np.empty(int(ncoll.value), dtype=_capi.COLL_DTYPE)
        conj = np.empty(int(nconj.value), dtype=_capi.CONJ_DTYPE)
        chk(L.csc_narrow_phase_get(h, coll.ctypes.data_as(C.c_void_p), len(coll), conj.ctypes.data_as(C.c_void_p), len(conj)))
        self.last_collisions = coll
        self.last_np_candidates = int(L.csc_narrow_phase_candidates(h))
        self.last_step_conjunctions = conj
        lap("narrow")

        # Outcome resolution (:1189-1345). coll is sorted by (time, i, j).
        oc = outcome.success
        interrupt_time = None
        if len(coll):
            interrupt_time = float(coll["time"][0])
            oc = outcome.collision
        if summ.n_ste:
            if interrupt_time is None or float(summ.ste_time) < interrupt_time:
                interrupt_time = float(summ.ste_time)
                if self._with_exit() and summ.ste_ev_idx == 0:
                    oc = outcome.exit
                else:
                    oc = outcome.reentry

        if oc == outcome.success:
            self._time = self._time.add_d(delta_t0)
            chk(L.csc_copy_final_state(h, n, d_final, d_state))
            if self.host_sync:
                self.download_state()
            self._int_info = None
            if with_conj:
                self._append_conj(conj)
        else:
            chk(L.csc_dense_propagate(h, n, self._order, interrupt_time, d_tc_row, d_nsub, d_txyz, d_tvr, d_tco, d_final))
            rc = L.csc_check_flags(h)
            if rc == -7:
                raise ValueError("The computation of dense_propagate() could not be performed because no timesteps have been taken for a particle")
            chk(rc)
            self._time = self._time.add_d(interrupt_time)
            chk(L.csc_copy_final_state(h, n, d_final, d_state))
            if self.host_sync:
                self.download_state()
            if oc == outcome.collision:
                self._int_info = (int(coll["i"][0]), int(coll["j"][0]))
            else:
                self._int_info = int(summ.ste_idx)
            if with_conj:
                first = self._append_conj(conj)
                new = self._conj[first:]
                k = int(np.searchsorted(new["time"], self._time.hi, side="left"))
                self._set_conj(self._conj[: first + k])
        lap("finish")
        if timing:
            self.stage_ms = ms
        self._last = (nchunks, delta_t, delta_t0)
        return oc

    def _set_conj(self, arr):
        # Always a NEW array (src/sim_narrow_phase.cpp:1000-1011): old views stay valid.
        a = np.array(arr, dtype=_capi.CONJ_DTYPE)
        a.flags.writeable = False
        self._conj = a

    def _append_conj(self, new):
        # append_conj_data (src/sim_propagate.cpp:1644-1701): new records sorted by time.
        first = len(self._conj)
        self._set_conj(np.concatenate([self._conj, new]))
        return first

    def bp_nodes_visited(self):
        """Measurement aid: BVH nodes tested by the broad phase of the last step()."""
        n = self.nparts
        nchunks = self._last[0]
        d = self._dev
        v = C.c_uint64()
        self._ctx._chk(self._ctx.L.csc_broad_phase_stats(self._ctx.h, n, nchunks, d["srt_lbs"].ptr, d["srt_ubs"].ptr, d["vidx"].ptr, d["active"].ptr, C.byref(v)))
        return int(v.value)

    # ---- dense output / propagate_until (:1386-1603) ----------------------------------
    def _dense_to(self, t_rel):
        n = self.nparts
        L, h, chk = self._ctx.L, self._ctx.h, self._ctx._chk
        d_state = self._buf("state", (n, 7), np.float64)
        d_final = self._buf("final_state", (n, 6), np.float64)
        d = self._dev
        chk(L.csc_dense_propagate(h, n, self._order, t_rel, d["tc_row"].ptr, d["nsub"].ptr, d["tcs_xyz"].ptr, d["tcs_vr"].ptr, d["tcoords"].ptr, d_final))
        rc = L.csc_check_flags(h)
        if rc == -7:
            raise ValueError("The computation of dense_propagate() could not be performed because no timesteps have been taken for a particle")
        chk(rc)
        chk(L.csc_copy_final_state(h, n, d_final, d_state))
        if self.host_sync:
            self.download_state()

    def propagate_until(self, t):
        if not math.isfinite(t) or _DFloat(t).lt(self._time):
            raise ValueError(
                "The final time passed to the propagate_until() function must be finite and not less than "
                "the current simulation time, but a value of {} was provided instead".format(t)
            )
        final_t = _DFloat(t)
        if final_t.eq(self._time):
            return outcome.time_limit
        while True:
            orig_t = self._time
            cur_oc = self.step()
            if cur_oc == outcome.success:
                if not self._time.lt(final_t):
                    self._dense_to(final_t.sub(orig_t).hi)
                    self._time = final_t
                    return outcome.time_limit
            elif cur_oc != outcome.err_nf_state:
                if final_t.lt(self._time):
                    self._dense_to(final_t.sub(orig_t).hi)
                    self._time = final_t
                    self._int_info = None
                    return outcome.time_limit
                return cur_oc
            else:
                return cur_oc
