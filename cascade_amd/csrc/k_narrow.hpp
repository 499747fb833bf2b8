// cascade_amd/csrc/k_narrow.hpp
//
// K7: narrow phase, two kernels.
//   k_np_fex   one lane per broad-phase pair; for every overlapping pair of substeps inside
//              the chunk the lane translates the x/y/z Taylor polynomials to a common origin,
//              builds the distance-square polynomial in registers and runs the fast exclusion
//              checks. Intervals that are NOT excluded (rare) are appended to a survivor list.
//   k_np_roots one lane per survivor: rebuilds the same polynomial (same operations, same
//              bits) and runs root isolation + TOMS748 refinement, recording collisions and
//              conjunctions. Keeping this divergent, register-hungry code out of the first
//              kernel is what lets the hot kernel run at a useful occupancy.
// K8: dense output (Horner of the 6 coordinates).
//
// Replaces:
//   src/sim_narrow_phase.cpp:495-1023  narrow_phase_parallel
//   src/sim_narrow_phase.cpp:208-422   run_poly_root_finding (slow path in dev_math.hpp)
//   src/sim_jit.cpp:64-113, 118-218    pta_cfunc / ssdiff3_cfunc (register versions here)
//   src/sim_propagate.cpp:1386-1503    dense_propagate
// Traffic per pair-interval: 2 x 512 B of x/y/z Taylor coefficients (4 aligned cache lines
// each; the vx/vy/vz/r rows are only read for recorded conjunctions), 2 x 16 B substep
// times, 16 B pair record; ~1.5 kFLOP. L2/HBM bound: rows are stored in Morton order, so the
// rows gathered by one wave (64 pairs of neighbouring queries) are neighbours in memory.
#pragma once

#include "dev_math.hpp"
#include "k_bvh.hpp"
#include "k_chord.hpp"
#include "k_propagate.hpp"

namespace csc
{

struct coll_rec {
    std::uint64_t i, j;
    double time;
};

// include/cascade/sim.hpp:90-104
struct conj_rec {
    std::uint64_t i, j;
    double time, dist;
    double state_i[6], state_j[6];
};
static_assert(sizeof(conj_rec) == 128, "conjunction layout");

constexpr int NP_FLAT_SHARDS = 32; // = DT_SHARDS of k_bp_dual.hpp
// k_np_prefilter_flat: 256 threads, 4 pairs each, workgroups looping over the tiles.
constexpr unsigned NP_FLAT_BLOCK = 256, NP_FLAT_PPT = 4;
constexpr unsigned NP_FLAT_TILE = NP_FLAT_BLOCK * NP_FLAT_PPT; // pairs per workgroup

struct np_params {
    std::uint32_t n;       // particles
    std::uint32_t nchunks; // chunks (after a possible STE shrink)
    // The chunks may be a block of a longer superstep (chunks dealt to several GPUs): chunk c here
    // is chunk chunk_first + c of nchunks_global, which is what its time window follows.
    std::uint32_t chunk_first, nchunks_global;
    double ct, delta_t;    // delta_t after a possible STE shrink (src/sim_narrow_phase.cpp:560)
    double init_hi, init_lo;
    double conj_thresh2;
    const double *state;          // [n][7] (sizes)
    const std::uint8_t *active;   // [n] bit0 coll, bit1 conj
    const std::uint64_t *tc_row;  // [n] first CSR row of particle i
    const std::uint32_t *nsub;    // [n] number of rows of particle i
    const double *tcs_xyz;        // CSR rows of tc_xyz_stride(P)
    const double *tcs_vr;         // CSR rows of tc_vr_stride(P)
    const double *tcoords;        // CSR rows of 2
    // broad-phase output (sorted positions within the chunk), see k_bvh.hpp
    const uint2 *order;           // [norder] {pair block, time chunk}, group-major (k_pl_order)
    std::uint32_t norder;
    const std::uint32_t *pb_count; // pairs in each block
    const uint2 *pairs;           // [block][PB_SIZE] {pos_i, pos_j}
    // ... or the flat pair lists of the dual traversal (k_bp_dual.hpp): [NP_FLAT_SHARDS][flat_cap]
    const unsigned long long *flat_pairs;
    std::uint32_t flat_cap;
    std::uint32_t flat_cnt[NP_FLAT_SHARDS];          // pairs of each list
    std::uint32_t flat_tile_off[NP_FLAT_SHARDS + 1]; // first tile (= workgroup, NP_FLAT_TILE pairs) of each list
    std::uint32_t flat_xcd_run;                      // consecutive tiles handled by the same XCD (0: dispatch order)
    const std::uint32_t *vidx;    // [C][n] sorted position -> original index
    const float4 *srt_recs;       // [C][n][4] sampled-trajectory records in sorted order
    double thr_extra;             // conjunction threshold (0 when conjunctions are not tracked)
    int conj_fast;                // root searches with the polynomials in registers (np_slow_path_regs)
    // candidates: pairs that survive the pre-filter, {pi, pj, chunk, 0} with ORIGINAL indices
    uint4 *cand;
    std::uint32_t cand_cap;
    coll_rec *coll;
    conj_rec *conj;
    std::uint32_t coll_cap, conj_cap;
    uint4 *surv;             // [surv_cap][2] {pi, pj, chunk, flags} {it_i, it_j, 0, 0}; flags bit0 coll, bit1 conj
    std::uint32_t surv_cap;
    uint4 *defer;            // [defer_cap][2] windows that need a polynomial translation (same record, flags 0)
    std::uint32_t defer_cap;
    std::uint32_t *counters; // [0] n_coll, [1] n_conj, [2] warnings, [3] n survivors, [4] n deferred windows, [5] n candidates
};

CSC_DEV dfl load_tcoord(const double *tcoords, std::uint64_t row)
{
    return dfl{tcoords[row * 2u], tcoords[row * 2u + 1u]};
}

// Substep range [first, last) covering a chunk (src/sim_propagate.cpp:275-287).
CSC_DEV void substep_range(const double *tcoords, std::uint64_t off, std::uint32_t ns, dfl cb, dfl ce,
                           std::uint32_t &first, std::uint32_t &last)
{
    // upper_bound(cb): first s with cb < tcoords[s]
    std::uint32_t lo = 0, len = ns;
    while (len > 0) {
        const std::uint32_t half = len / 2, mid = lo + half;
        if (!df_lt(cb, load_tcoord(tcoords, off + mid))) {
            lo = mid + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    first = lo;
    // lower_bound(ce) from first: first s with !(tcoords[s] < ce)
    len = ns - lo;
    while (len > 0) {
        const std::uint32_t half = len / 2, mid = lo + half;
        if (df_lt(load_tcoord(tcoords, off + mid), ce)) {
            lo = mid + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    last = lo + (lo != ns ? 1u : 0u);
}

// out(x) = cf(x + a) in registers; same operation order as poly_translate_a (dev_math.hpp).
template <int P>
CSC_DEV void translate_a_regs(double *out, const double *cf, double a)
{
    double ap[P + 1];
    ap[0] = 1.;
    ap[1] = a;
#pragma unroll
    for (int i = 2; i <= P; ++i) {
        if (i % 2 == 0) {
            ap[i] = ap[i / 2] * ap[i / 2];
        } else {
            ap[i] = ap[i / 2 + 1] * ap[i / 2];
        }
    }
    constexpr binom_tab bt = make_binom_tab();
#pragma unroll
    for (int i = 0; i <= P; ++i) {
        double acc = 0;
#pragma unroll
        for (int k = i; k <= P; ++k) {
            acc += (cf[k] * bt.v[k * BINOM_N + (k - i)]) * ap[k - i];
        }
        out[i] = acc;
    }
}

// Coefficient i of p*p as written in src/sim_jit.cpp:175-202 (plain mul/add, no fma).
template <int I>
CSC_DEV double psquare(const double *p)
{
    if constexpr (I == 0) {
        return p[0] * p[0];
    } else if constexpr (I % 2 == 0) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j <= I / 2 - 1; ++j) {
            acc += p[I - j] * p[j];
        }
        return 2. * acc + p[I / 2] * p[I / 2];
    } else {
        double acc = 0;
#pragma unroll
        for (int j = 0; j <= (I - 1) / 2; ++j) {
            acc += p[I - j] * p[j];
        }
        return 2. * acc;
    }
}

template <int P, int I>
struct sq_all {
    static CSC_DEV void run(double *out, const double *d)
    {
        out[I] = psquare<I>(d);
        if constexpr (I < P) {
            sq_all<P, I + 1>::run(out, d);
        }
    }
};

// Loads coordinate c (0..2) of row `row`, translated by delta when delta != 0. The row is
// 16-byte aligned: 128-bit loads, indices resolved at compile time once c is unrolled.
template <int P, bool TR = true>
CSC_DEV void load_poly(double *out, const double *xyz, std::uint64_t row, int c, double delta)
{
    const double2 *src = reinterpret_cast<const double2 *>(xyz + row * static_cast<std::uint64_t>(tc_xyz_stride(P)));
    double tmp[P + 1];
    const int e0 = c * (P + 1);
#pragma unroll
    for (int o = 0; o <= P; ++o) {
        const int e = e0 + o;
        const double2 v = src[e >> 1];
        tmp[o] = (e & 1) ? v.y : v.x;
    }
    if constexpr (TR) {
        if (delta != 0.) {
            translate_a_regs<P>(out, tmp, delta);
            return;
        }
    }
#pragma unroll
    for (int o = 0; o <= P; ++o) {
        out[o] = tmp[o];
    }
}

// Runtime-order load of any of the 6 coordinates (slow path; private memory).
__device__ inline void load_poly_rt(double *out, const double *xyz, const double *vr, std::uint64_t row, int c,
                                    double delta, int P)
{
    const double *src = (c < 3) ? xyz + row * static_cast<std::uint64_t>(tc_xyz_stride(P)) + c * (P + 1)
                                : vr + row * static_cast<std::uint64_t>(tc_vr_stride(P)) + (c - 3) * (P + 1);
    if (delta != 0.) {
        double tmp[RF_MAX_ORDER + 1];
        for (int o = 0; o <= P; ++o) {
            tmp[o] = src[o];
        }
        poly_translate_a(out, tmp, delta, P);
    } else {
        for (int o = 0; o <= P; ++o) {
            out[o] = src[o];
        }
    }
}

// Slow path: the fast exclusion check did not exclude a collision and/or a conjunction in
// this interval. ss: the distance-square polynomial with its ORIGINAL constant term.
template <int P>
CSC_DEV_NOINLINE void np_slow_path(const np_params &q, const double *ss_in, bool do_coll, bool do_conj, double rsum2,
                                   double rf_int, dfl lb_rf, std::uint32_t pi, std::uint32_t pj, std::uint64_t row_i,
                                   std::uint64_t row_j, double delta_i, double delta_j)
{
    double ss[P + 1], roots[P + 1];
    for (int o = 0; o <= P; ++o) {
        ss[o] = ss_in[o];
    }
    int warn = 0;
    if (do_coll) {
        const double c0 = ss[0];
        ss[0] = c0 - rsum2;
        int nr = isolate_and_refine(ss, P, rf_int, -1, roots, P + 1, &warn);
        if (nr < 0) {
            nr = -1 - nr;
        }
        for (int k = 0; k < nr && k <= P; ++k) {
            const double tcoll = df_add(lb_rf, df(roots[k])).hi;
            if (!finite_d(tcoll)) {
                warn |= 1;
                continue;
            }
            const std::uint32_t slot = wave_append_slot(&q.counters[0]);
            if (slot < q.coll_cap) {
                q.coll[slot] = coll_rec{pi, pj, tcoll};
            }
        }
        ss[0] = c0;
    }
    if (do_conj) {
        double der[P + 1];
        for (int i = 0; i < P; ++i) {
            der[i] = (i + 1) * ss[i + 1];
        }
        der[P] = 0;
        int nr = isolate_and_refine(der, P, rf_int, 1, roots, P + 1, &warn);
        if (nr < 0) {
            nr = -1 - nr;
        }
        for (int k = 0; k < nr && k <= P; ++k) {
            const double conj_tm = df_add(df(0.), df(roots[k])).hi;
            if (!finite_d(conj_tm)) {
                warn |= 1;
                continue;
            }
            const double conj_dist2 = poly_eval(ss, conj_tm, P);
            if (!finite_d(conj_dist2)) {
                warn |= 1;
                continue;
            }
            if (conj_dist2 < q.conj_thresh2) {
                conj_rec c;
                c.i = pi;
                c.j = pj;
                const dfl init_time{q.init_hi, q.init_lo};
                c.time = df_add(init_time, df_add(lb_rf, df(conj_tm))).hi;
                c.dist = sqrt(dmax(conj_dist2, 0.));
                double poly[P + 1];
                for (int cc = 0; cc < 6; ++cc) {
                    load_poly_rt(poly, q.tcs_xyz, q.tcs_vr, row_i, cc, delta_i, P);
                    c.state_i[cc] = poly_eval(poly, conj_tm, P);
                    load_poly_rt(poly, q.tcs_xyz, q.tcs_vr, row_j, cc, delta_j, P);
                    c.state_j[cc] = poly_eval(poly, conj_tm, P);
                }
                const std::uint32_t slot = wave_append_slot(&q.counters[1]);
                if (slot < q.conj_cap) {
                    q.conj[slot] = c;
                }
            }
        }
    }
    if (warn) {
        atomicOr(&q.counters[2], static_cast<std::uint32_t>(warn));
    }
}

// The record of a conjunction at conj_tm after the window start (src/sim_narrow_phase.cpp:886-947):
// the twelve coordinates by Horner on the translated polynomials, in registers (same operations
// as load_poly_rt + poly_eval).
template <int P>
CSC_DEV void np_conj_write(const np_params &q, conj_rec *out, double conj_tm, double conj_dist2, dfl lb_rf,
                           std::uint32_t pi, std::uint32_t pj, std::uint64_t row_i, std::uint64_t row_j, double delta_i,
                           double delta_j)
{
    out->i = pi;
    out->j = pj;
    const dfl init_time{q.init_hi, q.init_lo};
    out->time = df_add(init_time, df_add(lb_rf, df(conj_tm))).hi;
    out->dist = sqrt(dmax(conj_dist2, 0.));
    const double *xi = q.tcs_xyz + row_i * static_cast<std::uint64_t>(tc_xyz_stride(P));
    const double *xj = q.tcs_xyz + row_j * static_cast<std::uint64_t>(tc_xyz_stride(P));
    const double *vi = q.tcs_vr + row_i * static_cast<std::uint64_t>(tc_vr_stride(P));
    const double *vj = q.tcs_vr + row_j * static_cast<std::uint64_t>(tc_vr_stride(P));
#pragma unroll 1
    for (int cc = 0; cc < 12; ++cc) {
        const int c6 = cc >> 1;
        const bool second = (cc & 1) != 0;
        const double *src = (c6 < 3) ? (second ? xj : xi) + c6 * (P + 1) : (second ? vj : vi) + (c6 - 3) * (P + 1);
        const double delta = second ? delta_j : delta_i;
        double raw[P + 1], poly[P + 1];
#pragma unroll
        for (int o = 0; o <= P; ++o) {
            raw[o] = src[o];
        }
        if (delta != 0.) {
            translate_a_regs<P>(poly, raw, delta);
        } else {
#pragma unroll
            for (int o = 0; o <= P; ++o) {
                poly[o] = raw[o];
            }
        }
        const double v = poly_eval_n<P>(poly, conj_tm, P);
        (second ? out->state_j : out->state_i)[c6] = v;
    }
}

// np_slow_path with the polynomials in registers (isolate_and_refine_n, np_conj_write): the
// same operations in the same order, hence the same events bit for bit, at ~1/40 of the time --
// in a conjunction-heavy superstep nearly every candidate pair gets here.
template <int P>
CSC_DEV_NOINLINE void np_slow_path_regs(const np_params &q, const double *ss_in, bool do_coll, bool do_conj, double rsum2,
                                        double rf_int, dfl lb_rf, std::uint32_t pi, std::uint32_t pj, std::uint64_t row_i,
                                        std::uint64_t row_j, double delta_i, double delta_j)
{
    double ss[P + 1], roots[P + 1];
#pragma unroll
    for (int o = 0; o <= P; ++o) {
        ss[o] = ss_in[o];
    }
    int warn = 0;
    if (do_coll) {
        const double c0 = ss[0];
        ss[0] = c0 - rsum2;
        int nr = isolate_and_refine_n<P>(ss, rf_int, -1, roots, P + 1, &warn);
        if (nr < 0) {
            nr = -1 - nr;
        }
        for (int k = 0; k < nr && k <= P; ++k) {
            const double tcoll = df_add(lb_rf, df(roots[k])).hi;
            if (!finite_d(tcoll)) {
                warn |= 1;
                continue;
            }
            const std::uint32_t slot = wave_append_slot(&q.counters[0]);
            if (slot < q.coll_cap) {
                q.coll[slot] = coll_rec{pi, pj, tcoll};
            }
        }
        ss[0] = c0;
    }
    if (do_conj) {
        double der[P + 1];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            der[i] = (i + 1) * ss[i + 1];
        }
        der[P] = 0;
        int nr = isolate_and_refine_n<P>(der, rf_int, 1, roots, P + 1, &warn);
        if (nr < 0) {
            nr = -1 - nr;
        }
        for (int k = 0; k < nr && k <= P; ++k) {
            const double conj_tm = df_add(df(0.), df(roots[k])).hi;
            if (!finite_d(conj_tm)) {
                warn |= 1;
                continue;
            }
            const double conj_dist2 = poly_eval_n<P>(ss, conj_tm, P);
            if (!finite_d(conj_dist2)) {
                warn |= 1;
                continue;
            }
            if (conj_dist2 < q.conj_thresh2) {
                const std::uint32_t slot = wave_append_slot(&q.counters[1]);
                if (slot < q.conj_cap) {
                    np_conj_write<P>(q, q.conj + slot, conj_tm, conj_dist2, lb_rf, pi, pj, row_i, row_j, delta_i, delta_j);
                }
            }
        }
    }
    if (warn) {
        atomicOr(&q.counters[2], static_cast<std::uint32_t>(warn));
    }
}

// Root-finding window of one pair of substeps (src/sim_narrow_phase.cpp:691-716).
struct np_window {
    dfl lb_rf;
    double delta_i, delta_j, rf_int;
    dfl tend_i, tend_j;
    bool ok;
};

CSC_DEV np_window np_make_window(const np_params &q, dfl chunk_begin, dfl chunk_end, std::uint64_t off_i,
                                 std::uint32_t it_i, std::uint64_t off_j, std::uint32_t it_j)
{
    np_window w;
    const dfl ss_start_i = (it_i == 0u) ? df(0.) : load_tcoord(q.tcoords, off_i + it_i - 1u);
    const dfl ss_start_j = (it_j == 0u) ? df(0.) : load_tcoord(q.tcoords, off_j + it_j - 1u);
    w.tend_i = load_tcoord(q.tcoords, off_i + it_i);
    w.tend_j = load_tcoord(q.tcoords, off_j + it_j);
    const dfl lb_i = df_max(chunk_begin, ss_start_i), ub_i = df_min(chunk_end, w.tend_i);
    const dfl lb_j = df_max(chunk_begin, ss_start_j), ub_j = df_min(chunk_end, w.tend_j);
    const dfl ub_rf = df_min(ub_i, ub_j);
    w.lb_rf = df_max(lb_i, lb_j);
    w.delta_i = df_sub(w.lb_rf, ss_start_i).hi;
    w.delta_j = df_sub(w.lb_rf, ss_start_j).hi;
    w.rf_int = df_sub(ub_rf, w.lb_rf).hi;
    w.ok = finite_d(w.delta_i) && finite_d(w.delta_j) && finite_d(w.rf_int) && !(w.delta_i < 0) && !(w.delta_j < 0)
           && !(w.rf_int < 0);
    return w;
}

// Distance-square polynomial (ssdiff3, src/sim_jit.cpp:118-218), coordinate by coordinate.
template <int P, bool TR = true>
CSC_DEV void np_build_ss(double *ss, const double *xyz, std::uint64_t row_i, std::uint64_t row_j, double delta_i,
                         double delta_j)
{
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        double d[P + 1];
        {
            double a[P + 1], b[P + 1];
            load_poly<P, TR>(a, xyz, row_i, cc, delta_i);
            load_poly<P, TR>(b, xyz, row_j, cc, delta_j);
#pragma unroll
            for (int o = 0; o <= P; ++o) {
                d[o] = a[o] - b[o];
            }
        }
        double sq[P + 1];
        sq_all<P, 0>::run(sq, d);
#pragma unroll
        for (int o = 0; o <= P; ++o) {
            ss[o] = (cc == 0) ? sq[o] : ss[o] + sq[o];
        }
    }
}

CSC_DEV dfl np_chunk_begin(const np_params &q, std::uint32_t c)
{
    return df(q.ct * static_cast<double>(q.chunk_first + c));
}
// get_chunk_begin_end (src/sim_basic.cpp:87-103)
CSC_DEV dfl np_chunk_end(const np_params &q, std::uint32_t c)
{
    const std::uint32_t g = q.chunk_first + c;
    return df((g == q.nchunks_global - 1u) ? q.delta_t : q.ct * static_cast<double>(g + 1u));
}

// The fast exclusion checks on one window, given its distance-square polynomial (clobbered).
// Windows that are not excluded are appended to the survivor list. Returns false if the pair
// must be abandoned (non-finite interval evaluation, src/sim_narrow_phase.cpp:845-852).
template <int P>
CSC_DEV bool np_check_window(const np_params &q, double *ss, bool coll_act, bool conj_act, double rsum2, double rf_int,
                             std::uint32_t pi, std::uint32_t pj, std::uint32_t c, std::uint32_t it_i, std::uint32_t it_j)
{
    bool need_coll = false, need_conj = false, bail = false;
    if (coll_act) {
        const double c0 = ss[0];
        ss[0] = c0 - rsum2;
        need_coll = !fex_excluded<P>(ss, rf_int);
        ss[0] = c0;
    }
    if (conj_act) {
        const ivl d2 = iv_poly_eval<P>(ss, ivl{0., rf_int});
        if (!finite_d(d2.lo) || !finite_d(d2.hi)) {
            atomicOr(&q.counters[2], 32u);
            bail = true;
        } else if (d2.lo < q.conj_thresh2) {
#pragma unroll
            for (int o = 0; o < P; ++o) {
                ss[o] = (o + 1) * ss[o + 1];
            }
            ss[P] = 0;
            need_conj = !fex_excluded<P>(ss, rf_int);
        }
    }
    if (need_coll || (need_conj && !bail)) {
        const std::uint32_t slot = atomicAdd(&q.counters[3], 1u);
        if (slot < q.surv_cap) {
            q.surv[2u * slot] = make_uint4(pi, pj, c, (need_coll ? 1u : 0u) | ((need_conj && !bail) ? 2u : 0u));
            q.surv[2u * slot + 1u] = make_uint4(it_i, it_j, 0u, 0u);
        }
    }
    return !bail;
}

CSC_DEV double np_rsum2(const np_params &q, std::uint32_t pi, std::uint32_t pj)
{
    const double p_rad_i = q.state[static_cast<std::uint64_t>(pi) * 7u + 6u];
    const double p_rad_j = q.state[static_cast<std::uint64_t>(pj) * 7u + 6u];
    return (p_rad_i + p_rad_j) * (p_rad_i + p_rad_j);
}

// One lane per pair. Windows whose two substeps both start at the window start (no polynomial
// translation: always the case when every particle takes a single substep per superstep) are
// checked here; the others are deferred to k_np_fex_tr, whose translation code needs twice the
// registers, so that this kernel keeps a useful occupancy.
// Pre-filter (no counterpart in the reference: a rigorous geometric exclusion that runs before
// its polynomial test and cannot change the detected events). Propagation leaves, per particle
// and chunk, the positions at the 5 times cb + s (ce - cb) / 4 and a radius dev >= (distance of
// the true trajectory from the chord through consecutive samples) + the particle's size + a
// margin for single-precision storage. For t in one quarter of the chunk
//   |p_i(t) - p_j(t)| >= |chord_i(t) - chord_j(t)| - dev_i - dev_j,
// and the difference of two chords over the same quarter is itself a segment, so the minimum
// of the right-hand side is a point-segment distance. A pair whose bound exceeds the
// conjunction threshold in all four quarters can have neither a collision (distance <= r_i +
// r_j) nor a recordable conjunction (distance < conj_thresh) anywhere in the chunk. Measured
// on the 1M-object LEO population: 0.1 % of the AABB pairs survive.
// One lane per pair, pair blocks in group-major order; FILTER = false passes every pair on
// (used to check that the filter does not change any result).
// The geometric exclusion test for the particles at sorted positions pa, pb of the chunk whose
// arrays start at offset co. Returns false when the pair can be dropped.
// (np_prefilter_core, chord_keep: k_chord.hpp -- shared with the tile kernel of the broad phase)
CSC_DEV bool np_prefilter_keep(const np_params &q, std::size_t co, std::uint32_t pa, std::uint32_t pb)
{
    return chord_keep(q.srt_recs, co, pa, pb, static_cast<float>(q.thr_extra));
}

template <bool FILTER>
__global__ __launch_bounds__(64) void k_np_prefilter(const np_params q)
{
    const std::uint32_t wg = xcd_remap<64>(blockIdx.x, gridDim.x);
    const uint2 ent = q.order[wg / (PB_SIZE / 64)];
    const std::uint32_t blk = ent.x, c = ent.y, t = (wg % (PB_SIZE / 64)) * 64u + threadIdx.x;
    if (c >= q.nchunks || t >= q.pb_count[blk]) {
        // (chunk dropped by a terminal-event shrink of the superstep, src/sim_narrow_phase.cpp:560)
        return;
    }
    const uint2 rec = q.pairs[static_cast<std::size_t>(blk) * PB_SIZE + t];
    const std::size_t co = static_cast<std::size_t>(c) * q.n;
    bool keep = true;
    if constexpr (FILTER) {
        keep = np_prefilter_keep(q, co, rec.x, rec.y);
    }
    if (keep) {
        const std::uint32_t slot = atomicAdd(&q.counters[5], 1u);
        if (slot < q.cand_cap) {
            q.cand[slot] = make_uint4(q.vidx[co + rec.x], q.vidx[co + rec.y], c, 0u);
        }
    }
}

// The same over the flat pair lists of the dual traversal (k_bp_dual.hpp): NP_FLAT_SHARDS lists
// of packed pairs pos_a | pos_b << 28 | chunk << 56; workgroup w handles NP_FLAT_TILE pairs of the shard
// whose tile range [flat_tile_off[s], flat_tile_off[s + 1]) contains w.
// Candidates are appended through a buffer in LDS and one device atomic per flush: one atomic per
// candidate on the list's counter serialises at ~10 ns each (same address), which WAS this
// kernel's time: 0.32 ms for 27 k candidates, 4.3 ms for 424 k (1M objects, 1 and 8 chunks),
// whatever the record size (a 32-byte quantised copy of the records, tested first, changed
// nothing and was removed again) or the number of pairs in flight per lane. Since then: 0.22 and
// 2.3 ms = 10 ps per pair whatever the population (62 k to 4M objects: it is not a cache effect),
// 57 % of it VALU issue (204 instructions per 64 pairs). Loading each record with four lanes and
// handing it over through LDS (16 instead of 64 cache lines per load instruction) was tried and
// changed nothing either (0.24 / 2.3 ms): the L1 lookup rate is not the bound.
constexpr unsigned NP_FLAT_CBUF = 2 * NP_FLAT_TILE; // flushed before a tile when more than half full

template <bool FILTER>
__global__ __launch_bounds__(NP_FLAT_BLOCK) void k_np_prefilter_flat(const np_params q)
{
    __shared__ uint4 s_cand[NP_FLAT_CBUF];
    __shared__ std::uint32_t s_n, s_base;
    if (threadIdx.x == 0) {
        s_n = 0;
    }
    __syncthreads();
    auto flush = [&]() { // (whole workgroup)
        const std::uint32_t nb = s_n;
        if (threadIdx.x == 0 && nb != 0u) {
            s_base = atomicAdd(&q.counters[5], nb);
        }
        __syncthreads();
        const std::uint32_t base = s_base;
        for (std::uint32_t k = threadIdx.x; k < nb; k += NP_FLAT_BLOCK) {
            if (base + k < q.cand_cap) {
                q.cand[base + k] = s_cand[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_n = 0;
        }
        __syncthreads();
    };
    const std::uint32_t ntiles = q.flat_tile_off[NP_FLAT_SHARDS];
    for (std::uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        std::uint32_t wg = tile;
        if (q.flat_xcd_run != 0u) {
            // Workgroups are dealt to the 8 XCDs round-robin: give each XCD runs of consecutive tiles
            // of the list, so that its L2 sees one neighbourhood of trajectory records at a time.
            const std::uint32_t span = 8u * q.flat_xcd_run, full = (ntiles / span) * span;
            if (wg < full) {
                const std::uint32_t r = wg % span;
                wg = (wg / span) * span + (r % 8u) * q.flat_xcd_run + r / 8u;
            }
        }
        int s = 0;
        while (s + 1 < NP_FLAT_SHARDS && wg >= q.flat_tile_off[s + 1]) {
            ++s;
        }
        const std::uint32_t cnt = q.flat_cnt[s]; // (> 0: the shard has a tile)
        const std::uint32_t tb = (wg - q.flat_tile_off[s]) * NP_FLAT_TILE + threadIdx.x;
        const unsigned long long *list = q.flat_pairs + static_cast<std::size_t>(s) * q.flat_cap;
        // All loads of the NP_FLAT_PPT pairs of a lane are issued before the first use (clamped
        // indices, no branches in between).
        unsigned long long rec[NP_FLAT_PPT];
        bool keep[NP_FLAT_PPT];
#pragma unroll
        for (unsigned k = 0; k < NP_FLAT_PPT; ++k) {
            const std::uint32_t t = tb + k * NP_FLAT_BLOCK;
            keep[k] = t < cnt;
            rec[k] = list[keep[k] ? t : cnt - 1u];
        }
        std::uint32_t pa[NP_FLAT_PPT], pb[NP_FLAT_PPT], c[NP_FLAT_PPT];
#pragma unroll
        for (unsigned k = 0; k < NP_FLAT_PPT; ++k) {
            pa[k] = static_cast<std::uint32_t>(rec[k]) & 0xfffffffu;
            pb[k] = static_cast<std::uint32_t>(rec[k] >> 28) & 0xfffffffu;
            c[k] = static_cast<std::uint32_t>(rec[k] >> 56);
            // (chunk dropped by a terminal-event shrink of the superstep, src/sim_narrow_phase.cpp:560)
            keep[k] = keep[k] && c[k] < q.nchunks;
        }
#pragma unroll
        for (unsigned k = 0; k < NP_FLAT_PPT; ++k) {
            if (keep[k]) {
                const std::size_t co = static_cast<std::size_t>(c[k]) * q.n;
                bool kk = true;
                if constexpr (FILTER) {
                    kk = np_prefilter_keep(q, co, pa[k], pb[k]);
                }
                if (kk) {
                    // (i < j; the tile kernel of the broad phase records positions in no particular order)
                    const std::uint32_t oa = q.vidx[co + pa[k]], ob = q.vidx[co + pb[k]];
                    s_cand[atomicAdd(&s_n, 1u)] = make_uint4(oa < ob ? oa : ob, oa < ob ? ob : oa, c[k], 0u);
                }
            }
        }
        __syncthreads();
        // Every thread reads the count BEFORE any thread can start the next tile's appends (a
        // faster wave's atomicAdd would otherwise split the workgroup's decision: mismatched
        // barriers inside flush(), lost candidates).
        const bool do_flush = s_n > NP_FLAT_TILE;
        __syncthreads();
        if (do_flush) {
            flush();
        }
    }
    // (after the loop's last barrier nobody writes s_n any more: uniform)
    if (s_n != 0u) {
        flush();
    }
}

template <int P>
__global__ __launch_bounds__(64) void k_np_fex(const np_params q)
{
    // One lane per candidate pair.
    const std::uint32_t k = xcd_remap(blockIdx.x, gridDim.x) * 64u + threadIdx.x;
    const std::uint32_t nc = q.counters[5] < q.cand_cap ? q.counters[5] : q.cand_cap;
    if (k >= nc) {
        return;
    }
    const uint4 rec = q.cand[k];
    const std::uint32_t c = rec.z;
    if (c >= q.nchunks) {
        return;
    }
    const dfl chunk_begin = np_chunk_begin(q, c), chunk_end = np_chunk_end(q, c);
    {
        const std::uint32_t pi = rec.x, pj = rec.y;
        const std::uint8_t ai = q.active[pi], aj = q.active[pj];
        const bool coll_act = ((ai | aj) & 1u) != 0, conj_act = ((ai | aj) & 2u) != 0;
        const double rsum2 = coll_act ? np_rsum2(q, pi, pj) : 0.;

        const std::uint64_t off_i = q.tc_row[pi], off_j = q.tc_row[pj];
        const std::uint32_t ns_i = q.nsub[pi], ns_j = q.nsub[pj];
        std::uint32_t it_i, end_i, it_j, end_j;
        substep_range(q.tcoords, off_i, ns_i, chunk_begin, chunk_end, it_i, end_i);
        substep_range(q.tcoords, off_j, ns_j, chunk_begin, chunk_end, it_j, end_j);

        while (it_i != end_i && it_j != end_j) {
            const np_window w = np_make_window(q, chunk_begin, chunk_end, off_i, it_i, off_j, it_j);
            if (!w.ok) {
                atomicOr(&q.counters[2], 16u);
                break;
            }
            if (w.delta_i != 0. || w.delta_j != 0.) {
                const std::uint32_t slot = atomicAdd(&q.counters[4], 1u);
                if (slot < q.defer_cap) {
                    q.defer[2u * slot] = make_uint4(pi, pj, c, 0u);
                    q.defer[2u * slot + 1u] = make_uint4(it_i, it_j, 0u, 0u);
                }
            } else {
                double ss[P + 1];
                np_build_ss<P, false>(ss, q.tcs_xyz, off_i + it_i, off_j + it_j, 0., 0.);
                if (!np_check_window<P>(q, ss, coll_act, conj_act, rsum2, w.rf_int, pi, pj, c, it_i, it_j)) {
                    break;
                }
            }
            if (df_lt(w.tend_i, w.tend_j)) {
                ++it_i;
            } else if (df_lt(w.tend_j, w.tend_i)) {
                ++it_j;
            } else {
                ++it_i;
                ++it_j;
            }
        }
    }
}

// One lane per deferred window (at least one of the two polynomials needs a translation).
// NOTE: a pair whose window hits the non-finite bail-out is abandoned by the reference for the
// rest of the chunk; here only that window is dropped (the warning flag is raised either way).
template <int P>
__global__ __launch_bounds__(64) void k_np_fex_tr(const np_params q)
{
    const std::uint32_t k = xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const std::uint32_t nd = q.counters[4] < q.defer_cap ? q.counters[4] : q.defer_cap;
    if (k >= nd) {
        return;
    }
    const uint4 r0 = q.defer[2u * k], r1 = q.defer[2u * k + 1u];
    const std::uint32_t pi = r0.x, pj = r0.y, c = r0.z;
    const std::uint8_t ai = q.active[pi], aj = q.active[pj];
    const bool coll_act = ((ai | aj) & 1u) != 0, conj_act = ((ai | aj) & 2u) != 0;
    const double rsum2 = coll_act ? np_rsum2(q, pi, pj) : 0.;
    const std::uint64_t off_i = q.tc_row[pi], off_j = q.tc_row[pj];
    const np_window w = np_make_window(q, np_chunk_begin(q, c), np_chunk_end(q, c), off_i, r1.x, off_j, r1.y);
    double ss[P + 1];
    np_build_ss<P, true>(ss, q.tcs_xyz, off_i + r1.x, off_j + r1.y, w.delta_i, w.delta_j);
    np_check_window<P>(q, ss, coll_act, conj_act, rsum2, w.rf_int, pi, pj, c, r1.x, r1.y);
}

// One lane per survivor of k_np_fex: same window, same polynomial, then the root finding.
template <int P>
__global__ __launch_bounds__(64) void k_np_roots(const np_params q)
{
    const std::uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const std::uint32_t ns = q.counters[3] < q.surv_cap ? q.counters[3] : q.surv_cap;
    if (k >= ns) {
        return;
    }
    const uint4 r0 = q.surv[2u * k], r1 = q.surv[2u * k + 1u];
    const std::uint32_t pi = r0.x, pj = r0.y, c = r0.z;
    const std::uint64_t off_i = q.tc_row[pi], off_j = q.tc_row[pj];
    const np_window w = np_make_window(q, np_chunk_begin(q, c), np_chunk_end(q, c), off_i, r1.x, off_j, r1.y);
    const double rsum2 = np_rsum2(q, pi, pj);
    double ss[P + 1];
    np_build_ss<P>(ss, q.tcs_xyz, off_i + r1.x, off_j + r1.y, w.delta_i, w.delta_j);
    const bool do_coll = (r0.w & 1u) != 0;
    const bool do_conj = (r0.w & 2u) != 0;
    if (q.conj_fast) {
        np_slow_path_regs<P>(q, ss, do_coll, do_conj, rsum2, w.rf_int, w.lb_rf, pi, pj, off_i + r1.x, off_j + r1.y, w.delta_i,
                             w.delta_j);
    } else {
        np_slow_path<P>(q, ss, do_coll, do_conj, rsum2, w.rf_int, w.lb_rf, pi, pj, off_i + r1.x, off_j + r1.y, w.delta_i,
                        w.delta_j);
    }
}

// K8: dense_propagate (src/sim_propagate.cpp:1401-1500). out: [n][6].
template <int P>
__global__ __launch_bounds__(256) void k_dense(std::uint32_t n, double t, const std::uint64_t *tc_row,
                                               const std::uint32_t *nsub, const double *tcs_xyz, const double *tcs_vr,
                                               const double *tcoords, double *out, int *err_flag)
{
    const std::uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const std::uint64_t off = tc_row[i];
    const std::uint32_t ns = nsub[i];
    if (ns == 0u) {
        atomicOr(err_flag, 1);
        return;
    }
    const dfl dt = df(t);
    // lower_bound(dt): first s with !(tcoords[s] < dt)
    std::uint32_t lo = 0, len = ns;
    while (len > 0) {
        const std::uint32_t half = len / 2, mid = lo + half;
        if (df_lt(load_tcoord(tcoords, off + mid), dt)) {
            lo = mid + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    if (lo == ns) {
        lo = ns - 1u;
    }
    const dfl ss_start = (lo == 0u) ? df(0.) : load_tcoord(tcoords, off + lo - 1u);
    const double eval_tm = df_sub(dt, ss_start).hi;
    const double *bxyz = tcs_xyz + (off + lo) * static_cast<std::uint64_t>(tc_xyz_stride(P));
    const double *bvr = tcs_vr + (off + lo) * static_cast<std::uint64_t>(tc_vr_stride(P));
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double *p = (c < 3) ? bxyz + c * (P + 1) : bvr + (c - 3) * (P + 1);
        double acc = p[P];
#pragma unroll
        for (int o = 1; o <= P; ++o) {
            acc = p[P - o] + acc * eval_tm;
        }
        out[static_cast<std::uint64_t>(i) * 6u + c] = acc;
    }
}

} // namespace csc
