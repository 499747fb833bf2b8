// oracle/orc_capi.cpp
//
// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT (see oracle/orc.hpp).
// Plain-C entry points over the oracle so that tests/ and bench.py's
// cpu_baseline leg can drive it through ctypes.

#include "orc_sim.hpp"

#include <cstdio>
#include <memory>

namespace
{

thread_local std::string g_last_error;

template <typename F>
int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return -1000;
    } catch (...) {
        g_last_error = "unknown exception";
        return -1000;
    }
}

orc::sim *S(void *h)
{
    return static_cast<orc::sim *>(h);
}

} // namespace

extern "C" {

const char *orc_last_error()
{
    return g_last_error.c_str();
}

int orc_num_threads()
{
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#if defined(_OPENMP)
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// ---- scalar / polynomial helpers ------------------------------------------------

std::uint64_t orc_morton4(std::uint64_t a, std::uint64_t b, std::uint64_t c, std::uint64_t d)
{
    return orc::morton4_16(a, b, c, d);
}

std::uint64_t orc_disc_single_coord(float x, float mn, float mx)
{
    return orc::disc_single_coord(x, mn, mx);
}

void orc_ival_horner(const double *cf, std::uint32_t order, double lo, double hi, double *out2)
{
    const auto r = orc::poly_eval(cf, orc::ival(lo, hi), order);
    out2[0] = r.lower;
    out2[1] = r.upper;
}

void orc_poly_translate_a(double *out, const double *cf, double a, std::uint32_t order)
{
    orc::poly_translate_a(out, cf, a, order);
}

void orc_poly_ssdiff3(double *out, const double *in6, std::uint32_t order)
{
    orc::poly_ssdiff3(out, in6, order);
}

std::uint32_t orc_fex_check(const double *cf, double h, std::uint32_t order)
{
    return orc::fex_check(cf, h, order);
}

std::uint32_t orc_rtscc(const double *cf, std::uint32_t order)
{
    std::vector<double> a(order + 1u), b(order + 1u);
    std::uint32_t n = 0;
    orc::poly_rtscc(a.data(), b.data(), &n, cf, order);
    return n;
}

// Roots of poly in [0, rf_int) with direction filter; returns the number of roots.
int orc_root_find(const double *poly, std::uint32_t order, double rf_int, int direction, double *roots, int cap)
{
    orc::rf_scratch sc;
    orc::rf_stats st;
    int n = 0;
    orc::run_poly_root_finding(poly, order, rf_int, direction, sc, st, [&](double r) {
        if (n < cap) {
            roots[n] = r;
        }
        ++n;
    });
    return n;
}

double orc_dfloat_add(double ahi, double alo, double bhi, double blo, double *lo_out)
{
    const auto r = orc::dfloat(ahi, alo) + orc::dfloat(bhi, blo);
    *lo_out = r.lo;
    return r.hi;
}

// ---- stage functions on caller-provided arrays -----------------------------------

int orc_morton_encode(std::uint64_t n, const float *lbs, const float *ubs, const float *glb, float *gub_inout,
                      std::uint64_t *codes)
{
    return guarded([&]() {
        orc::morton_encode_chunk(n, lbs, ubs, glb, gub_inout, codes);
        return 0;
    });
}

void orc_sort_indices(std::uint64_t n, const std::uint64_t *codes, std::uint32_t *vidx)
{
    orc::sort_indices(n, codes, vidx);
}

// Returns the number of nodes (<= 2n - 1), or -1 if cap is too small.
std::int64_t orc_bvh_build(std::uint64_t n, const std::uint64_t *srt_codes, const float *srt_lbs, const float *srt_ubs,
                           void *nodes_out, std::uint64_t cap)
{
    std::vector<orc::bvh_node> tree;
    orc::bvh_build(n, srt_codes, srt_lbs, srt_ubs, tree);
    if (tree.size() > cap) {
        return -1;
    }
    std::memcpy(nodes_out, tree.data(), tree.size() * sizeof(orc::bvh_node));
    return static_cast<std::int64_t>(tree.size());
}

int orc_bvh_verify(std::uint64_t n, const std::uint64_t *srt_codes, const float *srt_lbs, const float *srt_ubs,
                   const void *nodes, std::uint64_t n_nodes)
{
    const auto msg = orc::bvh_verify(n, srt_codes, srt_lbs, srt_ubs, static_cast<const orc::bvh_node *>(nodes), n_nodes);
    if (msg.empty()) {
        return 0;
    }
    g_last_error = msg;
    return -1;
}

// Sorted (i, j) pairs; returns the number of pairs (may exceed cap, in which case only cap are written).
std::int64_t orc_broad_phase(std::uint64_t n, const void *nodes, const float *srt_lbs, const float *srt_ubs,
                             const std::uint32_t *vidx, const char *coll_active, const char *conj_active,
                             std::uint32_t *pairs_out, std::uint64_t cap)
{
    std::vector<std::pair<std::uint32_t, std::uint32_t>> bp;
    orc::broad_phase(n, static_cast<const orc::bvh_node *>(nodes), srt_lbs, srt_ubs, vidx, coll_active, conj_active, bp);
    std::sort(bp.begin(), bp.end());
    for (std::size_t k = 0; k < bp.size() && k < cap; ++k) {
        pairs_out[2u * k] = bp[k].first;
        pairs_out[2u * k + 1u] = bp[k].second;
    }
    return static_cast<std::int64_t>(bp.size());
}

std::int64_t orc_broad_phase_brute(std::uint64_t n, const float *lbs, const float *ubs, const char *coll_active,
                                   const char *conj_active, std::uint32_t *pairs_out, std::uint64_t cap)
{
    std::vector<std::pair<std::uint32_t, std::uint32_t>> bp;
    orc::broad_phase_brute(n, lbs, ubs, coll_active, conj_active, bp);
    std::sort(bp.begin(), bp.end());
    for (std::size_t k = 0; k < bp.size() && k < cap; ++k) {
        pairs_out[2u * k] = bp[k].first;
        pairs_out[2u * k + 1u] = bp[k].second;
    }
    return static_cast<std::int64_t>(bp.size());
}

// ---- sim object ------------------------------------------------------------------

void *orc_sim_new(const double *state, std::uint64_t nparts, double ct, std::uint32_t n_par_ct, double conj_thresh,
                  double min_coll_radius, double tol, int dyn_kind, double mu, double J2, double Re, double exit_radius,
                  int reentry_kind, const double *reentry3)
{
    void *ret = nullptr;
    guarded([&]() {
        orc::sim_params p;
        p.ct = ct;
        p.n_par_ct = n_par_ct;
        p.conj_thresh = conj_thresh;
        p.min_coll_radius = min_coll_radius;
        p.tcfg.tol = (tol == 0.) ? std::numeric_limits<double>::epsilon() : tol;
        p.tcfg.dyn.kind = dyn_kind;
        p.tcfg.dyn.mu = mu;
        p.tcfg.dyn.J2 = J2;
        p.tcfg.dyn.Re = Re;
        p.tcfg.exit_radius = exit_radius;
        p.tcfg.reentry_kind = reentry_kind;
        for (int i = 0; i < 3; ++i) {
            p.tcfg.reentry[i] = reentry3 ? reentry3[i] : 0.;
        }
        ret = new orc::sim(std::vector<double>(state, state + nparts * 7u), p);
        return 0;
    });
    return ret;
}

// Runtime parameters [nparts][npars] (BSTAR per particle when drag is on; cascade.py/dynamics/
// _simple_earth.py:367) and, optionally, the 12 constants of the density fit (alpha, beta per
// metre, gamma in metres).
int orc_sim_set_pars(void *h, const double *pars, std::uint64_t n, const double *dens12)
{
    return guarded([&]() {
        auto &p = S(h)->prm;
        p.pars.assign(pars, pars + n);
        if (dens12) {
            for (int k = 0; k < 4; ++k) {
                p.tcfg.dyn.dens_alpha[k] = dens12[k];
                p.tcfg.dyn.dens_beta[k] = dens12[4 + k];
                p.tcfg.dyn.dens_gamma[k] = dens12[8 + k];
            }
        }
        return 0;
    });
}

// General dynamics as a tape (see orc::tape_op): switches the simulation to DYN_TAPE.
int orc_sim_set_tape(void *h, const orc::tape_op *ops, std::uint64_t nops, std::uint32_t nslots, const std::int32_t *rhs7,
                     std::uint32_t npars)
{
    return guarded([&]() {
        auto &d = S(h)->prm.tcfg.dyn;
        for (std::uint64_t k = 0; k < nops; ++k) {
            const auto &o = ops[k];
            const bool two = o.op == orc::TP_SINCOS;
            if (o.op < orc::TP_ADD || o.op > orc::TP_CONST || o.dst < 7 || static_cast<std::uint32_t>(o.dst) + (two ? 1u : 0u) >= nslots
                || o.a < 0 || o.a >= o.dst || o.b >= o.dst || (o.b <= -2 && static_cast<std::uint32_t>(-2 - o.b) >= npars)) {
                throw std::invalid_argument("Malformed dynamics tape");
            }
        }
        d.tape.assign(ops, ops + nops);
        d.tape_nslots = nslots;
        d.tape_npars = npars;
        for (int c = 0; c < 7; ++c) {
            if (rhs7[c] < 0 || static_cast<std::uint32_t>(rhs7[c]) >= nslots) {
                throw std::invalid_argument("Malformed dynamics tape");
            }
            d.tape_rhs[c] = rhs7[c];
        }
        d.kind = orc::DYN_TAPE;
        return 0;
    });
}

void orc_sim_free(void *h)
{
    delete S(h);
}

int orc_sim_step(void *h)
{
    return guarded([&]() { return static_cast<int>(S(h)->step()); });
}

int orc_sim_propagate_until(void *h, double t)
{
    return guarded([&]() { return static_cast<int>(S(h)->propagate_until(t)); });
}

std::uint64_t orc_sim_nparts(void *h)
{
    return S(h)->get_nparts();
}

std::uint32_t orc_sim_order(void *h)
{
    return S(h)->prm.tcfg.order;
}

void orc_sim_get_state(void *h, double *out)
{
    std::memcpy(out, S(h)->m_state.data(), S(h)->m_state.size() * sizeof(double));
}

int orc_sim_set_state(void *h, const double *in, std::uint64_t nparts)
{
    return guarded([&]() {
        std::vector<double> st(in, in + nparts * 7u);
        S(h)->verify_state_vector(st);
        S(h)->m_state = std::move(st);
        return 0;
    });
}

// Raw overwrite without validation (mirrors writes through get_state_data()).
void orc_sim_poke_state(void *h, const double *in)
{
    std::memcpy(S(h)->m_state.data(), in, S(h)->m_state.size() * sizeof(double));
}

double orc_sim_get_time(void *h)
{
    return S(h)->m_time.hi;
}

void orc_sim_get_dtime(void *h, double *hi_lo)
{
    hi_lo[0] = S(h)->m_time.hi;
    hi_lo[1] = S(h)->m_time.lo;
}

void orc_sim_set_time(void *h, double t)
{
    S(h)->m_time = orc::dfloat(t);
}

void orc_sim_set_ct(void *h, double ct)
{
    S(h)->prm.ct = ct;
}

void orc_sim_set_n_par_ct(void *h, std::uint32_t n)
{
    S(h)->prm.n_par_ct = n;
}

void orc_sim_set_conj_thresh(void *h, double v)
{
    S(h)->prm.conj_thresh = v;
}

void orc_sim_set_min_coll_radius(void *h, double v)
{
    S(h)->prm.min_coll_radius = v;
}

void orc_sim_set_whitelists(void *h, const std::uint64_t *coll, std::uint64_t ncoll, const std::uint64_t *conj,
                            std::uint64_t nconj)
{
    S(h)->m_coll_whitelist.clear();
    S(h)->m_conj_whitelist.clear();
    for (std::uint64_t i = 0; i < ncoll; ++i) {
        S(h)->m_coll_whitelist.insert(coll[i]);
    }
    for (std::uint64_t i = 0; i < nconj; ++i) {
        S(h)->m_conj_whitelist.insert(conj[i]);
    }
}

std::uint64_t orc_sim_nconj(void *h)
{
    return S(h)->m_det_conj.size();
}

void orc_sim_get_conj(void *h, void *out)
{
    std::memcpy(out, S(h)->m_det_conj.data(), S(h)->m_det_conj.size() * sizeof(orc::conjunction));
}

void orc_sim_reset_conj(void *h)
{
    S(h)->m_det_conj.clear();
}

void orc_sim_interrupt_info(void *h, int *kind, std::uint64_t *i, std::uint64_t *j, double *t)
{
    *kind = S(h)->m_int_kind;
    *i = S(h)->m_int_i;
    *j = S(h)->m_int_j;
    *t = S(h)->m_int_t;
}

// ---- stage data of the last step -------------------------------------------------

std::uint32_t orc_sim_nchunks(void *h)
{
    return S(h)->nchunks;
}

double orc_sim_delta_t(void *h)
{
    return S(h)->delta_t;
}

void orc_sim_get_aabbs(void *h, std::uint32_t chunk, float *lbs, float *ubs)
{
    const auto n = S(h)->get_nparts();
    std::memcpy(lbs, &S(h)->lbs[chunk * n * 4u], n * 4u * sizeof(float));
    std::memcpy(ubs, &S(h)->ubs[chunk * n * 4u], n * 4u * sizeof(float));
}

// NOTE: the upper bound is the value AFTER the in-place nextafter bump of the Morton stage.
void orc_sim_get_global_aabb(void *h, std::uint32_t chunk, float *lb4, float *ub4)
{
    for (int i = 0; i < 4; ++i) {
        lb4[i] = S(h)->global_lb[chunk][i];
        ub4[i] = S(h)->global_ub[chunk][i];
    }
}

void orc_sim_get_codes(void *h, std::uint32_t chunk, std::uint64_t *codes, std::uint32_t *vidx)
{
    const auto n = S(h)->get_nparts();
    std::memcpy(codes, &S(h)->mcodes[chunk * n], n * sizeof(std::uint64_t));
    std::memcpy(vidx, &S(h)->vidx[chunk * n], n * sizeof(std::uint32_t));
}

void orc_sim_get_sorted(void *h, std::uint32_t chunk, std::uint64_t *srt_codes, float *srt_lbs, float *srt_ubs)
{
    const auto n = S(h)->get_nparts();
    std::memcpy(srt_codes, &S(h)->srt_mcodes[chunk * n], n * sizeof(std::uint64_t));
    std::memcpy(srt_lbs, &S(h)->srt_lbs[chunk * n * 4u], n * 4u * sizeof(float));
    std::memcpy(srt_ubs, &S(h)->srt_ubs[chunk * n * 4u], n * 4u * sizeof(float));
}

std::uint64_t orc_sim_bvh_size(void *h, std::uint32_t chunk)
{
    return S(h)->bvh_trees[chunk].size();
}

void orc_sim_get_bvh(void *h, std::uint32_t chunk, void *nodes)
{
    std::memcpy(nodes, S(h)->bvh_trees[chunk].data(), S(h)->bvh_trees[chunk].size() * sizeof(orc::bvh_node));
}

std::uint64_t orc_sim_bp_size(void *h, std::uint32_t chunk)
{
    return S(h)->bp_coll[chunk].size();
}

void orc_sim_get_bp(void *h, std::uint32_t chunk, std::uint32_t *pairs)
{
    const auto &bp = S(h)->bp_coll[chunk];
    for (std::size_t k = 0; k < bp.size(); ++k) {
        pairs[2u * k] = bp[k].first;
        pairs[2u * k + 1u] = bp[k].second;
    }
}

void orc_sim_get_active(void *h, char *coll_active, char *conj_active)
{
    const auto n = S(h)->get_nparts();
    std::memcpy(coll_active, S(h)->coll_active.data(), n);
    std::memcpy(conj_active, S(h)->conj_active.data(), n);
}

// Number of substeps per particle; returns the total.
std::uint64_t orc_sim_get_nsub(void *h, std::uint32_t *nsub)
{
    const auto n = S(h)->get_nparts();
    std::uint64_t tot = 0;
    for (std::size_t i = 0; i < n; ++i) {
        const auto s = S(h)->s_data[i].tcoords.size();
        if (nsub) {
            nsub[i] = static_cast<std::uint32_t>(s);
        }
        tot += s;
    }
    return tot;
}

// CSR export: tcs[total][7][order+1], tcoords[total][2] (hi, lo).
void orc_sim_get_tcs(void *h, double *tcs, double *tcoords)
{
    const auto n = S(h)->get_nparts();
    std::size_t off = 0, toff = 0;
    for (std::size_t i = 0; i < n; ++i) {
        const auto &sd = S(h)->s_data[i];
        std::memcpy(tcs + off, sd.tcs.data(), sd.tcs.size() * sizeof(double));
        off += sd.tcs.size();
        for (const auto &t : sd.tcoords) {
            tcoords[toff++] = t.hi;
            tcoords[toff++] = t.lo;
        }
    }
}

void orc_sim_get_final_state(void *h, double *out)
{
    std::memcpy(out, S(h)->final_state.data(), S(h)->final_state.size() * sizeof(double));
}

std::uint64_t orc_sim_ncoll(void *h)
{
    return S(h)->coll_vec.size();
}

// out: [ncoll] records of (u64 i, u64 j, f64 t).
void orc_sim_get_coll(void *h, void *out)
{
    std::memcpy(out, S(h)->coll_vec.data(), S(h)->coll_vec.size() * sizeof(orc::collision));
}

// Conjunctions detected in the last step before the append/erase logic, per chunk.
std::uint64_t orc_sim_step_nconj(void *h)
{
    std::uint64_t tot = 0;
    for (unsigned c = 0; c < S(h)->nchunks; ++c) {
        tot += S(h)->conj_vecs[c].size();
    }
    return tot;
}

void orc_sim_get_step_conj(void *h, void *out)
{
    auto *o = static_cast<orc::conjunction *>(out);
    for (unsigned c = 0; c < S(h)->nchunks; ++c) {
        for (const auto &cj : S(h)->conj_vecs[c]) {
            *o++ = cj;
        }
    }
}

// out[7]: propagate, morton_sort, bvh, broad, narrow, total, n_bp_pairs
void orc_sim_stage_times(void *h, double *out)
{
    const auto &t = S(h)->times;
    out[0] = t.propagate;
    out[1] = t.morton_sort;
    out[2] = t.bvh;
    out[3] = t.broad;
    out[4] = t.narrow;
    out[5] = t.total;
    out[6] = t.n_bp_pairs;
}

} // extern "C"

extern "C" {
double orc_det_pow(double x, double y)
{
    return orc::det_pow(x, y);
}
double orc_rhofac(std::uint32_t p)
{
    return orc::rhofac(p);
}
}

extern "C" {
// out[5]: max bisection depth, max work-list size, warnings, fex failures, fex passes (cumulative)
void orc_sim_rf_stats(void *h, std::uint64_t *out)
{
    auto *s = static_cast<orc::sim *>(h);
    out[0] = s->rf_max_depth;
    out[1] = s->rf_max_wlist;
    out[2] = s->rf_warn;
    out[3] = s->rf_fex_fail;
    out[4] = s->rf_fex_pass;
}
}
