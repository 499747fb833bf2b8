// oracle/orc_sim.hpp
//
// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT (see oracle/orc.hpp).
//
// CPU restatement of cascade::sim (include/cascade/sim.hpp:84-513) around the
// stage functions of orc.hpp: ctor validation (src/sim_basic.cpp:359-615),
// step() (src/sim_propagate.cpp:516-1345), dense_propagate (:1386-1503),
// propagate_until (:1505-1603), append_conj_data (:1644-1701).
#pragma once

#include "orc.hpp"

#include <chrono>
#include <unordered_set>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace orc
{

struct sim_params {
    double ct = 1;
    std::uint32_t n_par_ct = 1;
    double conj_thresh = 0;
    double min_coll_radius = 0;
    taylor_cfg tcfg;
    std::vector<double> pars; // [N][npars] runtime parameters (BSTAR when drag is on), zero if empty
};

struct stage_times {
    double propagate = 0, morton_sort = 0, bvh = 0, broad = 0, narrow = 0, total = 0;
    double n_bp_pairs = 0;
};

class sim
{
public:
    using size_type = std::size_t;

    std::vector<double> m_state; // [N][7]
    std::vector<double> m_pars;
    sim_params prm;
    std::unordered_set<size_type> m_coll_whitelist, m_conj_whitelist;
    std::vector<conjunction> m_det_conj;
    dfloat m_time;

    // interrupt info: kind 0 none, 1 collision (i, j), 2 reentry/exit (i), 3 nf state (i, t)
    int m_int_kind = 0;
    size_type m_int_i = 0, m_int_j = 0;
    double m_int_t = 0;

    // --- per-step scratch (sim_data, include/cascade/detail/sim_data.hpp:90-318) ---
    double delta_t = 0;
    unsigned nchunks = 0;
    std::vector<double> final_state;
    std::vector<step_data> s_data;
    std::vector<float> lbs, ubs, srt_lbs, srt_ubs;
    std::vector<std::uint64_t> mcodes, srt_mcodes;
    std::vector<std::uint32_t> vidx;
    std::vector<std::array<float, 4>> global_lb, global_ub;
    std::vector<std::vector<bvh_node>> bvh_trees;
    std::vector<std::vector<std::pair<std::uint32_t, std::uint32_t>>> bp_coll;
    std::vector<char> coll_active, conj_active;
    std::vector<collision> coll_vec;
    std::vector<std::vector<conjunction>> conj_vecs;
    std::vector<std::tuple<size_type, double, std::uint32_t>> ste_vec;
    std::vector<std::tuple<size_type, double>> err_nf_state_vec;
    stage_times times;
    std::uint64_t rf_max_depth = 0, rf_max_wlist = 0, rf_warn = 0, rf_fex_fail = 0, rf_fex_pass = 0;
    bool keep_stage_data = true;

    size_type get_nparts() const
    {
        return m_state.size() / 7u;
    }

    // finalise_ctor validation: src/sim_basic.cpp:359-501 (messages verbatim).
    sim(std::vector<double> state, const sim_params &p) : m_state(std::move(state)), prm(p)
    {
        if (m_state.size() % 7u != 0u) {
            throw std::invalid_argument("The size of the state vector is " + std::to_string(m_state.size())
                                        + ", which is not a multiple of 7");
        }
        if (!std::isfinite(prm.ct) || prm.ct <= 0) {
            throw std::invalid_argument("The collisional timestep must be finite and positive");
        }
        if (prm.n_par_ct == 0u) {
            throw std::invalid_argument(
                "The number of collisional timesteps to be processed in parallel cannot be zero");
        }
        if (!std::isfinite(prm.conj_thresh) || prm.conj_thresh < 0) {
            throw std::invalid_argument("The conjunction threshold value is invalid");
        }
        if (std::isnan(prm.min_coll_radius) || prm.min_coll_radius < 0) {
            throw std::invalid_argument("The minimum collisional radius cannot be NaN or negative");
        }
        prm.tcfg.order = order_from_tol(prm.tcfg.tol);
        verify_state_vector(m_state);
    }

    // src/sim_basic.cpp:686-760
    void verify_state_vector(const std::vector<double> &st) const
    {
        if (st.size() % 7u != 0u) {
            throw std::invalid_argument("The size of the state vector is " + std::to_string(st.size())
                                        + ", which is not a multiple of 7");
        }
        const auto nparts = st.size() / 7u;
        const auto &tc = prm.tcfg;
        for (size_type pidx = 0; pidx < nparts; ++pidx) {
            const double x = st[pidx * 7u], y = st[pidx * 7u + 1u], z = st[pidx * 7u + 2u];
            if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) {
                throw std::invalid_argument(
                    "An non-finite position was detected in the state vector of the particle at index "
                    + std::to_string(pidx));
            }
            if (tc.reentry_kind == 1) {
                if (x * x + y * y + z * z < tc.reentry[0] * tc.reentry[0]) {
                    throw std::invalid_argument("The particle at index " + std::to_string(pidx)
                                                + " is inside the spherical central body");
                }
            } else if (tc.reentry_kind == 2) {
                const double a = tc.reentry[0], b = tc.reentry[1], c = tc.reentry[2];
                if (x * x / (a * a) + y * y / (b * b) + z * z / (c * c) < 1) {
                    throw std::invalid_argument("The particle at index " + std::to_string(pidx)
                                                + " is inside the ellipsoidal central body");
                }
            }
            if (tc.with_exit()) {
                if (x * x + y * y + z * z >= tc.exit_radius * tc.exit_radius) {
                    throw std::invalid_argument("The particle at index " + std::to_string(pidx)
                                                + " is outside the exit radius");
                }
            }
            if (!std::isfinite(st[pidx * 7u + 3u]) || !std::isfinite(st[pidx * 7u + 4u])
                || !std::isfinite(st[pidx * 7u + 5u])) {
                throw std::invalid_argument(
                    "An non-finite velocity was detected in the state vector of the particle at index "
                    + std::to_string(pidx));
            }
            if (!std::isfinite(st[pidx * 7u + 6u]) || st[pidx * 7u + 6u] < 0) {
                throw std::invalid_argument("An invalid particle size of " + std::to_string(st[pidx * 7u + 6u])
                                            + " was detected for the particle at index " + std::to_string(pidx));
            }
        }
    }

    // --- step(): src/sim_propagate.cpp:516-1345 ---
    outcome step()
    {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        auto lap = [&t0]() {
            const auto t1 = clk::now();
            const double d = std::chrono::duration<double>(t1 - t0).count();
            t0 = t1;
            return d;
        };
        const auto t_begin = clk::now();

        if (get_nparts() == 0u) {
            throw std::invalid_argument("Cannot integrate a simulation with no particles");
        }
        nchunks = prm.n_par_ct;
        delta_t = prm.ct * nchunks;
        if (!std::isfinite(delta_t) || delta_t <= 0) {
            throw std::invalid_argument("An invalid superstep size of " + std::to_string(delta_t) + " was inferred");
        }
        const auto nparts = get_nparts();
        const auto order = prm.tcfg.order;
        const auto init_time = m_time;
        const double ct = prm.ct;
        const unsigned nchunks0 = nchunks;
        const double delta_t0 = delta_t;

        s_data.resize(nparts);
        lbs.resize(static_cast<size_type>(nchunks) * nparts * 4u);
        ubs.resize(lbs.size());
        srt_lbs.resize(lbs.size());
        srt_ubs.resize(lbs.size());
        mcodes.resize(static_cast<size_type>(nchunks) * nparts);
        srt_mcodes.resize(mcodes.size());
        vidx.resize(mcodes.size());
        final_state.resize(nparts * 6u);
        constexpr auto finf = std::numeric_limits<float>::infinity();
        global_lb.assign(nchunks, {finf, finf, finf, finf});
        global_ub.assign(nchunks, {-finf, -finf, -finf, -finf});
        bvh_trees.resize(nchunks);
        bp_coll.resize(nchunks);
        coll_active.resize(nparts);
        conj_active.resize(nparts);
        conj_vecs.resize(nchunks);
        ste_vec.clear();
        err_nf_state_vec.clear();

        const bool with_conj = (prm.conj_thresh != 0);
        const bool coll_wl_empty = m_coll_whitelist.empty();
        const bool conj_wl_empty = m_conj_whitelist.empty();

        // Propagation + AABBs (:650-1121).
        std::string err_msg;
#pragma omp parallel
        {
            taylor_work wk;
            std::vector<std::array<float, 4>> loc_lb(nchunks0, {finf, finf, finf, finf}),
                loc_ub(nchunks0, {-finf, -finf, -finf, -finf});
            std::vector<std::tuple<size_type, double, std::uint32_t>> loc_ste;
            std::vector<std::tuple<size_type, double>> loc_nf;
#pragma omp for schedule(dynamic, 64)
            for (std::int64_t ip = 0; ip < static_cast<std::int64_t>(nparts); ++ip) {
                const auto pidx = static_cast<size_type>(ip);
                try {
                    // :639-647
                    coll_active[pidx] = static_cast<char>((m_state[pidx * 7u + 6u] > prm.min_coll_radius)
                                                          && (coll_wl_empty || m_coll_whitelist.count(pidx) == 1u));
                    conj_active[pidx]
                        = static_cast<char>(with_conj && (conj_wl_empty || m_conj_whitelist.count(pidx) == 1u));
                    const std::uint32_t npars = prm.tcfg.dyn.npars();
                    wk.pars = (npars != 0u && (pidx + 1u) * npars <= prm.pars.size()) ? &prm.pars[pidx * npars] : nullptr;
                    wk.par0 = wk.pars ? wk.pars[0] : 0.;
                    const auto res = propagate_particle(prm.tcfg, wk, &m_state[pidx * 7u], init_time, delta_t0,
                                                        s_data[pidx], &final_state[pidx * 6u]);
                    if (res.status == 2) {
                        loc_nf.emplace_back(pidx, res.nf_last_t);
                        continue;
                    }
                    if (res.status == 1) {
                        loc_ste.emplace_back(pidx, res.ev_time, res.ev_idx);
                    }
                    for (unsigned chunk_idx = 0; chunk_idx < nchunks0; ++chunk_idx) {
                        const auto [cb, ce] = get_chunk_begin_end(chunk_idx, ct, nchunks0, delta_t0);
                        float *lb = &lbs[(static_cast<size_type>(chunk_idx) * nparts + pidx) * 4u];
                        float *ub = &ubs[(static_cast<size_type>(chunk_idx) * nparts + pidx) * 4u];
                        compute_particle_aabb(s_data[pidx], order, cb, ce, m_state[pidx * 7u + 6u], prm.conj_thresh,
                                              lb, ub, pidx);
                        for (int j = 0; j < 4; ++j) {
                            loc_lb[chunk_idx][j] = std::min(loc_lb[chunk_idx][j], lb[j]);
                            loc_ub[chunk_idx][j] = std::max(loc_ub[chunk_idx][j], ub[j]);
                        }
                    }
                } catch (const std::exception &e) {
#pragma omp critical
                    if (err_msg.empty()) {
                        err_msg = e.what();
                    }
                }
            }
#pragma omp critical
            {
                for (unsigned c = 0; c < nchunks0; ++c) {
                    for (int j = 0; j < 4; ++j) {
                        global_lb[c][j] = std::min(global_lb[c][j], loc_lb[c][j]);
                        global_ub[c][j] = std::max(global_ub[c][j], loc_ub[c][j]);
                    }
                }
                ste_vec.insert(ste_vec.end(), loc_ste.begin(), loc_ste.end());
                err_nf_state_vec.insert(err_nf_state_vec.end(), loc_nf.begin(), loc_nf.end());
            }
        }
        if (!err_msg.empty()) {
            throw std::invalid_argument(err_msg);
        }
        // Deterministic container order (the reference's concurrent_vector order is not).
        std::sort(ste_vec.begin(), ste_vec.end());
        std::sort(err_nf_state_vec.begin(), err_nf_state_vec.end());
        times.propagate = lap();

        // :1127-1145
        if (!err_nf_state_vec.empty()) {
            const auto nf_it = std::min_element(err_nf_state_vec.cbegin(), err_nf_state_vec.cend(),
                                                [](const auto &a, const auto &b) { return std::get<1>(a) < std::get<1>(b); });
            m_int_kind = 3;
            m_int_i = std::get<0>(*nf_it);
            m_int_t = std::get<1>(*nf_it);
            return outcome::err_nf_state;
        }

        // :1148-1171
        auto ste_it = ste_vec.cend();
        if (!ste_vec.empty()) {
            ste_it = std::min_element(ste_vec.cbegin(), ste_vec.cend(),
                                      [](const auto &a, const auto &b) { return std::get<1>(a) < std::get<1>(b); });
            delta_t = std::min(std::get<1>(*ste_it), delta_t);
            const double nc = std::ceil(delta_t / ct);
            if (!(nc >= 1.)) {
                throw std::invalid_argument("The recomputed number of chunks after the triggering of a stopping terminal "
                                            "event cannot be zero (this likely indicates that the simulation was restarted "
                                            "without resolving a stopping terminal event)");
            }
            nchunks = static_cast<unsigned>(nc);
        }

        // Morton encoding and sorting (:399-511).
        for (unsigned c = 0; c < nchunks; ++c) {
            const size_type off = static_cast<size_type>(c) * nparts;
            morton_encode_chunk(nparts, &lbs[off * 4u], &ubs[off * 4u], global_lb[c].data(), global_ub[c].data(),
                                &mcodes[off]);
            sort_indices(nparts, &mcodes[off], &vidx[off]);
#pragma omp parallel for
            for (std::int64_t ip = 0; ip < static_cast<std::int64_t>(nparts); ++ip) {
                const auto pidx = static_cast<size_type>(ip);
                const auto sorted_idx = vidx[off + pidx];
                for (int i = 0; i < 4; ++i) {
                    srt_lbs[(off + pidx) * 4u + i] = lbs[(off + sorted_idx) * 4u + i];
                    srt_ubs[(off + pidx) * 4u + i] = ubs[(off + sorted_idx) * 4u + i];
                }
                srt_mcodes[off + pidx] = mcodes[off + sorted_idx];
            }
        }
        times.morton_sort = lap();

        // BVH (src/sim_bvh.cpp:89-496).
#pragma omp parallel for schedule(dynamic, 1)
        for (std::int64_t c = 0; c < static_cast<std::int64_t>(nchunks); ++c) {
            const size_type off = static_cast<size_type>(c) * nparts;
            bvh_build(nparts, &srt_mcodes[off], &srt_lbs[off * 4u], &srt_ubs[off * 4u], bvh_trees[c]);
        }
        times.bvh = lap();

        // Broad phase (src/sim_broad_phase.cpp:44-218).
        double tot_bp = 0;
        for (unsigned c = 0; c < nchunks; ++c) {
            const size_type off = static_cast<size_type>(c) * nparts;
            auto &bp = bp_coll[c];
            bp.clear();
#pragma omp parallel
            {
                std::vector<std::pair<std::uint32_t, std::uint32_t>> loc;
#pragma omp for schedule(dynamic, 256) nowait
                for (std::int64_t ip = 0; ip < static_cast<std::int64_t>(nparts); ++ip) {
                    broad_phase(nparts, bvh_trees[c].data(), &srt_lbs[off * 4u], &srt_ubs[off * 4u], &vidx[off],
                                coll_active.data(), conj_active.data(), loc, static_cast<size_type>(ip),
                                static_cast<size_type>(ip) + 1u);
                }
#pragma omp critical
                bp.insert(bp.end(), loc.begin(), loc.end());
            }
            std::sort(bp.begin(), bp.end());
            tot_bp += static_cast<double>(bp.size());
        }
        times.n_bp_pairs = tot_bp;
        times.broad = lap();

        // Narrow phase (src/sim_narrow_phase.cpp:495-1023).
        const double conj_thresh2 = prm.conj_thresh * prm.conj_thresh;
        coll_vec.clear();
        for (unsigned c = 0; c < nchunks; ++c) {
            conj_vecs[c].clear();
            // NOTE: uses the (possibly redefined) delta_t / nchunks, :560.
            const auto [cb, ce] = get_chunk_begin_end(c, ct, nchunks, delta_t);
            const dfloat chunk_begin(cb), chunk_end(ce);
            const auto &bpc = bp_coll[c];
#pragma omp parallel
            {
                np_scratch sc;
                std::vector<collision> loc_coll;
                std::vector<conjunction> loc_conj;
#pragma omp for schedule(dynamic, 512) nowait
                for (std::int64_t k = 0; k < static_cast<std::int64_t>(bpc.size()); ++k) {
                    const auto pi = bpc[k].first, pj = bpc[k].second;
                    narrow_phase_pair(pi, pj, s_data[pi], s_data[pj], order, m_state[pi * 7u + 6u],
                                      m_state[pj * 7u + 6u], coll_active[pi], coll_active[pj], conj_active[pi],
                                      conj_active[pj], chunk_begin, chunk_end, init_time, conj_thresh2, sc, loc_coll,
                                      loc_conj);
                }
#pragma omp critical
                {
                    rf_max_depth = std::max(rf_max_depth, sc.st.max_depth);
                    rf_max_wlist = std::max(rf_max_wlist, sc.st.max_wlist);
                    rf_warn += sc.st.n_warn;
                    rf_fex_fail += sc.st.n_fex_fail;
                    rf_fex_pass += sc.st.n_fex_pass;
                    coll_vec.insert(coll_vec.end(), loc_coll.begin(), loc_coll.end());
                    conj_vecs[c].insert(conj_vecs[c].end(), loc_conj.begin(), loc_conj.end());
                }
            }
        }
        // Deterministic container order: (time, i, j).
        std::sort(coll_vec.begin(), coll_vec.end(), [](const collision &a, const collision &b) {
            return std::tie(a.time, a.i, a.j) < std::tie(b.time, b.i, b.j);
        });
        times.narrow = lap();

        // Outcome resolution (:1189-1345).
        outcome oc = outcome::success;
        const double *interrupt_time = nullptr;
        auto coll_it = coll_vec.cend();
        if (!coll_vec.empty()) {
            coll_it = std::min_element(coll_vec.cbegin(), coll_vec.cend(),
                                       [](const collision &a, const collision &b) { return a.time < b.time; });
            interrupt_time = &coll_it->time;
            oc = outcome::collision;
        }
        if (!ste_vec.empty()) {
            const double *new_itime_ptr = &std::get<1>(*ste_it);
            if (interrupt_time == nullptr || *new_itime_ptr < *interrupt_time) {
                interrupt_time = new_itime_ptr;
                if (prm.tcfg.with_exit() && prm.tcfg.exit_event_idx() == std::get<2>(*ste_it)) {
                    oc = outcome::exit;
                } else {
                    oc = outcome::reentry;
                }
            }
        }

        if (oc == outcome::success) {
            m_time = m_time + delta_t0;
            copy_from_final_state();
            m_int_kind = 0;
            if (with_conj) {
                append_conj_data();
            }
        } else {
            dense_propagate(*interrupt_time);
            m_time = m_time + *interrupt_time;
            copy_from_final_state();
            switch (oc) {
                case outcome::collision:
                    m_int_kind = 1;
                    m_int_i = coll_it->i;
                    m_int_j = coll_it->j;
                    break;
                default:
                    m_int_kind = 2;
                    m_int_i = std::get<0>(*ste_it);
            }
            if (with_conj) {
                const auto first_new = append_conj_data();
                const auto conj_it = std::lower_bound(m_det_conj.begin() + static_cast<std::ptrdiff_t>(first_new),
                                                      m_det_conj.end(), m_time.hi,
                                                      [](const conjunction &c, double tgt) { return c.time < tgt; });
                m_det_conj.erase(conj_it, m_det_conj.end());
            }
        }
        times.total = std::chrono::duration<double>(clk::now() - t_begin).count();
        return oc;
    }

    // append_conj_data: :1644-1701 (sorted by (time, i, j) for determinism).
    size_type append_conj_data()
    {
        const auto first = m_det_conj.size();
        for (unsigned c = 0; c < nchunks; ++c) {
            m_det_conj.insert(m_det_conj.end(), conj_vecs[c].begin(), conj_vecs[c].end());
        }
        std::sort(m_det_conj.begin() + static_cast<std::ptrdiff_t>(first), m_det_conj.end(),
                  [](const conjunction &a, const conjunction &b) {
                      return std::tie(a.time, a.i, a.j) < std::tie(b.time, b.i, b.j);
                  });
        return first;
    }

    // :1610-1630
    void copy_from_final_state()
    {
        const auto nparts = get_nparts();
        for (size_type p = 0; p < nparts; ++p) {
            for (int j = 0; j < 6; ++j) {
                m_state[p * 7u + j] = final_state[p * 6u + j];
            }
        }
    }

    // dense_propagate: :1386-1503
    void dense_propagate(double t)
    {
        const auto order = prm.tcfg.order;
        const auto nparts = get_nparts();
        const dfloat dt(t);
        const size_type np1 = order + 1u;
        for (size_type pidx = 0; pidx < nparts; ++pidx) {
            const auto &sd = s_data[pidx];
            const auto &tcoords = sd.tcoords;
            if (tcoords.empty()) {
                throw std::invalid_argument("The computation of dense_propagate() for particle " + std::to_string(pidx)
                                            + " could not be performed because no timesteps have been taken for this particle");
            }
            auto it = std::lower_bound(tcoords.begin(), tcoords.end(), dt);
            it -= (it == tcoords.end());
            const dfloat ss_start = (it == tcoords.begin()) ? dfloat(0.) : *(it - 1);
            const double eval_tm = static_cast<double>(dt - ss_start);
            const auto ss_idx = static_cast<size_type>(it - tcoords.begin());
            const double *base = sd.tcs.data() + ss_idx * 7u * np1;
            for (int c = 0; c < 6; ++c) {
                final_state[pidx * 6u + c] = poly_eval(base + c * np1, eval_tm, order);
            }
        }
    }

    // propagate_until: :1505-1603
    outcome propagate_until(double t)
    {
        if (!std::isfinite(t) || dfloat(t) < m_time) {
            throw std::invalid_argument("The final time passed to the propagate_until() function must be finite and not less "
                                        "than the current simulation time, but a value of "
                                        + std::to_string(t) + " was provided instead");
        }
        if (dfloat(t) == m_time) {
            return outcome::time_limit;
        }
        const dfloat final_t(t);
        while (true) {
            const auto orig_t = m_time;
            const auto cur_oc = step();
            if (cur_oc == outcome::success) {
                if (m_time >= final_t) {
                    dense_propagate(static_cast<double>(final_t - orig_t));
                    m_time = final_t;
                    copy_from_final_state();
                    return outcome::time_limit;
                }
            } else if (cur_oc != outcome::err_nf_state) {
                if (m_time > final_t) {
                    dense_propagate(static_cast<double>(final_t - orig_t));
                    m_time = final_t;
                    copy_from_final_state();
                    m_int_kind = 0;
                    return outcome::time_limit;
                } else {
                    return cur_oc;
                }
            } else {
                return cur_oc;
            }
        }
    }
};

} // namespace orc
