// tests/cpp/test_sim_api.cpp -- CPU-only: construction, named arguments, getters/setters, error
// texts, particle removal, state replacement of cascade::sim. Restates test/sim_api.cpp of the
// reference (defaults :32-51, kwargs :54-83, copy/move :86-187, error strings :191-236,
// remove_particles :239-304, set_new_state_pars :306-411, setters :413-462) for the parts that
// do not need heyoka expressions. No step() is taken: no GPU needed.
#include <limits>
#include <sstream>
#include <variant>

#include <cascade/sim.hpp>

#include "check.hpp"

using namespace cascade;

int main()
{
    using whitelist_t = sim::whitelist_t;
    // Defaults.
    {
        sim s;
        REQUIRE(s.get_nparts() == 0u);
        REQUIRE(s.get_ct() == 1.);
        REQUIRE(s.get_n_par_ct() == 1u);
        REQUIRE(s.get_tol() == std::numeric_limits<double>::epsilon());
        REQUIRE(!s.get_high_accuracy());
        REQUIRE(!s.get_compact_mode());
        REQUIRE(s.get_npars() == 0u);
        REQUIRE(std::get<double>(s.get_reentry_radius()) == 0.);
        REQUIRE(s.get_exit_radius() == 0.);
        REQUIRE(s.get_conj_thresh() == 0.);
        REQUIRE(s.get_min_coll_radius() == 0.);
        REQUIRE(s.get_coll_whitelist().empty());
        REQUIRE(s.get_conj_whitelist().empty());
        REQUIRE(s.get_conjunctions().empty());
        REQUIRE(!s.get_interrupt_info());
        REQUIRE(s.get_time() == 0.);
        REQUIRE(s.get_dyn() == dynamics::kepler());
    }
    // Named arguments round trip.
    {
        sim s({1., .001, .001, .001, 1., .001, .001}, .5, kw::dyn = dynamics::kepler(2.), kw::reentry_radius = .1,
              kw::exit_radius = 10., kw::tol = 1e-12, kw::high_accuracy = true, kw::compact_mode = true,
              kw::n_par_ct = 3, kw::conj_thresh = .1, kw::min_coll_radius = .2, kw::coll_whitelist = whitelist_t{1, 2},
              kw::conj_whitelist = whitelist_t{3, 4});
        REQUIRE(s.get_nparts() == 1u);
        REQUIRE(s.get_ct() == .5);
        REQUIRE(s.get_dyn().mu == 2.);
        REQUIRE(std::get<double>(s.get_reentry_radius()) == .1);
        REQUIRE(s.get_exit_radius() == 10.);
        REQUIRE(s.get_tol() == 1e-12);
        REQUIRE(s.get_high_accuracy());
        REQUIRE(s.get_compact_mode());
        REQUIRE(s.get_n_par_ct() == 3u);
        REQUIRE(s.get_conj_thresh() == .1);
        REQUIRE(s.get_min_coll_radius() == .2);
        REQUIRE(s.get_coll_whitelist() == whitelist_t{1, 2});
        REQUIRE(s.get_conj_whitelist() == whitelist_t{3, 4});

        // Copy / move keep everything.
        sim s2(s);
        REQUIRE(s2.get_state() == s.get_state());
        REQUIRE(s2.get_n_par_ct() == 3u);
        REQUIRE(s2.get_coll_whitelist() == whitelist_t{1, 2});
        REQUIRE(s2.get_state_data() != s.get_state_data());
        sim s3(std::move(s2));
        REQUIRE(s3.get_conj_whitelist() == whitelist_t{3, 4});
        REQUIRE(s3.get_exit_radius() == 10.);
        sim s4;
        s4 = s3;
        REQUIRE(s4.get_ct() == .5);
        s4 = std::move(s3);
        REQUIRE(s4.get_tol() == 1e-12);

        sim e({1., .001, .001, .001, 1., .001, .001}, .5, kw::reentry_radius = std::vector{.1, .2, .3});
        REQUIRE((std::get<std::vector<double>>(e.get_reentry_radius()) == std::vector{.1, .2, .3}));
    }
    // Error modes (exact texts of the reference).
    REQUIRE_THROWS_MSG(sim({1., .001, .001, .001, 1., .001}, .5), std::invalid_argument,
                       "The size of the state vector is 6, which is not a multiple of 7");
    REQUIRE_THROWS_MSG(sim({}, 0), std::invalid_argument,
                       "The collisional timestep must be finite and positive, but it is 0 instead");
    REQUIRE_THROWS_MSG(sim({}, std::numeric_limits<double>::infinity()), std::invalid_argument,
                       "The collisional timestep must be finite and positive, but it is inf instead");
    REQUIRE_THROWS_MSG(
        sim({}, .5, kw::reentry_radius = std::vector{.1}), std::invalid_argument,
        "The reentry_radius argument must be either a scalar (for a spherical central body) or a vector of 3 "
        "elements (for a triaxial ellipsoid), but instead it is a vector of 1 element(s)");
    REQUIRE_THROWS_MSG(
        sim({}, .5, kw::reentry_radius = std::vector{1., 2., 0.}), std::invalid_argument,
        "A non-finite or non-positive value was detected among the 3 semiaxes of the central body: [1, 2, 0]");
    REQUIRE_THROWS_MSG(sim({}, .5, kw::reentry_radius = -1), std::invalid_argument,
                       "The reentry radius must be finite and non-negative, but it is -1 instead");
    REQUIRE_THROWS_MSG(sim({}, .5, kw::exit_radius = -1), std::invalid_argument,
                       "The exit radius must be finite and non-negative, but it is -1 instead");
    REQUIRE_THROWS_MSG(sim({}, .5, kw::n_par_ct = 0), std::invalid_argument,
                       "The number of collisional timesteps to be processed in parallel cannot be zero");
    REQUIRE_THROWS_MSG(sim({}, .5, kw::conj_thresh = -1.), std::invalid_argument,
                       "The conjunction threshold value -1 is invalid: it must be finite and non-negative");
    REQUIRE_THROWS_MSG(sim({}, .5, kw::min_coll_radius = -1.), std::invalid_argument,
                       "The minimum collisional radius cannot be NaN or negative, but the invalid value -1 was provided");
    REQUIRE_THROWS_MSG(sim({1., 0., 0., 0., 1., 0., 0.}, .5, kw::pars = std::vector{1.}), std::invalid_argument,
                       "The input array of parameter values must be empty when the number of parameters in the dynamics "
                       "is zero");
    // State validation (src/sim_basic.cpp:686-760; test/invalid_state.cpp:46-54).
    REQUIRE_THROWS_MSG(sim({1., 0., 0., 0., 1., 0., -1.}, .5), std::invalid_argument,
                       "An invalid particle size of -1 was detected for the particle at index 0");
    REQUIRE_THROWS_MSG(sim({.1, 0., 0., 0., 1., 0., 0.}, .5, kw::reentry_radius = 1.), std::invalid_argument,
                       "The particle at index 0 is inside the spherical central body");
    REQUIRE_THROWS_MSG(sim({.1, 0., 0., 0., 1., 0., 0.}, .5, kw::reentry_radius = std::vector{1., 1., 1.}),
                       std::invalid_argument, "The particle at index 0 is inside the ellipsoidal central body");
    REQUIRE_THROWS_MSG(sim({20., 0., 0., 0., 1., 0., 0.}, .5, kw::exit_radius = 10.), std::invalid_argument,
                       "The particle at index 0 is outside the exit radius 10");
    REQUIRE_THROWS_MSG(sim().step(), std::invalid_argument, "Cannot integrate a simulation with no particles");

    // remove_particles.
    {
        sim s({1., 0., 0., 0., 1., 0., .1, 2., 0., 0., 0., 1., 0., .2, 3., 0., 0., 0., 1., 0., .3}, .5);
        s.remove_particles({});
        REQUIRE(s.get_nparts() == 3u);
        const auto old_ptr = s._get_state_ptr();
        s.remove_particles({1, 1});
        REQUIRE(s.get_nparts() == 2u);
        REQUIRE(s.get_state()[0] == 1. && s.get_state()[7] == 3.);
        REQUIRE(old_ptr->size() == 21u); // the old vector stays alive and untouched
        REQUIRE_THROWS_MSG(s.remove_particles({0, 2}), std::invalid_argument,
                           "An invalid vector of indices was passed to the function for particle removal: [0, 2]");
        s.remove_particles({0, 1});
        REQUIRE(s.get_nparts() == 0u);
    }
    // set_new_state_pars.
    {
        sim s({1., 0., 0., 0., 1., 0., .1}, .5, kw::exit_radius = 10.);
        s.set_new_state_pars({2., 0., 0., 0., 1., 0., .1, 3., 0., 0., 0., 1., 0., .1});
        REQUIRE(s.get_nparts() == 2u);
        REQUIRE_THROWS_MSG(s.set_new_state_pars({20., 0., 0., 0., 1., 0., .1}), std::invalid_argument,
                           "The particle at index 0 is outside the exit radius 10");
        REQUIRE(s.get_nparts() == 2u);
        REQUIRE_THROWS_MSG(s.set_new_state_pars({2., 0., 0., 0., 1., 0., .1}, {1.}), std::invalid_argument,
                           "The input array of parameter values must be empty when the number of parameters in the "
                           "dynamics is zero");
    }
    // Setters.
    {
        sim s;
        s.set_time(1.5);
        REQUIRE(s.get_time() == 1.5);
        REQUIRE_THROWS_MSG(s.set_time(std::numeric_limits<double>::infinity()), std::invalid_argument,
                           "Cannot set the simulation time to the non-finite value inf");
        s.set_ct(.1);
        REQUIRE(s.get_ct() == .1);
        REQUIRE_THROWS_MSG(s.set_ct(-1.), std::invalid_argument,
                           "The collisional timestep must be finite and positive, but it is -1 instead");
        s.set_n_par_ct(7);
        REQUIRE(s.get_n_par_ct() == 7u);
        REQUIRE_THROWS_MSG(s.set_n_par_ct(0), std::invalid_argument,
                           "The number of collisional timesteps to be processed in parallel cannot be zero");
        s.set_conj_thresh(.3);
        REQUIRE(s.get_conj_thresh() == .3);
        s.set_min_coll_radius(std::numeric_limits<double>::infinity());
        REQUIRE(std::isinf(s.get_min_coll_radius()));
        s.set_coll_whitelist({5});
        s.set_conj_whitelist({6});
        REQUIRE(s.get_coll_whitelist() == whitelist_t{5});
        REQUIRE(s.get_conj_whitelist() == whitelist_t{6});
        std::ostringstream oss;
        oss << s;
        REQUIRE(oss.str() == "Total number of particles: 0\nCollisional timestep     : 0.1\n");
    }
    std::printf("test_sim_api: %d checks passed\n", g_checks);
    // Dynamics given as expressions (test/sim_api.cpp:198-212; benchmark/haumea.cpp:114-139).
    {
        using eqs_t = std::vector<std::pair<expression, expression>>;
        const std::vector<double> st = {1., .001, .001, .001, 1., .001, .001};
        REQUIRE_THROWS_MSG(sim(st, .5, kw::dyn = eqs_t{{}}), std::invalid_argument,
                           "6 dynamical equations are expected, but 1 were provided instead");
        auto dyn = dynamics::kepler();
        REQUIRE(dyn.kind == dynamics::dyn_t::kepler && dyn.size() == 6u);
        dyn[0].first = "foo"_var;
        REQUIRE(dyn.kind == dynamics::dyn_t::tape); // touched: no longer the built-in right-hand side
        REQUIRE_THROWS_MSG(sim(st, .5, kw::dyn = dyn), std::invalid_argument,
                           "The LHS of the dynamics at index 0 must be a variable named \"x\", but instead it "
                           "is the expression \"foo\"");
        dyn[0].first = "x"_var;
        dyn[0].second += "a"_var;
        REQUIRE_THROWS_MSG(sim(st, .5, kw::dyn = dyn), std::invalid_argument,
                           "The RHS of the differential equation for the variable \"x\" contains the invalid variables "
                           "[\"a\"] (the allowed variables are [\"x\", \"y\", \"z\", \"vx\", \"vy\", \"vz\"])");
        // A perturbation added the way the reference's drivers do, with a runtime parameter.
        auto d2 = dynamics::kepler(3.);
        const auto [x, y, z] = make_vars("x", "y", "z");
        const auto pert = par(1) * diff(pow(x * x + y * y + z * z, -1.5) * sin(time_expr), z);
        d2[5].second += pert;
        sim s(st, .5, kw::dyn = d2, kw::pars = std::vector<double>{0.25, 0.5});
        REQUIRE(s.get_npars() == 2u);
        REQUIRE(s.get_pars() == (std::vector<double>{0.25, 0.5}));
        REQUIRE(s.get_dyn().kind == dynamics::dyn_t::tape);
        REQUIRE(s.get_dyn() == dynamics::dyn_t(d2));
        REQUIRE(!(s.get_dyn() == dynamics::kepler(3.)));
        // The built-in factories keep their tag until touched.
        REQUIRE(dynamics::simple_earth(true, false, false, false).kind == (dynamics::dyn_t::j2 | dynamics::dyn_t::drag));
        REQUIRE(dynamics::simple_earth().kind == dynamics::dyn_t::tape); // C22S22 is on by default
        REQUIRE(dynamics::simple_earth(true, true, true, true, true, true, true, true).npars == 2u);
        REQUIRE_THROWS_MSG(dynamics::simple_earth(true, false, false, true, false, false, false, true, true), std::invalid_argument,
                           "simple_earth(): the thermonets density model (a neural network evaluated by heyoka) is not "
                           "available in this build; use the exponential density fit");
    }
    return 0;
}
