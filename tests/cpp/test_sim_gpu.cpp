// tests/cpp/test_sim_gpu.cpp -- needs an MI355X. Known-answer tests of the reference run through
// the C++ class: test/single_collision.cpp:34-115 (polar collision, propagate_until before and
// after it), cascade.py/test.py:155-184 (conjunction counts 0 / 1 / 6 from literal initial
// conditions), test/reentry.cpp / domain_exit.cpp (radius hit to 1e-14), test/invalid_state.cpp.
#include <cmath>
#include <limits>

#include <cascade/sim.hpp>

#include "check.hpp"

using namespace cascade;

int main()
{
    const double pi = 3.14159265358979323846;
    const double psize = 1.57e-8;
    const auto [x1, v1] = kep_to_cart({1.05, .05, pi / 2, 0, 1.23, 0}, 1);
    const auto [x2, v2] = kep_to_cart({1.05, .05, pi / 2, 0, 4.56, 0}, 1);
    const std::vector<double> s0{x1[0], x1[1], x1[2], v1[0], v1[1], v1[2], psize,
                                 x2[0], x2[1], x2[2], v2[0], v2[1], v2[2], psize};
    for (auto n_par_ct : {1u, 3u}) {
        // single collision polar
        {
            sim s(s0, 0.23, kw::n_par_ct = n_par_ct);
            while (s.step() != outcome::collision) {
            }
            const auto &ii = std::get<0>(*s.get_interrupt_info());
            REQUIRE(ii[0] == 0u && ii[1] == 1u);
            const double *p = s.get_state_data();
            const double dx = p[0] - p[7], dy = p[1] - p[8], dz = p[2] - p[9];
            REQUIRE(dx * dx + dy * dy + dz * dz - 4 * psize * psize < std::numeric_limits<double>::epsilon() * 10);
        }
        // single collision propagate_until
        {
            sim s(s0, 0.23, kw::n_par_ct = n_par_ct);
            auto oc = s.propagate_until(10);
            REQUIRE(oc == outcome::collision);
            REQUIRE(s.get_time() < 9);
            const double *p = s.get_state_data();
            const double dx = p[0] - p[7], dy = p[1] - p[8], dz = p[2] - p[9];
            REQUIRE(dx * dx + dy * dy + dz * dz - 4 * psize * psize < std::numeric_limits<double>::epsilon() * 10);

            std::copy(s0.begin(), s0.end(), s.get_state_data());
            s.set_time(0);
            oc = s.propagate_until(0.5);
            REQUIRE(oc == outcome::time_limit);
            REQUIRE(!s.get_interrupt_info());
            REQUIRE(s.get_time() == .5);

            std::copy(s0.begin(), s0.end(), s.get_state_data());
            s.set_time(0);
            oc = s.propagate_until(1.57);
            REQUIRE(oc == outcome::time_limit);
            REQUIRE(std::abs(s.get_time() - 1.57) <= 1e-15);
            REQUIRE(!s.get_interrupt_info());
        }
    }
    // Conjunction counts from literal initial conditions (cascade.py/test.py:162-184).
    {
        sim s({0.3342377271241684, 0.942488801930755, 0.0, 0.009424730938644767, -0.003342321565239028,
               0.9999500004176654, psize, -0.15179985849820377, -0.988411251938142, 0.0, 6.052273379648475e-17,
               -9.295060541060909e-18, 1.000000000001, psize},
              0.23, kw::conj_thresh = psize * 100000000);
        const auto cv0 = s._get_conjunctions_ptr();
        REQUIRE(cv0->empty());
        s.propagate_until(2.0);
        const auto cv1 = s._get_conjunctions_ptr();
        REQUIRE(cv1->size() == 1u);
        const auto first = (*cv1)[0];
        s.propagate_until(20.0);
        const auto cv2 = s._get_conjunctions_ptr();
        REQUIRE(cv2->size() == 6u);
        // Older handles stay valid and their records unchanged (the vector is either appended to
        // in place or replaced by a larger one, src/sim_narrow_phase.cpp:999-1011).
        REQUIRE(!cv1->empty());
        REQUIRE(((*cv1)[0].time == first.time && (*cv1)[0].dist == first.dist));
        for (std::size_t k = 1; k < cv2->size(); ++k) {
            REQUIRE((*cv2)[k - 1].time <= (*cv2)[k].time);
        }
        s.reset_conjunctions();
        REQUIRE(s.get_conjunctions().empty());
        REQUIRE(cv2->size() == 6u);
    }
    // Reentry on a sphere and domain exit, radius hit to 1e-14 (test/reentry.cpp, domain_exit.cpp).
    {
        const auto [x, v] = kep_to_cart({1., .5, 0.3, 0.1, 0.2, pi}, 1); // apocentre 1.5, pericentre 0.5
        sim s({x[0], x[1], x[2], v[0], v[1], v[2], psize}, 0.23, kw::reentry_radius = .75);
        outcome oc;
        while ((oc = s.step()) == outcome::success) {
        }
        REQUIRE(oc == outcome::reentry);
        REQUIRE(std::get<1>(*s.get_interrupt_info()) == 0u);
        const double *p = s.get_state_data();
        REQUIRE(std::abs(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - .75 * .75) < 1e-14);
        // Possible re-trigger of the same event if it is not resolved (test/reentry.cpp:60-66).
        try {
            s.step();
        } catch (const std::invalid_argument &ia) {
            REQUIRE(std::string(ia.what()).find("The recomputed number of chunks after the triggering of a stopping terminal")
                    != std::string::npos);
        }

        const auto [xe, ve] = kep_to_cart({1., .5, 0.3, 0.1, 0.2, 0.}, 1);
        sim se({xe[0], xe[1], xe[2], ve[0], ve[1], ve[2], psize}, 0.23, kw::exit_radius = 1.25, kw::n_par_ct = 3);
        while ((oc = se.step()) == outcome::success) {
        }
        REQUIRE(oc == outcome::exit);
        p = se.get_state_data();
        REQUIRE(std::abs(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] - 1.25 * 1.25) < 1e-14);
    }
    // Invalid states (test/invalid_state.cpp:46-63): nothing changes.
    {
        sim s(s0, 0.23);
        s.get_state_data()[6] = -1;
        REQUIRE_THROWS_MSG(s.step(), std::invalid_argument,
                           "An invalid particle size of -1 was detected for the particle at index 0");
        s.get_state_data()[6] = psize;
        s.get_state_data()[0] = std::numeric_limits<double>::infinity();
        const auto st = s.get_state();
        REQUIRE(s.step() == outcome::err_nf_state);
        REQUIRE(s.get_time() == 0.);
        REQUIRE(s.get_state() == st);
        REQUIRE(std::get<0>(std::get<2>(*s.get_interrupt_info())) == 0u);
    }
    std::printf("test_sim_gpu: %d checks passed\n", g_checks);
    return 0;
}
