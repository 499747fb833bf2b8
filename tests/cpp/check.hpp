// tests/cpp/check.hpp -- minimal assertion helpers for the C++ host tests (the reference uses
// Catch2, which is not available here).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

static int g_checks = 0;

#define REQUIRE(...)                                                                                                   \
    do {                                                                                                               \
        ++g_checks;                                                                                                    \
        if (!(__VA_ARGS__)) {                                                                                               \
            std::fprintf(stderr, "%s:%d: REQUIRE failed: %s\n", __FILE__, __LINE__, #__VA_ARGS__);                         \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

#define REQUIRE_THROWS_MSG(expr, extype, msg)                                                                          \
    do {                                                                                                               \
        ++g_checks;                                                                                                    \
        bool thrown_ = false;                                                                                          \
        try {                                                                                                          \
            (void)(expr);                                                                                              \
        } catch (const extype &e_) {                                                                                   \
            thrown_ = true;                                                                                            \
            if (std::string(e_.what()) != std::string(msg)) {                                                          \
                std::fprintf(stderr, "%s:%d: wrong message:\n  got:      %s\n  expected: %s\n", __FILE__, __LINE__,    \
                             e_.what(), std::string(msg).c_str());                                                     \
                std::exit(1);                                                                                          \
            }                                                                                                          \
        }                                                                                                              \
        if (!thrown_) {                                                                                                \
            std::fprintf(stderr, "%s:%d: no " #extype " thrown by: %s\n", __FILE__, __LINE__, #expr);                  \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

// Keplerian elements {a, e, i, om, Om, nu} -> Cartesian, argument order of test/keputils.hpp:34.
inline std::pair<std::vector<double>, std::vector<double>> kep_to_cart(const std::vector<double> &kep, double mu)
{
    const double a = kep[0], e = kep[1], i = kep[2], om = kep[3], Om = kep[4], nu = kep[5];
    const double p = a * (1 - e * e);
    const double r = p / (1 + e * std::cos(nu));
    const double h = std::sqrt(mu * p);
    const double x = r * (std::cos(Om) * std::cos(om + nu) - std::sin(Om) * std::sin(om + nu) * std::cos(i));
    const double y = r * (std::sin(Om) * std::cos(om + nu) + std::cos(Om) * std::sin(om + nu) * std::cos(i));
    const double z = r * (std::sin(i) * std::sin(om + nu));
    const double vx = x * h * e / (r * p) * std::sin(nu)
                      - h / r * (std::cos(Om) * std::sin(om + nu) + std::sin(Om) * std::cos(om + nu) * std::cos(i));
    const double vy = y * h * e / (r * p) * std::sin(nu)
                      - h / r * (std::sin(Om) * std::sin(om + nu) - std::cos(Om) * std::cos(om + nu) * std::cos(i));
    const double vz = z * h * e / (r * p) * std::sin(nu) + h / r * std::sin(i) * std::cos(om + nu);
    return {{x, y, z}, {vx, vy, vz}};
}
