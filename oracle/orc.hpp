// oracle/orc.hpp
//
// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT.
//
// CPU restatement (plain C++17, no third-party dependencies) of the reference's
// per-superstep pipeline cascade::sim::step(), used as the parity oracle for the
// HIP kernels and (bench.py cpu_baseline leg) as the timed CPU port. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline may link or call this code.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference). Arithmetic that lives in third-party code absent from the
// reference tree is restated from its published algorithm:
//   * heyoka >= 7.4 (cascade_devel.yml:14): Taylor coefficient recurrences, order
//     and step-size rule, propagate_for clamping, terminal-event detection, dfloat,
//     fex_check, poly_rtscc, poly_translate_1.
//   * Boost.Math >= 1.73 toms748_solve (src/sim_narrow_phase.cpp:179).
//   * oneTBB parallel_sort (src/sim_propagate.cpp:483): unstable; restated as a
//     STABLE sort (tie order cannot change tree topology or the pair set).
// For those pieces bit-level parity with the reference is UNPINNED (the reference
// holds no golden numbers for them; its tests compare against heyoka at run time).
// What pins this oracle: tests/test_oracle_kat.py (the reference's own known-answer
// tests: conjunction counts, collision/reentry/exit tolerances, interrupt order),
// closed-form two-body propagation, and oracle/_ref (the reference's own
// mortonND_LUT.h and ival.hpp compiled from where they lie) for Morton codes and
// interval arithmetic.
//
// Deliberate, documented choices where the reference's arithmetic is unpinned
// (the HIP kernels make the same choices so that most stages agree bit for bit):
//   * pow(u, -1.5) at order 0 is evaluated as 1 / (u * sqrt(u)) (IEEE ops only);
//   * divisions by small integers are multiplications by the rounded reciprocal;
//   * convolution sums run in ascending index order with explicit fma().
#pragma once

#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

namespace orc
{

// ---------------------------------------------------------------------------
// Double-length floats. Restates heyoka::detail::dfloat<double> (EXTERNAL):
// unevaluated normalised sum hi + lo, error-free transforms for + and -,
// lexicographic ordering, cast to double == hi. Used at
// src/sim_propagate.cpp:724-725, src/sim_narrow_phase.cpp:691-716.
// ---------------------------------------------------------------------------
struct dfloat {
    double hi = 0, lo = 0;
    dfloat() = default;
    explicit dfloat(double h) : hi(h), lo(0) {}
    dfloat(double h, double l) : hi(h), lo(l) {}
    explicit operator double() const
    {
        return hi;
    }
};

inline std::pair<double, double> eft_add_knuth(double a, double b)
{
    const double x = a + b;
    const double z = x - a;
    const double y = (a - (x - z)) + (b - z);
    return {x, y};
}

inline dfloat df_normalise(double a, double b)
{
    // Dekker's fast two-sum (|a| >= |b| expected).
    const double x = a + b;
    const double y = b - (x - a);
    return dfloat(x, y);
}

inline dfloat operator+(const dfloat &x, const dfloat &y)
{
    auto [xx, yy] = eft_add_knuth(x.hi, y.hi);
    yy += x.lo + y.lo;
    return df_normalise(xx, yy);
}

inline dfloat operator-(const dfloat &x)
{
    return dfloat(-x.hi, -x.lo);
}

inline dfloat operator-(const dfloat &x, const dfloat &y)
{
    return x + (-y);
}

inline dfloat operator+(const dfloat &x, double y)
{
    return x + dfloat(y);
}

inline dfloat operator-(const dfloat &x, double y)
{
    return x + dfloat(-y);
}

inline bool operator<(const dfloat &a, const dfloat &b)
{
    return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
}
inline bool operator>(const dfloat &a, const dfloat &b)
{
    return b < a;
}
inline bool operator==(const dfloat &a, const dfloat &b)
{
    return a.hi == b.hi && a.lo == b.lo;
}
inline bool operator>=(const dfloat &a, const dfloat &b)
{
    return !(a < b);
}
inline bool df_isfinite(const dfloat &a)
{
    return std::isfinite(a.hi) && std::isfinite(a.lo);
}

// ---------------------------------------------------------------------------
// Interval arithmetic: restates src/detail/ival.hpp:25-62 (min/max of the 4
// products, no directed rounding).
// ---------------------------------------------------------------------------
struct ival {
    double lower, upper;
    ival() : lower(0), upper(0) {}
    explicit ival(double v) : lower(v), upper(v) {}
    ival(double l, double u) : lower(l), upper(u) {}
};

inline ival operator+(ival a, ival b)
{
    return ival(a.lower + b.lower, a.upper + b.upper);
}
inline ival operator+(double a, ival b)
{
    return ival(a) + b;
}
inline ival operator*(ival a, ival b)
{
    const double t1 = a.lower * b.lower;
    const double t2 = a.lower * b.upper;
    const double t3 = a.upper * b.lower;
    const double t4 = a.upper * b.upper;
    const double l = std::min(std::min(t1, t2), std::min(t3, t4));
    const double u = std::max(std::max(t1, t2), std::max(t3, t4));
    return ival(l, u);
}

// ---------------------------------------------------------------------------
// Dynamics descriptors (replace the heyoka expression vector of kw::dyn,
// include/cascade/sim.hpp:213-222).
// ---------------------------------------------------------------------------
// kind is a bit mask: Kepler plus the selected perturbations of simple_earth().
enum dyn_kind : int {
    DYN_KEPLER = 0,    // src/sim_dynamics.cpp:25-42
    DYN_KEPLER_J2 = 1, // bit 0: J2, cascade.py/dynamics/_simple_earth.py:169-178
    DYN_DRAG = 2,      // bit 1: drag with the 4-term exponential density fit, :365-387 and :16-46
    DYN_TAPE = 0x100,  // general right-hand sides as a tape of Taylor-series operations (see tape_op)
};

// General dynamics. The reference hands arbitrary heyoka expressions to heyoka's JIT
// (include/cascade/sim.hpp:213-222, src/sim_basic.cpp:405-469); here they arrive lowered to a
// tape (include/cascade/expression.hpp, the dynamics definition = input data of the test). Slots
// 0..5 = x, y, z, vx, vy, vz, slot 6 = r; operand b is a slot (>= 0), the immediate c (-1) or
// runtime parameter -2 - b of the particle.
struct tape_op {
    std::int32_t op, dst, a, b;
    double c;
};
enum : std::int32_t {
    TP_ADD = 1, TP_SUB, TP_RSUB, TP_MUL, TP_DIV, TP_RDIV, TP_SQR, TP_POW, TP_SQRT, TP_EXP, TP_SINCOS, TP_TIME, TP_CONST
};

struct dynamics {
    int kind = DYN_KEPLER;
    double mu = 1.;
    double J2 = 0.;
    double Re = 1.;
    // Density fit rho(h) = sum_k alpha_k exp(-(h - gamma_k) beta_k), h = r - Re (the fitted
    // constants of _simple_earth.py:24-39 / benchmark/data/best_fit_density.txt, SI units).
    double dens_alpha[4] = {1.01709935e-06, 7.86443375e-01, 7.50341883e-09, 8.63934252e-14};
    double dens_beta[4] = {4.63822910e-02 / 1000, 1.86080048e-01 / 1000, 2.48667176e-02 / 1000, 4.81080852e-03 / 1000};
    double dens_gamma[4] = {5.01594516e00 * 1000, 2.28429809e01 * 1000, 4.27957829e00 * 1000, 1.56291673e-01 * 1000};
    // kind & DYN_TAPE
    std::vector<tape_op> tape;
    std::uint32_t tape_nslots = 0, tape_npars = 0;
    std::int32_t tape_rhs[7] = {0, 0, 0, 0, 0, 0, 0};
    std::uint32_t npars() const
    {
        if (kind & DYN_TAPE) {
            return tape_npars;
        }
        return (kind & DYN_DRAG) ? 1u : 0u; // par[0] = BSTAR (:367)
    }
};

enum class outcome : int { success = 0, time_limit, collision, reentry, exit, err_nf_state };

// ---------------------------------------------------------------------------
// Polynomial helpers for root finding: restate src/sim_narrow_phase.cpp:67-140
// and the JIT helpers of src/sim_jit.cpp.
// ---------------------------------------------------------------------------
template <typename T>
inline int sgn(T val)
{
    return (T(0) < val) - (val < T(0));
}

// src/sim_narrow_phase.cpp:80-91
inline double poly_eval(const double *a, double x, std::uint32_t n)
{
    double ret = a[n];
    for (std::uint32_t i = 1; i <= n; ++i) {
        ret = a[n - i] + ret * x;
    }
    return ret;
}

inline ival poly_eval(const double *a, ival x, std::uint32_t n)
{
    ival ret(a[n]);
    for (std::uint32_t i = 1; i <= n; ++i) {
        ret = a[n - i] + ret * x;
    }
    return ret;
}

// src/sim_narrow_phase.cpp:96-108
inline double poly_eval_1(const double *a, double x, std::uint32_t n)
{
    double ret1 = a[n] * n;
    for (std::uint32_t i = 1; i < n; ++i) {
        ret1 = a[n - i] * (n - i) + ret1 * x;
    }
    return ret1;
}

// src/sim_narrow_phase.cpp:116-124
inline void poly_rescale(double *ret, const double *a, double scal, std::uint32_t n)
{
    double cur_f = 1;
    for (std::uint32_t i = 0; i <= n; ++i) {
        ret[i] = cur_f * a[i];
        cur_f *= scal;
    }
}

// src/sim_narrow_phase.cpp:130-140
inline void poly_rescale_p2(double *ret, const double *a, std::uint32_t n)
{
    double cur_f = 1;
    for (std::uint32_t i = 0; i <= n; ++i) {
        ret[n - i] = cur_f * a[n - i];
        cur_f *= 2;
    }
}

// Binomial coefficients table C(n, k), n,k <= 64, as doubles (exact up to 2^53).
struct binom_table {
    static constexpr int N = 41;
    double c[N][N];
    binom_table()
    {
        for (int n = 0; n < N; ++n) {
            for (int k = 0; k < N; ++k) {
                c[n][k] = 0;
            }
            c[n][0] = 1;
            for (int k = 1; k <= n; ++k) {
                c[n][k] = c[n - 1][k - 1] + (k <= n - 1 ? c[n - 1][k] : 0.);
            }
        }
    }
};
inline const binom_table &binoms()
{
    static const binom_table t;
    return t;
}

// pta_cfunc: src/sim_jit.cpp:64-113. c'_i = sum_{k=i}^n c_k * C(k, k-i) * a^(k-i),
// powers of a by the repeated-squaring scheme of :76-87; the sum (heyoka sum(),
// association unpinned) runs in ascending k.
inline void poly_translate_a(double *out, const double *cf, double a, std::uint32_t n)
{
    double a_pows[64];
    a_pows[0] = 1.;
    a_pows[1] = a;
    for (std::uint32_t i = 2; i <= n; ++i) {
        if (i % 2u == 0u) {
            a_pows[i] = a_pows[i / 2u] * a_pows[i / 2u];
        } else {
            a_pows[i] = a_pows[i / 2u + 1u] * a_pows[i / 2u];
        }
    }
    const auto &bc = binoms();
    for (std::uint32_t i = 0; i <= n; ++i) {
        double acc = 0;
        for (std::uint32_t k = i; k <= n; ++k) {
            acc += (cf[k] * bc.c[k][k - i]) * a_pows[k - i];
        }
        out[i] = acc;
    }
}

// poly_translate_1 (heyoka detail, EXTERNAL; src/sim_jit.cpp:244):
// ret[i] = sum_{k>=i} C(k, i) cf[k], ascending k.
inline void poly_translate_1(double *ret, const double *cf, std::uint32_t n)
{
    const auto &bc = binoms();
    for (std::uint32_t i = 0; i <= n; ++i) {
        double acc = cf[i];
        for (std::uint32_t k = i + 1u; k <= n; ++k) {
            acc += bc.c[k][i] * cf[k];
        }
        ret[i] = acc;
    }
}

// poly_rtscc (heyoka detail, EXTERNAL; src/sim_jit.cpp:235): out1 = reverse(cf),
// out2(x) = out1(x + 1), n_sc = number of sign changes in out2 ignoring zeros.
inline void poly_rtscc(double *out1, double *out2, std::uint32_t *n_sc, const double *cf, std::uint32_t n)
{
    for (std::uint32_t i = 0; i <= n; ++i) {
        out1[i] = cf[n - i];
    }
    poly_translate_1(out2, out1, n);
    std::uint32_t count = 0;
    int last = 0;
    for (std::uint32_t i = 0; i <= n; ++i) {
        const int s = sgn(out2[i]);
        if (s != 0) {
            if (last != 0 && s != last) {
                ++count;
            }
            last = s;
        }
    }
    *n_sc = count;
}

// fex_check (heyoka detail llvm_add_fex_check, EXTERNAL; src/sim_jit.cpp:234):
// interval-Horner enclosure of the polynomial over [0, h] (h >= 0 here); result 1
// iff the enclosure is sign-definite and non-zero. NaNs give sgn 0 => not excluded.
inline std::uint32_t fex_check(const double *cf, double h, std::uint32_t n)
{
    const ival hi = (h >= 0) ? ival(0., h) : ival(h, 0.);
    const ival enc = poly_eval(cf, hi, n);
    const int s_lo = sgn(enc.lower), s_hi = sgn(enc.upper);
    return (s_lo == s_hi && s_lo != 0) ? 1u : 0u;
}

// ssdiff3_cfunc: src/sim_jit.cpp:118-218. Sum of the squares of three polynomial
// differences in the truncated power series algebra.
inline void poly_ssdiff3(double *out, const double *in6, std::uint32_t n)
{
    const std::uint32_t np1 = n + 1u;
    double d[3][64];
    for (int c = 0; c < 3; ++c) {
        for (std::uint32_t i = 0; i <= n; ++i) {
            d[c][i] = in6[c * np1 + i] - in6[(c + 3) * np1 + i];
        }
    }
    auto psquare = [&](const double *p, std::uint32_t i) {
        if (i % 2u == 0u) {
            if (i == 0u) {
                return p[0] * p[0];
            }
            double acc = 0;
            for (std::uint32_t j = 0; j <= i / 2u - 1u; ++j) {
                acc += p[i - j] * p[j];
            }
            return 2. * acc + p[i / 2u] * p[i / 2u];
        } else {
            double acc = 0;
            for (std::uint32_t j = 0; j <= (i - 1u) / 2u; ++j) {
                acc += p[i - j] * p[j];
            }
            return 2. * acc;
        }
    };
    for (std::uint32_t i = 0; i <= n; ++i) {
        out[i] = (psquare(d[0], i) + psquare(d[1], i)) + psquare(d[2], i);
    }
}

// ---------------------------------------------------------------------------
// TOMS 748 bracketed root finder. Restates the published algorithm of
// Boost.Math toms748_solve (EXTERNAL, Boost >= 1.73) as called from
// src/sim_narrow_phase.cpp:145-201 with eps_tolerance<double>() and a limit
// of 100 iterations.
// ---------------------------------------------------------------------------
namespace t748
{

struct fpoly {
    const double *p;
    std::uint32_t n;
    double operator()(double x) const
    {
        return poly_eval(p, x, n);
    }
};

inline bool tol_eps(double a, double b)
{
    const double eps = 4 * std::numeric_limits<double>::epsilon();
    return std::fabs(a - b) <= eps * std::min(std::fabs(a), std::fabs(b));
}

inline double safe_div(double num, double denom, double r)
{
    if (std::fabs(denom) < 1) {
        if (std::fabs(denom * std::numeric_limits<double>::max()) <= std::fabs(num)) {
            return r;
        }
    }
    return num / denom;
}

template <typename F>
inline void bracket(F &f, double &a, double &b, double c, double &fa, double &fb, double &d, double &fd)
{
    const double tol = std::numeric_limits<double>::epsilon() * 2;
    if ((b - a) < 2 * tol * a) {
        c = a + (b - a) / 2;
    } else if (c <= a + std::fabs(a) * tol) {
        c = a + std::fabs(a) * tol;
    } else if (c >= b - std::fabs(b) * tol) {
        c = b - std::fabs(b) * tol;
    }
    const double fc = f(c);
    if (fc == 0) {
        a = c;
        fa = 0;
        d = 0;
        fd = 0;
        return;
    }
    if (sgn(fa) * sgn(fc) < 0) {
        d = b;
        fd = fb;
        b = c;
        fb = fc;
    } else {
        d = a;
        fd = fa;
        a = c;
        fa = fc;
    }
}

inline double secant_interpolate(double a, double b, double fa, double fb)
{
    const double tol = std::numeric_limits<double>::epsilon() * 5;
    const double c = a - (fa / (fb - fa)) * (b - a);
    if ((c <= a + std::fabs(a) * tol) || (c >= b - std::fabs(b) * tol)) {
        return (a + b) / 2;
    }
    return c;
}

inline double quadratic_interpolate(double a, double b, double d, double fa, double fb, double fd, unsigned count)
{
    const double dmax = std::numeric_limits<double>::max();
    double B = safe_div(fb - fa, b - a, dmax);
    double A = safe_div(fd - fb, d - b, dmax);
    A = safe_div(A - B, d - a, 0.);
    if (A == 0) {
        return secant_interpolate(a, b, fa, fb);
    }
    double c = (sgn(A) * sgn(fa) > 0) ? a : b;
    for (unsigned i = 1; i <= count; ++i) {
        c -= safe_div(fa + (B + A * (c - b)) * (c - a), B + A * (2 * c - a - b), 1 + c - a);
    }
    if ((c <= a) || (c >= b)) {
        c = secant_interpolate(a, b, fa, fb);
    }
    return c;
}

inline double cubic_interpolate(double a, double b, double d, double e, double fa, double fb, double fd, double fe)
{
    const double q11 = (d - e) * fd / (fe - fd);
    const double q21 = (b - d) * fb / (fd - fb);
    const double q31 = (a - b) * fa / (fb - fa);
    const double d21 = (b - d) * fd / (fd - fb);
    const double d31 = (a - b) * fb / (fb - fa);
    const double q22 = (d21 - q11) * fb / (fe - fb);
    const double q32 = (d31 - q21) * fa / (fd - fa);
    const double d32 = (d31 - q21) * fd / (fd - fa);
    const double q33 = (d32 - q22) * fa / (fe - fa);
    double c = q31 + q32 + q33 + a;
    if ((c <= a) || (c >= b)) {
        c = quadratic_interpolate(a, b, d, fa, fb, fd, 3);
    }
    return c;
}

// Returns {a, b, err} with err != 0 for domain errors (errno policy of the reference).
template <typename F>
inline std::tuple<double, double, int> solve(F f, double ax, double bx, std::uintmax_t &max_iter)
{
    // 3-argument front end: two evaluations are charged to the budget.
    max_iter -= 2;
    const double fax = f(ax), fbx = f(bx);

    std::uintmax_t count = max_iter;
    double a = ax, b = bx, fa = fax, fb = fbx, c, u, fu, a0, b0, d, fd, e, fe;
    const double mu = 0.5;

    if (!(a < b)) {
        max_iter += 2;
        return {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN(), EDOM};
    }
    if (tol_eps(a, b) || (fa == 0) || (fb == 0)) {
        max_iter = 0;
        if (fa == 0) {
            b = a;
        } else if (fb == 0) {
            a = b;
        }
        max_iter += 2;
        return {a, b, 0};
    }
    if (sgn(fa) * sgn(fb) > 0) {
        max_iter += 2;
        return {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN(), EDOM};
    }
    fe = e = fd = 1e5;
    d = 0;

    if (fa != 0) {
        c = secant_interpolate(a, b, fa, fb);
        bracket(f, a, b, c, fa, fb, d, fd);
        --count;
        if (count && (fa != 0) && !tol_eps(a, b)) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
            e = d;
            fe = fd;
            bracket(f, a, b, c, fa, fb, d, fd);
            --count;
        }
    }

    const double min_diff = std::numeric_limits<double>::min() * 32;
    auto prof_f = [&]() {
        return (std::fabs(fa - fb) < min_diff) || (std::fabs(fa - fd) < min_diff) || (std::fabs(fa - fe) < min_diff)
               || (std::fabs(fb - fd) < min_diff) || (std::fabs(fb - fe) < min_diff)
               || (std::fabs(fd - fe) < min_diff);
    };

    while (count && (fa != 0) && !tol_eps(a, b)) {
        a0 = a;
        b0 = b;
        bool prof = prof_f();
        if (prof) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
        } else {
            c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
        }
        e = d;
        fe = fd;
        bracket(f, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        prof = prof_f();
        if (prof) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 3);
        } else {
            c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
        }
        bracket(f, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        if (std::fabs(fa) < std::fabs(fb)) {
            u = a;
            fu = fa;
        } else {
            u = b;
            fu = fb;
        }
        c = u - 2 * (fu / (fb - fa)) * (b - a);
        if (std::fabs(c - u) > (b - a) / 2) {
            c = a + (b - a) / 2;
        }
        e = d;
        fe = fd;
        bracket(f, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        if ((b - a) < mu * (b0 - a0)) {
            continue;
        }
        e = d;
        fe = fd;
        bracket(f, a, b, a + (b - a) / 2, fa, fb, d, fd);
        --count;
    }
    max_iter -= count;
    if (fa == 0) {
        b = a;
    } else if (fb == 0) {
        a = b;
    }
    max_iter += 2;
    return {a, b, 0};
}

} // namespace t748

// src/sim_narrow_phase.cpp:145-201.
inline std::tuple<double, int> bracketed_root_find(const double *poly, std::uint32_t order, double lb, double ub)
{
    if (std::isfinite(lb) && std::isfinite(ub) && ub > lb) {
        ub = std::nextafter(ub, lb);
    }
    constexpr std::uintmax_t iter_limit = 100;
    auto max_iter = iter_limit;
    const auto [pa, pb, err] = t748::solve(t748::fpoly{poly, order}, lb, ub, max_iter);
    const double ret = (pa + pb) / 2;
    if (err > 0) {
        return {0., err};
    }
    if (max_iter < iter_limit) {
        return {ret, 0};
    }
    return {ret, -1};
}

struct rf_stats {
    std::uint64_t n_fex_pass = 0, n_fex_fail = 0, n_warn = 0, max_depth = 0, max_wlist = 0;
};

// Scratch for real-root isolation (replaces np_data::pwrap / wlist / isol,
// include/cascade/detail/sim_data.hpp:236-294).
struct rf_scratch {
    struct witem {
        double lb, ub;
        std::vector<double> p;
        std::uint32_t depth = 0;
    };
    std::vector<witem> wlist;
    std::vector<std::pair<double, double>> isol;
    std::vector<double> tmp, tmp1, tmp2;
    void prepare(std::uint32_t order)
    {
        tmp.resize(order + 1u);
        tmp1.resize(order + 1u);
        tmp2.resize(order + 1u);
    }
};

// run_poly_root_finding: src/sim_narrow_phase.cpp:208-422. on_root(double t)
// receives every accepted root already referred to [0, rf_int).
template <typename OnRoot>
inline void run_poly_root_finding(const double *poly, std::uint32_t order, double rf_int, int direction,
                                  rf_scratch &sc, rf_stats &stats, OnRoot &&on_root)
{
    if (fex_check(poly, rf_int, order) != 0u) {
        ++stats.n_fex_pass;
        return;
    }
    ++stats.n_fex_fail;

    sc.prepare(order);
    sc.isol.clear();
    sc.wlist.clear();

    auto add_root = [&](double root) {
        if (!std::isfinite(root)) {
            ++stats.n_warn;
            return;
        }
        bool accept_root = true;
        if (direction != 0) {
            const auto der = poly_eval_1(poly, root, order);
            if (!std::isfinite(der)) {
                ++stats.n_warn;
                return;
            }
            if (sgn(der) != direction) {
                accept_root = false;
            }
        }
        if (accept_root) {
            on_root(root);
        }
    };

    poly_rescale(sc.tmp.data(), poly, rf_int, order);
    sc.wlist.push_back({0., 1., sc.tmp, 0u});

    bool loop_failed = false;

    do {
        const double lb = sc.wlist.back().lb;
        const double ub = sc.wlist.back().ub;
        const std::uint32_t depth = sc.wlist.back().depth;
        stats.max_depth = std::max<std::uint64_t>(stats.max_depth, depth);
        sc.tmp = std::move(sc.wlist.back().p);
        sc.wlist.pop_back();

        if (sc.tmp[0] == 0
            && std::all_of(sc.tmp.data() + 1, sc.tmp.data() + 1 + order, [](double x) { return std::isfinite(x); })) {
            add_root(lb * rf_int);
        }

        std::uint32_t n_sc = 0;
        poly_rtscc(sc.tmp1.data(), sc.tmp2.data(), &n_sc, sc.tmp.data(), order);

        if (n_sc == 1u) {
            sc.isol.emplace_back(lb, ub);
        } else if (n_sc > 1u) {
            poly_rescale_p2(sc.tmp1.data(), sc.tmp.data(), order);
            poly_translate_1(sc.tmp2.data(), sc.tmp1.data(), order);
            const double mid = (lb + ub) / 2;
            sc.wlist.push_back({lb, mid, sc.tmp1, depth + 1u});
            sc.wlist.push_back({mid, ub, sc.tmp2, depth + 1u});
            stats.max_wlist = std::max<std::uint64_t>(stats.max_wlist, sc.wlist.size());
        }

        if (sc.wlist.size() > 250u || sc.isol.size() > order) {
            ++stats.n_warn;
            loop_failed = true;
            break;
        }
    } while (!sc.wlist.empty());

    if (!sc.isol.empty() && !loop_failed) {
        sc.tmp1.resize(order + 1u);
        poly_rescale(sc.tmp1.data(), poly, rf_int, order);
        for (auto &[lb, ub] : sc.isol) {
            const auto [root, cflag] = bracketed_root_find(sc.tmp1.data(), order, lb, ub);
            if (cflag == 0) {
                add_root(root * rf_int);
            } else {
                ++stats.n_warn;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Adaptive Taylor integrator. Restates what heyoka's taylor_adaptive<double>
// does for the 7-variable system built in src/sim_basic.cpp:461-578
// (6 user equations + r' = (x vx + y vy + z vz) / r, exit/reentry terminal
// events) when driven by propagate_for() at src/sim_propagate.cpp:760,1011.
// ---------------------------------------------------------------------------
struct taylor_cfg {
    dynamics dyn;
    double tol = std::numeric_limits<double>::epsilon();
    std::uint32_t order = 20;
    // Events (src/sim_basic.cpp:509-578). reentry_kind: 0 none, 1 sphere, 2 ellipsoid.
    double exit_radius = 0;
    int reentry_kind = 0;
    double reentry[3] = {0, 0, 0};

    bool with_exit() const
    {
        return exit_radius > 0;
    }
    bool with_reentry() const
    {
        return reentry_kind != 0;
    }
    // src/sim_basic.cpp:658-676
    std::uint32_t exit_event_idx() const
    {
        return 0;
    }
    std::uint32_t reentry_event_idx() const
    {
        return static_cast<std::uint32_t>(with_exit());
    }
};

// heyoka's order rule (EXTERNAL): p = max(2, ceil(-ln(tol)/2 + 1)).
inline std::uint32_t order_from_tol(double tol)
{
    const double o = std::ceil(-std::log(tol) / 2 + 1);
    return o < 2 ? 2u : static_cast<std::uint32_t>(o);
}

struct taylor_work {
    std::uint32_t order = 0;
    // tc[c][o], c in x,y,z,vx,vy,vz,r.
    std::vector<double> tc;
    std::vector<double> u, w, g, inv;
    std::vector<double> aux[20];
    double par0 = 0; // first runtime parameter of the particle being integrated (BSTAR)
    const double *pars = nullptr; // all runtime parameters of that particle (tape dynamics)
    double time0 = 0;             // simulation time at the start of the substep (tape dynamics)
    std::vector<double> ws;       // tape slots [nslots][order + 1]
    rf_scratch rfs;
    rf_stats rf_st;
    void prepare(std::uint32_t p)
    {
        if (order == p && !tc.empty()) {
            return;
        }
        order = p;
        tc.assign(7u * (p + 1u), 0.);
        u.assign(p + 1u, 0.);
        w.assign(p + 1u, 0.);
        g.assign(p + 1u, 0.);
        inv.assign(p + 2u, 0.);
        for (std::uint32_t i = 1; i <= p + 1u; ++i) {
            inv[i] = 1. / static_cast<double>(i);
        }
        for (auto &a : aux) {
            a.assign(p + 1u, 0.);
        }
    }
    double *c(int i)
    {
        return tc.data() + static_cast<std::size_t>(i) * (order + 1u);
    }
};

// Cauchy product coefficient n of a*b (ascending j, fma).
inline double conv(const double *a, const double *b, std::uint32_t n)
{
    double acc = 0;
    for (std::uint32_t j = 0; j <= n; ++j) {
        acc = std::fma(a[j], b[n - j], acc);
    }
    return acc;
}

// Coefficient n of a*a using the symmetric half-sum.
inline double conv_sq(const double *a, std::uint32_t n)
{
    if (n == 0u) {
        return a[0] * a[0];
    }
    double acc = 0;
    for (std::uint32_t j = 0; 2u * j < n; ++j) {
        acc = std::fma(a[j], a[n - j], acc);
    }
    acc = acc + acc;
    if (n % 2u == 0u) {
        acc = std::fma(a[n / 2u], a[n / 2u], acc);
    }
    return acc;
}

inline double det_exp(double t);
inline double det_pow(double x, double y);

// a0^c at order 0 of a power series: exact arithmetic (multiplications, one sqrt, one division)
// when 2c is a small integer -- every exponent of the reference's dynamics is (-3/2, -5/2,
// -7/2, ...) --, det_pow otherwise.
inline double pow_order0(double a0, double c)
{
    const double k2 = 2. * c;
    if (k2 == std::floor(k2) && std::fabs(k2) <= 64.) {
        const int k = static_cast<int>(k2), m = k < 0 ? -k : k;
        double r = (m % 2) ? std::sqrt(a0) : 1.;
        for (int i = 0; i < m / 2; ++i) {
            r = r * a0;
        }
        return k < 0 ? 1. / r : r;
    }
    return det_pow(a0, c);
}

// Tape dynamics: Taylor coefficients of all slots, order by order (classical AD recurrences,
// SURVEY appendix A), then x_k^[n+1] = rhs_k^[n] / (n + 1).
inline void tape_compute_tcs(const taylor_cfg &cfg, taylor_work &wk, const double st[7]);

// Normalised Taylor coefficients of the 7 state variables at the current state
// (heyoka AD recurrences, EXTERNAL). Kepler RHS:
// src/sim_dynamics.cpp:36-41; r equation: src/sim_basic.cpp:469.
inline void compute_tcs(const taylor_cfg &cfg, taylor_work &wk, const double st[7])
{
    if (cfg.dyn.kind & DYN_TAPE) {
        tape_compute_tcs(cfg, wk, st);
        return;
    }
    const std::uint32_t p = wk.order;
    double *x = wk.c(0), *y = wk.c(1), *z = wk.c(2), *vx = wk.c(3), *vy = wk.c(4), *vz = wk.c(5), *r = wk.c(6);
    double *u = wk.u.data(), *w = wk.w.data();
    const double *inv = wk.inv.data();
    x[0] = st[0];
    y[0] = st[1];
    z[0] = st[2];
    vx[0] = st[3];
    vy[0] = st[4];
    vz[0] = st[5];
    r[0] = st[6];

    const double mu = cfg.dyn.mu;
    double inv_u0 = 0;
    const double inv_r0 = 1. / r[0];

    // J2 auxiliaries (kind 1): see compute below.
    const bool j2 = (cfg.dyn.kind & DYN_KEPLER_J2) != 0;
    const bool drag = (cfg.dyn.kind & DYN_DRAG) != 0;
    // Drag auxiliaries: s = sqrt(u), e_k = exp(-beta_k (s - Re - gamma_k)), dens = sum alpha_k e_k,
    // vv = |v|^2, mv = sqrt(vv), fd = (dens * mv) * (BSTAR / ref_density).
    double *sq = wk.aux[6].data(), *dens = wk.aux[7].data(), *vv = wk.aux[8].data(), *mv = wk.aux[9].data(),
           *fd = wk.aux[10].data();
    double *ek[4] = {wk.aux[11].data(), wk.aux[12].data(), wk.aux[13].data(), wk.aux[14].data()};
    const double cdrag = wk.par0 / (0.1570 / cfg.dyn.Re);
    double inv_2s0 = 0, inv_2mv0 = 0;
    double *w5 = wk.aux[0].data(), *z2 = wk.aux[1].data(), *q = wk.aux[2].data(), *fxy = wk.aux[3].data(),
           *fz = wk.aux[4].data(), *w2 = wk.aux[5].data();
    const double kj2 = 1.5 * cfg.dyn.J2 * cfg.dyn.mu * cfg.dyn.Re * cfg.dyn.Re;

    for (std::uint32_t n = 0; n < p; ++n) {
        // u = x^2 + y^2 + z^2
        u[n] = (conv_sq(x, n) + conv_sq(y, n)) + conv_sq(z, n);
        // w = u^(-3/2)
        if (n == 0u) {
            inv_u0 = 1. / u[0];
            w[0] = 1. / (u[0] * std::sqrt(u[0]));
        } else {
            double acc = 0;
            for (std::uint32_t j = 0; j < n; ++j) {
                // n*alpha - j*(alpha + 1), alpha = -3/2
                const double cf = 0.5 * static_cast<double>(j) - 1.5 * static_cast<double>(n);
                acc = std::fma(cf * u[n - j], w[j], acc);
            }
            w[n] = (acc * inv_u0) * inv[n];
        }
        double ax = -mu * conv(x, w, n);
        double ay = -mu * conv(y, w, n);
        double az = -mu * conv(z, w, n);
        if (j2) {
            // J2 acceleration (cascade.py/dynamics/_simple_earth.py:169-178):
            //   k = 1.5 J2 mu Re^2, r2 = u, magJ2 = k / r2^(5/2)
            //   ax += k x/r^5 (5 z^2/r^2 - 1), ay likewise, az += k z/r^5 (5 z^2/r^2 - 3)
            // with w5 = u^(-5/2) = w * (1/u) where w2 = 1/u.
            // w2 = u^-1
            if (n == 0u) {
                w2[0] = inv_u0;
            } else {
                double acc = 0;
                for (std::uint32_t j = 0; j < n; ++j) {
                    acc = std::fma(u[n - j], w2[j], acc);
                }
                w2[n] = -(acc * inv_u0);
            }
            w5[n] = conv(w, w2, n);
            z2[n] = conv_sq(z, n);
            // q = 5 z^2 / u
            q[n] = 5. * conv(z2, w2, n);
            // fxy = k * w5 * (q - 1), fz = k * w5 * (q - 3)
            {
                double a1 = 0, a2 = 0;
                for (std::uint32_t j = 0; j <= n; ++j) {
                    const double qm1 = (n - j == 0u) ? q[0] - 1. : q[n - j];
                    const double qm3 = (n - j == 0u) ? q[0] - 3. : q[n - j];
                    a1 = std::fma(w5[j], qm1, a1);
                    a2 = std::fma(w5[j], qm3, a2);
                }
                fxy[n] = kj2 * a1;
                fz[n] = kj2 * a2;
            }
            ax += conv(x, fxy, n);
            ay += conv(y, fxy, n);
            az += conv(z, fz, n);
        }
        if (drag) {
            // Drag acceleration (cascade.py/dynamics/_simple_earth.py:365-387), spherical Earth:
            //   a -= (rho(r - Re) / ref_density) BSTAR |v| v,   ref_density = 0.1570 / Re.
            // sqrt series: c = sqrt(a): c[n] = (a[n] - sum_{j=1}^{n-1} c[j] c[n-j]) / (2 c[0]).
            if (n == 0u) {
                sq[0] = std::sqrt(u[0]);
                inv_2s0 = 1. / (2. * sq[0]);
            } else {
                double acc = u[n];
                for (std::uint32_t j = 1; j < n; ++j) {
                    acc = std::fma(-sq[j], sq[n - j], acc);
                }
                sq[n] = acc * inv_2s0;
            }
            double dsum = 0;
            for (int k = 0; k < 4; ++k) {
                // e = exp(g), g[0] = -beta (s[0] - Re - gamma), g[m] = -beta s[m]:
                // e[n] = (1 / n) sum_{j=1}^{n} j g[j] e[n-j].
                const double mb = -cfg.dyn.dens_beta[k];
                if (n == 0u) {
                    ek[k][0] = det_exp(mb * ((sq[0] - cfg.dyn.Re) - cfg.dyn.dens_gamma[k]));
                } else {
                    double acc = 0;
                    for (std::uint32_t j = 1; j <= n; ++j) {
                        acc = std::fma((static_cast<double>(j) * mb) * sq[j], ek[k][n - j], acc);
                    }
                    ek[k][n] = acc * inv[n];
                }
                dsum = std::fma(cfg.dyn.dens_alpha[k], ek[k][n], dsum);
            }
            dens[n] = dsum;
            vv[n] = (conv_sq(vx, n) + conv_sq(vy, n)) + conv_sq(vz, n);
            if (n == 0u) {
                mv[0] = std::sqrt(vv[0]);
                inv_2mv0 = 1. / (2. * mv[0]);
            } else {
                double acc = vv[n];
                for (std::uint32_t j = 1; j < n; ++j) {
                    acc = std::fma(-mv[j], mv[n - j], acc);
                }
                mv[n] = acc * inv_2mv0;
            }
            fd[n] = conv(dens, mv, n) * cdrag;
            ax -= conv(fd, vx, n);
            ay -= conv(fd, vy, n);
            az -= conv(fd, vz, n);
        }
        // r' = (x vx + y vy + z vz) / r, with q_r[m] = (m + 1) r[m + 1].
        const double s = (conv(x, vx, n) + conv(y, vy, n)) + conv(z, vz, n);
        double acc = s;
        for (std::uint32_t j = 1; j <= n; ++j) {
            const double qr = static_cast<double>(n - j + 1u) * r[n - j + 1u];
            acc = std::fma(-r[j], qr, acc);
        }
        const double qn = acc * inv_r0;

        const double f = inv[n + 1u];
        x[n + 1u] = vx[n] * f;
        y[n + 1u] = vy[n] * f;
        z[n + 1u] = vz[n] * f;
        vx[n + 1u] = ax * f;
        vy[n + 1u] = ay * f;
        vz[n + 1u] = az * f;
        r[n + 1u] = qn * f;
    }
}

// Deterministic pow(x, y) for x > 0 built from IEEE basic operations, fma, frexp,
// ldexp and rint only (|relative error| < 1e-14 over the range used here). The
// reference's own pow is library-dependent (SLEEF in heyoka's batch lanes, libm in the
// scalar remainder), so the step size is not pinned at the last bit anyway; using one
// spelled-out algorithm here and in the HIP kernel makes substep boundaries agree bit
// for bit between oracle and GPU.
inline double det_log(double x)
{
    int e;
    double m = std::frexp(x, &e);
    if (m < 0.70710678118654752) {
        m = m * 2.;
        e -= 1;
    }
    const double s = (m - 1.) / (m + 1.);
    const double s2 = s * s;
    double p = 1. / 23.;
    p = std::fma(p, s2, 1. / 21.);
    p = std::fma(p, s2, 1. / 19.);
    p = std::fma(p, s2, 1. / 17.);
    p = std::fma(p, s2, 1. / 15.);
    p = std::fma(p, s2, 1. / 13.);
    p = std::fma(p, s2, 1. / 11.);
    p = std::fma(p, s2, 1. / 9.);
    p = std::fma(p, s2, 1. / 7.);
    p = std::fma(p, s2, 1. / 5.);
    p = std::fma(p, s2, 1. / 3.);
    p = std::fma(p, s2, 1.);
    const double at = s * p;
    return std::fma(static_cast<double>(e), 0.69314718055994530942, 2. * at);
}

inline double det_exp(double t)
{
    const double k = std::rint(t * 1.44269504088896340736);
    double r = std::fma(-k, 0.693147180369123816490, t);
    r = std::fma(-k, 1.90821492927058770002e-10, r);
    double p = 1. / 87178291200.;
    p = std::fma(p, r, 1. / 6227020800.);
    p = std::fma(p, r, 1. / 479001600.);
    p = std::fma(p, r, 1. / 39916800.);
    p = std::fma(p, r, 1. / 3628800.);
    p = std::fma(p, r, 1. / 362880.);
    p = std::fma(p, r, 1. / 40320.);
    p = std::fma(p, r, 1. / 5040.);
    p = std::fma(p, r, 1. / 720.);
    p = std::fma(p, r, 1. / 120.);
    p = std::fma(p, r, 1. / 24.);
    p = std::fma(p, r, 1. / 6.);
    p = std::fma(p, r, 0.5);
    p = std::fma(p, r, 1.);
    p = std::fma(p, r, 1.);
    return std::ldexp(p, static_cast<int>(k));
}

inline double det_pow(double x, double y)
{
    if (!(x > 0.)) {
        // 0 -> 0 (y > 0 here), NaN -> NaN
        return x;
    }
    if (!(x <= std::numeric_limits<double>::max())) {
        return x; // +inf
    }
    return det_exp(y * det_log(x));
}

inline void tape_compute_tcs(const taylor_cfg &cfg, taylor_work &wk, const double st[7])
{
    const auto &d = cfg.dyn;
    const std::uint32_t p = wk.order;
    const std::size_t L = p + 1u;
    wk.ws.assign(static_cast<std::size_t>(d.tape_nslots) * L, 0.);
    double *ws = wk.ws.data();
    for (int c = 0; c < 7; ++c) {
        ws[static_cast<std::size_t>(c) * L] = st[c];
    }
    for (std::uint32_t n = 0; n < p; ++n) {
        const double dn = static_cast<double>(n);
        for (const auto &o : d.tape) {
            const double *a = ws + static_cast<std::size_t>(o.a) * L;
            double *r = ws + static_cast<std::size_t>(o.dst) * L;
            const bool bs = o.b < 0;
            const double sc = o.b == -1 ? o.c : ((o.b <= -2 && wk.pars) ? wk.pars[-2 - o.b] : 0.);
            const double *b = bs ? nullptr : ws + static_cast<std::size_t>(o.b) * L;
            switch (o.op) {
                case TP_ADD:
                    r[n] = bs ? (n == 0u ? a[0] + sc : a[n]) : a[n] + b[n];
                    break;
                case TP_SUB:
                    r[n] = bs ? (n == 0u ? a[0] - sc : a[n]) : a[n] - b[n];
                    break;
                case TP_RSUB:
                    r[n] = n == 0u ? sc - a[0] : -a[n];
                    break;
                case TP_MUL:
                    r[n] = bs ? a[n] * sc : conv(a, b, n);
                    break;
                case TP_DIV:
                    if (bs) {
                        r[n] = a[n] / sc;
                    } else {
                        double acc = a[n];
                        for (std::uint32_t j = 1; j <= n; ++j) {
                            acc = std::fma(-b[j], r[n - j], acc);
                        }
                        r[n] = acc / b[0];
                    }
                    break;
                case TP_RDIV: {
                    double acc = n == 0u ? sc : 0.;
                    for (std::uint32_t j = 1; j <= n; ++j) {
                        acc = std::fma(-a[j], r[n - j], acc);
                    }
                    r[n] = acc / a[0];
                    break;
                }
                case TP_SQR:
                    r[n] = conv_sq(a, n);
                    break;
                case TP_POW:
                    if (n == 0u) {
                        r[0] = pow_order0(a[0], o.c);
                    } else {
                        double acc = 0;
                        for (std::uint32_t j = 0; j < n; ++j) {
                            const double cf = dn * o.c - static_cast<double>(j) * (o.c + 1.);
                            acc = std::fma(cf * a[n - j], r[j], acc);
                        }
                        r[n] = acc / (dn * a[0]);
                    }
                    break;
                case TP_SQRT:
                    if (n == 0u) {
                        r[0] = std::sqrt(a[0]);
                    } else {
                        double acc = a[n];
                        for (std::uint32_t j = 1; j < n; ++j) {
                            acc = std::fma(-r[j], r[n - j], acc);
                        }
                        r[n] = acc / (2. * r[0]);
                    }
                    break;
                case TP_EXP:
                    if (n == 0u) {
                        r[0] = det_exp(a[0]);
                    } else {
                        double acc = 0;
                        for (std::uint32_t j = 1; j <= n; ++j) {
                            acc = std::fma(static_cast<double>(j) * a[j], r[n - j], acc);
                        }
                        r[n] = acc / dn;
                    }
                    break;
                case TP_SINCOS: {
                    double *sn = r, *cs = r + L;
                    if (n == 0u) {
                        sn[0] = std::sin(a[0]);
                        cs[0] = std::cos(a[0]);
                    } else {
                        double as = 0, ac = 0;
                        for (std::uint32_t j = 1; j <= n; ++j) {
                            const double ja = static_cast<double>(j) * a[j];
                            as = std::fma(ja, cs[n - j], as);
                            ac = std::fma(ja, sn[n - j], ac);
                        }
                        sn[n] = as / dn;
                        cs[n] = -(ac / dn);
                    }
                    break;
                }
                case TP_TIME:
                    r[n] = n == 0u ? wk.time0 : (n == 1u ? 1. : 0.);
                    break;
                case TP_CONST:
                    r[n] = n == 0u ? sc : 0.;
                    break;
                default:
                    throw std::invalid_argument("Unknown operation in the dynamics tape");
            }
        }
        const double f = wk.inv[n + 1u];
        for (int c = 0; c < 7; ++c) {
            ws[static_cast<std::size_t>(c) * L + n + 1u] = ws[static_cast<std::size_t>(d.tape_rhs[c]) * L + n] * f;
        }
    }
    for (int c = 0; c < 7; ++c) {
        std::copy(ws + static_cast<std::size_t>(c) * L, ws + static_cast<std::size_t>(c + 1) * L, wk.c(c));
    }
}

// Jorba-Zou step-size rule as used by heyoka (EXTERNAL).
// Returns NaN if any of the norms is non-finite.
inline double rhofac(std::uint32_t p)
{
    return std::exp(-0.7 / static_cast<double>(p - 1u)) / (std::exp(1.) * std::exp(1.));
}

inline double determine_h(const taylor_work &wk)
{
    const std::uint32_t p = wk.order;
    double m0 = 0, mo = 0, mom1 = 0;
    bool bad = false;
    for (int c = 0; c < 7; ++c) {
        const double *t = wk.tc.data() + static_cast<std::size_t>(c) * (p + 1u);
        const double a0 = std::fabs(t[0]), ao = std::fabs(t[p]), aom1 = std::fabs(t[p - 1u]);
        if (!std::isfinite(a0) || !std::isfinite(ao) || !std::isfinite(aom1)) {
            bad = true;
        }
        m0 = std::max(m0, a0);
        mo = std::max(mo, ao);
        mom1 = std::max(mom1, aom1);
    }
    if (bad) {
        return std::numeric_limits<double>::quiet_NaN();
    }
    const double num = m0 < 1. ? 1. : m0;
    const double rho_o = det_pow(num / mo, 1. / static_cast<double>(p));
    const double rho_om1 = det_pow(num / mom1, 1. / static_cast<double>(p - 1u));
    const double rho_m = std::min(rho_o, rho_om1);
    return rho_m * rhofac(p);
}

inline double horner(const double *t, double h, std::uint32_t p)
{
    double acc = t[p];
    for (std::uint32_t o = 1; o <= p; ++o) {
        acc = std::fma(acc, h, t[p - o]);
    }
    return acc;
}

// Substep data of one particle (sim_data::step_data, include/cascade/detail/sim_data.hpp:144-159).
struct step_data {
    std::vector<double> tcs;     // [S][7][p+1]
    std::vector<dfloat> tcoords; // [S]
};

struct prop_result {
    // 0 time_limit, 1 stopping terminal event, 2 err_nf_state
    int status = 0;
    std::uint32_t ev_idx = 0;
    double ev_time = 0; // relative to the beginning of the superstep
    double nf_last_t = 0;
};

// Event polynomial g(tau) of terminal event ev (0-based in heyoka's event list),
// from the x, y, z Taylor coefficients: src/sim_basic.cpp:509-533.
inline void event_poly(const taylor_cfg &cfg, taylor_work &wk, bool is_exit)
{
    const std::uint32_t p = wk.order;
    const double *x = wk.c(0), *y = wk.c(1), *z = wk.c(2);
    double *g = wk.g.data();
    if (is_exit || cfg.reentry_kind == 1) {
        const double R = is_exit ? cfg.exit_radius : cfg.reentry[0];
        for (std::uint32_t n = 0; n <= p; ++n) {
            g[n] = (conv_sq(x, n) + conv_sq(y, n)) + conv_sq(z, n);
        }
        g[0] -= R * R;
    } else {
        const double a = cfg.reentry[0], b = cfg.reentry[1], c = cfg.reentry[2];
        const double kx = (b * c) * (b * c), ky = (a * c) * (a * c), kz = (a * b) * (a * b);
        for (std::uint32_t n = 0; n <= p; ++n) {
            g[n] = (kx * conv_sq(x, n) + ky * conv_sq(y, n)) + kz * conv_sq(z, n);
        }
        g[0] -= a * a * b * b * c * c;
    }
}

// Numerical integration of one particle over the superstep [0, delta_t], recording
// Taylor coefficients and substep end times: src/sim_propagate.cpp:927-1056 (scalar
// path; the batch path :650-877 is lane-wise identical), with heyoka's
// propagate_for()/step_impl() semantics restated (EXTERNAL).
inline prop_result propagate_particle(const taylor_cfg &cfg, taylor_work &wk, const double st_in[7],
                                      const dfloat &init_time, double delta_t, step_data &sd, double final_state[6])
{
    const std::uint32_t p = cfg.order;
    wk.prepare(p);
    sd.tcs.clear();
    sd.tcoords.clear();

    double st[7];
    for (int j = 0; j < 6; ++j) {
        st[j] = st_in[j];
    }
    // src/sim_propagate.cpp:160
    st[6] = std::sqrt(st[0] * st[0] + st[1] * st[1] + st[2] * st[2]);

    dfloat t = init_time;
    const dfloat t_end = init_time + delta_t;
    prop_result res;

    while (true) {
        const dfloat rem = t_end - t;
        const double max_dt = rem.hi > 0 ? rem.hi : 0.;

        wk.time0 = t.hi;
        compute_tcs(cfg, wk, st);
        double h = determine_h(wk);
        bool tlimit = false;
        if (!(h < max_dt)) {
            // NOTE: also taken when h is NaN; the state check below then reports
            // the non-finite state.
            if (!std::isnan(h)) {
                h = max_dt;
                tlimit = true;
            }
        }
        if (std::isnan(h)) {
            res.status = 2;
            res.nf_last_t = sd.tcoords.empty() ? 0. : static_cast<double>(sd.tcoords.back());
            return res;
        }

        // Terminal event detection in [0, h) (heyoka detect_events, EXTERNAL).
        bool ev_found = false;
        double ev_t = 0;
        std::uint32_t ev_idx = 0;
        auto scan_event = [&](bool is_exit, std::uint32_t idx, int direction) {
            event_poly(cfg, wk, is_exit);
            run_poly_root_finding(wk.g.data(), p, h, direction, wk.rfs, wk.rf_st, [&](double root) {
                if (!ev_found || root < ev_t) {
                    ev_found = true;
                    ev_t = root;
                    ev_idx = idx;
                }
            });
        };
        if (h > 0) {
            if (cfg.with_exit()) {
                scan_event(true, cfg.exit_event_idx(), 1);
            }
            if (cfg.with_reentry()) {
                scan_event(false, cfg.reentry_event_idx(), -1);
            }
        }
        if (ev_found) {
            h = ev_t;
            tlimit = false;
        }

        // Advance the state by Horner evaluation, advance the double-length time.
        double nst[7];
        bool finite = true;
        for (int c = 0; c < 7; ++c) {
            nst[c] = horner(wk.c(c), h, p);
            finite = finite && std::isfinite(nst[c]);
        }
        const dfloat nt = t + h;
        if (!finite || !df_isfinite(nt)) {
            res.status = 2;
            res.nf_last_t = sd.tcoords.empty() ? 0. : static_cast<double>(sd.tcoords.back());
            return res;
        }
        t = nt;
        for (int c = 0; c < 7; ++c) {
            st[c] = nst[c];
        }

        // The propagate callback: src/sim_propagate.cpp:969-1007.
        if (h != 0.) {
            const dfloat tc = t - init_time;
            if (!df_isfinite(tc)) {
                throw std::invalid_argument("A non-finite time coordinate was generated during the numerical "
                                            "integration of a particle");
            }
            sd.tcoords.push_back(tc);
            sd.tcs.insert(sd.tcs.end(), wk.tc.begin(), wk.tc.end());
        }

        if (ev_found) {
            res.status = 1;
            res.ev_idx = ev_idx;
            res.ev_time = static_cast<double>(t - init_time);
            break;
        }
        if (tlimit) {
            break;
        }
    }

    for (int j = 0; j < 6; ++j) {
        final_state[j] = st[j];
    }
    return res;
}

// get_chunk_begin_end: src/sim_basic.cpp:87-103.
inline std::array<double, 2> get_chunk_begin_end(unsigned chunk_idx, double ct, unsigned nchunks, double delta_t)
{
    const double cbegin = ct * chunk_idx;
    const double cend = (chunk_idx == nchunks - 1u) ? delta_t : (ct * (chunk_idx + 1u));
    if (!std::isfinite(cbegin) || !std::isfinite(cend) || !(cend > cbegin) || cbegin < 0 || cend > delta_t) {
        throw std::invalid_argument("Invalid chunk range [" + std::to_string(cbegin) + ", " + std::to_string(cend)
                                    + ")");
    }
    return {cbegin, cend};
}

// Substep range [first, last) covering a chunk: src/sim_propagate.cpp:275-287.
inline std::pair<std::size_t, std::size_t> substep_range(const std::vector<dfloat> &tcoords, const dfloat &chunk_begin,
                                                        const dfloat &chunk_end)
{
    const auto b = tcoords.begin(), e = tcoords.end();
    const auto it_begin = std::upper_bound(b, e, chunk_begin);
    auto it_end = std::lower_bound(it_begin, e, chunk_end);
    it_end += (it_end != e);
    return {static_cast<std::size_t>(it_begin - b), static_cast<std::size_t>(it_end - b)};
}

// compute_particle_aabb: src/sim_propagate.cpp:217-395. lb/ub point to 4 floats.
inline void compute_particle_aabb(const step_data &sd, std::uint32_t order, double chunk_begin_d, double chunk_end_d,
                                  double p_radius, double conj_thresh, float *lb, float *ub, std::size_t pidx)
{
    constexpr auto finf = std::numeric_limits<float>::infinity();
    for (int i = 0; i < 4; ++i) {
        lb[i] = finf;
        ub[i] = -finf;
    }
    if (!std::isfinite(p_radius) || p_radius < 0) {
        throw std::invalid_argument("An invalid particle size of " + std::to_string(p_radius)
                                    + " was detected for the particle at index " + std::to_string(pidx));
    }
    const double conj_radius = conj_thresh / 2;
    const dfloat chunk_begin(chunk_begin_d), chunk_end(chunk_end_d);
    const auto [s0, s1] = substep_range(sd.tcoords, chunk_begin, chunk_end);
    const std::size_t np1 = order + 1u;

    for (std::size_t s = s0; s < s1; ++s) {
        const dfloat ss_start = (s == 0u) ? dfloat(0.) : sd.tcoords[s - 1u];
        const dfloat ev_lb = std::max(chunk_begin, ss_start);
        const dfloat ev_ub = std::min(chunk_end, sd.tcoords[s]);
        const double h_int_lb = static_cast<double>(ev_lb - ss_start);
        const double h_int_ub = static_cast<double>(ev_ub - ss_start);
        const ival h_int(h_int_lb, h_int_ub);
        const double *base = sd.tcs.data() + s * 7u * np1;
        const int cidx[4] = {0, 1, 2, 6};
        for (int i = 0; i < 4; ++i) {
            ival v = poly_eval(base + cidx[i] * np1, h_int, order);
            v.lower -= std::max(p_radius, conj_radius);
            v.upper += std::max(p_radius, conj_radius);
            const float fl = std::nextafter(static_cast<float>(v.lower), -finf);
            const float fu = std::nextafter(static_cast<float>(v.upper), finf);
            if (!std::isfinite(fl) || !std::isfinite(fu)) {
                throw std::invalid_argument("The computation of the bounding box for the particle at index "
                                            + std::to_string(pidx) + " produced a non-finite bound");
            }
            lb[i] = std::min(lb[i], fl);
            ub[i] = std::max(ub[i], fu);
        }
    }
}

// disc_single_coord: src/sim_propagate.cpp:94-124 (all in f32).
inline std::uint64_t disc_single_coord(float x, float min, float max)
{
    const float isize = max - min;
    float rx = (x - min) / isize;
    rx = rx >= 0.f ? rx : 0.f;
    rx *= static_cast<float>(static_cast<std::uint64_t>(1) << 16);
    // NOTE: saturating conversion (the reference's cast is UB out of range; its
    // result is clamped below anyway).
    std::uint64_t retval;
    if (rx >= 18446744073709551616.f) {
        retval = ~static_cast<std::uint64_t>(0);
    } else {
        retval = static_cast<std::uint64_t>(rx);
    }
    return std::min(retval, static_cast<std::uint64_t>((static_cast<std::uint64_t>(1) << 16) - 1u));
}

// 4 x 16-bit Morton interleave: bit b of field k goes to result bit 4b + k.
// Restates MortonNDLutEncoder<4,16,8>::Encode (src/detail/mortonND_LUT.h:166-323);
// checked against the reference header itself in oracle/_ref (tests/test_oracle_ref_pin.py).
inline std::uint64_t morton4_16(std::uint64_t n0, std::uint64_t n1, std::uint64_t n2, std::uint64_t n3)
{
    auto spread = [](std::uint64_t v) {
        v &= 0xffffull;
        v = (v | (v << 24)) & 0x000000ff000000ffull;
        v = (v | (v << 12)) & 0x000f000f000f000full;
        v = (v | (v << 6)) & 0x0303030303030303ull;
        v = (v | (v << 3)) & 0x1111111111111111ull;
        return v;
    };
    return spread(n0) | (spread(n1) << 1) | (spread(n2) << 2) | (spread(n3) << 3);
}

// Morton codes of one chunk: src/sim_propagate.cpp:440-479. gub is bumped IN PLACE (:448-449).
inline void morton_encode_chunk(std::size_t nparts, const float *lbs, const float *ubs, const float glb[4],
                                float gub[4], std::uint64_t *mcodes)
{
    constexpr auto finf = std::numeric_limits<float>::infinity();
    for (int i = 0; i < 4; ++i) {
        gub[i] = std::nextafter(gub[i], finf);
        if (!std::isfinite(gub[i] - glb[i])) {
            throw std::invalid_argument("A global bounding box with non-finite boundaries and/or size was generated");
        }
    }
    for (std::size_t pidx = 0; pidx < nparts; ++pidx) {
        float ctr[4];
        for (int i = 0; i < 4; ++i) {
            ctr[i] = lbs[pidx * 4u + i] / 2 + ubs[pidx * 4u + i] / 2;
        }
        const auto n0 = disc_single_coord(ctr[0], glb[0], gub[0]);
        const auto n1 = disc_single_coord(ctr[1], glb[1], gub[1]);
        const auto n2 = disc_single_coord(ctr[2], glb[2], gub[2]);
        const auto n3 = disc_single_coord(ctr[3], glb[3], gub[3]);
        mcodes[pidx] = morton4_16(n0, n1, n2, n3);
    }
}

// Indirect sort: src/sim_propagate.cpp:483-486 (restated STABLE, see header).
inline void sort_indices(std::size_t nparts, const std::uint64_t *mcodes, std::uint32_t *vidx)
{
    for (std::size_t i = 0; i < nparts; ++i) {
        vidx[i] = static_cast<std::uint32_t>(i);
    }
    std::stable_sort(vidx, vidx + nparts, [mcodes](std::uint32_t a, std::uint32_t b) { return mcodes[a] < mcodes[b]; });
}

// bvh_node: include/cascade/detail/sim_data.hpp:193-207 (60 bytes).
struct bvh_node {
    std::uint32_t begin, end;
    std::int32_t parent, left, right;
    float lb[4], ub[4];
    std::uint32_t nn_level;
    int split_idx;
};
static_assert(sizeof(bvh_node) == 60, "bvh_node layout");

// construct_bvh_trees_parallel for one chunk: src/sim_bvh.cpp:129-489.
inline void bvh_build(std::size_t nparts, const std::uint64_t *srt_mcodes, const float *srt_lbs, const float *srt_ubs,
                      std::vector<bvh_node> &tree)
{
    constexpr auto finf = std::numeric_limits<float>::infinity();
    auto default_node = [&](std::uint32_t b, std::uint32_t e, std::int32_t parent, int split_idx) {
        bvh_node n;
        n.begin = b;
        n.end = e;
        n.parent = parent;
        n.left = -1;
        n.right = -1;
        for (int i = 0; i < 4; ++i) {
            n.lb[i] = finf;
            n.ub[i] = -finf;
        }
        n.nn_level = 0;
        n.split_idx = split_idx;
        return n;
    };

    tree.clear();
    tree.push_back(default_node(0, static_cast<std::uint32_t>(nparts), -1, 0));

    std::uint32_t cur_n_nodes = 1;
    std::vector<std::uint32_t> nc_buf, ps_buf, nplc_buf;

    while (cur_n_nodes != 0u) {
        const auto cur_tree_size = tree.size();
        const auto n_begin = cur_tree_size - cur_n_nodes;
        const auto n_end = cur_tree_size;
        auto nn_next_level = cur_n_nodes * 2u;
        nc_buf.resize(cur_n_nodes);
        ps_buf.resize(cur_n_nodes);
        nplc_buf.resize(cur_n_nodes);

        std::uint32_t n_leaf_nodes = 0;
        for (auto node_idx = n_begin; node_idx != n_end; ++node_idx) {
            auto &cur_node = tree[node_idx];
            bool is_leaf_node = false;
            const std::uint64_t *split_ptr = nullptr;
            const auto mcodes_begin = srt_mcodes + cur_node.begin;
            const auto mcodes_end = srt_mcodes + cur_node.end;

            if (cur_node.end - cur_node.begin > 1u && cur_node.split_idx <= 63) {
                auto find_split = [&]() {
                    const auto mask = static_cast<std::uint64_t>(1) << (63 - cur_node.split_idx);
                    return std::lower_bound(mcodes_begin, mcodes_end, 1u,
                                            [mask](std::uint64_t mcode, unsigned val) { return (mcode & mask) < val; });
                };
                split_ptr = find_split();
                while (split_ptr == mcodes_begin || split_ptr == mcodes_end) {
                    if (cur_node.split_idx == 63) {
                        is_leaf_node = true;
                        break;
                    }
                    ++cur_node.split_idx;
                    split_ptr = find_split();
                }
            } else {
                is_leaf_node = true;
            }

            if (is_leaf_node) {
                nc_buf[node_idx - n_begin] = 0;
                nplc_buf[node_idx - n_begin] = 0;
                ++n_leaf_nodes;
                for (auto pidx = cur_node.begin; pidx != cur_node.end; ++pidx) {
                    for (int i = 0; i < 4; ++i) {
                        cur_node.lb[i] = std::min(cur_node.lb[i], srt_lbs[pidx * 4u + i]);
                        cur_node.ub[i] = std::max(cur_node.ub[i], srt_ubs[pidx * 4u + i]);
                    }
                }
            } else {
                nc_buf[node_idx - n_begin] = 2;
                nplc_buf[node_idx - n_begin] = static_cast<std::uint32_t>(split_ptr - mcodes_begin);
            }
        }

        nn_next_level -= n_leaf_nodes * 2u;
        tree.resize(cur_tree_size + nn_next_level);

        std::uint32_t run = 0;
        for (std::size_t i = 0; i < nc_buf.size(); ++i) {
            run += nc_buf[i];
            ps_buf[i] = run;
        }

        for (auto node_idx = n_begin; node_idx != n_end; ++node_idx) {
            auto &cur_node = tree[node_idx];
            const auto nc = nc_buf[node_idx - n_begin];
            cur_node.nn_level = cur_n_nodes;
            if (nc != 0u) {
                const auto lsize = nplc_buf[node_idx - n_begin];
                const auto lc_idx = cur_tree_size + ps_buf[node_idx - n_begin] - 2u;
                cur_node.left = static_cast<std::int32_t>(lc_idx);
                cur_node.right = static_cast<std::int32_t>(lc_idx + 1u);
                tree[lc_idx] = default_node(cur_node.begin, cur_node.begin + lsize,
                                            static_cast<std::int32_t>(node_idx), cur_node.split_idx + 1);
                tree[lc_idx + 1u] = default_node(cur_node.begin + lsize, cur_node.end,
                                                 static_cast<std::int32_t>(node_idx), cur_node.split_idx + 1);
            }
        }
        cur_n_nodes = nn_next_level;
    }

    // Backwards pass for the internal nodes' AABBs: src/sim_bvh.cpp:411-485.
    auto n_begin = tree.size() - tree.back().nn_level;
    auto n_end = tree.size();
    if (n_begin != 0u) {
        auto new_n_end = n_begin;
        n_begin -= tree[n_begin - 1u].nn_level;
        n_end = new_n_end;
        while (true) {
            for (auto node_idx = n_begin; node_idx != n_end; ++node_idx) {
                auto &cur_node = tree[node_idx];
                if (cur_node.left != -1) {
                    const auto &lc = tree[static_cast<std::size_t>(cur_node.left)];
                    const auto &rc = tree[static_cast<std::size_t>(cur_node.right)];
                    for (int j = 0; j < 4; ++j) {
                        cur_node.lb[j] = std::min(lc.lb[j], rc.lb[j]);
                        cur_node.ub[j] = std::max(lc.ub[j], rc.ub[j]);
                    }
                }
            }
            if (n_begin == 0u) {
                break;
            }
            new_n_end = n_begin;
            n_begin -= tree[n_begin - 1u].nn_level;
            n_end = new_n_end;
        }
    }
}

// verify_bvh_trees_parallel invariants (src/sim_bvh.cpp:498-656), restated as a checker.
// Returns an empty string if the tree is valid.
inline std::string bvh_verify(std::size_t nparts, const std::uint64_t *srt_mcodes, const float *srt_lbs,
                              const float *srt_ubs, const bvh_node *tree, std::size_t n_nodes)
{
    if (n_nodes == 0u) {
        return "empty tree";
    }
    std::vector<char> covered(nparts, 0);
    for (std::size_t i = 0; i < n_nodes; ++i) {
        const auto &n = tree[i];
        if (!(n.end > n.begin) || n.end > nparts) {
            return "bad range at node " + std::to_string(i);
        }
        if (i == 0u) {
            if (n.parent != -1 || n.begin != 0u || n.end != nparts) {
                return "bad root";
            }
        } else {
            if (n.parent < 0 || static_cast<std::size_t>(n.parent) >= i) {
                return "bad parent at node " + std::to_string(i);
            }
            const auto &pn = tree[n.parent];
            if (pn.left != static_cast<std::int32_t>(i) && pn.right != static_cast<std::int32_t>(i)) {
                return "parent does not point to node " + std::to_string(i);
            }
        }
        float lb[4], ub[4];
        if (n.left == -1) {
            if (n.right != -1) {
                return "half leaf";
            }
            for (auto k = n.begin; k < n.end; ++k) {
                if (srt_mcodes[k] != srt_mcodes[n.begin]) {
                    return "leaf with different codes at node " + std::to_string(i);
                }
                if (covered[k]) {
                    return "particle in two leaves";
                }
                covered[k] = 1;
            }
            for (int j = 0; j < 4; ++j) {
                lb[j] = std::numeric_limits<float>::infinity();
                ub[j] = -lb[j];
            }
            for (auto k = n.begin; k < n.end; ++k) {
                for (int j = 0; j < 4; ++j) {
                    lb[j] = std::min(lb[j], srt_lbs[k * 4u + j]);
                    ub[j] = std::max(ub[j], srt_ubs[k * 4u + j]);
                }
            }
        } else {
            if (n.left <= static_cast<std::int32_t>(i) || n.right != n.left + 1
                || static_cast<std::size_t>(n.right) >= n_nodes) {
                return "bad children at node " + std::to_string(i);
            }
            const auto &lc = tree[n.left], &rc = tree[n.right];
            if (lc.begin != n.begin || lc.end != rc.begin || rc.end != n.end) {
                return "children ranges do not partition node " + std::to_string(i);
            }
            for (int j = 0; j < 4; ++j) {
                lb[j] = std::min(lc.lb[j], rc.lb[j]);
                ub[j] = std::max(lc.ub[j], rc.ub[j]);
            }
        }
        for (int j = 0; j < 4; ++j) {
            if (lb[j] != n.lb[j] || ub[j] != n.ub[j]) {
                return "bad AABB at node " + std::to_string(i);
            }
        }
    }
    for (std::size_t k = 0; k < nparts; ++k) {
        if (!covered[k]) {
            return "particle not covered by any leaf";
        }
    }
    return "";
}

// broad_phase_parallel for one chunk: src/sim_broad_phase.cpp:92-202.
inline void broad_phase(std::size_t nparts, const bvh_node *tree, const float *srt_lbs, const float *srt_ubs,
                        const std::uint32_t *vidx, const char *coll_active, const char *conj_active,
                        std::vector<std::pair<std::uint32_t, std::uint32_t>> &bp, std::size_t p_begin = 0,
                        std::size_t p_end = ~static_cast<std::size_t>(0))
{
    std::vector<std::int32_t> stack;
    if (p_end > nparts) {
        p_end = nparts;
    }
    for (std::size_t pidx = p_begin; pidx < p_end; ++pidx) {
        const auto orig_pidx = vidx[pidx];
        const auto coll_active_pidx = coll_active[orig_pidx];
        const auto conj_active_pidx = conj_active[orig_pidx];
        stack.clear();
        stack.push_back(0);
        const float *plb = srt_lbs + pidx * 4u, *pub = srt_ubs + pidx * 4u;
        do {
            const auto cur_node_idx = stack.back();
            stack.pop_back();
            const auto &cur_node = tree[static_cast<std::uint32_t>(cur_node_idx)];
            const auto &n_lb = cur_node.lb;
            const auto &n_ub = cur_node.ub;
            const bool overlap = (pub[0] >= n_lb[0] && plb[0] <= n_ub[0]) && (pub[1] >= n_lb[1] && plb[1] <= n_ub[1])
                                 && (pub[2] >= n_lb[2] && plb[2] <= n_ub[2])
                                 && (pub[3] >= n_lb[3] && plb[3] <= n_ub[3]);
            if (overlap) {
                if (cur_node.left == -1) {
                    for (auto i = cur_node.begin; i != cur_node.end; ++i) {
                        const auto orig_i = vidx[i];
                        if (orig_pidx >= orig_i) {
                            continue;
                        }
                        if (coll_active_pidx || conj_active_pidx || coll_active[orig_i] || conj_active[orig_i]) {
                            bp.emplace_back(orig_pidx, orig_i);
                        }
                    }
                } else {
                    stack.push_back(cur_node.left);
                    stack.push_back(cur_node.right);
                }
            }
        } while (!stack.empty());
    }
}

// verify_broad_phase_parallel (src/sim_broad_phase.cpp:221-321) restated as an
// O(N^2) checker over the UNSORTED AABBs: returns the exact overlap pairs.
inline void broad_phase_brute(std::size_t nparts, const float *lbs, const float *ubs, const char *coll_active,
                              const char *conj_active, std::vector<std::pair<std::uint32_t, std::uint32_t>> &out)
{
    for (std::size_t i = 0; i < nparts; ++i) {
        for (std::size_t j = i + 1u; j < nparts; ++j) {
            bool overlap = true;
            for (int k = 0; k < 4; ++k) {
                overlap = overlap && (ubs[i * 4u + k] >= lbs[j * 4u + k] && lbs[i * 4u + k] <= ubs[j * 4u + k]);
            }
            overlap = overlap && (coll_active[i] || conj_active[i] || coll_active[j] || conj_active[j]);
            if (overlap) {
                out.emplace_back(static_cast<std::uint32_t>(i), static_cast<std::uint32_t>(j));
            }
        }
    }
}

// conjunction: include/cascade/sim.hpp:90-104 (128 bytes; numpy dtype cascade.py/core.cpp:81).
struct conjunction {
    std::uint64_t i = 0, j = 0;
    double time = 0, dist = 0;
    double state_i[6] = {}, state_j[6] = {};
};
static_assert(sizeof(conjunction) == 128, "conjunction layout");

struct collision {
    std::uint64_t i, j;
    double time; // relative to the beginning of the superstep
};

struct np_scratch {
    std::vector<double> pbuf[14];
    std::vector<double> diff_input;
    rf_scratch rfs;
    rf_stats st;
    std::vector<double> tmp_conj;
    void prepare(std::uint32_t order)
    {
        for (auto &v : pbuf) {
            v.resize(order + 1u);
        }
        diff_input.resize(6u * (order + 1u));
    }
};

// Narrow phase for ONE broad-phase pair in one chunk: src/sim_narrow_phase.cpp:619-967.
inline void narrow_phase_pair(std::uint32_t pi, std::uint32_t pj, const step_data &sd_i, const step_data &sd_j,
                              std::uint32_t order, double p_rad_i, double p_rad_j, bool pi_coll_active,
                              bool pj_coll_active, bool pi_conj_active, bool pj_conj_active, const dfloat &chunk_begin,
                              const dfloat &chunk_end, const dfloat &init_time, double conj_thresh2, np_scratch &sc,
                              std::vector<collision> &coll_vec, std::vector<conjunction> &conj_vec)
{
    const bool pij_conj_active = pi_conj_active || pj_conj_active;
    const std::size_t np1 = order + 1u;
    sc.prepare(order);
    auto &xi_temp = sc.pbuf[0], &yi_temp = sc.pbuf[1], &zi_temp = sc.pbuf[2], &xj_temp = sc.pbuf[3],
         &yj_temp = sc.pbuf[4], &zj_temp = sc.pbuf[5], &ss_diff = sc.pbuf[6], &ss_diff_der = sc.pbuf[7],
         &vxi_temp = sc.pbuf[8], &vyi_temp = sc.pbuf[9], &vzi_temp = sc.pbuf[10], &vxj_temp = sc.pbuf[11],
         &vyj_temp = sc.pbuf[12], &vzj_temp = sc.pbuf[13];

    const auto [bi, ei] = substep_range(sd_i.tcoords, chunk_begin, chunk_end);
    const auto [bj, ej] = substep_range(sd_j.tcoords, chunk_begin, chunk_end);

    for (std::size_t it_i = bi, it_j = bj; it_i != ei && it_j != ej;) {
        const dfloat ss_start_i = (it_i == 0u) ? dfloat(0.) : sd_i.tcoords[it_i - 1u];
        const dfloat ss_start_j = (it_j == 0u) ? dfloat(0.) : sd_j.tcoords[it_j - 1u];
        const dfloat lb_i = std::max(chunk_begin, ss_start_i);
        const dfloat ub_i = std::min(chunk_end, sd_i.tcoords[it_i]);
        const dfloat lb_j = std::max(chunk_begin, ss_start_j);
        const dfloat ub_j = std::min(chunk_end, sd_j.tcoords[it_j]);
        const dfloat lb_rf = std::max(lb_i, lb_j);
        const dfloat ub_rf = std::min(ub_i, ub_j);
        const double delta_i = static_cast<double>(lb_rf - ss_start_i);
        const double delta_j = static_cast<double>(lb_rf - ss_start_j);
        const double rf_int = static_cast<double>(ub_rf - lb_rf);

        if (!std::isfinite(delta_i) || !std::isfinite(delta_j) || !std::isfinite(rf_int) || delta_i < 0 || delta_j < 0
            || rf_int < 0) {
            ++sc.st.n_warn;
            break;
        }

        const double *tcs_i = sd_i.tcs.data() + it_i * 7u * np1;
        const double *tcs_j = sd_j.tcs.data() + it_j * 7u * np1;
        const double *poly_xi = tcs_i, *poly_yi = tcs_i + np1, *poly_zi = tcs_i + 2u * np1;
        const double *poly_vxi = pij_conj_active ? tcs_i + 3u * np1 : nullptr;
        const double *poly_vyi = pij_conj_active ? tcs_i + 4u * np1 : nullptr;
        const double *poly_vzi = pij_conj_active ? tcs_i + 5u * np1 : nullptr;
        const double *poly_xj = tcs_j, *poly_yj = tcs_j + np1, *poly_zj = tcs_j + 2u * np1;
        const double *poly_vxj = pij_conj_active ? tcs_j + 3u * np1 : nullptr;
        const double *poly_vyj = pij_conj_active ? tcs_j + 4u * np1 : nullptr;
        const double *poly_vzj = pij_conj_active ? tcs_j + 5u * np1 : nullptr;

        if (delta_i != 0) {
            poly_translate_a(xi_temp.data(), poly_xi, delta_i, order);
            poly_xi = xi_temp.data();
            poly_translate_a(yi_temp.data(), poly_yi, delta_i, order);
            poly_yi = yi_temp.data();
            poly_translate_a(zi_temp.data(), poly_zi, delta_i, order);
            poly_zi = zi_temp.data();
            if (pij_conj_active) {
                poly_translate_a(vxi_temp.data(), poly_vxi, delta_i, order);
                poly_vxi = vxi_temp.data();
                poly_translate_a(vyi_temp.data(), poly_vyi, delta_i, order);
                poly_vyi = vyi_temp.data();
                poly_translate_a(vzi_temp.data(), poly_vzi, delta_i, order);
                poly_vzi = vzi_temp.data();
            }
        }
        if (delta_j != 0) {
            poly_translate_a(xj_temp.data(), poly_xj, delta_j, order);
            poly_xj = xj_temp.data();
            poly_translate_a(yj_temp.data(), poly_yj, delta_j, order);
            poly_yj = yj_temp.data();
            poly_translate_a(zj_temp.data(), poly_zj, delta_j, order);
            poly_zj = zj_temp.data();
            if (pij_conj_active) {
                poly_translate_a(vxj_temp.data(), poly_vxj, delta_j, order);
                poly_vxj = vxj_temp.data();
                poly_translate_a(vyj_temp.data(), poly_vyj, delta_j, order);
                poly_vyj = vyj_temp.data();
                poly_translate_a(vzj_temp.data(), poly_vzj, delta_j, order);
                poly_vzj = vzj_temp.data();
            }
        }

        double *di = sc.diff_input.data();
        std::copy(poly_xi, poly_xi + np1, di);
        std::copy(poly_yi, poly_yi + np1, di + np1);
        std::copy(poly_zi, poly_zi + np1, di + 2u * np1);
        std::copy(poly_xj, poly_xj + np1, di + 3u * np1);
        std::copy(poly_yj, poly_yj + np1, di + 4u * np1);
        std::copy(poly_zj, poly_zj + np1, di + 5u * np1);

        double *ss_diff_ptr = ss_diff.data();
        poly_ssdiff3(ss_diff_ptr, di, order);
        const double orig_const_cf = ss_diff_ptr[0];

        if (pi_coll_active || pj_coll_active) {
            ss_diff_ptr[0] -= (p_rad_i + p_rad_j) * (p_rad_i + p_rad_j);
            run_poly_root_finding(ss_diff_ptr, order, rf_int, -1, sc.rfs, sc.st, [&](double root) {
                const double tcoll = static_cast<double>(lb_rf + root);
                if (!std::isfinite(tcoll)) {
                    ++sc.st.n_warn;
                    return;
                }
                coll_vec.push_back({pi, pj, tcoll});
            });
        }

        if (pij_conj_active) {
            ss_diff_ptr[0] = orig_const_cf;
            const ival dist2_ieval = poly_eval(ss_diff_ptr, ival(0., rf_int), order);
            if (!std::isfinite(dist2_ieval.lower) || !std::isfinite(dist2_ieval.upper)) {
                ++sc.st.n_warn;
                break;
            }
            if (dist2_ieval.lower < conj_thresh2) {
                double *der = ss_diff_der.data();
                for (std::uint32_t i = 0; i < order; ++i) {
                    der[i] = (i + 1u) * ss_diff_ptr[i + 1u];
                }
                der[order] = 0;
                sc.tmp_conj.clear();
                run_poly_root_finding(der, order, rf_int, 1, sc.rfs, sc.st, [&](double root) {
                    // lb_rf passed as 0 (:879).
                    const double t = static_cast<double>(dfloat(0.) + root);
                    if (!std::isfinite(t)) {
                        ++sc.st.n_warn;
                        return;
                    }
                    sc.tmp_conj.push_back(t);
                });
                for (const double conj_tm : sc.tmp_conj) {
                    const double conj_dist2 = poly_eval(ss_diff_ptr, conj_tm, order);
                    if (!std::isfinite(conj_dist2)) {
                        ++sc.st.n_warn;
                        continue;
                    }
                    if (conj_dist2 < conj_thresh2) {
                        conjunction c;
                        c.i = pi;
                        c.j = pj;
                        c.time = static_cast<double>(init_time + (lb_rf + conj_tm));
                        c.dist = std::sqrt(std::max(conj_dist2, 0.));
                        const double *pis[6] = {poly_xi, poly_yi, poly_zi, poly_vxi, poly_vyi, poly_vzi};
                        const double *pjs[6] = {poly_xj, poly_yj, poly_zj, poly_vxj, poly_vyj, poly_vzj};
                        for (int k = 0; k < 6; ++k) {
                            c.state_i[k] = poly_eval(pis[k], conj_tm, order);
                            c.state_j[k] = poly_eval(pjs[k], conj_tm, order);
                        }
                        conj_vec.push_back(c);
                    }
                }
            }
        }

        if (sd_i.tcoords[it_i] < sd_j.tcoords[it_j]) {
            ++it_i;
        } else if (sd_j.tcoords[it_j] < sd_i.tcoords[it_i]) {
            ++it_j;
        } else {
            ++it_i;
            ++it_j;
        }
    }
}

} // namespace orc
