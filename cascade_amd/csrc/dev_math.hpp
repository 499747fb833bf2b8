// cascade_amd/csrc/dev_math.hpp
//
// Device-side arithmetic shared by the HIP kernels: double-length floats, interval
// arithmetic, the deterministic pow of the step-size rule, polynomial helpers and the
// real-root isolation / bracketed refinement used by terminal-event detection and the
// narrow phase. Everything is built from IEEE basic operations with contraction
// disabled (-ffp-contract=off), so results do not depend on the compiler's mood.
//
// Reference (paths relative to the esa/cascade tree):
//   src/detail/ival.hpp:25-62                      interval arithmetic
//   src/sim_narrow_phase.cpp:67-140                poly_eval / poly_eval_1 / rescale
//   src/sim_narrow_phase.cpp:145-201               bracketed_root_find (Boost toms748_solve)
//   src/sim_narrow_phase.cpp:208-422               run_poly_root_finding
//   src/sim_jit.cpp:64-113, 118-218, 234-244       pta / ssdiff3 / fex_check / rtscc / pt1
#pragma once

// Kernels that are not templates are defined once, in the main translation unit; the per-order
// translation units (cascade_order.hip, -DCSC_ORDER=P) see the same headers and give them
// internal linkage (unused there, dropped by the compiler).
#ifdef CSC_ORDER
#define CSC_GLOBAL static __global__
#else
#define CSC_GLOBAL __global__
#endif


#ifdef __HIPCC_RTC__
// Run-time compilation (hipRTC: the compiled tapes of user dynamics): the HIP runtime header is
// built in and the C++ standard headers are not available; these are the few names used from them.
namespace std
{
using int8_t = signed char;
using uint8_t = unsigned char;
using int32_t = int;
using uint32_t = unsigned int;
using int64_t = long long;
using uint64_t = unsigned long long;
using size_t = decltype(sizeof(0));
} // namespace std
#define DBL_MAX __DBL_MAX__
#define DBL_MIN __DBL_MIN__
#define DBL_EPSILON __DBL_EPSILON__
#define FLT_MAX __FLT_MAX__
#else
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#endif

namespace csc
{

#define CSC_DEV __device__ __forceinline__
#define CSC_DEV_NOINLINE __device__ __noinline__

// MI355X: 8 XCDs, each with its own L2; the dispatcher deals workgroups to XCDs round-robin
// by linear workgroup id. Kernels whose neighbouring workgroups share data (tree nodes,
// Taylor rows of Morton neighbours) remap the hardware id so that every XCD works on runs of
// CH consecutive logical workgroups (its L2 then sees the reuse), while the runs themselves
// stay interleaved across the XCDs (so that dense and sparse regions of the population are
// spread evenly). The ragged tail keeps the identity mapping.
#ifndef CSC_XCD_RUN
#define CSC_XCD_RUN 16
#endif
template <unsigned CH = CSC_XCD_RUN>
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb)
{
    constexpr unsigned NXCD = 8, SPAN = NXCD * CH;
    const unsigned full = (nb / SPAN) * SPAN;
    if (b >= full) {
        return b;
    }
    const unsigned span = b / SPAN, r = b % SPAN;
    return span * SPAN + (r % NXCD) * CH + r / NXCD;
}

// std::min / std::max semantics (NOT fmin/fmax): (b < a) ? b : a.
CSC_DEV double dmin(double a, double b)
{
    return (b < a) ? b : a;
}
CSC_DEV double dmax(double a, double b)
{
    return (a < b) ? b : a;
}
CSC_DEV float fminf_std(float a, float b)
{
    return (b < a) ? b : a;
}
CSC_DEV float fmaxf_std(float a, float b)
{
    return (a < b) ? b : a;
}
CSC_DEV int sgn(double v)
{
    return (0. < v) - (v < 0.);
}
CSC_DEV bool finite_d(double v)
{
    return fabs(v) <= DBL_MAX;
}

// ---------------------------------------------------------------------------
// Double-length floats (heyoka::detail::dfloat<double> semantics, EXTERNAL).
// ---------------------------------------------------------------------------
struct dfl {
    double hi, lo;
};

CSC_DEV dfl df(double h)
{
    return dfl{h, 0.};
}

CSC_DEV dfl df_add(dfl x, dfl y)
{
    // Knuth two-sum on the high parts.
    const double xx = x.hi + y.hi;
    const double z = xx - x.hi;
    double yy = (x.hi - (xx - z)) + (y.hi - z);
    yy += x.lo + y.lo;
    // Normalise (Dekker fast two-sum).
    const double h = xx + yy;
    const double l = yy - (h - xx);
    return dfl{h, l};
}

CSC_DEV dfl df_neg(dfl x)
{
    return dfl{-x.hi, -x.lo};
}
CSC_DEV dfl df_sub(dfl x, dfl y)
{
    return df_add(x, df_neg(y));
}
CSC_DEV bool df_lt(dfl a, dfl b)
{
    return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
}
CSC_DEV dfl df_max(dfl a, dfl b)
{
    return df_lt(a, b) ? b : a;
}
CSC_DEV dfl df_min(dfl a, dfl b)
{
    return df_lt(b, a) ? b : a;
}
CSC_DEV bool df_finite(dfl a)
{
    return finite_d(a.hi) && finite_d(a.lo);
}

// ---------------------------------------------------------------------------
// Deterministic pow for the Jorba-Zou step-size rule (same spelled-out algorithm as
// the test oracle; see DESIGN.md "step size").
// ---------------------------------------------------------------------------
CSC_DEV double det_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752) {
        m = m * 2.;
        e -= 1;
    }
    const double s = (m - 1.) / (m + 1.);
    const double s2 = s * s;
    double p = 1. / 23.;
    p = fma(p, s2, 1. / 21.);
    p = fma(p, s2, 1. / 19.);
    p = fma(p, s2, 1. / 17.);
    p = fma(p, s2, 1. / 15.);
    p = fma(p, s2, 1. / 13.);
    p = fma(p, s2, 1. / 11.);
    p = fma(p, s2, 1. / 9.);
    p = fma(p, s2, 1. / 7.);
    p = fma(p, s2, 1. / 5.);
    p = fma(p, s2, 1. / 3.);
    p = fma(p, s2, 1.);
    const double at = s * p;
    return fma(static_cast<double>(e), 0.69314718055994530942, 2. * at);
}

CSC_DEV double det_exp(double t)
{
    const double k = rint(t * 1.44269504088896340736);
    double r = fma(-k, 0.693147180369123816490, t);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1. / 87178291200.;
    p = fma(p, r, 1. / 6227020800.);
    p = fma(p, r, 1. / 479001600.);
    p = fma(p, r, 1. / 39916800.);
    p = fma(p, r, 1. / 3628800.);
    p = fma(p, r, 1. / 362880.);
    p = fma(p, r, 1. / 40320.);
    p = fma(p, r, 1. / 5040.);
    p = fma(p, r, 1. / 720.);
    p = fma(p, r, 1. / 120.);
    p = fma(p, r, 1. / 24.);
    p = fma(p, r, 1. / 6.);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.);
    p = fma(p, r, 1.);
    return ldexp(p, static_cast<int>(k));
}

CSC_DEV double det_pow(double x, double y)
{
    if (!(x > 0.)) {
        return x;
    }
    if (!(x <= DBL_MAX)) {
        return x;
    }
    return det_exp(y * det_log(x));
}

// ---------------------------------------------------------------------------
// Interval arithmetic (src/detail/ival.hpp:25-62).
// ---------------------------------------------------------------------------
struct ivl {
    double lo, hi;
};

CSC_DEV ivl iv_mul(ivl a, ivl b)
{
    const double t1 = a.lo * b.lo;
    const double t2 = a.lo * b.hi;
    const double t3 = a.hi * b.lo;
    const double t4 = a.hi * b.hi;
    return ivl{dmin(dmin(t1, t2), dmin(t3, t4)), dmax(dmax(t1, t2), dmax(t3, t4))};
}

// acc = c + acc * h   (src/sim_propagate.cpp:342-349, src/sim_narrow_phase.cpp:80-91)
CSC_DEV ivl iv_horner_step(double c, ivl acc, ivl h)
{
    const ivl m = iv_mul(acc, h);
    return ivl{c + m.lo, c + m.hi};
}

template <int P>
CSC_DEV ivl iv_poly_eval(const double *a, ivl h)
{
    ivl acc{a[P], a[P]};
#pragma unroll
    for (int i = 1; i <= P; ++i) {
        acc = iv_horner_step(a[P - i], acc, h);
    }
    return acc;
}

// The same enclosure for 0 <= h.lo <= h.hi (the sub-interval of a Taylor step that a chunk covers).
// There the extreme products follow from the signs of acc's bounds, and because rounding is
// monotonic the product selected IS the minimum (maximum) of iv_mul's four rounded products: the
// bounds are bit-identical, at 2 multiplications and 2 selects per step instead of 4 and 12. (A zero
// product may differ in sign; the addition of the next coefficient removes it, and a series of
// zeros ends in nextafterf, which does not see the sign. Non-finite coefficients can differ; the
// caller rejects a non-finite box either way.)
template <int P>
CSC_DEV ivl iv_poly_eval_nonneg(const double *a, ivl h)
{
    ivl acc{a[P], a[P]};
#pragma unroll
    for (int i = 1; i <= P; ++i) {
        // (selecting the factor, 1 compare + 2 v_cndmask, measured faster than v_min_f64 / v_max_f64
        // of both products: 3.13 against 3.30 ms for k_propagate_rows at 8 chunks)
        const double ml = acc.lo < 0. ? h.hi : h.lo;
        const double mh = acc.hi < 0. ? h.lo : h.hi;
        acc = ivl{a[P - i] + acc.lo * ml, a[P - i] + acc.hi * mh};
    }
    return acc;
}

// fex_check (heyoka llvm_add_fex_check, EXTERNAL; src/sim_jit.cpp:234).
template <int P>
CSC_DEV bool fex_excluded(const double *cf, double h)
{
    const ivl hi = (h >= 0.) ? ivl{0., h} : ivl{h, 0.};
    const ivl enc = iv_poly_eval<P>(cf, hi);
    const int s_lo = sgn(enc.lo), s_hi = sgn(enc.hi);
    return s_lo == s_hi && s_lo != 0;
}

// Slot in an output list for every lane that is active at the call, with ONE atomic per wave:
// the first active lane adds the number of active lanes to the counter. For appends in divergent
// code whose counter pointer the compiler cannot prove uniform (behind a reference in an
// out-of-line function): there it issues one atomic per lane, and ~10^7 atomics on one address
// take ~13 ns each -- longer than all the arithmetic of a conjunction-heavy narrow phase.
CSC_DEV std::uint32_t wave_append_slot(std::uint32_t *counter)
{
    const unsigned long long active = __ballot(1);
    const unsigned lane = __lane_id();
    std::uint32_t base = 0;
    if (lane == static_cast<unsigned>(__ffsll(static_cast<long long>(active)) - 1)) {
        base = atomicAdd(counter, static_cast<std::uint32_t>(__popcll(active)));
    }
    base = static_cast<std::uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
    return base + static_cast<std::uint32_t>(__popcll(active & ((1ull << lane) - 1ull)));
}

// ---------------------------------------------------------------------------
// Scalar polynomial helpers (runtime order; operate on arrays in private memory).
// ---------------------------------------------------------------------------
__device__ inline double poly_eval(const double *a, double x, int n)
{
    double ret = a[n];
    for (int i = 1; i <= n; ++i) {
        ret = a[n - i] + ret * x;
    }
    return ret;
}

__device__ inline double poly_eval_1(const double *a, double x, int n)
{
    double ret1 = a[n] * n;
    for (int i = 1; i < n; ++i) {
        ret1 = a[n - i] * (n - i) + ret1 * x;
    }
    return ret1;
}

// The same two evaluations with the order known at compile time (N > 0): fully unrolled, the
// coefficients stay in registers; N == 0 = runtime order n. Same operations in the same order.
template <int N>
__device__ __forceinline__ double poly_eval_n(const double *a, double x, int n)
{
    if constexpr (N > 0) {
        double ret = a[N];
#pragma unroll
        for (int i = 1; i <= N; ++i) {
            ret = a[N - i] + ret * x;
        }
        return ret;
    } else {
        return poly_eval(a, x, n);
    }
}

template <int N>
__device__ __forceinline__ double poly_eval_1_n(const double *a, double x)
{
    double ret1 = a[N] * N;
#pragma unroll
    for (int i = 1; i < N; ++i) {
        ret1 = a[N - i] * (N - i) + ret1 * x;
    }
    return ret1;
}

__device__ inline void poly_rescale(double *ret, const double *a, double scal, int n)
{
    double cur_f = 1;
    for (int i = 0; i <= n; ++i) {
        ret[i] = cur_f * a[i];
        cur_f *= scal;
    }
}

__device__ inline void poly_rescale_p2(double *ret, const double *a, int n)
{
    double cur_f = 1;
    for (int i = 0; i <= n; ++i) {
        ret[n - i] = cur_f * a[n - i];
        cur_f *= 2;
    }
}

// Binomial coefficients C(n, k) for n, k <= 24 = the largest Taylor order built (exact in double), row-major.
constexpr int BINOM_N = 25;
struct binom_tab {
    double v[BINOM_N * BINOM_N];
};
constexpr binom_tab make_binom_tab()
{
    binom_tab t{};
    for (int n = 0; n < BINOM_N; ++n) {
        for (int k = 0; k < BINOM_N; ++k) {
            t.v[n * BINOM_N + k] = 0.;
        }
        t.v[n * BINOM_N] = 1.;
        for (int k = 1; k <= n; ++k) {
            t.v[n * BINOM_N + k] = t.v[(n - 1) * BINOM_N + k - 1] + (k <= n - 1 ? t.v[(n - 1) * BINOM_N + k] : 0.);
        }
    }
    return t;
}
// Single translation unit (unity build, see cascade_hip.hip): one definition.
__constant__ binom_tab c_binom_tab = make_binom_tab();
#define c_binom c_binom_tab.v

// ret[i] = sum_{k >= i} C(k, i) cf[k], ascending k (heyoka poly_translate_1, EXTERNAL).
__device__ inline void poly_translate_1(double *ret, const double *cf, int n)
{
    for (int i = 0; i <= n; ++i) {
        double acc = cf[i];
        for (int k = i + 1; k <= n; ++k) {
            acc += c_binom[k * BINOM_N + i] * cf[k];
        }
        ret[i] = acc;
    }
}

// poly_rtscc (heyoka, EXTERNAL): out1 = reverse(cf), out2(x) = out1(x + 1); returns the
// number of sign changes in out2 ignoring zeros.
__device__ inline int poly_rtscc(double *out1, double *out2, const double *cf, int n)
{
    for (int i = 0; i <= n; ++i) {
        out1[i] = cf[n - i];
    }
    poly_translate_1(out2, out1, n);
    int count = 0, last = 0;
    for (int i = 0; i <= n; ++i) {
        const int s = sgn(out2[i]);
        if (s != 0) {
            if (last != 0 && s != last) {
                ++count;
            }
            last = s;
        }
    }
    return count;
}

// pta_cfunc (src/sim_jit.cpp:64-113): out(x) = cf(x + a).
__device__ inline void poly_translate_a(double *out, const double *cf, double a, int n)
{
    double a_pows[BINOM_N];
    a_pows[0] = 1.;
    a_pows[1] = a;
    for (int i = 2; i <= n; ++i) {
        if (i % 2 == 0) {
            a_pows[i] = a_pows[i / 2] * a_pows[i / 2];
        } else {
            a_pows[i] = a_pows[i / 2 + 1] * a_pows[i / 2];
        }
    }
    for (int i = 0; i <= n; ++i) {
        double acc = 0;
        for (int k = i; k <= n; ++k) {
            acc += (cf[k] * c_binom[k * BINOM_N + (k - i)]) * a_pows[k - i];
        }
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------
// TOMS 748 (published algorithm of Boost.Math toms748_solve, EXTERNAL), called as in
// src/sim_narrow_phase.cpp:145-201: eps_tolerance<double>(), at most 100 iterations,
// result = midpoint of the final bracket.
// ---------------------------------------------------------------------------
namespace t748
{

__device__ inline bool tol_eps(double a, double b)
{
    const double eps = 4 * DBL_EPSILON;
    return fabs(a - b) <= eps * dmin(fabs(a), fabs(b));
}

__device__ inline double safe_div(double num, double denom, double r)
{
    if (fabs(denom) < 1) {
        if (fabs(denom * DBL_MAX) <= fabs(num)) {
            return r;
        }
    }
    return num / denom;
}

template <int N = 0>
__device__ __forceinline__ void bracket(const double *p, int n, double &a, double &b, double c, double &fa, double &fb,
                               double &d, double &fd)
{
    const double tol = DBL_EPSILON * 2;
    if ((b - a) < 2 * tol * a) {
        c = a + (b - a) / 2;
    } else if (c <= a + fabs(a) * tol) {
        c = a + fabs(a) * tol;
    } else if (c >= b - fabs(b) * tol) {
        c = b - fabs(b) * tol;
    }
    const double fc = poly_eval_n<N>(p, c, n);
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

__device__ inline double secant_interpolate(double a, double b, double fa, double fb)
{
    const double tol = DBL_EPSILON * 5;
    const double c = a - (fa / (fb - fa)) * (b - a);
    if ((c <= a + fabs(a) * tol) || (c >= b - fabs(b) * tol)) {
        return (a + b) / 2;
    }
    return c;
}

__device__ inline double quadratic_interpolate(double a, double b, double d, double fa, double fb, double fd,
                                               unsigned count)
{
    double B = safe_div(fb - fa, b - a, DBL_MAX);
    double A = safe_div(fd - fb, d - b, DBL_MAX);
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

__device__ inline double cubic_interpolate(double a, double b, double d, double e, double fa, double fb, double fd,
                                           double fe)
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

// Returns an error flag (non-zero = domain error); [a, b] is the final bracket and
// used the number of iterations charged (incl. the two end-point evaluations).
template <int N = 0>
__device__ __forceinline__ int solve(const double *p, int n, double ax, double bx, double &a_out, double &b_out, int &used)
{
    const int limit = 100;
    int max_iter = limit - 2;
    const double fax = poly_eval_n<N>(p, ax, n), fbx = poly_eval_n<N>(p, bx, n);
    int count = max_iter;
    double a = ax, b = bx, fa = fax, fb = fbx, c, u, fu, a0, b0, d, fd, e, fe;
    const double mu = 0.5;

    if (!(a < b)) {
        used = limit;
        return 1;
    }
    if (tol_eps(a, b) || (fa == 0) || (fb == 0)) {
        if (fa == 0) {
            b = a;
        } else if (fb == 0) {
            a = b;
        }
        a_out = a;
        b_out = b;
        used = 2;
        return 0;
    }
    if (sgn(fa) * sgn(fb) > 0) {
        used = limit;
        return 1;
    }
    fe = e = fd = 1e5;
    d = 0;

    if (fa != 0) {
        c = secant_interpolate(a, b, fa, fb);
        bracket<N>(p, n, a, b, c, fa, fb, d, fd);
        --count;
        if (count && (fa != 0) && !tol_eps(a, b)) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
            e = d;
            fe = fd;
            bracket<N>(p, n, a, b, c, fa, fb, d, fd);
            --count;
        }
    }

    const double min_diff = DBL_MIN * 32;
    while (count && (fa != 0) && !tol_eps(a, b)) {
        a0 = a;
        b0 = b;
        bool prof = (fabs(fa - fb) < min_diff) || (fabs(fa - fd) < min_diff) || (fabs(fa - fe) < min_diff)
                    || (fabs(fb - fd) < min_diff) || (fabs(fb - fe) < min_diff) || (fabs(fd - fe) < min_diff);
        if (prof) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 2);
        } else {
            c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
        }
        e = d;
        fe = fd;
        bracket<N>(p, n, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        prof = (fabs(fa - fb) < min_diff) || (fabs(fa - fd) < min_diff) || (fabs(fa - fe) < min_diff)
               || (fabs(fb - fd) < min_diff) || (fabs(fb - fe) < min_diff) || (fabs(fd - fe) < min_diff);
        if (prof) {
            c = quadratic_interpolate(a, b, d, fa, fb, fd, 3);
        } else {
            c = cubic_interpolate(a, b, d, e, fa, fb, fd, fe);
        }
        bracket<N>(p, n, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        if (fabs(fa) < fabs(fb)) {
            u = a;
            fu = fa;
        } else {
            u = b;
            fu = fb;
        }
        c = u - 2 * (fu / (fb - fa)) * (b - a);
        if (fabs(c - u) > (b - a) / 2) {
            c = a + (b - a) / 2;
        }
        e = d;
        fe = fd;
        bracket<N>(p, n, a, b, c, fa, fb, d, fd);
        if ((0 == --count) || (fa == 0) || tol_eps(a, b)) {
            break;
        }
        if ((b - a) < mu * (b0 - a0)) {
            continue;
        }
        e = d;
        fe = fd;
        bracket<N>(p, n, a, b, a + (b - a) / 2, fa, fb, d, fd);
        --count;
    }
    max_iter -= count;
    if (fa == 0) {
        b = a;
    } else if (fb == 0) {
        a = b;
    }
    a_out = a;
    b_out = b;
    used = max_iter + 2;
    return 0;
}

} // namespace t748

// ---------------------------------------------------------------------------
// Real-root isolation + refinement on [0, rf_int) with direction filter
// (run_poly_root_finding, src/sim_narrow_phase.cpp:208-422), AFTER the fast exclusion
// check has failed to exclude the interval.
//
// MI355X design note: the reference keeps a LIFO work-list of up to 250 bisected
// polynomials per thread (42 KB). Here a lane keeps no work-list at all: because the
// list is LIFO and the right half is pushed last, the pending set is exactly "the left
// siblings of the right-turns on the current path", so the traversal is driven by a
// path word (one bit per level); a pending left sibling's polynomial is rebuilt by replaying the
// same rescale/translate sequence from the root polynomial, which reproduces the
// reference's arithmetic operation for operation.
//
// Roots are appended to roots[] (at most max_roots); returns the number of roots, or
// -1 if the isolation failed (work-list > 250 or more isolating intervals than the
// order), in which case roots found at interval lower bounds may already have been
// appended (as in the reference, :329-336 precede the failure check :373-381);
// n_lb_roots counts those.
// ---------------------------------------------------------------------------
constexpr int RF_PATH_WORDS = 18;                  // 1152 bits
constexpr int RF_MAX_DEPTH = RF_PATH_WORDS * 64 - 8; // deeper than any interval a double can halve to

struct rf_path {
    unsigned long long w[RF_PATH_WORDS];
    __device__ void clear()
    {
        for (int i = 0; i < RF_PATH_WORDS; ++i) {
            w[i] = 0;
        }
    }
    __device__ bool get(int i) const
    {
        return (w[i >> 6] >> (i & 63)) & 1ull;
    }
    __device__ void set(int i, bool v)
    {
        const unsigned long long m = 1ull << (i & 63);
        if (v) {
            w[i >> 6] |= m;
        } else {
            w[i >> 6] &= ~m;
        }
    }
};

constexpr int RF_MAX_ORDER = 20;

// poly: order+1 coefficients (NOT rescaled). direction: -1, 0, +1.
__device__ __forceinline__ int isolate_and_refine(const double *poly, int order, double rf_int, int direction, double *roots,
                                         int max_roots, int *warn)
{
    double root_poly[RF_MAX_ORDER + 1]; // poly rescaled to [0, 1)
    double tmp[RF_MAX_ORDER + 1], tmp1[RF_MAX_ORDER + 1], tmp2[RF_MAX_ORDER + 1];
    double isol_lb[RF_MAX_ORDER + 1], isol_ub[RF_MAX_ORDER + 1];
    int n_isol = 0, n_roots = 0;

    auto add_root = [&](double root) {
        if (!finite_d(root)) {
            *warn |= 1;
            return;
        }
        if (direction != 0) {
            const double der = poly_eval_1(poly, root, order);
            if (!finite_d(der)) {
                *warn |= 1;
                return;
            }
            if (sgn(der) != direction) {
                return;
            }
        }
        if (n_roots < max_roots) {
            roots[n_roots] = root;
        } else {
            *warn |= 2;
        }
        ++n_roots;
    };

    poly_rescale(root_poly, poly, rf_int, order);
    for (int i = 0; i <= order; ++i) {
        tmp[i] = root_poly[i];
    }

    rf_path path;
    path.clear();
    int depth = 0;       // depth of the current node (root = 0)
    int n_pending = 0;   // == wlist.size() after popping the current node
    double lb = 0., ub = 1.;
    bool loop_failed = false;

    while (true) {
        // --- process the current node (tmp, [lb, ub)) ---
        if (tmp[0] == 0) {
            bool all_fin = true;
            for (int i = 1; i <= order; ++i) {
                all_fin = all_fin && finite_d(tmp[i]);
            }
            if (all_fin) {
                add_root(lb * rf_int);
            }
        }
        const int n_sc = poly_rtscc(tmp1, tmp2, tmp, order);
        bool descend = false;
        if (n_sc == 1) {
            if (n_isol <= order) {
                isol_lb[n_isol] = lb;
                isol_ub[n_isol] = ub;
            }
            ++n_isol;
        } else if (n_sc > 1) {
            descend = true;
            n_pending += 2;
        }
        // NOTE: the depth guard has no counterpart in the reference (whose work-list holds the
        // polynomials themselves); it only bounds the path word and cannot trigger before an
        // interval has been halved ~1100 times, i.e. far past double precision.
        if (n_pending > 250 || n_isol > order || depth >= RF_MAX_DEPTH) {
            *warn |= 4;
            loop_failed = true;
            break;
        }
        if (descend) {
            // Next node = right child: 2^n q((x + 1) / 2).
            poly_rescale_p2(tmp1, tmp, order);
            poly_translate_1(tmp2, tmp1, order);
            for (int i = 0; i <= order; ++i) {
                tmp[i] = tmp2[i];
            }
            const double mid = (lb + ub) / 2;
            lb = mid;
            path.set(depth, true);
            ++depth;
            --n_pending; // popped the right child
            continue;
        }
        if (n_pending == 0) {
            break;
        }
        // Pop the deepest pending left sibling: find the deepest right-turn.
        int k = depth - 1;
        while (k >= 0 && !path.get(k)) {
            --k;
        }
        // k >= 0 is guaranteed by n_pending > 0.
        path.set(k, false);
        depth = k + 1;
        --n_pending;
        // Replay from the root along path[0 .. depth).
        for (int i = 0; i <= order; ++i) {
            tmp[i] = root_poly[i];
        }
        lb = 0.;
        ub = 1.;
        for (int lvl = 0; lvl < depth; ++lvl) {
            const double mid = (lb + ub) / 2;
            poly_rescale_p2(tmp1, tmp, order);
            if (path.get(lvl)) {
                poly_translate_1(tmp2, tmp1, order);
                for (int i = 0; i <= order; ++i) {
                    tmp[i] = tmp2[i];
                }
                lb = mid;
            } else {
                for (int i = 0; i <= order; ++i) {
                    tmp[i] = tmp1[i];
                }
                ub = mid;
            }
        }
    }

    if (n_isol > 0 && !loop_failed) {
        for (int k = 0; k < n_isol; ++k) {
            double lo = isol_lb[k], up = isol_ub[k];
            if (finite_d(lo) && finite_d(up) && up > lo) {
                up = nextafter(up, lo);
            }
            double a, b;
            int used;
            const int err = t748::solve(root_poly, order, lo, up, a, b, used);
            if (err != 0 || !(used < 100)) {
                *warn |= 8;
                continue;
            }
            add_root(((a + b) / 2) * rf_int);
        }
    }
    return loop_failed ? -1 - n_roots : n_roots;
}

template <int N>
__device__ __forceinline__ void poly_rescale_n(double *ret, const double *a, double scal)
{
    double cur_f = 1;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        ret[i] = cur_f * a[i];
        cur_f *= scal;
    }
}

template <int N>
__device__ __forceinline__ void poly_rescale_p2_n(double *ret, const double *a)
{
    double cur_f = 1;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        ret[N - i] = cur_f * a[N - i];
        cur_f *= 2;
    }
}

template <int N>
__device__ __forceinline__ void poly_translate_1_n(double *ret, const double *cf)
{
    constexpr binom_tab bt = make_binom_tab();
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        double acc = cf[i];
#pragma unroll
        for (int k = i + 1; k <= N; ++k) {
            acc += bt.v[k * BINOM_N + i] * cf[k];
        }
        ret[i] = acc;
    }
}

template <int N>
__device__ __forceinline__ int poly_rtscc_n(double *out1, double *out2, const double *cf)
{
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        out1[i] = cf[N - i];
    }
    poly_translate_1_n<N>(out2, out1);
    int count = 0, last = 0;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        const int s = sgn(out2[i]);
        if (s != 0) {
            if (last != 0 && s != last) {
                ++count;
            }
            last = s;
        }
    }
    return count;
}

// The same search with the order known at compile time: every coefficient loop is unrolled, so
// the polynomials (root_poly, tmp, tmp1, tmp2 and the caller's poly) live in registers instead
// of private memory, where each access of the runtime-order version above is a scratch round
// trip and a translation by 1 costs ~10^5 cycles. Statement for statement the function above
// (kept as the general fallback and as the comparison in the tests): same operations, same
// order, same bits.
template <int N>
__device__ __forceinline__ int isolate_and_refine_n(const double *poly, double rf_int, int direction, double *roots,
                                                    int max_roots, int *warn)
{
    constexpr int order = N;
    double root_poly[N + 1]; // poly rescaled to [0, 1)
    double tmp[N + 1], tmp1[N + 1], tmp2[N + 1];
    double isol_lb[N + 1], isol_ub[N + 1];
    int n_isol = 0, n_roots = 0;

    auto add_root = [&](double root) {
        if (!finite_d(root)) {
            *warn |= 1;
            return;
        }
        if (direction != 0) {
            const double der = poly_eval_1_n<N>(poly, root);
            if (!finite_d(der)) {
                *warn |= 1;
                return;
            }
            if (sgn(der) != direction) {
                return;
            }
        }
        if (n_roots < max_roots) {
            roots[n_roots] = root;
        } else {
            *warn |= 2;
        }
        ++n_roots;
    };

    poly_rescale_n<N>(root_poly, poly, rf_int);
#pragma unroll
    for (int i = 0; i <= order; ++i) {
        tmp[i] = root_poly[i];
    }

    rf_path path;
    path.clear();
    int depth = 0;       // depth of the current node (root = 0)
    int n_pending = 0;   // == wlist.size() after popping the current node
    double lb = 0., ub = 1.;
    bool loop_failed = false;

    while (true) {
        // --- process the current node (tmp, [lb, ub)) ---
        if (tmp[0] == 0) {
            bool all_fin = true;
#pragma unroll
            for (int i = 1; i <= order; ++i) {
                all_fin = all_fin && finite_d(tmp[i]);
            }
            if (all_fin) {
                add_root(lb * rf_int);
            }
        }
        const int n_sc = poly_rtscc_n<N>(tmp1, tmp2, tmp);
        bool descend = false;
        if (n_sc == 1) {
            if (n_isol <= order) {
                isol_lb[n_isol] = lb;
                isol_ub[n_isol] = ub;
            }
            ++n_isol;
        } else if (n_sc > 1) {
            descend = true;
            n_pending += 2;
        }
        // NOTE: the depth guard has no counterpart in the reference (whose work-list holds the
        // polynomials themselves); it only bounds the path word and cannot trigger before an
        // interval has been halved ~1100 times, i.e. far past double precision.
        if (n_pending > 250 || n_isol > order || depth >= RF_MAX_DEPTH) {
            *warn |= 4;
            loop_failed = true;
            break;
        }
        if (descend) {
            // Next node = right child: 2^n q((x + 1) / 2).
            poly_rescale_p2_n<N>(tmp1, tmp);
            poly_translate_1_n<N>(tmp2, tmp1);
#pragma unroll
            for (int i = 0; i <= order; ++i) {
                tmp[i] = tmp2[i];
            }
            const double mid = (lb + ub) / 2;
            lb = mid;
            path.set(depth, true);
            ++depth;
            --n_pending; // popped the right child
            continue;
        }
        if (n_pending == 0) {
            break;
        }
        // Pop the deepest pending left sibling: find the deepest right-turn.
        int k = depth - 1;
        while (k >= 0 && !path.get(k)) {
            --k;
        }
        // k >= 0 is guaranteed by n_pending > 0.
        path.set(k, false);
        depth = k + 1;
        --n_pending;
        // Replay from the root along path[0 .. depth).
#pragma unroll
        for (int i = 0; i <= order; ++i) {
            tmp[i] = root_poly[i];
        }
        lb = 0.;
        ub = 1.;
        for (int lvl = 0; lvl < depth; ++lvl) {
            const double mid = (lb + ub) / 2;
            poly_rescale_p2_n<N>(tmp1, tmp);
            if (path.get(lvl)) {
                poly_translate_1_n<N>(tmp2, tmp1);
#pragma unroll
                for (int i = 0; i <= order; ++i) {
                    tmp[i] = tmp2[i];
                }
                lb = mid;
            } else {
#pragma unroll
                for (int i = 0; i <= order; ++i) {
                    tmp[i] = tmp1[i];
                }
                ub = mid;
            }
        }
    }

    if (n_isol > 0 && !loop_failed) {
        for (int k = 0; k < n_isol; ++k) {
            double lo = isol_lb[k], up = isol_ub[k];
            if (finite_d(lo) && finite_d(up) && up > lo) {
                up = nextafter(up, lo);
            }
            double a, b;
            int used;
            const int err = t748::solve<N>(root_poly, order, lo, up, a, b, used);
            if (err != 0 || !(used < 100)) {
                *warn |= 8;
                continue;
            }
            add_root(((a + b) / 2) * rf_int);
        }
    }
    return loop_failed ? -1 - n_roots : n_roots;
}

} // namespace csc
