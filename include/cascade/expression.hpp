// include/cascade/expression.hpp
//
// Symbolic right-hand sides for cascade::sim's kw::dyn.
//
// The reference takes its dynamics as a vector of (variable, heyoka expression) pairs
// (esa/cascade include/cascade/sim.hpp:213-222; built by dynamics::kepler, src/sim_dynamics.cpp:25-42,
// cascade.py/dynamics/_simple_earth.py, benchmark/haumea.cpp:105-139, benchmark/2022_leo_pop.cpp:218-287)
// and lets heyoka JIT-compile a Taylor integrator for them. heyoka and LLVM are not part of this
// build: this header is a small expression system of its own (numbers, the six state variables,
// runtime parameters par[i], time, + - * /, pow with a constant exponent, sqrt, exp, sin, cos,
// symbolic differentiation) whose equations are LOWERED TO A TAPE of Taylor-series operations
// (cascade::tape_t). The tape is what the GPU integrates: the propagation kernel interprets it order
// by order with the classical automatic-differentiation recurrences, one
// object per lane. Kepler, J2 and drag written with the factory functions keep their hand-unrolled
// kernels; everything else (J3, J4, C22/S22, Sun, Moon, SRP, the Haumea ring, user dynamics) runs
// through the tape.
#ifndef CASCADE_EXPRESSION_HPP
#define CASCADE_EXPRESSION_HPP

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <ostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

namespace cascade
{

// ---- the tape -----------------------------------------------------------------------------
// One operation on truncated Taylor series ("slots"). Slots 0..5 are x, y, z, vx, vy, vz, slot 6
// is r = sqrt(x^2 + y^2 + z^2) (src/sim_basic.cpp:461-469), every operation defines one new slot
// (TP_SINCOS two: dst = sin, dst + 1 = cos). Operand a is always a slot. Operand b is a slot
// (b >= 0), the immediate c (b == -1) or runtime parameter -2 - b of the object (b <= -2).
// Layout shared with the C ABI (csc_tape_op, include/cascade_hip.h).
enum tape_code : std::int32_t {
    TP_ADD = 1,  // a + b
    TP_SUB,      // a - b
    TP_RSUB,     // b - a         (b scalar)
    TP_MUL,      // a * b
    TP_DIV,      // a / b
    TP_RDIV,     // b / a         (b scalar)
    TP_SQR,      // a * a
    TP_POW,      // a ^ c         (real exponent c)
    TP_SQRT,     // sqrt(a)
    TP_EXP,      // exp(a)
    TP_SINCOS,   // dst = sin(a), dst + 1 = cos(a)
    TP_TIME,     // the simulation time
    TP_CONST     // the scalar b as a series
};

struct tape_op {
    std::int32_t op, dst, a, b;
    double c;
};

struct tape_t {
    std::vector<tape_op> ops;
    std::uint32_t nslots = 7;
    // Slot whose series is the right-hand side of x, y, z, vx, vy, vz, r.
    std::array<std::int32_t, 7> rhs = {};
    std::uint32_t npars = 0;

    [[nodiscard]] bool empty() const
    {
        return ops.empty();
    }
    friend bool operator==(const tape_t &a, const tape_t &b)
    {
        if (a.nslots != b.nslots || a.rhs != b.rhs || a.npars != b.npars || a.ops.size() != b.ops.size()) {
            return false;
        }
        return a.ops.empty() || std::memcmp(a.ops.data(), b.ops.data(), a.ops.size() * sizeof(tape_op)) == 0;
    }
};

// ---- expressions --------------------------------------------------------------------------
namespace detail
{

enum class ex_kind : int { num, var, par, time, add, sub, mul, div, neg, pow, sqrt, exp, sin, cos };

struct ex_node {
    ex_kind k = ex_kind::num;
    double val = 0;        // num: value; pow: exponent
    std::uint32_t idx = 0; // par: index
    std::string name;      // var: name
    std::shared_ptr<const ex_node> a, b;
};

using ex_ptr = std::shared_ptr<const ex_node>;

inline ex_ptr mk(ex_kind k, ex_ptr a = {}, ex_ptr b = {}, double val = 0, std::uint32_t idx = 0, std::string name = {})
{
    auto n = std::make_shared<ex_node>();
    n->k = k;
    n->a = std::move(a);
    n->b = std::move(b);
    n->val = val;
    n->idx = idx;
    n->name = std::move(name);
    return n;
}

} // namespace detail

class expression
{
    detail::ex_ptr m_p;

public:
    expression() : m_p(detail::mk(detail::ex_kind::num)) {}
    expression(double v) : m_p(detail::mk(detail::ex_kind::num, {}, {}, v)) {} // NOLINT (implicit on purpose)
    explicit expression(std::string name) : m_p(detail::mk(detail::ex_kind::var, {}, {}, 0, 0, std::move(name))) {}
    explicit expression(detail::ex_ptr p) : m_p(std::move(p)) {}

    [[nodiscard]] const detail::ex_node &node() const
    {
        return *m_p;
    }
    [[nodiscard]] const detail::ex_ptr &ptr() const
    {
        return m_p;
    }
    [[nodiscard]] bool is_num() const
    {
        return m_p->k == detail::ex_kind::num;
    }
    [[nodiscard]] bool is_num(double v) const
    {
        return is_num() && m_p->val == v;
    }
    [[nodiscard]] bool is_var() const
    {
        return m_p->k == detail::ex_kind::var;
    }

    expression &operator+=(const expression &o);
    expression &operator-=(const expression &o);
    expression &operator*=(const expression &o);
    expression &operator/=(const expression &o);
};

// Runtime parameter i of the object (heyoka's par[i]) and the simulation time (heyoka's time).
inline expression par(std::uint32_t i)
{
    return expression(detail::mk(detail::ex_kind::par, {}, {}, 0, i));
}
struct par_t {
    expression operator[](std::uint32_t i) const
    {
        return cascade::par(i);
    }
};
inline const expression time_expr = expression(detail::mk(detail::ex_kind::time));

// "x"_var (heyoka::literals).
inline namespace literals
{
inline expression operator""_var(const char *s, std::size_t n)
{
    return expression(std::string(s, n));
}
} // namespace literals

template <typename... Names>
inline std::array<expression, sizeof...(Names)> make_vars(const Names &...names)
{
    return {expression(std::string(names))...};
}

inline expression operator-(const expression &a)
{
    using detail::ex_kind;
    if (a.is_num()) {
        return expression(-a.node().val);
    }
    if (a.node().k == ex_kind::neg) {
        return expression(a.node().a);
    }
    return expression(detail::mk(ex_kind::neg, a.ptr()));
}
inline expression operator+(const expression &a)
{
    return a;
}
inline expression operator+(const expression &a, const expression &b)
{
    using detail::ex_kind;
    if (a.is_num() && b.is_num()) {
        return expression(a.node().val + b.node().val);
    }
    if (a.is_num(0.)) {
        return b;
    }
    if (b.is_num(0.)) {
        return a;
    }
    if (b.node().k == ex_kind::neg) {
        return expression(detail::mk(ex_kind::sub, a.ptr(), b.node().a));
    }
    return expression(detail::mk(ex_kind::add, a.ptr(), b.ptr()));
}
inline expression operator-(const expression &a, const expression &b)
{
    using detail::ex_kind;
    if (a.is_num() && b.is_num()) {
        return expression(a.node().val - b.node().val);
    }
    if (a.is_num(0.)) {
        return -b;
    }
    if (b.is_num(0.)) {
        return a;
    }
    if (b.node().k == ex_kind::neg) {
        return expression(detail::mk(ex_kind::add, a.ptr(), b.node().a));
    }
    return expression(detail::mk(ex_kind::sub, a.ptr(), b.ptr()));
}
inline expression operator*(const expression &a, const expression &b)
{
    using detail::ex_kind;
    if (a.is_num() && b.is_num()) {
        return expression(a.node().val * b.node().val);
    }
    if (a.is_num(0.) || b.is_num(0.)) {
        return expression(0.);
    }
    if (a.is_num(1.)) {
        return b;
    }
    if (b.is_num(1.)) {
        return a;
    }
    if (a.is_num(-1.)) {
        return -b;
    }
    if (b.is_num(-1.)) {
        return -a;
    }
    // Pull unary minus and nested numeric factors outwards: keeps derivatives compact.
    if (a.node().k == ex_kind::neg && b.node().k == ex_kind::neg) {
        return expression(a.node().a) * expression(b.node().a);
    }
    if (a.is_num() && b.node().k == ex_kind::mul && b.node().a->k == ex_kind::num) {
        return expression(a.node().val * b.node().a->val) * expression(b.node().b);
    }
    if (b.is_num()) {
        return b * a; // numbers first
    }
    return expression(detail::mk(ex_kind::mul, a.ptr(), b.ptr()));
}
inline expression operator/(const expression &a, const expression &b)
{
    using detail::ex_kind;
    if (a.is_num() && b.is_num()) {
        return expression(a.node().val / b.node().val);
    }
    if (a.is_num(0.)) {
        return expression(0.);
    }
    if (b.is_num(1.)) {
        return a;
    }
    return expression(detail::mk(ex_kind::div, a.ptr(), b.ptr()));
}
inline expression &expression::operator+=(const expression &o)
{
    return *this = *this + o;
}
inline expression &expression::operator-=(const expression &o)
{
    return *this = *this - o;
}
inline expression &expression::operator*=(const expression &o)
{
    return *this = *this * o;
}
inline expression &expression::operator/=(const expression &o)
{
    return *this = *this / o;
}

inline expression sqrt(const expression &a)
{
    if (a.is_num()) {
        return expression(std::sqrt(a.node().val));
    }
    return expression(detail::mk(detail::ex_kind::sqrt, a.ptr()));
}
// Power with a constant real exponent.
inline expression pow(const expression &a, double e)
{
    if (a.is_num()) {
        return expression(std::pow(a.node().val, e));
    }
    if (e == 0.) {
        return expression(1.);
    }
    if (e == 1.) {
        return a;
    }
    if (e == 0.5) {
        return sqrt(a);
    }
    return expression(detail::mk(detail::ex_kind::pow, a.ptr(), {}, e));
}
inline expression pow(const expression &a, const expression &e)
{
    if (!e.is_num()) {
        throw std::invalid_argument("pow(): only constant exponents are supported");
    }
    return pow(a, e.node().val);
}
inline expression exp(const expression &a)
{
    if (a.is_num()) {
        return expression(std::exp(a.node().val));
    }
    return expression(detail::mk(detail::ex_kind::exp, a.ptr()));
}
inline expression sin(const expression &a)
{
    if (a.is_num()) {
        return expression(std::sin(a.node().val));
    }
    return expression(detail::mk(detail::ex_kind::sin, a.ptr()));
}
inline expression cos(const expression &a)
{
    if (a.is_num()) {
        return expression(std::cos(a.node().val));
    }
    return expression(detail::mk(detail::ex_kind::cos, a.ptr()));
}
// Sum of a list (heyoka's sum()).
inline expression sum(const std::vector<expression> &v)
{
    expression acc(0.);
    for (const auto &e : v) {
        acc = acc + e;
    }
    return acc;
}

namespace detail
{

inline expression diff_impl(const ex_ptr &p, const std::string &v, std::unordered_map<const ex_node *, expression> &memo)
{
    if (auto it = memo.find(p.get()); it != memo.end()) {
        return it->second;
    }
    expression r(0.);
    const expression A = p->a ? expression(p->a) : expression(0.), B = p->b ? expression(p->b) : expression(0.);
    switch (p->k) {
        case ex_kind::num:
        case ex_kind::par:
        case ex_kind::time:
            break;
        case ex_kind::var:
            r = expression(p->name == v ? 1. : 0.);
            break;
        case ex_kind::add:
            r = diff_impl(p->a, v, memo) + diff_impl(p->b, v, memo);
            break;
        case ex_kind::sub:
            r = diff_impl(p->a, v, memo) - diff_impl(p->b, v, memo);
            break;
        case ex_kind::neg:
            r = -diff_impl(p->a, v, memo);
            break;
        case ex_kind::mul:
            r = diff_impl(p->a, v, memo) * B + A * diff_impl(p->b, v, memo);
            break;
        case ex_kind::div: {
            const auto da = diff_impl(p->a, v, memo), db = diff_impl(p->b, v, memo);
            r = da / B - (A * db) / (B * B);
            break;
        }
        case ex_kind::pow:
            r = (expression(p->val) * pow(A, p->val - 1.)) * diff_impl(p->a, v, memo);
            break;
        case ex_kind::sqrt:
            r = diff_impl(p->a, v, memo) / (expression(2.) * expression(p));
            break;
        case ex_kind::exp:
            r = expression(p) * diff_impl(p->a, v, memo);
            break;
        case ex_kind::sin:
            r = cos(A) * diff_impl(p->a, v, memo);
            break;
        case ex_kind::cos:
            r = -(sin(A) * diff_impl(p->a, v, memo));
            break;
    }
    memo.emplace(p.get(), r);
    return r;
}

inline void collect(const ex_ptr &p, std::set<std::string> &vars, std::uint32_t &npars, std::set<const ex_node *> &seen)
{
    if (!p || !seen.insert(p.get()).second) {
        return;
    }
    if (p->k == ex_kind::var) {
        vars.insert(p->name);
    } else if (p->k == ex_kind::par) {
        npars = std::max(npars, p->idx + 1u);
    }
    collect(p->a, vars, npars, seen);
    collect(p->b, vars, npars, seen);
}

inline void print(std::ostream &os, const ex_ptr &p)
{
    switch (p->k) {
        case ex_kind::num:
            os << p->val;
            break;
        case ex_kind::var:
            os << p->name;
            break;
        case ex_kind::par:
            os << "p" << p->idx;
            break;
        case ex_kind::time:
            os << "t";
            break;
        case ex_kind::neg:
            os << "-";
            print(os, p->a);
            break;
        case ex_kind::add:
        case ex_kind::sub:
        case ex_kind::mul:
        case ex_kind::div: {
            const char *s = p->k == ex_kind::add ? " + " : p->k == ex_kind::sub ? " - " : p->k == ex_kind::mul ? " * " : " / ";
            os << "(";
            print(os, p->a);
            os << s;
            print(os, p->b);
            os << ")";
            break;
        }
        case ex_kind::pow:
            print(os, p->a);
            os << "**" << p->val;
            break;
        default: {
            const char *s = p->k == ex_kind::sqrt ? "sqrt" : p->k == ex_kind::exp ? "exp" : p->k == ex_kind::sin ? "sin" : "cos";
            os << s << "(";
            print(os, p->a);
            os << ")";
        }
    }
}

} // namespace detail

// d e / d var (heyoka's diff()).
inline expression diff(const expression &e, const expression &var)
{
    if (!var.is_var()) {
        throw std::invalid_argument("diff(): the second argument must be a variable");
    }
    std::unordered_map<const detail::ex_node *, expression> memo;
    return detail::diff_impl(e.ptr(), var.node().name, memo);
}
inline std::set<std::string> get_variables(const expression &e)
{
    std::set<std::string> vars;
    std::uint32_t np = 0;
    std::set<const detail::ex_node *> seen;
    detail::collect(e.ptr(), vars, np, seen);
    return vars;
}
inline std::uint32_t get_param_size(const expression &e)
{
    std::set<std::string> vars;
    std::uint32_t np = 0;
    std::set<const detail::ex_node *> seen;
    detail::collect(e.ptr(), vars, np, seen);
    return np;
}
inline std::ostream &operator<<(std::ostream &os, const expression &e)
{
    detail::print(os, e.ptr());
    return os;
}

// ---- lowering -----------------------------------------------------------------------------
namespace detail
{

// A lowered operand: slot >= 0, or a scalar (immediate / parameter).
struct operand {
    std::int32_t slot = -1;
    bool imm = false;
    double c = 0;
    std::int32_t par = -1;
    [[nodiscard]] bool scalar() const
    {
        return slot < 0;
    }
    [[nodiscard]] std::int32_t code() const
    {
        return slot >= 0 ? slot : (imm ? -1 : -2 - par);
    }
};

class lowerer
{
    tape_t &m_t;
    std::unordered_map<const ex_node *, operand> m_memo;
    std::map<std::tuple<std::int32_t, std::int32_t, std::int32_t, std::uint64_t>, std::int32_t> m_cse;

    static std::uint64_t bits(double c)
    {
        std::uint64_t u;
        std::memcpy(&u, &c, sizeof u);
        return u;
    }
    std::int32_t emit(std::int32_t op, std::int32_t a, std::int32_t b, double c, int nout = 1)
    {
        const auto key = std::make_tuple(op, a, b, bits(c));
        if (auto it = m_cse.find(key); it != m_cse.end()) {
            return it->second;
        }
        const auto dst = static_cast<std::int32_t>(m_t.nslots);
        m_t.nslots += static_cast<std::uint32_t>(nout);
        m_t.ops.push_back(tape_op{op, dst, a, b, c});
        m_cse.emplace(key, dst);
        return dst;
    }
    std::int32_t as_slot(const operand &o)
    {
        return o.slot >= 0 ? o.slot : emit(TP_CONST, 0, o.code(), o.imm ? o.c : 0.);
    }
    operand slot(std::int32_t s)
    {
        operand o;
        o.slot = s;
        return o;
    }
    // a (slot) op b, or its scalar variants.
    operand binary(std::int32_t op, std::int32_t rop, operand a, operand b, bool commutes)
    {
        if (a.scalar() && b.scalar()) {
            a = slot(as_slot(a));
        }
        if (a.scalar()) {
            // scalar op series
            if (commutes) {
                return slot(emit(op, b.slot, a.code(), a.imm ? a.c : 0.));
            }
            return slot(emit(rop, b.slot, a.code(), a.imm ? a.c : 0.));
        }
        if (b.scalar()) {
            return slot(emit(op, a.slot, b.code(), b.imm ? b.c : 0.));
        }
        if (commutes && b.slot < a.slot) {
            std::swap(a, b);
        }
        if (op == TP_MUL && a.slot == b.slot) {
            return slot(emit(TP_SQR, a.slot, 0, 0.));
        }
        return slot(emit(op, a.slot, b.slot, 0.));
    }
    // Small positive integer powers by multiplications (a general power divides by a[0], which
    // may be zero: z^4 on the equator).
    std::int32_t ipow(std::int32_t base, unsigned n)
    {
        if (n == 1u) {
            return base;
        }
        const std::int32_t h = ipow(base, n / 2u);
        std::int32_t r = emit(TP_SQR, h, 0, 0.);
        if (n % 2u) {
            const std::int32_t lo = std::min(r, base), hi = std::max(r, base);
            r = emit(TP_MUL, lo, hi, 0.);
        }
        return r;
    }

public:
    explicit lowerer(tape_t &t) : m_t(t) {}

    operand lower(const ex_ptr &p)
    {
        if (auto it = m_memo.find(p.get()); it != m_memo.end()) {
            return it->second;
        }
        static const char *names[6] = {"x", "y", "z", "vx", "vy", "vz"};
        operand r;
        switch (p->k) {
            case ex_kind::num:
                r.imm = true;
                r.c = p->val;
                break;
            case ex_kind::par:
                r.par = static_cast<std::int32_t>(p->idx);
                m_t.npars = std::max(m_t.npars, p->idx + 1u);
                break;
            case ex_kind::var: {
                int s = -1;
                for (int i = 0; i < 6; ++i) {
                    if (p->name == names[i]) {
                        s = i;
                    }
                }
                if (s < 0) {
                    throw std::invalid_argument("Unknown variable \"" + p->name + "\" in the dynamics");
                }
                r = slot(s);
                break;
            }
            case ex_kind::time:
                r = slot(emit(TP_TIME, 0, 0, 0.));
                break;
            case ex_kind::add: {
                const operand la = lower(p->a), lb = lower(p->b); // left operand first
                r = binary(TP_ADD, TP_ADD, la, lb, true);
                break;
            }
            case ex_kind::sub: {
                const operand la = lower(p->a), lb = lower(p->b); // left operand first
                r = binary(TP_SUB, TP_RSUB, la, lb, false);
                break;
            }
            case ex_kind::mul: {
                const operand la = lower(p->a), lb = lower(p->b); // left operand first
                r = binary(TP_MUL, TP_MUL, la, lb, true);
                break;
            }
            case ex_kind::div: {
                const operand la = lower(p->a), lb = lower(p->b); // left operand first
                r = binary(TP_DIV, TP_RDIV, la, lb, false);
                break;
            }
            case ex_kind::neg: {
                const operand a = lower(p->a);
                r = slot(emit(TP_MUL, as_slot(a), -1, -1.));
                break;
            }
            case ex_kind::pow: {
                const std::int32_t a = as_slot(lower(p->a));
                const double e = p->val;
                if (e == std::floor(e) && e >= 2. && e <= 64.) {
                    r = slot(ipow(a, static_cast<unsigned>(e)));
                } else {
                    r = slot(emit(TP_POW, a, 0, e));
                }
                break;
            }
            case ex_kind::sqrt:
                r = slot(emit(TP_SQRT, as_slot(lower(p->a)), 0, 0.));
                break;
            case ex_kind::exp:
                r = slot(emit(TP_EXP, as_slot(lower(p->a)), 0, 0.));
                break;
            case ex_kind::sin:
            case ex_kind::cos: {
                const std::int32_t s = emit(TP_SINCOS, as_slot(lower(p->a)), 0, 0., 2);
                r = slot(p->k == ex_kind::sin ? s : s + 1);
                break;
            }
        }
        m_memo.emplace(p.get(), r);
        return r;
    }
    std::int32_t lower_to_slot(const ex_ptr &p)
    {
        return as_slot(lower(p));
    }
    // r' = (x vx + y vy + z vz) / r (src/sim_basic.cpp:469).
    std::int32_t r_equation()
    {
        const std::int32_t a = emit(TP_MUL, 0, 3, 0.), b = emit(TP_MUL, 1, 4, 0.), c = emit(TP_MUL, 2, 5, 0.);
        const std::int32_t s = emit(TP_ADD, emit(TP_ADD, a, b, 0.), c, 0.);
        return emit(TP_DIV, s, 6, 0.);
    }
};

} // namespace detail

// Lowers the six equations (LHS must be the variables x, y, z, vx, vy, vz in this order) to a
// tape, appending the equation of r. Validation texts: src/sim_basic.cpp:411-451.
inline tape_t lower_dynamics(const std::vector<std::pair<expression, expression>> &dyn)
{
    static const char *names[6] = {"x", "y", "z", "vx", "vy", "vz"};
    if (dyn.size() != 6u) {
        throw std::invalid_argument("6 dynamical equations are expected, but " + std::to_string(dyn.size())
                                    + " were provided instead");
    }
    tape_t t;
    detail::lowerer L(t);
    for (unsigned i = 0; i < 6u; ++i) {
        const auto &[var, eq] = dyn[i];
        if (!var.is_var() || var.node().name != names[i]) {
            std::ostringstream oss;
            oss << "The LHS of the dynamics at index " << i << " must be a variable named \"" << names[i]
                << "\", but instead it is the expression \"" << var << "\"";
            throw std::invalid_argument(oss.str());
        }
        std::vector<std::string> bad;
        for (const auto &v : get_variables(eq)) {
            bool ok = false;
            for (const auto *n : names) {
                ok = ok || v == n;
            }
            if (!ok) {
                bad.push_back(v);
            }
        }
        if (!bad.empty()) {
            std::ostringstream oss;
            oss << "The RHS of the differential equation for the variable \"" << names[i]
                << "\" contains the invalid variables [";
            for (std::size_t k = 0; k < bad.size(); ++k) {
                oss << (k ? ", " : "") << "\"" << bad[k] << "\"";
            }
            oss << "] (the allowed variables are [\"x\", \"y\", \"z\", \"vx\", \"vy\", \"vz\"])";
            throw std::invalid_argument(oss.str());
        }
        t.npars = std::max(t.npars, get_param_size(eq));
        t.rhs[i] = L.lower_to_slot(eq.ptr());
    }
    t.rhs[6] = L.r_equation();
    return t;
}

} // namespace cascade

#endif
