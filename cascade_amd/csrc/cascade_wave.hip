// cascade_amd/csrc/cascade_wave.hip
//
// Translation unit of k_propagate_wave (one object over 8 to 64 lanes, series in LDS): the pass
// that fuses a dynamics tape into sum-of-products operations, the scheduler that packs them into
// rows of independent operations, a host evaluator of the result (tests), and the launcher.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "wave_api.hpp"

namespace csc
{

namespace
{

// ---- fusion ------------------------------------------------------------------------------
//
// The value of a tape slot as an expression over MATERIALISED slots (series that exist in LDS):
//     [par *] sum_t c_t A_t [B_t]
// a linear term when B < 0, a product of two series otherwise; the constant series `one`
// (1, 0, 0, ...) carries additive constants. Sums, differences, scalings and products of the tape
// only build such expressions; an operation is emitted (and its slot materialised) when somebody
// needs the series itself: a function argument, a factor of a product that is not a plain scaled
// series, a right-hand side, several users of something that is not cheap to repeat.
struct term {
    double c;
    std::int32_t A, B;
};
struct expan {
    std::int32_t par = -1; // runtime parameter that multiplies (2 p) or divides (2 p + 1) the sum
    std::vector<term> t;
};
struct fop {
    std::int32_t dst = 0, kind = WK_FUSED;
    expan e;
    std::int32_t a = 0; // operand of a function, old numbering
    double c = 0.;
    std::int32_t par = -1; // (WK_RDIV: the numerator is this runtime parameter)
};

struct fuser {
    const tape_op *tape;
    std::uint32_t nops;
    std::int32_t one;
    std::vector<expan> E;
    std::vector<char> defined, is_rhs;
    std::vector<std::vector<std::pair<std::uint32_t, int>>> uses; // (operation, 0: as a, 1: as b)
    std::vector<fop> out;
    bool ok = true;

    static expan mat(std::int32_t s)
    {
        expan e;
        e.t.push_back({1., s, -1});
        return e;
    }
    [[nodiscard]] bool is_mat(std::int32_t s) const
    {
        const expan &e = E[static_cast<std::size_t>(s)];
        return e.par < 0 && e.t.size() == 1u && e.t[0].c == 1. && e.t[0].A == s && e.t[0].B < 0;
    }
    static std::uint32_t nk(const expan &e)
    {
        std::uint32_t k = 0;
        for (const term &t : e.t) {
            k += t.B >= 0 ? 1u : 0u;
        }
        return k;
    }
    static std::uint32_t nm(const expan &e)
    {
        return static_cast<std::uint32_t>(e.t.size()) - nk(e);
    }
    static bool single_lin(const expan &e)
    {
        return e.par < 0 && e.t.size() == 1u && e.t[0].B < 0;
    }
    static bool too_big(const expan &e)
    {
        return nk(e) > WAVE_CONV_MAX || nm(e) > WAVE_LIN_MAX;
    }
    std::int32_t new_tmp()
    {
        E.emplace_back();
        defined.push_back(1);
        is_rhs.push_back(0);
        uses.emplace_back();
        return static_cast<std::int32_t>(E.size() - 1u);
    }
    void materialise(std::int32_t s)
    {
        if (is_mat(s)) {
            return;
        }
        fop f;
        f.dst = s;
        f.kind = WK_FUSED;
        f.e = E[static_cast<std::size_t>(s)];
        out.push_back(f);
        E[static_cast<std::size_t>(s)] = mat(s);
    }
    void emit(std::int32_t dst, std::int32_t kind, const expan &e, std::int32_t a, double c = 0., std::int32_t par = -1)
    {
        fop f;
        f.dst = dst;
        f.kind = kind;
        f.e = e;
        f.a = a;
        f.c = c;
        f.par = par;
        out.push_back(f);
        E[static_cast<std::size_t>(dst)] = mat(dst);
    }
    static expan scaled(const expan &e, double c)
    {
        expan r = e;
        for (term &t : r.t) {
            t.c *= c;
        }
        return r;
    }
    // ea + sign eb (same parameter factor), like terms merged.
    static expan added(const expan &ea, const expan &eb, double sign)
    {
        expan r;
        r.par = ea.par;
        auto put = [&r](double c, std::int32_t A, std::int32_t B) {
            if (B >= 0 && B < A) {
                std::swap(A, B);
            }
            for (term &t : r.t) {
                if (t.A == A && t.B == B) {
                    t.c += c;
                    return;
                }
            }
            r.t.push_back({c, A, B});
        };
        for (const term &t : ea.t) {
            put(t.c, t.A, t.B);
        }
        for (const term &t : eb.t) {
            put(sign * t.c, t.A, t.B);
        }
        r.t.erase(std::remove_if(r.t.begin(), r.t.end(), [](const term &t) { return t.c == 0.; }), r.t.end());
        return r;
    }
    // The constant b of an operation (immediate or runtime parameter) as an expression.
    expan scalar(const tape_op &o)
    {
        expan e;
        if (o.b == -1) {
            e.t.push_back({o.c, one, -1});
            return e;
        }
        // A runtime parameter added to something: a series of its own, so that sums need not mix factors.
        const std::int32_t tmp = new_tmp();
        expan p;
        p.par = 2 * (-2 - o.b);
        p.t.push_back({1., one, -1});
        E[static_cast<std::size_t>(tmp)] = p;
        materialise(tmp);
        return mat(tmp);
    }
    [[nodiscard]] bool need_mat(std::int32_t dst) const
    {
        const expan &e = E[static_cast<std::size_t>(dst)];
        const auto &us = uses[static_cast<std::size_t>(dst)];
        if (is_rhs[static_cast<std::size_t>(dst)] || us.empty()) {
            return true;
        }
        const bool triv = single_lin(e);
        for (const auto &[k, role] : us) {
            const tape_op &u = tape[k];
            switch (u.op) {
                case TP_POW:
                case TP_SQRT:
                case TP_EXP:
                case TP_SINCOS:
                case TP_RDIV:
                    return true;
                case TP_DIV:
                    if (role == 1) {
                        return true;
                    }
                    break; // (a numerator may be a whole expression; csc decides when it gets there)
                case TP_MUL:
                    if (u.b >= 0 && !triv) {
                        return true;
                    }
                    break;
                case TP_SQR:
                    if (!triv) {
                        return true;
                    }
                    break;
                default:
                    break;
            }
        }
        return us.size() > 1u && (nk(e) > 0u || nm(e) > 3u || e.par >= 0);
    }
    void settle(std::int32_t dst)
    {
        if (need_mat(dst)) {
            materialise(dst);
        }
    }

    void run(std::uint32_t nslots, const std::int32_t rhs[7])
    {
        one = static_cast<std::int32_t>(nslots);
        E.assign(nslots + 1u, expan{});
        defined.assign(nslots + 1u, 0);
        is_rhs.assign(nslots + 1u, 0);
        uses.assign(nslots + 1u, {});
        for (int s = 0; s < 7; ++s) {
            E[s] = mat(s);
            defined[s] = 1;
            is_rhs[static_cast<std::size_t>(rhs[s])] = 1;
        }
        E[static_cast<std::size_t>(one)] = mat(one);
        defined[static_cast<std::size_t>(one)] = 1;
        for (std::uint32_t k = 0; k < nops; ++k) {
            const tape_op &o = tape[k];
            const bool two = o.op == TP_ADD || o.op == TP_SUB || o.op == TP_MUL || o.op == TP_DIV;
            if (o.op != TP_TIME && o.op != TP_CONST) {
                uses[static_cast<std::size_t>(o.a)].push_back({k, 0});
                if (two && o.b >= 0) {
                    uses[static_cast<std::size_t>(o.b)].push_back({k, 1});
                }
            }
        }
        for (std::uint32_t k = 0; k < nops && ok; ++k) {
            const tape_op o = tape[k];
            const std::size_t d = static_cast<std::size_t>(o.dst);
            const bool reads_a = o.op != TP_TIME && o.op != TP_CONST;
            const bool reads_b = (o.op == TP_ADD || o.op == TP_SUB || o.op == TP_MUL || o.op == TP_DIV) && o.b >= 0;
            if (defined[d] || (o.op == TP_SINCOS && defined[d + 1u]) || (reads_a && !defined[static_cast<std::size_t>(o.a)])
                || (reads_b && !defined[static_cast<std::size_t>(o.b)])) {
                ok = false; // (not in single-assignment form: the interpreter takes it)
                return;
            }
            defined[d] = 1;
            const std::size_t a = static_cast<std::size_t>(o.a);
            switch (o.op) {
                case TP_ADD:
                case TP_SUB:
                case TP_RSUB: {
                    const double sign = o.op == TP_ADD ? 1. : -1.;
                    if (o.op == TP_RSUB || o.b < 0) {
                        // series +- constant, constant - series
                        if (E[a].par >= 0) {
                            materialise(o.a);
                        }
                        const expan cst = scalar(o);
                        E[d] = o.op == TP_RSUB ? added(cst, E[a], -1.) : added(E[a], cst, sign);
                        if (too_big(E[d])) {
                            materialise(o.a);
                            E[d] = o.op == TP_RSUB ? added(cst, E[a], -1.) : added(E[a], cst, sign);
                        }
                    } else {
                        const std::size_t b = static_cast<std::size_t>(o.b);
                        if (E[a].par != E[b].par) {
                            if (E[a].par >= 0) {
                                materialise(o.a);
                            }
                            if (E[b].par >= 0) {
                                materialise(o.b);
                            }
                        }
                        E[d] = added(E[a], E[b], sign);
                        while (too_big(E[d])) {
                            // The larger operand becomes a series of its own.
                            const bool am = is_mat(o.a), bm = is_mat(o.b);
                            if (am && bm) {
                                break;
                            }
                            const bool pick_a = !am && (bm || E[a].t.size() >= E[b].t.size());
                            materialise(pick_a ? o.a : o.b);
                            E[d] = added(E[a], E[b], sign);
                        }
                    }
                    settle(o.dst);
                    break;
                }
                case TP_MUL:
                case TP_SQR:
                    if (o.op == TP_MUL && o.b < 0) {
                        if (o.b == -1) {
                            E[d] = scaled(E[a], o.c);
                        } else {
                            if (E[a].par >= 0) {
                                materialise(o.a);
                            }
                            E[d] = E[a];
                            E[d].par = 2 * (-2 - o.b);
                        }
                    } else {
                        const std::int32_t bb = o.op == TP_SQR ? o.a : o.b;
                        if (!single_lin(E[a])) {
                            materialise(o.a);
                        }
                        if (!single_lin(E[static_cast<std::size_t>(bb)])) {
                            materialise(bb);
                        }
                        const term ta = E[a].t[0], tb = E[static_cast<std::size_t>(bb)].t[0];
                        expan e;
                        // (a factor that is a multiple of the constant series makes it a linear term)
                        if (ta.A == one) {
                            e.t.push_back({ta.c * tb.c, tb.A, -1});
                        } else if (tb.A == one) {
                            e.t.push_back({ta.c * tb.c, ta.A, -1});
                        } else {
                            e.t.push_back({ta.c * tb.c, std::min(ta.A, tb.A), std::max(ta.A, tb.A)});
                        }
                        E[d] = e;
                    }
                    settle(o.dst);
                    break;
                case TP_DIV:
                    if (o.b == -1) {
                        E[d] = scaled(E[a], 1. / o.c);
                        settle(o.dst);
                    } else if (o.b <= -2) {
                        if (E[a].par >= 0) {
                            materialise(o.a);
                        }
                        E[d] = E[a];
                        E[d].par = 2 * (-2 - o.b) + 1;
                        settle(o.dst);
                    } else {
                        materialise(o.b);
                        if (nk(E[a]) + 1u > WAVE_CONV_MAX || E[a].par >= 0) {
                            materialise(o.a);
                        }
                        emit(o.dst, WK_DIV, E[a], o.b);
                    }
                    break;
                case TP_RDIV:
                    materialise(o.a);
                    emit(o.dst, WK_RDIV, expan{}, o.a, o.c, o.b <= -2 ? -2 - o.b : -1);
                    break;
                case TP_SQRT:
                    materialise(o.a);
                    emit(o.dst, WK_SQRT, E[a], o.a);
                    break;
                case TP_POW:
                    materialise(o.a);
                    emit(o.dst, WK_POW, expan{}, o.a, o.c);
                    break;
                case TP_EXP:
                    materialise(o.a);
                    emit(o.dst, WK_EXP, expan{}, o.a);
                    break;
                case TP_SINCOS:
                    materialise(o.a);
                    defined[d + 1u] = 1;
                    emit(o.dst, WK_SIN, expan{}, o.a);
                    emit(o.dst + 1, WK_COS, expan{}, o.a);
                    break;
                case TP_TIME:
                    emit(o.dst, WK_TIME, expan{}, 0);
                    break;
                default: { // TP_CONST
                    expan e;
                    if (o.b == -1) {
                        e.t.push_back({o.c, one, -1});
                    } else {
                        e.par = 2 * (-2 - o.b);
                        e.t.push_back({1., one, -1});
                    }
                    E[d] = e;
                    settle(o.dst);
                    break;
                }
            }
        }
    }
};

// One fused operation in the kernel's terms (new slot numbers).
struct h_conv {
    std::uint32_t A = 0, B = 0, jlo = 0, hoff = 0; // sum over j = jlo .. n - hoff of (n wa + j wb + wc) A[j] B[n - j]
    double wa = 0., wb = 0., wc = 0.;
    bool none = true;
};
struct h_lin {
    std::uint32_t slot = 0;
    double coef = 0.;
};
struct h_hdr {
    std::uint32_t dst = 0, flags = 0;
};
struct sched_item {
    h_hdr hdr{};
    h_lin lin[WAVE_LIN_MAX]{};
    h_conv conv[WAVE_CONV_MAX]{};
    std::uint32_t nk = 0;
    wave_op0 z{};
    std::vector<std::int32_t> reads; // slots whose coefficient of the SAME order it needs
    std::int32_t dst = 0;
};

// The kernel's units: byte offsets into an object's series (L coefficients each), the lower
// limit of a sum folded into the offsets and the constant weight.
wave_hdr enc_hdr(const h_hdr &h, std::uint32_t nk_row, bool weighted_row, std::uint32_t L)
{
    wave_hdr o{};
    o.dst = h.dst * L * 8u;
    o.flags = h.flags;
    o.nk = nk_row;
    o.weighted = weighted_row ? 1u : 0u;
    return o;
}
wave_lin enc_lin(const h_lin &l, std::uint32_t L)
{
    wave_lin o{};
    o.off = l.slot * L * 8u;
    o.coef = l.coef;
    return o;
}
wave_conv enc_conv(const h_conv &c, std::uint32_t L)
{
    wave_conv o{};
    if (c.none) {
        o.a = 0xfffu << 20; // (an empty range whatever the order)
        return o;
    }
    o.a = (c.A * L * 8u + 8u * c.jlo) | ((c.jlo + c.hoff) << 20);
    o.b = static_cast<std::int32_t>(c.B * L * 8u) - 8 * static_cast<std::int32_t>(c.jlo);
    o.wa = c.wa;
    o.wb = c.wb;
    o.wc = c.wc + static_cast<double>(c.jlo) * c.wb;
    return o;
}

void put_unit(std::vector<std::uint32_t> &blk, std::size_t unit, const void *src)
{
    std::memcpy(blk.data() + unit * 4u, src, 16);
}

// Rows by list scheduling: an operation is ready when everything it reads at the same order is
// in an earlier row; among the ready ones those with the longest chain behind them go first.
std::vector<std::vector<std::uint32_t>> schedule(const std::vector<sched_item> &items, std::uint32_t nslots, int ng, std::uint32_t &depth,
                                                 std::vector<std::uint32_t> &levels)
{
    const std::size_t n = items.size();
    std::vector<std::int32_t> prod(nslots + 1u, -1);
    for (std::size_t i = 0; i < n; ++i) {
        prod[static_cast<std::size_t>(items[i].dst)] = static_cast<std::int32_t>(i);
    }
    std::vector<std::vector<std::uint32_t>> deps(n), succ(n);
    for (std::size_t i = 0; i < n; ++i) {
        for (const std::int32_t s : items[i].reads) {
            const std::int32_t p = prod[static_cast<std::size_t>(s)];
            if (p >= 0 && static_cast<std::size_t>(p) != i) {
                deps[i].push_back(static_cast<std::uint32_t>(p));
                succ[static_cast<std::size_t>(p)].push_back(static_cast<std::uint32_t>(i));
            }
        }
    }
    // (producers come before their users in `items`: one backward sweep gives the chain lengths)
    std::vector<std::uint32_t> tail(n, 1u), lvl(n, 1u);
    for (std::size_t i = n; i-- > 0;) {
        for (const std::uint32_t j : succ[i]) {
            tail[i] = std::max(tail[i], tail[j] + 1u);
        }
    }
    depth = 0;
    for (std::size_t i = 0; i < n; ++i) {
        for (const std::uint32_t j : deps[i]) {
            lvl[i] = std::max(lvl[i], lvl[j] + 1u);
        }
        depth = std::max(depth, lvl[i]);
    }
    levels = lvl;
    std::vector<char> wt(n, 0);
    for (std::size_t i = 0; i < n; ++i) {
        for (std::uint32_t k = 0; k < items[i].nk; ++k) {
            wt[i] = wt[i] || items[i].conv[k].wa != 0. || items[i].conv[k].wb != 0.;
        }
    }
    std::vector<std::int32_t> row_of(n, -1);
    std::vector<std::vector<std::uint32_t>> rows;
    std::size_t left = n;
    while (left != 0u) {
        const std::int32_t r = static_cast<std::int32_t>(rows.size());
        std::vector<std::uint32_t> ready;
        for (std::size_t i = 0; i < n; ++i) {
            if (row_of[i] >= 0) {
                continue;
            }
            bool ok = true;
            for (const std::uint32_t j : deps[i]) {
                ok = ok && row_of[j] >= 0 && row_of[j] < r;
            }
            if (ok) {
                ready.push_back(static_cast<std::uint32_t>(i));
            }
        }
        std::stable_sort(ready.begin(), ready.end(), [&](std::uint32_t x, std::uint32_t y) {
            // (weighted sums together: a row without them reads half the schedule and spends a third of the instructions per term)
            if (tail[x] != tail[y]) {
                return tail[x] > tail[y];
            }
            if (wt[x] != wt[y]) {
                return wt[x] > wt[y];
            }
            return items[x].nk > items[y].nk;
        });
        if (ready.size() > static_cast<std::size_t>(ng)) {
            ready.resize(static_cast<std::size_t>(ng));
        }
        for (const std::uint32_t i : ready) {
            row_of[i] = r;
        }
        left -= ready.size();
        rows.push_back(std::move(ready));
    }
    return rows;
}

} // namespace

bool wave_build_plan(const tape_op *tape, std::uint32_t nops, std::uint32_t nslots, const std::int32_t rhs[7], int ng_force,
                     std::uint32_t order, wave_plan &out)
{
    out = wave_plan{};
    out.tape_nslots = nslots;
    for (int c = 0; c < 7; ++c) {
        out.tape_rhs[c] = rhs[c];
    }
    fuser fz{};
    fz.tape = tape;
    fz.nops = nops;
    fz.run(nslots, rhs);
    if (!fz.ok) {
        return false;
    }
    // New slot numbers: the state, the emitted operations in order, the constant series.
    std::vector<std::int32_t> remap(fz.E.size(), -1);
    for (int s = 0; s < 7; ++s) {
        remap[s] = s;
    }
    const std::size_t nf = fz.out.size();
    for (std::size_t i = 0; i < nf; ++i) {
        remap[static_cast<std::size_t>(fz.out[i].dst)] = static_cast<std::int32_t>(7u + i);
    }
    const std::uint32_t one = static_cast<std::uint32_t>(7u + nf), nslots_new = one + 1u, spare = nslots_new;
    remap[static_cast<std::size_t>(fz.one)] = static_cast<std::int32_t>(one);
    if (static_cast<std::uint64_t>(nslots_new + 2u) * (order + 1u) * 8u >= (1u << 20)) {
        return false; // (byte offsets of the series have 20 bits)
    }
    auto slot = [&](std::int32_t s) { return static_cast<std::uint32_t>(remap[static_cast<std::size_t>(s)]); };
    for (int c = 0; c < 7; ++c) {
        if (remap[static_cast<std::size_t>(rhs[c])] < 0) {
            return false; // (a right-hand side nobody defined)
        }
        out.rhs[c] = remap[static_cast<std::size_t>(rhs[c])];
    }

    std::vector<sched_item> items(nf);
    for (std::size_t i = 0; i < nf; ++i) {
        const fop &f = fz.out[i];
        sched_item &it = items[i];
        it.dst = static_cast<std::int32_t>(7u + i);
        it.hdr.dst = static_cast<std::uint32_t>(it.dst);
        std::uint32_t m = 0;
        auto add_conv = [&](std::uint32_t A, std::uint32_t B, std::uint32_t jlo, std::uint32_t hoff, double wa, double wb, double wc) {
            h_conv &c = it.conv[it.nk++];
            c.none = false;
            c.A = A;
            c.B = B;
            c.jlo = jlo;
            c.hoff = hoff;
            c.wa = wa;
            c.wb = wb;
            c.wc = wc;
            // Same-order reads: A[n] when the sum reaches j = n, B[n] when it starts at j = 0.
            if (hoff == 0u && A != it.hdr.dst) {
                it.reads.push_back(static_cast<std::int32_t>(A));
            }
            if (jlo == 0u && B != it.hdr.dst) {
                it.reads.push_back(static_cast<std::int32_t>(B));
            }
        };
        for (const term &t : f.e.t) {
            if (t.B < 0) {
                it.lin[m].slot = slot(t.A);
                it.lin[m].coef = t.c;
                ++m;
                if (slot(t.A) != one) {
                    it.reads.push_back(static_cast<std::int32_t>(slot(t.A)));
                }
            } else {
                add_conv(slot(t.A), slot(t.B), 0, 0, 0., 0., t.c);
            }
        }
        const std::uint32_t a = f.kind == WK_FUSED || f.kind == WK_TIME ? 0u : slot(f.a);
        it.z.kind = f.kind;
        it.z.a = static_cast<std::int32_t>(a);
        it.z.par = -1;
        it.z.c = f.c;
        switch (f.kind) {
            case WK_FUSED:
                it.z.par = f.e.par >= 0 ? f.e.par / 2 : -1;
                it.z.pad = f.e.par >= 0 ? f.e.par % 2 : 0; // (1: the parameter divides)
                break;
            case WK_DIV: // c[n] = (num[n] - sum_{j >= 1} b[j] c[n - j]) / b0
                add_conv(a, it.hdr.dst, 1, 0, 0., 0., -1.);
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            case WK_RDIV: // c[n] = -sum_{j >= 1} a[j] c[n - j] / a0
                add_conv(a, it.hdr.dst, 1, 0, 0., 0., -1.);
                it.z.par = f.par;
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            case WK_SQRT: // c[n] = (a[n] - sum_{1 <= j < n} c[j] c[n - j]) / (2 c0)
                add_conv(it.hdr.dst, it.hdr.dst, 1, 1, 0., 0., -1.);
                break;
            case WK_POW: // c[n] = sum_{j < n} (n alpha - j (alpha + 1)) c[j] a[n - j] / (n a0)
                add_conv(it.hdr.dst, a, 0, 1, f.c, -(f.c + 1.), 0.);
                it.hdr.flags |= WF_DIVN;
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            case WK_EXP: // c[n] = sum_{j >= 1} j a[j] c[n - j] / n
                add_conv(a, it.hdr.dst, 1, 0, 0., 1., 0.);
                it.hdr.flags |= WF_DIVN;
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            case WK_SIN: // s[n] = sum j a[j] c[n - j] / n, c[n] = -sum j a[j] s[n - j] / n
                add_conv(a, it.hdr.dst + 1u, 1, 0, 0., 1., 0.);
                it.hdr.flags |= WF_DIVN;
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            case WK_COS:
                add_conv(a, it.hdr.dst - 1u, 1, 0, 0., -1., 0.);
                it.hdr.flags |= WF_DIVN;
                it.reads.push_back(static_cast<std::int32_t>(a));
                break;
            default: // WK_TIME
                it.hdr.flags |= WF_TIME;
                break;
        }
    }
    out.fused_ops = static_cast<std::uint32_t>(nf);

    // The layout: groups per object and waves per workgroup. Throughput goes with the objects of
    // a wave over the rows of a substep, times the waves a compute unit holds (8 are as many as
    // the kernel's registers allow).
    const std::uint32_t L = order + 1u;
    double best_score = -1.;
    wave_plan best;
    int env_ng = 0;
    if (const char *e = std::getenv("CASCADE_WAVE_NG")) {
        env_ng = std::atoi(e);
    }
    if (ng_force == 0 && (env_ng == 1 || env_ng == 2 || env_ng == 4 || env_ng == 8)) {
        ng_force = env_ng;
    }
    int env_lpg = 0;
    if (const char *e = std::getenv("CASCADE_WAVE_LPG")) {
        env_lpg = std::atoi(e);
    }
    for (const int ng : {8, 4, 2, 1}) {
        if (ng_force != 0 && ng != ng_force) {
            continue;
        }
        std::uint32_t depth = 0;
        std::vector<std::uint32_t> levels;
        const auto rows = schedule(items, nslots_new, ng, depth, levels);
        const std::uint32_t kp = WAVE_CONV_MAX; // (the kernel requests every unit of a row, used or not)
        wave_plan p = out;
        p.ng = static_cast<std::uint32_t>(ng);
        p.nrows = static_cast<std::uint32_t>(rows.size());
        p.nslots = nslots_new;
        p.one = one;
        p.depth = depth;
        p.row_units = 1u + WAVE_LIN_MAX + 2u * kp;
        // (zeros in front, the series and a spare one, one factor per place of a row, the event polynomial)
        p.obj_doubles = (WAVE_OBJ_PAD + (nslots_new + 1u) * L + p.nrows * p.ng + L + 2u) & ~1u;
        if (ng == 1) {
            // Eight objects per wave read the same offsets of their own series in one instruction, 16
            // banks each: with the objects 16 banks apart (mod 64) the four objects of a half-wave
            // cover the 64 banks exactly. (2532 doubles per object put them 36 banks apart: every
            // access a two- to three-way conflict, LDS busy half of the kernel's time.)
            while (p.obj_doubles % 32u != 8u) {
                p.obj_doubles += 2u;
            }
        }
        p.blk.assign(static_cast<std::size_t>(p.nrows) * p.row_units * p.ng * 4u, 0u);
        p.ops0.assign(static_cast<std::size_t>(p.nrows) * p.ng, wave_op0{});
        p.place_level.assign(static_cast<std::size_t>(p.nrows) * p.ng, 0u);
        for (std::uint32_t r = 0; r < p.nrows; ++r) {
            std::uint32_t rk = 0;
            bool wrow = false;
            for (const std::uint32_t i : rows[r]) {
                rk = std::max(rk, items[i].nk);
                for (std::uint32_t k = 0; k < items[i].nk; ++k) {
                    wrow = wrow || items[i].conv[k].wa != 0. || items[i].conv[k].wb != 0.;
                }
            }
            for (std::uint32_t g = 0; g < p.ng; ++g) {
                sched_item empty;
                empty.hdr.dst = spare; // (an empty place: the spare slot behind the series takes what it computes)
                empty.z.par = -1;      // (and it has no runtime parameter: the parameter array may be null)
                const sched_item &it = g < rows[r].size() ? items[rows[r][g]] : empty;
                const wave_hdr h = enc_hdr(it.hdr, rk, wrow, L);
                const std::size_t u0 = static_cast<std::size_t>(r) * p.row_units * p.ng + g;
                put_unit(p.blk, u0, &h);
                for (std::uint32_t m = 0; m < WAVE_LIN_MAX; ++m) {
                    const wave_lin l = enc_lin(it.lin[m], L);
                    put_unit(p.blk, u0 + static_cast<std::size_t>(1u + m) * p.ng, &l);
                }
                for (std::uint32_t k = 0; k < kp; ++k) {
                    const wave_conv c = enc_conv(k < WAVE_CONV_MAX ? it.conv[k] : h_conv{}, L);
                    put_unit(p.blk, u0 + static_cast<std::size_t>(5u + 2u * k) * p.ng, &c);
                    put_unit(p.blk, u0 + static_cast<std::size_t>(6u + 2u * k) * p.ng, reinterpret_cast<const char *>(&c) + 16);
                }
                p.ops0[static_cast<std::size_t>(r) * p.ng + g] = it.z;
                p.place_level[static_cast<std::size_t>(r) * p.ng + g] = g < rows[r].size() ? levels[rows[r][g]] : 0u;
            }
        }
        // Lanes per operation: 4 put twice the objects into a wave for the same rows (an object
        // needs 8 lanes at least: one per state variable).
        for (const std::uint32_t lpg : {4u, 8u}) {
            if (lpg * p.ng < 8u || (env_lpg != 0 && static_cast<std::uint32_t>(env_lpg) != lpg)) {
                continue;
            }
            p.lpg = lpg;
            // Waves per workgroup: what puts the most waves on a compute unit (fewer when equal).
            unsigned best_w = 0;
            for (std::uint32_t w = 1; w <= WAVE_WPB_MAX; ++w) {
                p.wpb = w;
                if (p.lds_bytes() <= WAVE_LDS_MAX && p.waves_per_cu() > best_w) {
                    best_w = p.waves_per_cu();
                }
            }
            if (best_w == 0u) {
                continue;
            }
            for (std::uint32_t w = 1; w <= WAVE_WPB_MAX; ++w) {
                p.wpb = w;
                if (p.lds_bytes() <= WAVE_LDS_MAX && p.waves_per_cu() == best_w) {
                    break;
                }
            }
            // Throughput goes with the objects of a wave over the rows of a substep, times the
            // waves a compute unit holds (8 are as many as the kernel's registers allow); a row
            // costs more with 4 lanes per operation (more trips through the sums).
            const double opw = static_cast<double>(64u / (lpg * p.ng));
            const double score = opw / (static_cast<double>(p.nrows) * (lpg == 4u ? 1.25 : 1.)) * static_cast<double>(std::min(8u, best_w));
            if (std::getenv("CASCADE_WAVE_DEBUG")) {
                std::fprintf(stderr, "[wave] ng %d x %u lanes: %u rows (depth %u), %zu + %zu bytes of LDS per wave / schedule, %u waves per workgroup, %u per CU, score %.3f\n",
                             ng, lpg, p.nrows, depth, p.wave_bytes(), p.blk_bytes(), p.wpb, best_w, score);
            }
            if (score > best_score) {
                best_score = score;
                best = p;
            }
        }
    }
    if (best_score < 0.) {
        return false;
    }
    out = std::move(best);
    if (std::getenv("CASCADE_WAVE_DEBUG")) {
        std::fprintf(stderr, "[wave] tape: %u operations, %u slots -> %u fused operations, %u series; %u groups per object\n", nops, nslots,
                     out.fused_ops, out.nslots, out.ng);
    }
    return true;
}

void wave_plan_eval(const wave_plan &p, std::uint32_t order, const double state[7], const double *pars, double time, double *outc)
{
    const std::uint32_t L = order + 1u, ng = p.ng;
    std::vector<double> W(static_cast<std::size_t>(p.nslots + 1u) * L, 0.), pm(static_cast<std::size_t>(p.nrows) * ng, 1.);
    for (int s = 0; s < 7; ++s) {
        W[static_cast<std::size_t>(s) * L] = state[s];
    }
    W[static_cast<std::size_t>(p.one) * L] = 1.;
    auto unit = [&](std::uint32_t r, std::uint32_t u, std::uint32_t g, void *dst, std::size_t bytes) {
        std::memcpy(dst, p.blk.data() + (static_cast<std::size_t>(r) * p.row_units * ng + static_cast<std::size_t>(u) * ng + g) * 4u, bytes);
    };
    for (std::uint32_t n = 0; n < order; ++n) {
        for (std::uint32_t r = 0; r < p.nrows; ++r) {
            // (the places of a row are independent: results go to LDS only after every place has read)
            std::vector<std::pair<std::size_t, double>> stores;
            for (std::uint32_t g = 0; g < ng; ++g) {
                wave_hdr h;
                unit(r, 0, g, &h, 16);
                double acc = 0.;
                for (std::uint32_t m = 0; m < WAVE_LIN_MAX; ++m) {
                    wave_lin l;
                    unit(r, 1u + m, g, &l, 16);
                    acc += l.coef * W[l.off / 8u + n];
                }
                for (std::uint32_t k = 0; k < h.nk; ++k) {
                    wave_conv c;
                    unit(r, 5u + 2u * k, g, &c, 16);
                    unit(r, 6u + 2u * k, g, reinterpret_cast<char *>(&c) + 16, 16);
                    // (term jj of the sum: j = jlo + jj, the lower limit already in the offsets and in wc)
                    const int span = static_cast<int>(n) - static_cast<int>(c.a >> 20);
                    const std::ptrdiff_t ia = static_cast<std::ptrdiff_t>((c.a & 0xfffffu) / 8u), ib = c.b / 8;
                    for (int jj = 0; jj <= span; ++jj) {
                        const double w = static_cast<double>(n) * c.wa + static_cast<double>(jj) * c.wb + c.wc;
                        acc += w * W[static_cast<std::size_t>(ia + jj)] * W[static_cast<std::size_t>(ib + static_cast<std::ptrdiff_t>(n) - jj)];
                    }
                }
                double res;
                if (n == 0u) {
                    const wave_op0 &o = p.ops0[static_cast<std::size_t>(r) * ng + g];
                    const double sv = o.par >= 0 ? pars[o.par] : o.c;
                    const double a0 = W[static_cast<std::size_t>(o.a) * L];
                    double pv = 1.;
                    res = acc;
                    switch (o.kind) {
                        case WK_DIV:
                            res = acc / a0;
                            pv = 1. / a0;
                            break;
                        case WK_SQRT:
                            res = std::sqrt(acc);
                            pv = 1. / (2. * res);
                            break;
                        case WK_POW:
                            res = std::pow(a0, o.c);
                            pv = 1. / a0;
                            break;
                        case WK_EXP:
                            res = std::exp(a0);
                            break;
                        case WK_SIN:
                            res = std::sin(a0);
                            break;
                        case WK_COS:
                            res = std::cos(a0);
                            break;
                        case WK_RDIV:
                            res = sv / a0;
                            pv = 1. / a0;
                            break;
                        case WK_TIME:
                            res = time;
                            break;
                        default:
                            pv = o.par >= 0 ? (o.pad != 0 ? 1. / sv : sv) : 1.;
                            res = acc * pv;
                            break;
                    }
                    pm[static_cast<std::size_t>(r) * ng + g] = pv;
                } else {
                    res = acc * pm[static_cast<std::size_t>(r) * ng + g];
                    if (h.flags & WF_DIVN) {
                        res /= static_cast<double>(n);
                    }
                    if (h.flags & WF_TIME) {
                        res = n == 1u ? 1. : 0.;
                    }
                }
                stores.emplace_back(h.dst / 8u + n, res);
            }
            for (const auto &[at, v] : stores) {
                W[at] = v;
            }
        }
        for (int s = 0; s < 7; ++s) {
            W[static_cast<std::size_t>(s) * L + n + 1u] = W[static_cast<std::size_t>(p.rhs[s]) * L + n] / static_cast<double>(n + 1u);
        }
    }
    for (int s = 0; s < 7; ++s) {
        for (std::uint32_t n = 0; n < L; ++n) {
            outc[static_cast<std::size_t>(s) * L + n] = W[static_cast<std::size_t>(s) * L + n];
        }
    }
}

std::uint64_t wave_plan_hash(const wave_plan &p)
{
    std::uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&h](const void *data, std::size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(data);
        for (std::size_t i = 0; i < bytes; ++i) {
            h = (h ^ b[i]) * 0x100000001b3ull;
        }
    };
    mix(p.blk.data(), p.blk_bytes());
    mix(p.ops0.data(), p.ops0.size() * sizeof(wave_op0));
    mix(p.rhs, sizeof(p.rhs));
    const std::uint32_t v[5] = {p.nrows, p.ng, p.nslots, p.one, p.row_units};
    mix(v, sizeof(v));
    return h;
}

// The schedule of a plan with one operation per row (ng = 1) as straight-line HIP source: what
// k_propagate_wave<.., 1, 8, .., CT> runs in place of its data-driven rows when the dynamics are
// known when the library is built (profiles/gen_compiled_tapes.py writes csrc/gen/compiled_tapes.hpp
// from it). Every lane of an object's 8 executes every operation; all 8 store the same result, so
// no lane ever reads from LDS what another lane wrote and the compiler may interleave the
// operations of a level freely. Offsets, trip counts, weights and 1 / n are literals; sums whose
// range is empty at an order are not emitted at all.
constexpr std::uint32_t EMIT_BATCH_LOADS = 56; // (36: 8.3 / 5.3 / 0.63 ms for default / J2 + drag / Haumea; 56: 7.9 / 3.6 / 0.48; 76: 7.8 / 3.6 / 0.59; 112: 9.7 / 3.6 / 0.46)

std::string wave_emit_source(const wave_plan &p, std::uint32_t order, int id)
{
    std::string o;
    char buf[640];
    auto put = [&o, &buf](const char *fmt, auto... args) {
        std::snprintf(buf, sizeof(buf), fmt, args...);
        o += buf;
    };
    auto lit = [](double v) {
        char b[64];
        if (v == 0.) {
            return std::string("0.");
        }
        std::snprintf(b, sizeof(b), "%a", v);
        return std::string(b);
    };
    const std::uint32_t L = order + 1u;
    auto unit = [&](std::uint32_t r, std::uint32_t u, void *dst, std::size_t bytes) {
        std::memcpy(dst, p.blk.data() + (static_cast<std::size_t>(r) * p.row_units + u) * 4u, bytes);
    };
    auto has_pm = [&](std::uint32_t r) {
        const wave_op0 &z = p.ops0[r];
        return z.kind == WK_DIV || z.kind == WK_SQRT || z.kind == WK_POW || z.kind == WK_RDIV || (z.kind == WK_FUSED && z.par >= 0);
    };
    // The operations by dependency level: those of a level are independent, so their loads are
    // written first, then their arithmetic, then their stores -- no store stands between the loads
    // of a level, and the compiler (which cannot tell Wb + k from Wb + 8 sub + k') may issue them
    // together instead of waiting for one operation's chain before it starts the next one's.
    std::uint32_t depth = 0;
    for (const std::uint32_t l : p.place_level) {
        depth = std::max(depth, l);
    }
    put("// plan hash %016llx: %u operations in %u levels, %u series\n", static_cast<unsigned long long>(wave_plan_hash(p)), p.nrows, depth, p.nslots);
    put("template <>\n__device__ __forceinline__ void wave_compiled_rows<%d>(char *Wb, int sub, const double *pars, double tmhi)\n{\n", id);
    o += "    const char *Ws = Wb + 8 * sub, *Wsb = Wb - 8 * sub;\n    const double dsub = static_cast<double>(sub);\n";
    o += "    (void)pars;\n    (void)tmhi;\n    (void)dsub;\n";
    for (std::uint32_t r = 0; r < p.nrows; ++r) {
        if (has_pm(r)) {
            put("    double pm%u = 1.;\n", r);
        }
    }
    for (std::uint32_t n = 0; n < order; ++n) {
        put("    // ---- order %u ----\n", n);
        // Batches: the operations of a level, cut where their operands would exceed EMIT_BATCH_LOADS doubles
        // (13 operations of one level at order 19 are 300 operands: more than the registers hold).
        std::vector<std::uint32_t> batch_of(p.nrows, 0u);
        std::uint32_t nbatch = 0;
        for (std::uint32_t lv = 1; lv <= depth; ++lv) {
            std::uint32_t in_batch = 0;
            ++nbatch;
            for (std::uint32_t r = 0; r < p.nrows; ++r) {
                if (p.place_level[r] != lv) {
                    continue;
                }
                wave_hdr h;
                unit(r, 0, &h, 16);
                std::uint32_t loads = 2u;
                for (std::uint32_t k = 0; k < h.nk; ++k) {
                    wave_conv c;
                    unit(r, 5u + 2u * k, &c, 16);
                    const int span = static_cast<int>(n) - static_cast<int>(c.a >> 20);
                    loads += span >= 0 ? 2u * static_cast<std::uint32_t>(span / 8 + 1) : 0u;
                }
                if (in_batch != 0u && in_batch + loads > EMIT_BATCH_LOADS) {
                    ++nbatch;
                    in_batch = 0;
                }
                in_batch += loads;
                batch_of[r] = nbatch;
            }
        }
        for (std::uint32_t lv = 1; lv <= nbatch; ++lv) {
            o += "    {\n";
            // -- loads --
            for (std::uint32_t r = 0; r < p.nrows; ++r) {
                if (batch_of[r] != lv) {
                    continue;
                }
                wave_hdr h;
                unit(r, 0, &h, 16);
                const wave_op0 &z = p.ops0[r];
                for (std::uint32_t k = 0; k < h.nk; ++k) {
                    wave_conv c;
                    unit(r, 5u + 2u * k, &c, 16);
                    const int span = static_cast<int>(n) - static_cast<int>(c.a >> 20);
                    if (span < 0) {
                        continue;
                    }
                    const long offa = static_cast<long>(c.a & 0xfffffu), offb = static_cast<long>(c.b) + 8l * static_cast<long>(n);
                    for (int t = 0; t <= span / 8; ++t) {
                        put("        const double a%u_%u_%d = *reinterpret_cast<const double *>(Ws + %ld), b%u_%u_%d = *reinterpret_cast<const double *>(Wsb + %ld);\n",
                            r, k, t, offa + 64l * t, r, k, t, offb - 64l * t);
                    }
                }
                for (std::uint32_t m = 0; m < WAVE_LIN_MAX; ++m) {
                    wave_lin l;
                    unit(r, 1u + m, &l, 16);
                    if (l.coef == 0. || (n > 0u && l.off == p.one * L * 8u)) {
                        continue;
                    }
                    put("        const double l%u_%u = *reinterpret_cast<const double *>(Wb + %lu);\n", r, m, static_cast<unsigned long>(l.off + 8u * n));
                }
                if (n == 0u && z.kind != WK_FUSED && z.kind != WK_SQRT && z.kind != WK_TIME) {
                    put("        const double z%u = *reinterpret_cast<const double *>(Wb + %lu);\n", r,
                        static_cast<unsigned long>(static_cast<std::uint32_t>(z.a) * L * 8u));
                }
            }
            // -- partial sums, written term by term ACROSS the sums of the batch (one dependent chain per sum:
            // in source order they alternate, which is what a lone wave on its SIMD needs to keep issuing),
            // then the sums over the 8 lanes, again step by step across the operations --
            struct conv_ref {
                std::uint32_t r, k;
                wave_conv c;
                int span;
                bool plain;
            };
            std::vector<conv_ref> cvs;
            std::vector<std::uint32_t> summed;
            for (std::uint32_t r = 0; r < p.nrows; ++r) {
                if (batch_of[r] != lv) {
                    continue;
                }
                wave_hdr h;
                unit(r, 0, &h, 16);
                put("        double acc%u = 0.;\n", r);
                bool any_sum = false;
                for (std::uint32_t k = 0; k < h.nk; ++k) {
                    conv_ref cr{};
                    cr.r = r;
                    cr.k = k;
                    unit(r, 5u + 2u * k, &cr.c, 16);
                    unit(r, 6u + 2u * k, reinterpret_cast<char *>(&cr.c) + 16, 16);
                    cr.span = static_cast<int>(n) - static_cast<int>(cr.c.a >> 20);
                    if (cr.span < 0) {
                        continue;
                    }
                    cr.plain = cr.c.wa == 0. && cr.c.wb == 0.;
                    cvs.push_back(cr);
                    any_sum = true;
                }
                if (any_sum) {
                    summed.push_back(r);
                }
            }
            for (int t = 0; t < 3; ++t) {
                for (const conv_ref &cr : cvs) {
                    if (t > cr.span / 8) {
                        continue;
                    }
                    // (the third trip of 8 lanes can leave the 21 coefficients of the series: those lanes are
                    // masked; everywhere else a lane beyond the range reads a zero on the A side)
                    const bool mask = 8 * (t + 1) > static_cast<int>(L);
                    const std::uint32_t r = cr.r, k = cr.k;
                    if (cr.plain) {
                        if (t == 0) {
                            put("        double sk%u_%u = a%u_%u_0 * b%u_%u_0;\n", r, k, r, k, r, k);
                        } else if (!mask) {
                            put("        sk%u_%u = fma(a%u_%u_%d, b%u_%u_%d, sk%u_%u);\n", r, k, r, k, t, r, k, t, r, k);
                        } else {
                            put("        sk%u_%u = sub <= %d ? fma(a%u_%u_%d, b%u_%u_%d, sk%u_%u) : sk%u_%u;\n", r, k, cr.span - 8 * t, r, k, t, r, k, t, r, k, r, k);
                        }
                    } else {
                        const double ct = static_cast<double>(n) * cr.c.wa + cr.c.wc + 8. * t * cr.c.wb;
                        std::snprintf(buf, sizeof(buf), "fma(dsub, %s, %s)", lit(cr.c.wb).c_str(), lit(ct).c_str());
                        const std::string W = buf;
                        if (t == 0) {
                            put("        double wk%u_%u = (%s * a%u_%u_0) * b%u_%u_0;\n", r, k, W.c_str(), r, k, r, k);
                        } else if (!mask) {
                            put("        wk%u_%u = fma(%s * a%u_%u_%d, b%u_%u_%d, wk%u_%u);\n", r, k, W.c_str(), r, k, t, r, k, t, r, k);
                        } else {
                            put("        wk%u_%u = sub <= %d ? fma(%s * a%u_%u_%d, b%u_%u_%d, wk%u_%u) : wk%u_%u;\n", r, k, cr.span - 8 * t, W.c_str(), r, k, t, r, k, t, r, k, r, k);
                        }
                    }
                }
            }
            for (const conv_ref &cr : cvs) {
                if (cr.plain) {
                    put("        acc%u = fma(%s, sk%u_%u, acc%u);\n", cr.r, lit(cr.c.wc).c_str(), cr.r, cr.k, cr.r);
                } else {
                    put("        acc%u += wk%u_%u;\n", cr.r, cr.r, cr.k);
                }
            }
            for (const char *ctrl : {"0xB1", "0x4E", "0x141"}) {
                for (const std::uint32_t r : summed) {
                    put("        acc%u += dpp_f64<%s>(acc%u);\n", r, ctrl, r);
                }
            }
            // -- linear terms, the functions (order 0) or the factors, the stores --
            for (std::uint32_t r = 0; r < p.nrows; ++r) {
                if (batch_of[r] != lv) {
                    continue;
                }
                wave_hdr h;
                unit(r, 0, &h, 16);
                const wave_op0 &z = p.ops0[r];
                for (std::uint32_t m = 0; m < WAVE_LIN_MAX; ++m) {
                    wave_lin l;
                    unit(r, 1u + m, &l, 16);
                    if (l.coef == 0. || (n > 0u && l.off == p.one * L * 8u)) {
                        continue;
                    }
                    put("        acc%u = fma(%s, l%u_%u, acc%u);\n", r, lit(l.coef).c_str(), r, m, r);
                }
                if (n == 0u) {
                    std::string sv = lit(z.c);
                    if (z.par >= 0) {
                        std::snprintf(buf, sizeof(buf), "pars[%d]", z.par);
                        sv = buf;
                    }
                    switch (z.kind) {
                        case WK_DIV:
                            put("        pm%u = 1. / z%u;\n        const double res%u = acc%u / z%u;\n", r, r, r, r, r);
                            break;
                        case WK_SQRT:
                            put("        const double res%u = sqrt(acc%u);\n        pm%u = 1. / (2. * res%u);\n", r, r, r, r);
                            break;
                        case WK_POW:
                            put("        pm%u = 1. / z%u;\n        const double res%u = pow_order0(z%u, %s);\n", r, r, r, r, lit(z.c).c_str());
                            break;
                        case WK_EXP:
                            put("        const double res%u = det_exp(z%u);\n", r, r);
                            break;
                        case WK_SIN:
                            put("        const double res%u = sin(z%u);\n", r, r);
                            break;
                        case WK_COS:
                            put("        const double res%u = cos(z%u);\n", r, r);
                            break;
                        case WK_RDIV:
                            put("        pm%u = 1. / z%u;\n        const double res%u = %s / z%u;\n", r, r, r, sv.c_str(), r);
                            break;
                        case WK_TIME:
                            put("        const double res%u = tmhi;\n", r);
                            break;
                        default:
                            if (z.par >= 0) {
                                put("        pm%u = %s%s;\n        const double res%u = acc%u * pm%u;\n", r, z.pad != 0 ? "1. / " : "", sv.c_str(), r, r, r);
                            } else {
                                put("        const double res%u = acc%u;\n", r, r);
                            }
                            break;
                    }
                } else if (h.flags & WF_TIME) {
                    put("        const double res%u = %s;\n", r, n == 1u ? "1." : "0.");
                } else {
                    std::snprintf(buf, sizeof(buf), "acc%u", r);
                    std::string e = buf;
                    if (has_pm(r)) {
                        std::snprintf(buf, sizeof(buf), "acc%u * pm%u", r, r);
                        e = buf;
                    }
                    if (h.flags & WF_DIVN) {
                        e = "(" + e + ") * " + lit(1. / static_cast<double>(n));
                    }
                    put("        const double res%u = %s;\n", r, e.c_str());
                }
            }
            for (std::uint32_t r = 0; r < p.nrows; ++r) {
                if (batch_of[r] != lv) {
                    continue;
                }
                wave_hdr h;
                unit(r, 0, &h, 16);
                put("        *reinterpret_cast<double *>(Wb + %lu) = res%u;\n", static_cast<unsigned long>(h.dst + 8u * n), r);
            }
            o += "    }\n";
        }
        // Order n of the right-hand sides gives order n + 1 of the state (every lane, for itself).
        o += "    {\n";
        for (int c = 0; c < 7; ++c) {
            put("        const double s%d = *reinterpret_cast<const double *>(Wb + %lu);\n", c,
                static_cast<unsigned long>((static_cast<std::uint32_t>(p.rhs[c]) * L + n) * 8u));
        }
        for (int c = 0; c < 7; ++c) {
            put("        *reinterpret_cast<double *>(Wb + %lu) = s%d * %s;\n", static_cast<unsigned long>((static_cast<std::uint32_t>(c) * L + n + 1u) * 8u), c,
                lit(1. / static_cast<double>(n + 1u)).c_str());
        }
        o += "    }\n";
    }
    o += "}\n";
    return o;
}

hipError_t wave_launch(const wave_plan &plan, const std::uint32_t *d_blk, const wave_op0 *d_ops0, bool events, const prop_params &prm,
                       hipStream_t stream)
{
    const int ng = static_cast<int>(plan.ng), lpg = static_cast<int>(plan.lpg);
    // Waves per workgroup: the plan's (what fills a compute unit), fewer when there would not be a
    // workgroup for each of the 256 compute units (1000 objects are 500 waves: 72 workgroups of 7
    // would leave two thirds of the GPU idle).
    const unsigned opw = 64u / (plan.lpg * plan.ng);
    const std::uint64_t waves = (static_cast<std::uint64_t>(prm.n) + opw - 1u) / opw;
    const unsigned wpb = static_cast<unsigned>(std::max<std::uint64_t>(1u, std::min<std::uint64_t>(plan.wpb, waves / 256u)));
    const unsigned opb = opw * wpb; // objects per workgroup
    const dim3 grid(static_cast<unsigned>((static_cast<std::uint64_t>(prm.n) + opb - 1u) / opb)), block(64u * wpb);
    const std::size_t lds_bytes = wpb * plan.wave_bytes() + plan.blk_bytes();
    wave_sched sc{};
    sc.blk = reinterpret_cast<const uint4 *>(d_blk);
    sc.ops0 = d_ops0;
    sc.nrows = plan.nrows;
    sc.nslots = plan.nslots;
    sc.obj_doubles = plan.obj_doubles;
    sc.row_units = plan.row_units;
    sc.one = plan.one;
    for (int c = 0; c < 7; ++c) {
        sc.rhs[c] = plan.rhs[c];
    }
#define CSC_WAVE_CASE(NGV, LPGV, EV)                                                                                    \
    {                                                                                                                   \
        auto kern = k_propagate_wave<20, NGV, LPGV, EV>;                                                                \
        if (lds_bytes > 48u * 1024u) {                                                                                  \
            const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      static_cast<int>(lds_bytes));                                     \
            if (rc != hipSuccess) {                                                                                     \
                return rc;                                                                                              \
            }                                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, prm, sc);                                              \
    }
#define CSC_WAVE_EV(NGV, LPGV)                                                                                          \
    if (events) {                                                                                                       \
        CSC_WAVE_CASE(NGV, LPGV, true)                                                                                  \
    } else {                                                                                                            \
        CSC_WAVE_CASE(NGV, LPGV, false)                                                                                 \
    }
    switch (ng * 16 + lpg) {
        case 1 * 16 + 8:
            CSC_WAVE_EV(1, 8)
            break;
        case 2 * 16 + 8:
            CSC_WAVE_EV(2, 8)
            break;
        case 4 * 16 + 8:
            CSC_WAVE_EV(4, 8)
            break;
        case 8 * 16 + 8:
            CSC_WAVE_EV(8, 8)
            break;
        case 2 * 16 + 4:
            CSC_WAVE_EV(2, 4)
            break;
        case 4 * 16 + 4:
            CSC_WAVE_EV(4, 4)
            break;
        case 8 * 16 + 4:
            CSC_WAVE_EV(8, 4)
            break;
        default:
            return hipErrorInvalidValue;
    }
#undef CSC_WAVE_EV
#undef CSC_WAVE_CASE
    return hipGetLastError();
}

} // namespace csc

// Diagnostic entry point (declared in include/cascade_hip.h): the fused, scheduled tape executed on
// the host. No GPU work; lets the CPU tests check the fusion against the oracle's coefficients.
extern "C" int csc_wave_plan_eval(const void *tape, uint32_t nops, uint32_t nslots, const int32_t *rhs7, int ng, uint32_t order,
                                  const double *state7, const double *pars, double time, double *out_coeffs, uint32_t *out_stats)
{
    if (tape == nullptr || rhs7 == nullptr || state7 == nullptr || out_coeffs == nullptr || order == 0u || order > 40u) {
        return 1;
    }
    csc::wave_plan plan;
    if (!csc::wave_build_plan(static_cast<const csc::tape_op *>(tape), nops, nslots, rhs7, ng, order, plan)) {
        return 2;
    }
    if (pars == nullptr) {
        // (what the kernel would do with a null parameter array: every place of every row is read)
        for (const csc::wave_op0 &o : plan.ops0) {
            if (o.par >= 0) {
                return 3;
            }
        }
    }
    csc::wave_plan_eval(plan, order, state7, pars, time, out_coeffs);
    if (out_stats != nullptr) {
        out_stats[0] = plan.fused_ops;
        out_stats[1] = plan.nslots;
        out_stats[2] = plan.depth;
        out_stats[3] = plan.nrows;
        out_stats[4] = plan.ng;
        out_stats[5] = static_cast<uint32_t>(plan.lds_bytes());
        out_stats[6] = plan.wpb;
        out_stats[7] = plan.lpg;
    }
    return 0;
}

// Diagnostic / build tool entry point: the straight-line source of the tape's fused schedule (one
// operation per row) as wave_compiled_rows<id>; returns the bytes needed (the text is written
// when it fits `cap`), 0 when the tape has no schedule. *hash: what identifies the schedule at run time.
extern "C" uint64_t csc_wave_emit_source(const void *tape, uint32_t nops, uint32_t nslots, const int32_t *rhs7, uint32_t order, int id,
                                         char *out, uint64_t cap, uint64_t *hash)
{
    csc::wave_plan plan;
    if (tape == nullptr || rhs7 == nullptr || !csc::wave_build_plan(static_cast<const csc::tape_op *>(tape), nops, nslots, rhs7, 1, order, plan)
        || plan.ng != 1u) {
        return 0;
    }
    const std::string src = csc::wave_emit_source(plan, order, id);
    if (hash != nullptr) {
        *hash = csc::wave_plan_hash(plan);
    }
    if (out != nullptr && src.size() + 1u <= cap) {
        std::memcpy(out, src.c_str(), src.size() + 1u);
    }
    return src.size() + 1u;
}
