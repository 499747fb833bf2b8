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
std::vector<std::vector<std::uint32_t>> schedule(const std::vector<sched_item> &items, std::uint32_t nslots, int ng, std::uint32_t &depth)
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
        const auto rows = schedule(items, nslots_new, ng, depth);
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
        p.blk.assign(static_cast<std::size_t>(p.nrows) * p.row_units * p.ng * 4u, 0u);
        p.ops0.assign(static_cast<std::size_t>(p.nrows) * p.ng, wave_op0{});
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
