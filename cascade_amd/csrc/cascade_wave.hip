// cascade_amd/csrc/cascade_wave.hip
//
// Translation unit of k_propagate_wave (one object over 8 to 64 lanes, series in LDS): the
// scheduler that turns a dynamics tape into rows of independent operations, and the launcher.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "wave_api.hpp"

namespace csc
{

namespace
{

constexpr std::uint32_t NOSLOT = 0xffffu;

std::uint32_t pack_f(int a1, int a2, int sg, std::uint32_t flags)
{
    return (static_cast<std::uint32_t>(a1) & 0xffu) | ((static_cast<std::uint32_t>(a2) & 0xffu) << 8) | ((static_cast<std::uint32_t>(sg) & 0xffu) << 16)
           | (flags << 24);
}

// One tape operation -> one or two (sine / cosine) scheduled operations.
struct sched_item {
    wave_op w{};
    wave_op0 z{};
    std::vector<std::int32_t> reads; // slots whose coefficient of the SAME order it needs
    std::int32_t dst = 0;
};

void lower_op(const tape_op &o, std::vector<sched_item> &out)
{
    const bool bs = o.b < 0;
    sched_item it;
    it.dst = o.dst;
    it.z.op = o.op;
    it.z.dst = o.dst;
    it.z.a = o.a;
    it.z.b = o.b;
    it.z.c = o.c;
    it.z.pm_kind = PM_ONE;
    it.z.pm_slot = 0;
    std::uint32_t sa = 0, sb = 0, l1 = 0, l2 = 0, jlo = 0, jhi_off = 0, fl = WF_VALID; // (no linear term: slot 0 x 0)
    int a1 = 0, a2 = 0, sg = 0;
    double wa = 0., wb = 0.;
    auto sum = [&](std::int32_t A, std::int32_t B, std::uint32_t lo, std::uint32_t hoff, int sign, bool wc) {
        sa = static_cast<std::uint32_t>(A);
        sb = static_cast<std::uint32_t>(B);
        jlo = lo;
        jhi_off = hoff;
        sg = sign;
        fl |= WF_SUM | (wc ? WF_WC : 0u);
        // Same-order reads: A[n] when the sum reaches j = n, B[n] when it starts at j = 0.
        if (hoff == 0u && A != o.dst) {
            it.reads.push_back(A);
        }
        if (lo == 0u && B != o.dst) {
            it.reads.push_back(B);
        }
    };
    auto lin1 = [&](std::int32_t s, int c) {
        l1 = static_cast<std::uint32_t>(s);
        a1 = c;
        it.reads.push_back(s);
    };
    auto lin2 = [&](std::int32_t s, int c) {
        l2 = static_cast<std::uint32_t>(s);
        a2 = c;
        it.reads.push_back(s);
    };
    switch (o.op) {
        case TP_ADD:
        case TP_SUB:
            lin1(o.a, 1);
            if (!bs) {
                lin2(o.b, o.op == TP_ADD ? 1 : -1);
            }
            break;
        case TP_RSUB:
            lin1(o.a, -1);
            break;
        case TP_MUL:
            if (bs) {
                lin1(o.a, 1);
                it.z.pm_kind = PM_SCALAR;
            } else {
                sum(o.a, o.b, 0, 0, 1, true);
            }
            break;
        case TP_SQR:
            sum(o.a, o.a, 0, 0, 1, true);
            break;
        case TP_DIV:
            lin1(o.a, 1);
            if (bs) {
                it.z.pm_kind = PM_INV_SCALAR;
            } else {
                sum(o.b, o.dst, 1, 0, -1, true);
                it.z.pm_kind = PM_INV_SLOT0;
                it.z.pm_slot = o.b;
            }
            break;
        case TP_RDIV:
            sum(o.a, o.dst, 1, 0, -1, true);
            it.z.pm_kind = PM_INV_SLOT0;
            it.z.pm_slot = o.a;
            break;
        case TP_POW:
            // c[n] = sum_{j < n} (n alpha - j (alpha + 1)) a[n - j] c[j] / (n a0)
            sum(o.dst, o.a, 0, 1, 1, false);
            wa = o.c;
            wb = -(o.c + 1.);
            fl |= WF_DIVN;
            it.z.pm_kind = PM_INV_SLOT0;
            it.z.pm_slot = o.a;
            break;
        case TP_SQRT:
            lin1(o.a, 1);
            sum(o.dst, o.dst, 1, 1, -1, true);
            it.z.pm_kind = PM_INV_2DST0;
            break;
        case TP_EXP:
            sum(o.a, o.dst, 1, 0, 1, false);
            wb = 1.;
            fl |= WF_DIVN;
            break;
        case TP_SINCOS: {
            // s[n] = sum j a[j] c[n - j] / n, c[n] = -sum j a[j] s[n - j] / n: two operations of one row level.
            sum(o.a, o.dst + 1, 1, 0, 1, false);
            wb = 1.;
            fl |= WF_DIVN;
            break;
        }
        case TP_TIME:
            fl |= WF_TIME;
            break;
        default: // TP_CONST: zero beyond order 0
            break;
    }
    it.w.s = sa | (sb << 16);
    it.w.l = l1 | (l2 << 16);
    it.w.d = static_cast<std::uint32_t>(o.dst) | (jlo << 16) | (jhi_off << 20);
    it.w.f = pack_f(a1, a2, sg, fl);
    it.w.wa = wa;
    it.w.wb = wb;
    out.push_back(it);
    if (o.op == TP_SINCOS) {
        sched_item c2;
        c2.dst = o.dst + 1;
        c2.z = it.z;
        c2.z.op = TP_WAVE_COS;
        c2.z.dst = o.dst + 1;
        c2.reads = it.reads;
        c2.w = it.w;
        c2.w.s = static_cast<std::uint32_t>(o.a) | (static_cast<std::uint32_t>(o.dst) << 16);
        c2.w.d = static_cast<std::uint32_t>(o.dst + 1) | (1u << 16);
        c2.w.f = pack_f(0, 0, -1, WF_VALID | WF_SUM | WF_DIVN);
        out.push_back(c2);
    }
}

} // namespace

int wave_pick_ng(std::uint32_t nslots, std::uint32_t nops, std::uint32_t order)
{
    if (const char *e = std::getenv("CASCADE_WAVE_NG")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8) {
            return v;
        }
    }
    // Upper bound of the LDS doubles of one object: series, one factor per operation (rows x ng
    // <= 2 x nops + ng), event polynomial.
    const std::size_t L = order + 1u;
    int best = 0;
    for (const int ng : {1, 2, 4, 8}) {
        const std::size_t obj = (static_cast<std::size_t>(nslots + 1u) * L + 2u * nops + 8u + L + 1u) * sizeof(double) + (2u * nops + 8u) * 32u / (64u / (8u * static_cast<unsigned>(ng)));
        const std::size_t per_wave = obj * (64u / (8u * static_cast<unsigned>(ng)));
        if (per_wave > WAVE_LDS_MAX) {
            continue;
        }
        if (best == 0) {
            best = ng;
        }
        // Fewer lanes per object use the lanes better; take the smallest layout that still puts
        // 8 waves on a compute unit (two per SIMD hide each other's LDS round trips).
        if (WAVE_LDS_MAX / per_wave >= 8u) {
            return ng;
        }
        best = ng;
    }
    return best;
}

void wave_build_plan(const tape_op *tape, std::uint32_t nops, std::uint32_t nslots, const std::int32_t rhs[7], int ng,
                     std::uint32_t order, wave_plan &out)
{
    std::vector<sched_item> items;
    for (std::uint32_t k = 0; k < nops; ++k) {
        lower_op(tape[k], items);
    }
    // Level of an operation = 1 + the highest level among the producers of what it reads at the
    // same order (the state slots 0..6 are known before the first row).
    std::vector<int> slot_level(nslots, 0), lvl(items.size(), 0);
    int max_level = 0;
    for (std::size_t k = 0; k < items.size(); ++k) {
        int lv = 0;
        for (const std::int32_t s : items[k].reads) {
            lv = std::max(lv, slot_level[static_cast<std::size_t>(s)]);
        }
        // Order 0 also reads what the factor pm needs (the operand's order 0: a same-level read as well).
        lvl[k] = lv + 1;
        slot_level[static_cast<std::size_t>(items[k].dst)] = lvl[k];
        max_level = std::max(max_level, lvl[k]);
    }
    out.ops.clear();
    out.ops0.clear();
    out.ng = static_cast<std::uint32_t>(ng);
    out.nslots = nslots;
    std::uint32_t rows = 0;
    for (int lv = 1; lv <= max_level; ++lv) {
        std::uint32_t in_row = 0;
        for (std::size_t k = 0; k < items.size(); ++k) {
            if (lvl[k] != lv) {
                continue;
            }
            if (in_row == 0u) {
                out.ops.resize(static_cast<std::size_t>(rows + 1u) * ng);
                out.ops0.resize(static_cast<std::size_t>(rows + 1u) * ng);
                for (int g = 0; g < ng; ++g) {
                    wave_op e{};
                    e.d = nslots; // (an empty place: the spare slot behind the series takes what it computes)
                    out.ops[static_cast<std::size_t>(rows) * ng + g] = e;
                    out.ops0[static_cast<std::size_t>(rows) * ng + g] = wave_op0{};
                }
            }
            out.ops[static_cast<std::size_t>(rows) * ng + in_row] = items[k].w;
            out.ops0[static_cast<std::size_t>(rows) * ng + in_row] = items[k].z;
            if (++in_row == static_cast<std::uint32_t>(ng)) {
                in_row = 0;
                ++rows;
            }
        }
        if (in_row != 0u) {
            ++rows;
        }
    }
    out.nrows = rows;
    if (std::getenv("CASCADE_WAVE_DEBUG")) {
        // operations per dependency level, by kind: s = with a convolution sum, l = linear only
        for (int lv = 1; lv <= max_level; ++lv) {
            int ns = 0, nl = 0;
            for (std::size_t k = 0; k < items.size(); ++k) {
                if (lvl[k] == lv) {
                    ((items[k].w.f >> 24) & WF_SUM) ? ++ns : ++nl;
                }
            }
            std::fprintf(stderr, "[wave] level %2d: %2d sums, %2d linear\n", lv, ns, nl);
        }
    }
    const std::uint32_t L = order + 1u;
    out.obj_doubles = ((nslots + 1u) * L + rows * static_cast<std::uint32_t>(ng) + L + 1u) & ~1u;
    for (int c = 0; c < 7; ++c) {
        out.rhs[c] = rhs[c];
    }
}

hipError_t wave_launch(int ng, bool events, const prop_params &prm, const wave_sched &sc, std::size_t lds_bytes, hipStream_t stream)
{
    const unsigned opw = 64u / (8u * static_cast<unsigned>(ng));
    const dim3 grid(static_cast<unsigned>((static_cast<std::uint64_t>(prm.n) + opw - 1u) / opw)), block(64);
#define CSC_WAVE_CASE(NGV, EV)                                                                                          \
    {                                                                                                                   \
        auto kern = k_propagate_wave<20, NGV, EV>;                                                                      \
        if (lds_bytes > 48u * 1024u) {                                                                                  \
            const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      static_cast<int>(lds_bytes));                                     \
            if (rc != hipSuccess) {                                                                                     \
                return rc;                                                                                              \
            }                                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, prm, sc);                                              \
    }
    switch (ng * 2 + (events ? 1 : 0)) {
        case 2:
            CSC_WAVE_CASE(1, false)
            break;
        case 3:
            CSC_WAVE_CASE(1, true)
            break;
        case 4:
            CSC_WAVE_CASE(2, false)
            break;
        case 5:
            CSC_WAVE_CASE(2, true)
            break;
        case 8:
            CSC_WAVE_CASE(4, false)
            break;
        case 9:
            CSC_WAVE_CASE(4, true)
            break;
        case 16:
            CSC_WAVE_CASE(8, false)
            break;
        case 17:
            CSC_WAVE_CASE(8, true)
            break;
        default:
            return hipErrorInvalidValue;
    }
#undef CSC_WAVE_CASE
    return hipGetLastError();
}

} // namespace csc
