// cascade_amd/csrc/k_propagate_wave.hpp
//
// K1 for every right-hand side that is not plain Kepler: ONE OBJECT OVER SEVERAL LANES.
//
// Replaces (like k_propagate.hpp) heyoka's JIT-compiled Taylor step (src/sim_basic.cpp:553-577,
// EXTERNAL), the propagate_for loop and the propagate callback (src/sim_propagate.cpp:650-877,
// 710-755) for dynamics given as a tape of Taylor-series operations (include/cascade/expression.hpp).
//
// Why. With one object per lane the 7 state series fill the register file and the auxiliary series
// of J2 + drag (17) or of the reference's default simple_earth() (~130) live in scratch or in a
// global workspace: 2 k dependent memory round trips per substep behind one wave per SIMD (16 ns
// per object-substep for J2 + drag, 570 ns for the tape interpreter). Here an object owns 8 NG
// lanes (NG = 1, 2, 4 or 8: 8 to 64 lanes) and ALL its series live in LDS, [slot][order]. The
// host (cascade_wave.hip) first FUSES the tape: sums, differences, scalings and products are
// folded into sum-of-products operations (x^2 + y^2 + z^2 is one operation, a quotient takes a
// whole expression as its numerator), which for the reference's default dynamics leaves 46 of 132
// operations, 54 of 140 series and 8 of 17 dependency levels. The fused operations are scheduled
// into rows of up to NG mutually independent ones; per Taylor order and row, each 8-lane group
// takes one operation: its lanes split the convolution sums (terms j = jlo + lane, + 8, + 16) and
// the linear terms, three DPP steps add the partial sums, and the coefficient is stored. Every
// automatic-differentiation recurrence (sum of products, quotient, power, square root,
// exponential, sine / cosine) has the form
//     c[n] = (sum_m a_m L_m[n] + sum_k sum_j (n wa_k + j wb_k + wc_k) A_k[j] B_k[n - j]) * pm * (1 / n)?
// so the row code is data-driven (wave_hdr / wave_lin / wave_conv) and the groups of a wave never
// diverge. Order 0 runs through the same rows; there the sums give the arguments, a switch applies
// the function itself and leaves the order-independent factor pm (1 / b0, 1 / (2 c0), a runtime
// parameter ...) in LDS.
//
// The summation order differs from oracle/orc.hpp::tape_compute_tcs (8 partial sums, then a tree),
// so the coefficients agree with the oracle to rounding (1e-15 relative), not bit for bit; the
// contract for propagation is 1e-9 relative with identical event sets (tests/test_gpu_wave.py).
//
// The kernel writes the dense-output rows of every substep (fixed slots, k_propagate.hpp's
// single-pass layout); the per-chunk boxes and trajectory samples come from the rows
// (k_chunk_boxes). Step-size control, terminal events, double-length time and the status records
// are those of propagate_object (k_propagate.hpp), restated for lanes that share an object.
#pragma once

#include "dev_math.hpp"
#include "k_propagate.hpp"

namespace csc
{

// The schedule, in 16-byte units [row][unit][group]: unit 0 the header, units 1..4 the linear
// terms (taken by lanes 0..3 of the group), units 5 + 2 k, 6 + 2 k the k-th convolution sum.
// Series are addressed by their byte offset within the object (slot x (P + 1) x 8).
struct wave_hdr {
    std::uint32_t dst, flags; // (dst: byte offset)
    std::uint32_t nk; // convolution sums of the ROW (the most of any of its operations)
    std::uint32_t weighted; // 1 when a sum of the ROW has weights that depend on n or j (wave_conv: wa, wb)
};
struct wave_lin {
    std::uint32_t off, pad; // (no term: offset 0 with coefficient 0)
    double coef;
};
// sum over j = jlo .. n - hoff of (n wa + j wb + wc) A[j] B[n - j], as terms jj = j - jlo = 0 .. n - (jlo + hoff):
// (two units: a, b, wc -- all a plain product needs -- and wa, wb, which a row without weighted sums never reads)
struct wave_conv {
    std::uint32_t a; // byte offset of A[jlo] | (jlo + hoff) << 20 (no sum: 0xfff << 20)
    std::int32_t b;  // byte offset of B[-jlo]: term jj reads B at b + 8 (n - jj)
    double wc, wa, wb; // weight of term jj: n wa + jj wb + wc (wc includes jlo wb)
};
enum : std::uint32_t { WF_TIME = 1, WF_DIVN = 2 };
// WAVE_OBJ_PAD doubles of zeros in front of every object's series (a term beyond the range of a
// sum reads up to 21 doubles before a series); P_SERIES: coefficients of a series at order 20.
constexpr std::uint32_t WAVE_LIN_MAX = 4, WAVE_CONV_MAX = 4, WAVE_OBJ_PAD = 24;
constexpr int P_SERIES = 21;
// Waves of a workgroup at most (they share the schedule in LDS; 7 x 20 KB of series + 13 KB of schedule fill a compute unit).
constexpr std::uint32_t WAVE_WPB_MAX = 8;
static_assert(sizeof(wave_hdr) == 16 && sizeof(wave_lin) == 16 && sizeof(wave_conv) == 32, "schedule layout");

// Order 0 of one scheduled operation, 32 bytes: what to do with the sums of order 0, and what to leave in pm.
struct wave_op0 {
    std::int32_t kind;
    std::int32_t a;   // operand whose order 0 the function takes (denominator, base, argument)
    std::int32_t par; // runtime parameter (>= 0) in place of c / as the factor of a sum of products
    std::int32_t pad; // (sum of products: 1 when the parameter divides)
    double c;         // exponent of a power, numerator of scalar / series
    double pad2;
};
enum : std::int32_t { WK_FUSED = 0, WK_DIV, WK_SQRT, WK_POW, WK_EXP, WK_SIN, WK_COS, WK_RDIV, WK_TIME };
static_assert(sizeof(wave_op0) == 32, "wave_op0 layout");

struct wave_sched {
    const uint4 *blk;     // [nrows][row_units][NG]
    const wave_op0 *ops0; // [nrows][NG]
    std::uint32_t nrows, nslots; // nslots: the series of an object, the constant series `one` included, the spare one behind them not
    std::uint32_t obj_doubles;   // LDS doubles per object
    std::uint32_t row_units, one;
    std::int32_t rhs[7];
};

template <int CTRL>
CSC_DEV double dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // (every lane has a source lane under the controls used here: the old value never shows)
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi2, lo2);
}

// Sum / maximum over the 8 lanes of a group (every lane gets the result).
CSC_DEV double group8_sum(double v)
{
    v += dpp_f64<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141>(v); // row_half_mirror: lane i <-> 7 - i of each 8
    return v;
}
CSC_DEV double group8_max(double v)
{
    v = dmax(v, dpp_f64<0xB1>(v));
    v = dmax(v, dpp_f64<0x4E>(v));
    v = dmax(v, dpp_f64<0x141>(v));
    return v;
}

CSC_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Terminal event on [0, h): the polynomial g (in LDS) through the interval test, the first node of
// the Descartes search and, when neither dismisses it, the full root isolation. (Inlined: out of
// line the kernel is charged the callee's registers all the same, and a call under a partial
// EXEC mask is what k_propagate.hpp avoids on purpose.)
template <int P>
CSC_DEV double wave_event_root(const double *gs, double h, int direction, int *flags)
{
    double g[P + 1];
#pragma unroll
    for (int n = 0; n <= P; ++n) {
        g[n] = gs[n];
    }
    if (fex_excluded<P>(g, h) || descartes_no_root<P>(g, h)) {
        return -1.;
    }
    return event_first_root<P>(g, h, direction, flags);
}

// 1 / n as the division would give it (a division costs ~35 instructions, and there were two per order).
__device__ const double wave_rcp[32] = {0.,      1. / 1,  1. / 2,  1. / 3,  1. / 4,  1. / 5,  1. / 6,  1. / 7,  1. / 8,  1. / 9,  1. / 10,
                                        1. / 11, 1. / 12, 1. / 13, 1. / 14, 1. / 15, 1. / 16, 1. / 17, 1. / 18, 1. / 19, 1. / 20, 1. / 21,
                                        1. / 22, 1. / 23, 1. / 24, 1. / 25, 1. / 26, 1. / 27, 1. / 28, 1. / 29, 1. / 30, 1. / 31};

CSC_DEV double u2d(std::uint32_t lo, std::uint32_t hi)
{
    return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}

// NK convolution sums of one operation at order n, the terms of each over the 8 lanes of the group in
// TR trips (lane `sub` of LPG takes jj = sub, sub + LPG, ... of 0 .. n - (jlo + hoff)); cv: the NK
// wave_conv units as loaded.
struct wave_conv_ctx {
    const char *Ws, *Wnb; // the object's series + 8 sub; + 8 n - 8 sub
    int n, sub;
    double dn, dsub;
};
// No lane is switched off: every coefficient of an order that has not been computed yet IS zero
// (the series are cleared at the start of a substep), so the terms beyond the range of a sum read
// a zero on the A side; what they read on the B side (lower slots of the same object, or the
// zeros in front of its first series) is finite. With 8 lanes the third trip can leave the
// series (terms 16 .. 23 of 21 coefficients): those lanes are masked.
template <int NK, int TR, int LPG, bool WT>
CSC_DEV double wave_conv_sums(const uint4 (&cv)[2 * WAVE_CONV_MAX], const wave_conv_ctx &c, double acc)
{
    double av[NK][TR], bv[NK][TR];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const char *pa = c.Ws + (cv[2 * k].x & 0xfffffu), *pb = c.Wnb + static_cast<int>(cv[2 * k].y);
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            av[k][t] = *reinterpret_cast<const double *>(pa + 8 * LPG * t);
            bv[k][t] = *reinterpret_cast<const double *>(pb - 8 * LPG * t);
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const double wc = u2d(cv[2 * k].z, cv[2 * k].w);
        if constexpr (WT) {
            const double wb = u2d(cv[2 * k + 1].z, cv[2 * k + 1].w);
            double w = fma(c.dsub, wb, fma(c.dn, u2d(cv[2 * k + 1].x, cv[2 * k + 1].y), wc));
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (LPG * (t + 1) > P_SERIES) {
                    const int lim = c.n - static_cast<int>(cv[2 * k].x >> 20) - c.sub;
                    const double p = fma(w * av[k][t], bv[k][t], acc);
                    acc = lim >= LPG * t ? p : acc;
                } else {
                    acc = fma(w * av[k][t], bv[k][t], acc);
                }
                w = fma(static_cast<double>(LPG), wb, w);
            }
        } else {
            // (constant weight: the plain sum of products, scaled once)
            double sk = 0.;
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (LPG * (t + 1) > P_SERIES) {
                    const int lim = c.n - static_cast<int>(cv[2 * k].x >> 20) - c.sub;
                    const double p = fma(av[k][t], bv[k][t], sk);
                    sk = lim >= LPG * t ? p : sk;
                } else {
                    sk = fma(av[k][t], bv[k][t], sk);
                }
            }
            acc = fma(wc, sk, acc);
        }
    }
    return acc;
}

template <int NK, int LPG, bool WT>
CSC_DEV double wave_conv_trips(const uint4 (&cv)[2 * WAVE_CONV_MAX], const wave_conv_ctx &c, double acc, int trips)
{
    switch (trips) {
        case 1:
            return wave_conv_sums<NK, 1, LPG, WT>(cv, c, acc);
        case 2:
            return wave_conv_sums<NK, 2, LPG, WT>(cv, c, acc);
        case 3:
            return wave_conv_sums<NK, 3, LPG, WT>(cv, c, acc);
        // (in two parts it was slower: 14.5 against 13.4 ms for the default dynamics)
        case 4:
            if constexpr (LPG < 8) {
                return wave_conv_sums<NK, 4, LPG, WT>(cv, c, acc);
            }
            return acc;
        default:
            if constexpr (LPG < 8) {
                return wave_conv_sums<NK, 5, LPG, WT>(cv, c, acc);
            }
            return acc;
    }
}

// The sums of a row: its units beyond the first (which came with the previous row) are requested
// here, the second unit of a sum only when the row has weighted sums.
template <int NK, int LPG>
CSC_DEV double wave_row_sums(uint4 (&cv)[2 * WAVE_CONV_MAX], const uint4 *rp, int ng, bool weighted, const wave_conv_ctx &c, double acc, int trips)
{
#pragma unroll
    for (int k = 1; k < NK; ++k) {
        cv[2 * k] = rp[(5 + 2 * k) * ng];
    }
    if (weighted) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            cv[2 * k + 1] = rp[(6 + 2 * k) * ng];
        }
        return wave_conv_trips<NK, LPG, true>(cv, c, acc, trips);
    }
    return wave_conv_trips<NK, LPG, false>(cv, c, acc, trips);
}

// Sum over the LPG lanes of a group (every lane gets the result).
template <int LPG>
CSC_DEV double group_sum(double v)
{
    v += dpp_f64<0xB1>(v); // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E>(v); // quad_perm [2, 3, 0, 1]
    if constexpr (LPG == 8) {
        v += dpp_f64<0x141>(v); // row_half_mirror: lane i <-> 7 - i of each 8
    }
    return v;
}

// Compiled tapes (csrc/gen/compiled_tapes.hpp, written by profiles/gen_compiled_tapes.py from
// cascade_wave.hip::wave_emit_source): orders 0 .. P - 1 of a schedule with one operation per row as
// straight-line code, for dynamics known when the library is built. CT = 0: none (the rows are
// interpreted from the schedule in LDS).
template <int CT>
__device__ __forceinline__ void wave_compiled_rows(char *Wb, int sub, const double *pars, double tmhi);

// A workgroup is 1 to 8 waves that share nothing but the copy of the schedule in LDS: after the
// copy every wave works through its own objects and synchronises with itself only.
template <int P, int NG, int LPG, bool EVENTS, int CT = 0>
// (a compiled tape runs one wave per workgroup and, its series filling the LDS, one wave per SIMD at most: 512 registers)
__global__ __launch_bounds__(CT != 0 ? 64 : 64 * WAVE_WPB_MAX, CT != 0 ? 1 : 2) void k_propagate_wave(const prop_params prm, const wave_sched sc)
{
    extern __shared__ double lds[];
    constexpr int L = P + 1;
    static_assert((LPG == 4 || LPG == 8) && LPG * NG >= 8 && LPG * NG <= 64, "lanes per object");
    constexpr int LPO = LPG * NG;  // lanes per object
    constexpr int OPW = 64 / LPO;  // objects per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    // sub / grp: the lane's place in the rows (LPG lanes per operation); sub8: its place among 8
    // aligned lanes, for the parts that give a lane to each state variable.
    const int sub = lane % LPG, grp = (lane / LPG) % NG, sub8 = lane & 7, ow = lane / LPO, lo = lane % LPO;
    const unsigned long long omask = (LPO == 64 ? ~0ull : ((1ull << LPO) - 1ull)) << (ow * LPO);
    static_assert(P + 1 == P_SERIES, "series length");
    double *W = lds + WAVE_OBJ_PAD + static_cast<std::size_t>(wv * OPW + ow) * sc.obj_doubles;
    char *Wb = reinterpret_cast<char *>(W);
    const char *Ws = Wb + 8 * sub;
    const double dsub = static_cast<double>(sub);
    double *pm = W + static_cast<std::size_t>(sc.nslots + 1u) * L; // [nrows][NG] (behind the series and one spare slot)
    double *gs = pm + static_cast<std::size_t>(sc.nrows) * NG; // [L] event polynomial / broadcast scratch
    // The schedule is read nrows x P times per substep: a copy in LDS behind the objects' series
    // (the 8 lanes of a group read the same 16 bytes).
    uint4 *lops = reinterpret_cast<uint4 *>(lds + static_cast<std::size_t>(wpb * OPW) * sc.obj_doubles);
    const std::uint32_t RUN = sc.row_units * NG; // 16-byte units per row
    static_assert(CT == 0 || (NG == 1 && LPG == 8), "a compiled tape runs one operation at a time, 8 lanes per object");
    if constexpr (CT == 0) {
        for (std::uint32_t k = threadIdx.x; k < sc.nrows * RUN; k += blockDim.x) {
            lops[k] = sc.blk[k];
        }
    }
    for (int e = lo; e < static_cast<int>(WAVE_OBJ_PAD); e += LPO) {
        W[e - static_cast<int>(WAVE_OBJ_PAD)] = 0.;
    }
    __syncthreads();
    // Clears the series of the object (the spare one included; 16 bytes per lane and store).
    const int clear_units = static_cast<int>(((sc.nslots + 1u) * L + 1u) / 2u);
    const auto clear_series = [&]() {
        double2 *z = reinterpret_cast<double2 *>(W);
        for (int e = lo; e < clear_units; e += LPO) {
            z[e] = double2{0., 0.};
        }
    };
    const uint4 *rowp = lops + grp;
    const int lin_unit = (1 + (sub & 3)) * NG;
    const wave_op0 *ops0 = sc.ops0 + grp;
    const dfl init_time{prm.init_hi, prm.init_lo};
    const dfl t_end = df_add(init_time, df(prm.delta_t));
    constexpr int XS = tc_xyz_stride(P), VS = tc_vr_stride(P);

    for (std::uint32_t base = (blockIdx.x * wpb + wv) * OPW; base < prm.n; base += gridDim.x * wpb * OPW) {
        const std::uint32_t k_obj = base + ow;
        const bool exists = k_obj < prm.n;
        const std::uint32_t kk = exists ? k_obj : prm.n - 1u;
        // Store pass (csc_propagate_store): object order[k], exact-fit rows from row_base[k], as many as pass 1 counted.
        const bool store = prm.nsub_in != nullptr;
        const std::uint32_t i = store ? (prm.order != nullptr ? prm.order[kk] : kk) : prm.first + kk;
        const double *sp = prm.state + static_cast<std::size_t>(i) * 7u;
        const double *pars = prm.pars + static_cast<std::size_t>(i) * prm.npars;
        clear_series();
        wave_lds_sync();
        if (lo < 6) {
            W[lo * L] = sp[lo];
        }
        if (lo == 7) {
            W[sc.one * L] = 1.; // the constant series (additive constants are its multiples)
        }
        wave_lds_sync();
        if (lo == 6) {
            const double x0 = W[0], y0 = W[L], z0 = W[2 * L];
            W[6 * L] = sqrt(x0 * x0 + y0 * y0 + z0 * z0); // src/sim_propagate.cpp:160
        }
        wave_lds_sync();

        dfl tm = init_time;
        dfl tc_prev = df(0.);
        int status = ST_TIME_LIMIT, flags = 0;
        std::uint32_t ns = 0, ev_idx = 0;
        double ev_time = 0., nf_last_t = 0.;
        bool alive = exists;
        const std::size_t row0 = store ? static_cast<std::size_t>(prm.row_base[kk]) : static_cast<std::size_t>(i - prm.row_first) * prm.row_cap;
        const std::uint32_t cap = store ? prm.nsub_in[i] : prm.row_cap;

        while (__any(alive)) {
            const dfl rem = df_sub(t_end, tm);
            const double max_dt = rem.hi > 0. ? rem.hi : 0.;

            // ---- orders 0 .. P - 1 of every operation; order n of the right-hand side gives order n + 1 of the state ----
            if constexpr (CT != 0) {
                wave_compiled_rows<CT>(Wb, sub, pars, tm.hi);
                wave_lds_sync();
            }
            for (int n = 0; CT == 0 && n < P; ++n) {
                static_assert(P < 31, "wave_rcp");
                const double dn = static_cast<double>(n), rn = n > 0 ? wave_rcp[n] : 1.;
                const int trips = __builtin_amdgcn_readfirstlane((n + LPG) / LPG);
                const bool order0 = __builtin_amdgcn_readfirstlane(n) == 0;
                const char *Wn = Wb + 8 * n;      // + offset of a series: its order n
                const char *Wnb = Wn - 8 * sub;   // ... its order n - sub
                uint4 nh = rowp[0], nl = rowp[lin_unit];
                uint4 ncv = rowp[5 * NG];
                for (std::uint32_t r = 0; r < sc.nrows; ++r) {
                    // (the next row's header, linear term and first sum are requested before this row's work starts)
                    const uint4 hd = nh, ln = nl;
                    uint4 cv[2 * WAVE_CONV_MAX];
                    const uint4 *rp = rowp + r * RUN;
                    const uint4 *np = rowp + (r + 1u < sc.nrows ? r + 1u : 0u) * RUN;
                    cv[0] = ncv;
                    nh = np[0];
                    nl = np[lin_unit];
                    ncv = np[5 * NG];
                    const std::uint32_t fl = hd.y;
                    const int nk = __builtin_amdgcn_readfirstlane(static_cast<int>(hd.z));
                    const bool weighted = __builtin_amdgcn_readfirstlane(static_cast<int>(hd.w)) != 0;
                    // No branches below except wave-uniform ones: an absent linear term reads offset 0
                    // with coefficient 0, an absent sum has an empty range, an empty place in a row
                    // writes to the spare slot behind the series, all lanes of a group store the (same) result.
                    const double pmv = pm[r * NG + grp]; // (order 0 writes it below and does not use it)
                    double acc = u2d(ln.z, ln.w) * *reinterpret_cast<const double *>(Wn + ln.x);
                    if constexpr (LPG == 8) {
                        acc = sub < 4 ? acc : 0.;
                    }
                    // The sums of the row, straight-line code per (sums, trips): every operand is
                    // requested before the first product, so a row costs one LDS round trip, not one per sum.
                    const wave_conv_ctx cc{Ws, Wnb, n, sub, dn, dsub};
                    switch (nk) {
                        case 1:
                            acc = wave_row_sums<1, LPG>(cv, rp, NG, weighted, cc, acc, trips);
                            break;
                        case 2:
                            acc = wave_row_sums<2, LPG>(cv, rp, NG, weighted, cc, acc, trips);
                            break;
                        case 3:
                            acc = wave_row_sums<3, LPG>(cv, rp, NG, weighted, cc, acc, trips);
                            break;
                        case 4:
                            acc = wave_row_sums<4, LPG>(cv, rp, NG, weighted, cc, acc, trips);
                            break;
                        default: // (a row without sums)
                            break;
                    }
                    acc = group_sum<LPG>(acc);
                    if (order0) {
                        // The functions themselves, and the order-independent factor of the recurrence.
                        const wave_op0 o = ops0[static_cast<std::size_t>(r) * NG];
                        const double sv = o.par >= 0 ? pars[o.par] : o.c;
                        const double a0 = W[o.a * L];
                        double res = acc, pv = 1.;
                        switch (o.kind) {
                            case WK_DIV:
                                res = acc / a0;
                                pv = 1. / a0;
                                break;
                            case WK_SQRT:
                                res = sqrt(acc);
                                pv = 1. / (2. * res);
                                break;
                            case WK_POW:
                                res = pow_order0(a0, o.c);
                                pv = 1. / a0;
                                break;
                            case WK_EXP:
                                res = det_exp(a0);
                                break;
                            case WK_SIN:
                                res = sin(a0);
                                break;
                            case WK_COS:
                                res = cos(a0);
                                break;
                            case WK_RDIV:
                                res = sv / a0;
                                pv = 1. / a0;
                                break;
                            case WK_TIME:
                                res = tm.hi;
                                break;
                            default: // WK_FUSED: a sum of products, times a runtime parameter if it has one
                                pv = o.par >= 0 ? (o.pad != 0 ? 1. / sv : sv) : 1.;
                                res = acc * pv;
                                break;
                        }
                        pm[r * NG + grp] = pv;
                        *reinterpret_cast<double *>(Wb + hd.x) = res;
                    } else {
                        double res = acc * pmv;
                        res = (fl & WF_DIVN) ? res * rn : res;
                        res = (fl & WF_TIME) ? (n == 1 ? 1. : 0.) : res;
                        *reinterpret_cast<double *>(Wb + hd.x + 8 * n) = res;
                    }
                    wave_lds_sync();
                }
                if (lo < 7) {
                    W[lo * L + n + 1] = W[sc.rhs[lo] * L + n] * wave_rcp[n + 1];
                }
                wave_lds_sync();
            }

            // ---- step size (Jorba-Zou; heyoka, EXTERNAL) ----
            double m0 = 0., mo = 0., mom1 = 0.;
            bool badl = false;
            if (sub8 < 7) {
                // (every 8 lanes of the object do it: no broadcast needed)
                const double a0 = fabs(W[sub8 * L]), ao = fabs(W[sub8 * L + P]), aom1 = fabs(W[sub8 * L + P - 1]);
                badl = !finite_d(a0) || !finite_d(ao) || !finite_d(aom1);
                m0 = a0;
                mo = ao;
                mom1 = aom1;
            }
            m0 = group8_max(m0);
            mo = group8_max(mo);
            mom1 = group8_max(mom1);
            bool bad = (__ballot(badl) & omask) != 0ull;
            bool stop_nf = false;
            double h = 0.;
            bool tlimit = false;
            if (bad) {
                stop_nf = true;
            } else {
                const double num = m0 < 1. ? 1. : m0;
                const double rho_o = det_pow(num / mo, 1. / static_cast<double>(P));
                const double rho_om1 = det_pow(num / mom1, 1. / static_cast<double>(P - 1));
                h = dmin(rho_o, rho_om1) * prm.rhofac;
                if (!(h < max_dt)) {
                    if (h != h) {
                        stop_nf = true;
                    }
                    h = max_dt;
                    tlimit = true;
                }
            }

            // ---- terminal events in [0, h) ----
            bool ev_found = false;
            if constexpr (EVENTS) {
                double ev_t = 0.;
                std::uint32_t idx = 0;
                // Where the object can be during the step, from the series of r (the 7th state
                // variable): r0 -+ sum |r_n| h^n. Nearly every step stays clear of both surfaces by
                // hundreds of kilometres; the event polynomial and its root search are for the others.
                double rlo, rhi;
                {
                    const double *c = W + 6 * L;
                    double sabs = fabs(c[P]);
#pragma unroll
                    for (int o = P - 1; o >= 1; --o) {
                        sabs = fma(sabs, h, fabs(c[o]));
                    }
                    const double dev = sabs * h;
                    rlo = c[0] - dev;
                    rhi = c[0] + dev;
                }
                for (int e = 0; e < 2; ++e) {
                    const bool on = e == 0 ? prm.has_exit != 0 : prm.has_reentry != 0;
                    if (!on) {
                        continue;
                    }
                    const double kx = e == 0 ? 1. : prm.re_k[0], ky = e == 0 ? 1. : prm.re_k[1], kz = e == 0 ? 1. : prm.re_k[2];
                    const double c0 = e == 0 ? prm.exit_c0 : prm.re_c0;
                    // g(t) = kx x^2 + ky y^2 + kz z^2 - c0 >= kmin |p|^2 - c0 (reentry), = |p|^2 - c0 (exit):
                    // no root while the bound keeps its sign, with a margin of 1e-9 for what separates
                    // the series of r from the norm of the truncated position polynomials (1e-16).
                    // (written so that a NaN anywhere keeps the search)
                    bool may = alive && !stop_nf && h > 0.;
                    if (e == 0) {
                        may = may && !(rhi * rhi < c0 * (1. - 1e-9));
                    } else {
                        const double kmin = dmin(kx, dmin(ky, kz));
                        may = may && !(rlo > 0. && kmin > 0. && kmin * (rlo * rlo) > c0 * (1. + 1e-9));
                    }
                    if (!__any(may)) {
                        idx = 1;
                        continue;
                    }
                    wave_lds_sync();
                    for (int n = lo; n <= P; n += LPO) {
                        double ax = 0., ay = 0., az = 0.;
                        for (int j = 0; j <= n; ++j) {
                            ax = fma(W[j], W[n - j], ax);
                            ay = fma(W[L + j], W[L + n - j], ay);
                            az = fma(W[2 * L + j], W[2 * L + n - j], az);
                        }
                        gs[n] = (kx * ax + ky * ay) + kz * az - (n == 0 ? c0 : 0.);
                    }
                    wave_lds_sync();
                    if (may) {
                        const double rt = wave_event_root<P>(gs, h, e == 0 ? 1 : -1, &flags);
                        if (rt >= 0. && (!ev_found || rt < ev_t)) {
                            ev_found = true;
                            ev_t = rt;
                            ev_idx = idx;
                        }
                    }
                    idx = 1;
                }
                if (ev_found) {
                    h = ev_t;
                    tlimit = false;
                }
            }

            // ---- advance: Horner of the 7 variables, double-length time ----
            double nv = 0.;
            bool finl = true;
            if (sub8 < 7) {
                const double *c = W + sub8 * L;
                double acc = c[P];
#pragma unroll
                for (int o = 1; o <= P; ++o) {
                    acc = fma(acc, h, c[P - o]);
                }
                nv = acc;
                finl = finite_d(acc);
            }
            const dfl nt = df_add(tm, df(h));
            const bool fin = (__ballot(!finl) & omask) == 0ull && df_finite(nt);
            if (!stop_nf && !fin) {
                stop_nf = true;
            }
            const bool step_ok = alive && !stop_nf;
            if (alive && stop_nf) {
                status = ST_NF;
                nf_last_t = tc_prev.hi;
                alive = false;
            }
            if (step_ok) {
                tm = nt;
            }
            // The propagate callback (src/sim_propagate.cpp:710-755): the row of this substep.
            const bool record = step_ok && h != 0.;
            const dfl tc = df_sub(tm, init_time);
            if (record) {
                if (!df_finite(tc)) {
                    flags |= ST_FLAG_BAD_TCOORD;
                }
                if (ns < cap) {
                    const std::size_t rw = row0 + ns;
                    double *dx = prm.tcs_xyz + rw * static_cast<std::size_t>(XS);
                    for (int e = lo; e < XS; e += LPO) {
                        dx[e] = e < 3 * L ? W[e] : 0.;
                    }
                    double *dv = prm.tcs_vr + rw * static_cast<std::size_t>(VS);
                    for (int e = lo; e < VS; e += LPO) {
                        dv[e] = W[3 * L + e];
                    }
                    if (lo == 0) {
                        prm.tcoords[rw * 2u] = tc.hi;
                        prm.tcoords[rw * 2u + 1u] = tc.lo;
                    }
                } else if (!store) {
                    flags |= ST_FLAG_ROW_OVERFLOW;
                }
                tc_prev = tc;
                ++ns;
            }
            // The next substep starts from series that are zero but for the state.
            const double keep = lo < 7 ? (step_ok ? nv : W[lo * L]) : 1.;
            wave_lds_sync();
            clear_series();
            wave_lds_sync();
            if (lo < 7) {
                W[lo * L] = keep;
            }
            if (lo == 7) {
                W[sc.one * L] = keep;
            }
            wave_lds_sync();
            if (step_ok) {
                if (ev_found) {
                    status = ST_STE;
                    ev_time = df_sub(tm, init_time).hi;
                    alive = false;
                } else if (tlimit) {
                    alive = false;
                } else if (ns >= prm.max_substeps) {
                    flags |= ST_FLAG_MAX_SUBSTEPS;
                    alive = false;
                }
            }
        }

        if (store) {
            if (exists && lo == 0) {
                prm.tc_row[i] = row0;
                if (ns != cap) {
                    atomicOr(prm.store_err, 1);
                }
            }
            wave_lds_sync();
            continue;
        }
        if (exists && lo == 0) {
            prm.nsub[i] = ns;
            prm.status[i] = status | flags;
            prm.ev_idx[i] = ev_idx;
            prm.ev_time[i] = ev_time;
            prm.nf_last_t[i] = nf_last_t;
            prm.tc_row[i] = row0;
        }
        if (exists && lo < 6) {
            prm.final_state[static_cast<std::size_t>(i) * 6u + lo] = W[lo * L];
        }
        wave_lds_sync();
    }
}

} // namespace csc
