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
// tape is scheduled on the host into rows of up to NG mutually independent operations; per Taylor
// order and row, each 8-lane group takes one operation: its lanes split the convolution sum
// (terms j = jlo + lane, + 8, + 16), three DPP steps add the partial sums, lane 0 stores the
// coefficient. Every automatic-differentiation recurrence (product, quotient, power, square
// root, exponential, sine / cosine) has the form
//     c[n] = (a1 L1[n] + a2 L2[n] + sg * sum_j (n wa + j wb + wc) A[j] B[n - j]) * pm * (1 / n)?
// so the row code is data-driven (wave_op) and the groups of a wave never diverge. Order 0
// (the transcendental functions themselves) is evaluated once per substep, one lane per
// operation, and leaves the order-independent factor pm (1 / b0, 1 / (2 c0), a scalar ...) in LDS.
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

// Orders >= 1 of one scheduled operation, 32 bytes.
struct wave_op {
    std::uint32_t s;   // sa | sb << 16: operands of the sum, A indexed by j, B by n - j
    std::uint32_t l;   // l1 | l2 << 16: linear terms (none: slot 0 with coefficient 0)
    std::uint32_t d;   // dst | jlo << 16 | jhi_off << 20 (sum over j = jlo .. n - jhi_off)
    std::uint32_t f;   // int8 a1 | int8 a2 << 8 | int8 sg << 16 | flags << 24
    double wa, wb;     // weight of term j: n wa + j wb + wc (wc = flag)
};
enum : std::uint32_t { WF_VALID = 1, WF_TIME = 2, WF_WC = 4, WF_DIVN = 8, WF_SUM = 16 };
static_assert(sizeof(wave_op) == 32, "wave_op layout");

// Order 0 of one scheduled operation, 32 bytes: the tape operation and what it leaves in pm.
struct wave_op0 {
    std::int32_t op, dst, a, b; // (op 0: nothing to do)
    double c;
    std::int32_t pm_kind, pm_slot;
};
enum : std::int32_t { PM_ONE = 0, PM_SCALAR, PM_INV_SCALAR, PM_INV_SLOT0, PM_INV_2DST0 };
constexpr std::int32_t TP_WAVE_COS = 100;
static_assert(sizeof(wave_op0) == 32, "wave_op0 layout");

struct wave_sched {
    const wave_op *ops;   // [nrows][NG]
    const wave_op0 *ops0; // [nrows][NG]
    std::uint32_t nrows, nslots;
    std::uint32_t obj_doubles; // LDS doubles per object
    std::int32_t rhs[7];
};

template <int CTRL>
CSC_DEV double dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
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

template <int P, int NG, bool EVENTS>
__global__ __launch_bounds__(64) void k_propagate_wave(const prop_params prm, const wave_sched sc)
{
    extern __shared__ double lds[];
    constexpr int L = P + 1;
    constexpr int LPO = 8 * NG;    // lanes per object
    constexpr int OPW = 64 / LPO;  // objects per wave
    const int lane = threadIdx.x;
    const int sub = lane & 7, grp = (lane >> 3) % NG, ow = lane / LPO, lo = lane % LPO;
    const unsigned long long omask = (LPO == 64 ? ~0ull : ((1ull << LPO) - 1ull)) << (ow * LPO);
    double *W = lds + static_cast<std::size_t>(ow) * sc.obj_doubles;
    double *pm = W + static_cast<std::size_t>(sc.nslots + 1u) * L; // [nrows][NG] (behind the series and one spare slot)
    double *gs = pm + static_cast<std::size_t>(sc.nrows) * NG; // [L] event polynomial / broadcast scratch
    // The schedule of the orders >= 1 is read nrows x (P - 1) times per substep: a copy in LDS,
    // behind the objects' series (2 x 16 bytes per row and group, the same address for 8 lanes).
    uint4 *lops = reinterpret_cast<uint4 *>(lds + static_cast<std::size_t>(OPW) * sc.obj_doubles);
    for (std::uint32_t k = lane; k < sc.nrows * NG * 2u; k += 64u) {
        lops[k] = reinterpret_cast<const uint4 *>(sc.ops)[k];
    }
    wave_lds_sync();
    const uint4 *ops = lops + 2 * grp;
    const wave_op0 *ops0 = sc.ops0 + grp;
    const dfl init_time{prm.init_hi, prm.init_lo};
    const dfl t_end = df_add(init_time, df(prm.delta_t));
    constexpr int XS = tc_xyz_stride(P), VS = tc_vr_stride(P);

    for (std::uint32_t base = blockIdx.x * OPW; base < prm.n; base += gridDim.x * OPW) {
        const std::uint32_t k_obj = base + ow;
        const bool exists = k_obj < prm.n;
        const std::uint32_t kk = exists ? k_obj : prm.n - 1u;
        // Store pass (csc_propagate_store): object order[k], exact-fit rows from row_base[k], as many as pass 1 counted.
        const bool store = prm.nsub_in != nullptr;
        const std::uint32_t i = store ? (prm.order != nullptr ? prm.order[kk] : kk) : prm.first + kk;
        const double *sp = prm.state + static_cast<std::size_t>(i) * 7u;
        const double *pars = prm.pars + static_cast<std::size_t>(i) * prm.npars;
        if (lo < 6) {
            W[lo * L] = sp[lo];
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

            // ---- order 0 of every operation, and its order-independent factor ----
            for (std::uint32_t r = 0; r < sc.nrows; ++r) {
                const wave_op0 o = ops0[static_cast<std::size_t>(r) * NG];
                if (o.op != 0) {
                    double sv = o.c;
                    if (o.b <= -2) {
                        sv = pars[-2 - o.b];
                    }
                    const double a0 = W[o.a * L];
                    const double b0 = o.b >= 0 ? W[o.b * L] : sv;
                    double res;
                    switch (o.op) {
                        case TP_ADD:
                            res = a0 + b0;
                            break;
                        case TP_SUB:
                            res = a0 - b0;
                            break;
                        case TP_RSUB:
                            res = sv - a0;
                            break;
                        case TP_MUL:
                            res = a0 * b0;
                            break;
                        case TP_DIV:
                            res = a0 / b0;
                            break;
                        case TP_RDIV:
                            res = sv / a0;
                            break;
                        case TP_SQR:
                            res = a0 * a0;
                            break;
                        case TP_POW:
                            res = pow_order0(a0, o.c);
                            break;
                        case TP_SQRT:
                            res = sqrt(a0);
                            break;
                        case TP_EXP:
                            res = det_exp(a0);
                            break;
                        case TP_SINCOS:
                            res = sin(a0);
                            break;
                        case TP_WAVE_COS: // (the cosine half of a TP_SINCOS: an operation of its own here)
                            res = cos(a0);
                            break;
                        case TP_TIME:
                            res = tm.hi;
                            break;
                        default: // TP_CONST
                            res = sv;
                            break;
                    }
                    double pv = 1.;
                    switch (o.pm_kind) {
                        case PM_SCALAR:
                            pv = sv;
                            break;
                        case PM_INV_SCALAR:
                            pv = 1. / sv;
                            break;
                        case PM_INV_SLOT0:
                            pv = 1. / W[o.pm_slot * L];
                            break;
                        case PM_INV_2DST0:
                            pv = 1. / (2. * res);
                            break;
                        default:
                            break;
                    }
                    if (sub == 0) {
                        W[o.dst * L] = res;
                        pm[r * NG + grp] = pv;
                    }
                }
                wave_lds_sync();
            }
            if (lo < 7) {
                W[lo * L + 1] = W[sc.rhs[lo] * L];
            }
            wave_lds_sync();

            // ---- orders 1 .. P - 1 ----
            for (int n = 1; n < P; ++n) {
                const double dn = static_cast<double>(n), rn = 1. / dn;
                const int trips = (n + 8) >> 3;
                uint4 nx0 = ops[0], nx1 = ops[1];
                for (std::uint32_t r = 0; r < sc.nrows; ++r) {
                    // (the next row's descriptor is requested before this row's work starts)
                    const uint4 o0 = nx0, o1 = nx1;
                    const std::uint32_t rn1 = r + 1u < sc.nrows ? r + 1u : 0u;
                    nx0 = ops[rn1 * (2 * NG)];
                    nx1 = ops[rn1 * (2 * NG) + 1];
                    const std::uint32_t of = o0.w, fl = of >> 24;
                    const int sa = o0.x & 0xffffu, sb = o0.x >> 16;
                    const int l1 = o0.y & 0xffffu, l2 = o0.y >> 16;
                    const int dst = o0.z & 0xffffu, jlo = (o0.z >> 16) & 0xf, jhi = n - static_cast<int>((o0.z >> 20) & 0xf);
                    const double a1 = static_cast<double>(static_cast<std::int8_t>(of & 0xffu));
                    const double a2 = static_cast<double>(static_cast<std::int8_t>((of >> 8) & 0xffu));
                    const double sg = static_cast<double>(static_cast<std::int8_t>((of >> 16) & 0xffu));
                    const double owa = __hiloint2double(static_cast<int>(o1.y), static_cast<int>(o1.x));
                    const double owb = __hiloint2double(static_cast<int>(o1.w), static_cast<int>(o1.z));
                    // No branches below except the wave-uniform one around the sum: a row without a
                    // linear term reads slot 0 with coefficient 0, an empty place in a row writes
                    // to the spare slot behind the series, all 8 lanes store the (same) result.
                    const double l1v = W[l1 * L + n], l2v = W[l2 * L + n], pmv = pm[r * NG + grp];
                    double acc = 0.;
                    if (__any((fl & WF_SUM) != 0u)) {
                        const double wn = fma(dn, owa, (fl & WF_WC) ? 1. : 0.);
                        const double *pa = W + sa * L, *pb = W + sb * L + n;
                        for (int t = 0; t < trips; ++t) {
                            const int j = jlo + sub + 8 * t;
                            const bool ok = (fl & WF_SUM) != 0u && j <= jhi;
                            const int jj = ok ? j : jlo;
                            const double av = pa[jj], bv = pb[-jj];
                            const double w = fma(static_cast<double>(jj), owb, wn);
                            acc = ok ? fma(w * av, bv, acc) : acc;
                        }
                        acc = group8_sum(acc);
                    }
                    double res = fma(sg, acc, fma(a2, l2v, a1 * l1v)) * pmv;
                    res = (fl & WF_DIVN) ? res * rn : res;
                    res = (fl & WF_TIME) ? (n == 1 ? 1. : 0.) : res;
                    W[dst * L + n] = res;
                    wave_lds_sync();
                }
                if (lo < 7) {
                    W[lo * L + n + 1] = W[sc.rhs[lo] * L + n] * (1. / static_cast<double>(n + 1));
                }
                wave_lds_sync();
            }

            // ---- step size (Jorba-Zou; heyoka, EXTERNAL) ----
            double m0 = 0., mo = 0., mom1 = 0.;
            bool badl = false;
            if (sub < 7) {
                // (every group of the object does it: no broadcast needed)
                const double a0 = fabs(W[sub * L]), ao = fabs(W[sub * L + P]), aom1 = fabs(W[sub * L + P - 1]);
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
                for (int e = 0; e < 2; ++e) {
                    const bool on = e == 0 ? prm.has_exit != 0 : prm.has_reentry != 0;
                    if (!on) {
                        continue;
                    }
                    const double kx = e == 0 ? 1. : prm.re_k[0], ky = e == 0 ? 1. : prm.re_k[1], kz = e == 0 ? 1. : prm.re_k[2];
                    const double c0 = e == 0 ? prm.exit_c0 : prm.re_c0;
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
                    if (alive && !stop_nf && h > 0.) {
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
            if (sub < 7) {
                const double *c = W + sub * L;
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
            wave_lds_sync();
            if (step_ok && grp == 0 && sub < 7) {
                W[sub * L] = nv;
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
