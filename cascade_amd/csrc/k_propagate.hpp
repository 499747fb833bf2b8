// cascade_amd/csrc/k_propagate.hpp
//
// K1/K2: one-object-per-lane adaptive Taylor propagation over the superstep with the
// per-chunk AABBs fused in (the coefficients never leave registers for that), and a
// second "store" pass that writes the Taylor coefficients for dense output into an
// exact-fit CSR layout once the substep counts are known.
//
// Replaces, for the built-in dynamics:
//   src/sim_propagate.cpp:134-211   init_scalar_ta / init_batch_ta
//   src/sim_propagate.cpp:650-877   batch_int_aabb integration loop + callback :710-755
//   src/sim_propagate.cpp:927-1056  scalar_int_aabb (same per lane)
//   src/sim_propagate.cpp:217-395   compute_particle_aabb (fused, pass 1)
//   heyoka taylor_adaptive(_batch)<double>::propagate_for (EXTERNAL): Taylor
//   coefficient recurrences, Jorba-Zou step size, terminal events, dfloat time.
//
// Layout. state [N][7] f64 (reference layout, sim.hpp:123). Pass 1 writes nsub[N],
// status/event records, final_state [N][6], lbs/ubs [C][N][4] f32. Pass 2 runs AFTER the
// Morton sort, one lane per particle in Morton order of chunk 0 (order[k]), and writes
// exact-fit CSR rows: row_base[k] + s holds
//   tcs_xyz [row][TC_XYZ_STRIDE]  x, y, z coefficients back to back (3 (P+1) doubles, padded
//                                 to 64 doubles = 512 B = 4 cache lines for P = 20): all the
//                                 narrow phase's fast path reads, 16-byte aligned;
//   tcs_vr  [row][4 (P+1)]        vx, vy, vz, r (conjunction states and dense output only);
//   tcoords [row][2]              substep end time (hi, lo) relative to the superstep start.
// tc_row[i] = first row of particle i. Rows of spatial neighbours are neighbours in memory,
// which is what keeps the narrow phase's gathers in L2.
//
// Roofline: pass 1 is FP64-VALU bound (~3.4 kFLOP per substep, no memory beyond 56 B
// in / 84 B out per particle); pass 2 is HBM-write bound (1176 + 16 B per substep).
#pragma once

#include "dev_math.hpp"

namespace csc
{

// Dynamics: Kepler plus a bit mask of perturbations (cascade.py/dynamics/_simple_earth.py).
enum : int {
    DYN_KEPLER = 0,
    DYN_J2 = 1,   // :169-178
    DYN_DRAG = 2, // :365-387 with the exponential density fit :16-46; par[0] = BSTAR
    // General right-hand sides (the reference's heyoka expressions, sim.hpp:213-222) lowered to a
    // tape of Taylor-series operations and interpreted order by order; see tape_compute_tcs.
    DYN_TAPE = 0x100,
    // Not a right-hand side: pass 1 writes its rows and leaves the per-chunk boxes and
    // trajectory samples to k_chunk_boxes, which computes them from the rows.
    DYN_NOBOX = 0x400,
};

// One tape operation (layout of csc_tape_op, include/cascade_hip.h). Slots 0..5 = x, y, z, vx,
// vy, vz, slot 6 = r; b is a slot (>= 0), the immediate c (-1) or runtime parameter -2 - b.
struct tape_op {
    std::int32_t op, dst, a, b;
    double c;
};
enum : std::int32_t {
    TP_ADD = 1, TP_SUB, TP_RSUB, TP_MUL, TP_DIV, TP_RDIV, TP_SQR, TP_POW, TP_SQRT, TP_EXP, TP_SINCOS, TP_TIME, TP_CONST
};

enum : int {
    ST_TIME_LIMIT = 0,
    ST_STE = 1,
    ST_NF = 2,
    ST_FLAG_BAD_AABB = 0x100,   // non-finite AABB bound (src/sim_propagate.cpp:367-383)
    ST_FLAG_BAD_TCOORD = 0x200, // non-finite time coordinate (:726-730)
    ST_FLAG_MAX_SUBSTEPS = 0x400,
    ST_FLAG_RF_WARN = 0x800,
    ST_FLAG_ROW_OVERFLOW = 0x1000, // single-pass mode: more substeps than row_cap
};

struct prop_params {
    const double *state; // [N][7]
    std::uint32_t n;
    double init_hi, init_lo;
    double delta_t, ct;
    std::uint32_t nchunks;
    double conj_thresh;
    double mu;
    double rhofac;
    // Perturbations: kj2 = 1.5 J2 mu Re^2; density fit rho(h) = sum alpha_k exp(-(h - gamma_k) beta_k).
    double kj2, Re;
    double dens_alpha[4], dens_beta[4], dens_gamma[4];
    const double *pars; // [N][npars] runtime parameters (BSTAR = pars[i * npars]); null if npars == 0
    std::uint32_t npars;
    // Tape dynamics (DYN_TAPE): the operations, the slot of each right-hand side and the
    // workspace, one block of tape_nslots * (P + 1) * 64 doubles per resident wave
    // ([slot][order][lane]: every access of a wave is one coalesced 512-byte line).
    const tape_op *tape;
    std::uint32_t tape_nops, tape_nslots;
    std::int32_t tape_rhs[7];
    double *tape_ws;
    // Terminal events (src/sim_basic.cpp:509-578): g = (kx x^2 + ky y^2) + kz z^2 - c0.
    int has_exit;
    double exit_c0;
    int has_reentry;
    double re_k[3];
    double re_c0;
    std::uint32_t max_substeps;
    // pass 1
    std::uint32_t *nsub;
    std::int32_t *status;
    std::uint32_t *ev_idx;
    double *ev_time;
    double *nf_last_t;
    double *final_state; // [N][6]
    float *lbs, *ubs;    // [C][N][4]
    float *recs;         // [C][N][16] sampled-trajectory records for the narrow-phase pre-filter (may be null)
    // Single-pass mode (row_cap > 0, pass 1 only): the Taylor rows are written during the
    // integration into the fixed slots [i * row_cap, (i + 1) * row_cap) of particle i
    // (tcs_xyz / tcs_vr / tcoords / tc_row below), and there is no second pass.
    std::uint32_t row_cap;
    // ... of particle i - row_first (a pass over a range of particles with rows of its own).
    std::uint32_t row_first;
    // Pass 1 over a range of particles (propagation sharded over GPUs): lane k handles particle
    // first + k, k < n, and the per-chunk arrays hold n_total particles per chunk.
    std::uint32_t first, n_total;
    // pass 2
    const std::uint32_t *nsub_in;   // [N] substeps pass 1 counted for particle i: the rows particle i owns
    int *store_err;                 // set when pass 2 does not repeat pass 1's substep count
    const std::uint32_t *order;     // [N] particle handled by lane k (null = identity)
    const std::uint64_t *row_base;  // [N] first CSR row of lane k
    std::uint64_t *tc_row;          // [N] out: first CSR row of particle i
    double *tcs_xyz;
    double *tcs_vr;
    double *tcoords;
};

// Row strides (doubles) of the two coefficient arrays.
constexpr int tc_xyz_stride(int P)
{
    return (3 * (P + 1) + 1) & ~1;
}
constexpr int tc_vr_stride(int P)
{
    return 4 * (P + 1);
}

// Coefficient n of a*b, ascending j, fma.
template <int N>
CSC_DEV double conv(const double *a, const double *b)
{
    double acc = 0;
#pragma unroll
    for (int j = 0; j <= N; ++j) {
        acc = fma(a[j], b[N - j], acc);
    }
    return acc;
}

// Coefficient n of a*a by the symmetric half-sum.
template <int N>
CSC_DEV double conv_sq(const double *a)
{
    if constexpr (N == 0) {
        return a[0] * a[0];
    } else {
        double acc = 0;
#pragma unroll
        for (int j = 0; 2 * j < N; ++j) {
            acc = fma(a[j], a[N - j], acc);
        }
        acc = acc + acc;
        if constexpr (N % 2 == 0) {
            acc = fma(a[N / 2], a[N / 2], acc);
        }
        return acc;
    }
}

template <int P>
struct tc_regs {
    double x[P + 1], y[P + 1], z[P + 1], vx[P + 1], vy[P + 1], vz[P + 1], r[P + 1];
};

// Per-step constants of the Taylor recurrences.
struct dyn_consts {
    double mu, inv_u0, inv_r0;
    double kj2;                                    // J2
    double Re, cdrag, inv_2s0, inv_2mv0;           // drag: cdrag = BSTAR / (0.1570 / Re)
    double alpha[4], mbeta[4], gamma[4];           // density fit (mbeta = -beta)
};

// Auxiliary series (orders 0..P-1); the members of perturbations that are not selected are
// never touched and cost nothing.
template <int P>
struct dyn_aux {
    double u[P], w[P];
    double w2[P], w5[P], z2[P], q[P], fxy[P], fz[P];                // J2
    double sq[P], ek[4][P], dens[P], vv[P], mv[P], fd[P];           // drag
};

template <int N>
CSC_DEV double conv_shift0(const double *a, const double *b, double b0)
{
    // sum_{j<=N} a[j] * (N - j == 0 ? b0 : b[N - j])
    double acc = 0;
#pragma unroll
    for (int j = 0; j <= N; ++j) {
        acc = fma(a[j], (N - j == 0) ? b0 : b[N - j], acc);
    }
    return acc;
}

// Order N of the recurrences: coefficients N of the auxiliary series, N + 1 of the state.
// Same operations in the same order as oracle/orc.hpp::compute_tcs.
template <int P, int N, int DYN>
struct dyn_order {
    static CSC_DEV void run(tc_regs<P> &t, dyn_aux<P> &a, dyn_consts &k)
    {
        double *u = a.u, *w = a.w;
        // u = x^2 + y^2 + z^2
        u[N] = (conv_sq<N>(t.x) + conv_sq<N>(t.y)) + conv_sq<N>(t.z);
        // w = u^(-3/2): w[n] = (1 / (n u0)) sum_{j<n} (n alpha - j (alpha + 1)) u[n-j] w[j]
        if constexpr (N == 0) {
            k.inv_u0 = 1. / u[0];
            w[0] = 1. / (u[0] * sqrt(u[0]));
        } else {
            double acc = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double cf = 0.5 * static_cast<double>(j) - 1.5 * static_cast<double>(N);
                acc = fma(cf * u[N - j], w[j], acc);
            }
            w[N] = (acc * k.inv_u0) * (1. / static_cast<double>(N));
        }
        double ax = -k.mu * conv<N>(t.x, w);
        double ay = -k.mu * conv<N>(t.y, w);
        double az = -k.mu * conv<N>(t.z, w);
        if constexpr ((DYN & DYN_J2) != 0) {
            // ax += kj2 x / r^5 (5 z^2 / r^2 - 1), ay likewise, az += kj2 z / r^5 (5 z^2 / r^2 - 3).
            if constexpr (N == 0) {
                a.w2[0] = k.inv_u0;
            } else {
                double acc = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    acc = fma(u[N - j], a.w2[j], acc);
                }
                a.w2[N] = -(acc * k.inv_u0);
            }
            a.w5[N] = conv<N>(w, a.w2);
            a.z2[N] = conv_sq<N>(t.z);
            a.q[N] = 5. * conv<N>(a.z2, a.w2);
            a.fxy[N] = k.kj2 * conv_shift0<N>(a.w5, a.q, a.q[0] - 1.);
            a.fz[N] = k.kj2 * conv_shift0<N>(a.w5, a.q, a.q[0] - 3.);
            ax += conv<N>(t.x, a.fxy);
            ay += conv<N>(t.y, a.fxy);
            az += conv<N>(t.z, a.fz);
        }
        if constexpr ((DYN & DYN_DRAG) != 0) {
            // a -= (rho(r - Re) / ref_density) BSTAR |v| v.
            if constexpr (N == 0) {
                a.sq[0] = sqrt(u[0]);
                k.inv_2s0 = 1. / (2. * a.sq[0]);
            } else {
                double acc = u[N];
#pragma unroll
                for (int j = 1; j < N; ++j) {
                    acc = fma(-a.sq[j], a.sq[N - j], acc);
                }
                a.sq[N] = acc * k.inv_2s0;
            }
            double dsum = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (N == 0) {
                    a.ek[e][0] = det_exp(k.mbeta[e] * ((a.sq[0] - k.Re) - k.gamma[e]));
                } else {
                    double acc = 0;
#pragma unroll
                    for (int j = 1; j <= N; ++j) {
                        acc = fma((static_cast<double>(j) * k.mbeta[e]) * a.sq[j], a.ek[e][N - j], acc);
                    }
                    a.ek[e][N] = acc * (1. / static_cast<double>(N == 0 ? 1 : N));
                }
                dsum = fma(k.alpha[e], a.ek[e][N], dsum);
            }
            a.dens[N] = dsum;
            a.vv[N] = (conv_sq<N>(t.vx) + conv_sq<N>(t.vy)) + conv_sq<N>(t.vz);
            if constexpr (N == 0) {
                a.mv[0] = sqrt(a.vv[0]);
                k.inv_2mv0 = 1. / (2. * a.mv[0]);
            } else {
                double acc = a.vv[N];
#pragma unroll
                for (int j = 1; j < N; ++j) {
                    acc = fma(-a.mv[j], a.mv[N - j], acc);
                }
                a.mv[N] = acc * k.inv_2mv0;
            }
            a.fd[N] = conv<N>(a.dens, a.mv) * k.cdrag;
            ax -= conv<N>(a.fd, t.vx);
            ay -= conv<N>(a.fd, t.vy);
            az -= conv<N>(a.fd, t.vz);
        }
        // r' = (x vx + y vy + z vz) / r
        const double s = (conv<N>(t.x, t.vx) + conv<N>(t.y, t.vy)) + conv<N>(t.z, t.vz);
        double acc = s;
#pragma unroll
        for (int j = 1; j <= N; ++j) {
            const double qr = static_cast<double>(N - j + 1) * t.r[N - j + 1];
            acc = fma(-t.r[j], qr, acc);
        }
        const double qn = acc * k.inv_r0;
        constexpr double f = 1. / static_cast<double>(N + 1);
        t.x[N + 1] = t.vx[N] * f;
        t.y[N + 1] = t.vy[N] * f;
        t.z[N + 1] = t.vz[N] * f;
        t.vx[N + 1] = ax * f;
        t.vy[N + 1] = ay * f;
        t.vz[N + 1] = az * f;
        t.r[N + 1] = qn * f;
        if constexpr (N + 1 < P) {
            dyn_order<P, N + 1, DYN>::run(t, a, k);
        }
    }
};

// Normalised Taylor coefficients (orders 1..P) of the 7 state variables; order 0 must be
// set by the caller. Kepler RHS src/sim_dynamics.cpp:36-41, r equation src/sim_basic.cpp:469,
// J2 and drag cascade.py/dynamics/_simple_earth.py:169-178, 365-387.
// a0^c at order 0 of a power series: exact arithmetic when 2c is a small integer (all the
// exponents of the reference's dynamics are -3/2, -5/2, -7/2, ...), det_pow otherwise. Same
// operations as oracle/orc.hpp::pow_order0.
CSC_DEV double pow_order0(double a0, double c)
{
    const double k2 = 2. * c;
    if (k2 == floor(k2) && fabs(k2) <= 64.) {
        const int k = static_cast<int>(k2), m = k < 0 ? -k : k;
        double r = (m % 2) ? sqrt(a0) : 1.;
        for (int i = 0; i < m / 2; ++i) {
            r = r * a0;
        }
        return k < 0 ? 1. / r : r;
    }
    return det_pow(a0, c);
}

// Per-object context of the tape interpreter.
struct tape_ctx {
    double *ws;         // this lane's column of the wave's workspace
    const double *pars; // runtime parameters of the object
    double time0;       // simulation time at the start of the substep
};

// Tape dynamics: coefficients of every slot, order by order (the automatic-differentiation
// recurrences of SURVEY appendix A; same operations in the same order as
// oracle/orc.hpp::tape_compute_tcs), then x_k^[n+1] = rhs_k^[n] / (n + 1). The tape is
// wave-uniform (scalar loads, scalar branches); the series live in the workspace
// [slot][order][lane], so every load and store of a wave is one coalesced line. Replaces the
// Taylor step function heyoka JIT-compiles from kw::dyn (src/sim_basic.cpp:553-577, EXTERNAL).
#define CSC_AT(slot, n) ws[(static_cast<std::size_t>(slot) * L + (n)) * 64u]

// Order N of every operation of the tape, then coefficient N + 1 of the state. The order is a
// template parameter: every inner loop has compile-time bounds, is fully unrolled, and its
// loads (independent of each other) are issued back to back -- one memory round trip per
// operation instead of one per term (2.5x on the interpreter's time).
template <int P, int N>
struct tape_order {
    static CSC_DEV void run(const prop_params &prm, const tape_ctx &tx)
    {
        constexpr int L = P + 1;
        constexpr int n = N;
        double *ws = tx.ws;
        const double dn = static_cast<double>(n);
        for (std::uint32_t k = 0; k < prm.tape_nops; ++k) {
            const tape_op o = prm.tape[k];
            const bool bs = o.b < 0;
            double sc = o.c;
            if (o.b <= -2) {
                sc = tx.pars[-2 - o.b];
            }
            double res;
            switch (o.op) {
                case TP_ADD:
                    res = bs ? (n == 0 ? CSC_AT(o.a, 0) + sc : CSC_AT(o.a, n)) : CSC_AT(o.a, n) + CSC_AT(o.b, n);
                    break;
                case TP_SUB:
                    res = bs ? (n == 0 ? CSC_AT(o.a, 0) - sc : CSC_AT(o.a, n)) : CSC_AT(o.a, n) - CSC_AT(o.b, n);
                    break;
                case TP_RSUB:
                    res = n == 0 ? sc - CSC_AT(o.a, 0) : -CSC_AT(o.a, n);
                    break;
                case TP_MUL:
                    if (bs) {
                        res = CSC_AT(o.a, n) * sc;
                    } else {
                        double acc = 0;
                        _Pragma("unroll")
                        for (int j = 0; j <= n; ++j) {
                            acc = fma(CSC_AT(o.a, j), CSC_AT(o.b, n - j), acc);
                        }
                        res = acc;
                    }
                    break;
                case TP_DIV:
                    if (bs) {
                        res = CSC_AT(o.a, n) / sc;
                    } else {
                        double acc = CSC_AT(o.a, n);
                        _Pragma("unroll")
                        for (int j = 1; j <= n; ++j) {
                            acc = fma(-CSC_AT(o.b, j), CSC_AT(o.dst, n - j), acc);
                        }
                        res = acc / CSC_AT(o.b, 0);
                    }
                    break;
                case TP_RDIV: {
                    double acc = n == 0 ? sc : 0.;
                    _Pragma("unroll")
                    for (int j = 1; j <= n; ++j) {
                        acc = fma(-CSC_AT(o.a, j), CSC_AT(o.dst, n - j), acc);
                    }
                    res = acc / CSC_AT(o.a, 0);
                    break;
                }
                case TP_SQR:
                    if (n == 0) {
                        const double a0 = CSC_AT(o.a, 0);
                        res = a0 * a0;
                    } else {
                        double acc = 0;
                        _Pragma("unroll")
                        for (int j = 0; 2 * j < n; ++j) {
                            acc = fma(CSC_AT(o.a, j), CSC_AT(o.a, n - j), acc);
                        }
                        acc = acc + acc;
                        if (n % 2 == 0) {
                            const double am = CSC_AT(o.a, n / 2);
                            acc = fma(am, am, acc);
                        }
                        res = acc;
                    }
                    break;
                case TP_POW:
                    if (n == 0) {
                        res = pow_order0(CSC_AT(o.a, 0), o.c);
                    } else {
                        double acc = 0;
                        _Pragma("unroll")
                        for (int j = 0; j < n; ++j) {
                            const double cf = dn * o.c - static_cast<double>(j) * (o.c + 1.);
                            acc = fma(cf * CSC_AT(o.a, n - j), CSC_AT(o.dst, j), acc);
                        }
                        res = acc / (dn * CSC_AT(o.a, 0));
                    }
                    break;
                case TP_SQRT:
                    if (n == 0) {
                        res = sqrt(CSC_AT(o.a, 0));
                    } else {
                        double acc = CSC_AT(o.a, n);
                        _Pragma("unroll")
                        for (int j = 1; j < n; ++j) {
                            acc = fma(-CSC_AT(o.dst, j), CSC_AT(o.dst, n - j), acc);
                        }
                        res = acc / (2. * CSC_AT(o.dst, 0));
                    }
                    break;
                case TP_EXP:
                    if (n == 0) {
                        res = det_exp(CSC_AT(o.a, 0));
                    } else {
                        double acc = 0;
                        _Pragma("unroll")
                        for (int j = 1; j <= n; ++j) {
                            acc = fma(static_cast<double>(j) * CSC_AT(o.a, j), CSC_AT(o.dst, n - j), acc);
                        }
                        res = acc / dn;
                    }
                    break;
                case TP_SINCOS: {
                    double cres;
                    if (n == 0) {
                        const double a0 = CSC_AT(o.a, 0);
                        res = sin(a0);
                        cres = cos(a0);
                    } else {
                        double as = 0, ac = 0;
                        _Pragma("unroll")
                        for (int j = 1; j <= n; ++j) {
                            const double ja = static_cast<double>(j) * CSC_AT(o.a, j);
                            as = fma(ja, CSC_AT(o.dst + 1, n - j), as);
                            ac = fma(ja, CSC_AT(o.dst, n - j), ac);
                        }
                        res = as / dn;
                        cres = -(ac / dn);
                    }
                    CSC_AT(o.dst + 1, n) = cres;
                    break;
                }
                case TP_TIME:
                    res = n == 0 ? tx.time0 : (n == 1 ? 1. : 0.);
                    break;
                default: // TP_CONST
                    res = n == 0 ? sc : 0.;
                    break;
            }
            CSC_AT(o.dst, n) = res;
        }
        const double f = 1. / static_cast<double>(n + 1);
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            CSC_AT(c, n + 1) = CSC_AT(prm.tape_rhs[c], n) * f;
        }
        if constexpr (N + 1 < P) {
            tape_order<P, N + 1>::run(prm, tx);
        }
    }
};

template <int P>
CSC_DEV void tape_compute_tcs(tc_regs<P> &t, const prop_params &prm, const tape_ctx &tx)
{
    constexpr int L = P + 1;
    double *ws = tx.ws;
    CSC_AT(0, 0) = t.x[0];
    CSC_AT(1, 0) = t.y[0];
    CSC_AT(2, 0) = t.z[0];
    CSC_AT(3, 0) = t.vx[0];
    CSC_AT(4, 0) = t.vy[0];
    CSC_AT(5, 0) = t.vz[0];
    CSC_AT(6, 0) = t.r[0];
    tape_order<P, 0>::run(prm, tx);
#pragma unroll
    for (int n = 1; n <= P; ++n) {
        t.x[n] = CSC_AT(0, n);
        t.y[n] = CSC_AT(1, n);
        t.z[n] = CSC_AT(2, n);
        t.vx[n] = CSC_AT(3, n);
        t.vy[n] = CSC_AT(4, n);
        t.vz[n] = CSC_AT(5, n);
        t.r[n] = CSC_AT(6, n);
    }
#undef CSC_AT
}

template <int P, int DYN>
CSC_DEV void compute_tcs(tc_regs<P> &t, const prop_params &prm, double bstar, const tape_ctx &tx)
{
    if constexpr ((DYN & DYN_TAPE) != 0) {
        tape_compute_tcs<P>(t, prm, tx);
        return;
    }
    dyn_aux<P> a;
    dyn_consts k;
    k.mu = prm.mu;
    k.inv_u0 = 0.;
    k.inv_r0 = 1. / t.r[0];
    k.kj2 = prm.kj2;
    k.Re = prm.Re;
    k.cdrag = bstar / (0.1570 / prm.Re);
    k.inv_2s0 = k.inv_2mv0 = 0.;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        k.alpha[e] = prm.dens_alpha[e];
        k.mbeta[e] = -prm.dens_beta[e];
        k.gamma[e] = prm.dens_gamma[e];
    }
    dyn_order<P, 0, DYN>::run(t, a, k);
}

template <int P>
CSC_DEV double horner_fma(const double *t, double h)
{
    double acc = t[P];
#pragma unroll
    for (int o = 1; o <= P; ++o) {
        acc = fma(acc, h, t[P - o]);
    }
    return acc;
}

template <int P, int N>
struct ev_poly_order {
    static CSC_DEV void run(const tc_regs<P> &t, double *g, double kx, double ky, double kz)
    {
        g[N] = (kx * conv_sq<N>(t.x) + ky * conv_sq<N>(t.y)) + kz * conv_sq<N>(t.z);
        if constexpr (N < P) {
            ev_poly_order<P, N + 1>::run(t, g, kx, ky, kz);
        }
    }
};

constexpr double ct_binom(int n, int k)
{
    double r = 1.;
    for (int i = 1; i <= k; ++i) {
        r = r * static_cast<double>(n - k + i) / static_cast<double>(i);
    }
    return r;
}

// First step of the root isolation, inline: true when the event polynomial g has NO root in
// [0, h) by Descartes' rule -- the rescaled polynomial does not vanish at 0 and its reversed,
// translated-by-1 form has no sign change. Same operations in the same order as the first
// node of isolate_and_refine (poly_rescale, poly_rtscc; src/sim_narrow_phase.cpp:300, 329-345),
// so the decision is the one the full search would take; it keeps the out-of-line search (and
// the register traffic around its call) for the steps that really have a candidate root. The
// interval test above fails for every object that moves radially towards the event surface
// during the step, i.e. for a few lanes of almost every wave of an eccentric population.
template <int P>
CSC_DEV bool descartes_no_root(const double *g, double h)
{
    double rs[P + 1];
    double cur_f = 1;
#pragma unroll
    for (int i = 0; i <= P; ++i) {
        rs[i] = cur_f * g[i];
        cur_f *= h;
    }
    if (rs[0] == 0) {
        return false;
    }
    int count = 0, last = 0;
#pragma unroll
    for (int i = 0; i <= P; ++i) {
        // out1[k] = rs[P - k]; out2[i] = sum_{k >= i} C(k, i) out1[k], ascending k
        double acc = rs[P - i];
#pragma unroll
        for (int k = i + 1; k <= P; ++k) {
            acc += ct_binom(k, i) * rs[P - k];
        }
        const int sg = sgn(acc);
        if (sg != 0) {
            if (last != 0 && sg != last) {
                ++count;
            }
            last = sg;
        }
    }
    return count == 0;
}

// Slow path of terminal-event detection: full root isolation on g over [0, h).
// Returns the earliest accepted root or a negative value if there is none.
template <int P>
CSC_DEV double event_first_root(const double *g_in, double h, int direction, int *warn)
{
    // (The runtime-order search on purpose: this out-of-line call sits in the middle of the
    // integration loop with every Taylor series live in registers, and the register-resident
    // isolate_and_refine_n needs so many registers that the compiler spills SGPRs into VGPR lanes
    // inside the callee -- lanes of the callers' registers that belong to particles which are
    // not taking part in the call. Seen as a substep counter reset in a neighbouring lane.)
    double g[P + 1];
    for (int i = 0; i <= P; ++i) {
        g[i] = g_in[i];
    }
    double roots[P + 1];
    int n = isolate_and_refine(g, P, h, direction, roots, P + 1, warn);
    if (n < 0) {
        n = -1 - n;
    }
    double best = -1.;
    for (int k = 0; k < n && k <= P; ++k) {
        if (best < 0. || roots[k] < best) {
            best = roots[k];
        }
    }
    return best;
}

template <int P, bool EVENTS, bool STORE, int DYN>
CSC_DEV void propagate_object(const prop_params &prm, const std::uint32_t lane_k, double *ws)
{
    if (lane_k >= prm.n) {
        return;
    }
    std::uint32_t i = prm.first + lane_k;
    if constexpr (STORE) {
        if (prm.order != nullptr) {
            i = prm.order[lane_k];
        }
    }

    tc_regs<P> t;
    const double *sp = prm.state + static_cast<std::size_t>(i) * 7u;
    t.x[0] = sp[0];
    t.y[0] = sp[1];
    t.z[0] = sp[2];
    t.vx[0] = sp[3];
    t.vy[0] = sp[4];
    t.vz[0] = sp[5];
    const double p_radius = sp[6];
    double bstar = 0.;
    if constexpr ((DYN & DYN_DRAG) != 0) {
        bstar = prm.pars[static_cast<std::size_t>(i) * prm.npars];
    }
    tape_ctx tx{ws, prm.pars + static_cast<std::size_t>(i) * prm.npars, 0.};
    // src/sim_propagate.cpp:160
    t.r[0] = sqrt(t.x[0] * t.x[0] + t.y[0] * t.y[0] + t.z[0] * t.z[0]);

    const dfl init_time{prm.init_hi, prm.init_lo};
    const dfl t_end = df_add(init_time, df(prm.delta_t));
    dfl tm = init_time;
    dfl tc_prev = df(0.); // end of the previous substep relative to init_time

    int status = ST_TIME_LIMIT;
    int flags = 0;
    std::uint32_t ns = 0;
    std::uint32_t ev_idx = 0;
    double ev_time = 0., nf_last_t = 0.;

    // Fused AABB state (pass 1): running box of the chunk being visited.
    const std::uint32_t C = prm.nchunks;
    std::uint32_t kc = 0;   // first chunk that may still overlap the coming substeps
    std::uint32_t kbox = ~0u; // chunk whose box is being accumulated
    float blo[4], bhi[4];
    const float finf = __builtin_huge_valf();
    const double infl = dmax(p_radius, prm.conj_thresh / 2);

    // Sampled-trajectory record of the chunk being visited (see np_prefilter in k_narrow.hpp):
    // positions at the 5 times cb + s (ce - cb) / 4, and a bound on |p''| over the chunk.
    float smp[15];
    std::uint32_t smask = 0;
    double sbound = 0., sabs = 0.;
    constexpr bool BOXES = !STORE && (DYN & DYN_NOBOX) == 0;
    const bool want_recs = BOXES && prm.recs != nullptr;
    if constexpr (BOXES) {
        if (want_recs) {
            // Chunks that are never completed keep an invalid record (dev = +inf: never excluded).
            for (std::uint32_t k = 0; k < C; ++k) {
                float4 *pr = reinterpret_cast<float4 *>(prm.recs) + (static_cast<std::size_t>(k) * prm.n_total + i) * 4u;
                pr[3] = make_float4(0.f, 0.f, 0.f, finf);
            }
        }
    }
    if constexpr (BOXES) {
        // Boxes of chunks that no substep reaches stay at +-inf (:238-244).
        for (std::uint32_t k = 0; k < C; ++k) {
            float4 *plo = reinterpret_cast<float4 *>(prm.lbs) + (static_cast<std::size_t>(k) * prm.n_total + i);
            float4 *phi = reinterpret_cast<float4 *>(prm.ubs) + (static_cast<std::size_t>(k) * prm.n_total + i);
            *plo = make_float4(finf, finf, finf, finf);
            *phi = make_float4(-finf, -finf, -finf, -finf);
        }
    }
    auto flush_box = [&]() {
        if constexpr (BOXES) {
            if (kbox != ~0u) {
                float4 *plo = reinterpret_cast<float4 *>(prm.lbs) + (static_cast<std::size_t>(kbox) * prm.n_total + i);
                float4 *phi = reinterpret_cast<float4 *>(prm.ubs) + (static_cast<std::size_t>(kbox) * prm.n_total + i);
                *plo = make_float4(blo[0], blo[1], blo[2], blo[3]);
                *phi = make_float4(bhi[0], bhi[1], bhi[2], bhi[3]);
                if (want_recs) {
                    const double cb = prm.ct * static_cast<double>(kbox);
                    const double ce = (kbox == C - 1u) ? prm.delta_t : prm.ct * static_cast<double>(kbox + 1u);
                    const double dts = (ce - cb) * 0.25;
                    // |p(t) - chord(t)| <= dt^2 / 8 * max |p''| on each quarter of the chunk, plus the
                    // particle's radius, plus a margin for the single-precision storage and
                    // arithmetic of the filter.
                    double dev = sbound * dts * dts * 0.125 + p_radius + 1e-6 * sabs;
                    float devf = (smask == 0x1fu && finite_d(dev)) ? nextafterf(static_cast<float>(dev), finf) : finf;
                    float4 *pr = reinterpret_cast<float4 *>(prm.recs) + (static_cast<std::size_t>(kbox) * prm.n_total + i) * 4u;
                    pr[0] = make_float4(smp[0], smp[1], smp[2], smp[3]);
                    pr[1] = make_float4(smp[4], smp[5], smp[6], smp[7]);
                    pr[2] = make_float4(smp[8], smp[9], smp[10], smp[11]);
                    pr[3] = make_float4(smp[12], smp[13], smp[14], devf);
                }
            }
        }
    };

    std::size_t row = 0;
    std::uint32_t ns_owned = 0; // pass 2: the rows [row, row + ns_owned) belong to this particle
    if constexpr (STORE) {
        row = prm.row_base[lane_k];
        prm.tc_row[i] = row;
        ns_owned = prm.nsub_in[i];
    } else {
        if (prm.row_cap != 0u) {
            row = static_cast<std::size_t>(i - prm.row_first) * prm.row_cap;
            prm.tc_row[i] = row;
        }
    }
    // One dense-output row: the callback of src/sim_propagate.cpp:710-755.
    auto write_row = [&](std::size_t r, const dfl &tc) {
        constexpr int XS = tc_xyz_stride(P), VS = tc_vr_stride(P);
        double2 *dxyz = reinterpret_cast<double2 *>(prm.tcs_xyz + r * static_cast<std::size_t>(XS));
        const double *axyz[3] = {t.x, t.y, t.z};
#pragma unroll
        for (int e = 0; e < XS; e += 2) {
            const double v0 = axyz[e / (P + 1)][e % (P + 1)];
            const double v1 = (e + 1 < 3 * (P + 1)) ? axyz[(e + 1) / (P + 1)][(e + 1) % (P + 1)] : 0.;
            dxyz[e / 2] = make_double2(v0, v1);
        }
        double2 *dvr = reinterpret_cast<double2 *>(prm.tcs_vr + r * static_cast<std::size_t>(VS));
        const double *avr[4] = {t.vx, t.vy, t.vz, t.r};
#pragma unroll
        for (int e = 0; e < VS; e += 2) {
            dvr[e / 2] = make_double2(avr[e / (P + 1)][e % (P + 1)], avr[(e + 1) / (P + 1)][(e + 1) % (P + 1)]);
        }
        prm.tcoords[r * 2u] = tc.hi;
        prm.tcoords[r * 2u + 1u] = tc.lo;
    };

    while (true) {
        const dfl rem = df_sub(t_end, tm);
        const double max_dt = rem.hi > 0. ? rem.hi : 0.;

        tx.time0 = tm.hi;
        compute_tcs<P, DYN>(t, prm, bstar, tx);

        // Jorba-Zou step size (heyoka, EXTERNAL).
        double m0 = 0, mo = 0, mom1 = 0;
        bool bad = false;
        {
            const double *arrs[7] = {t.x, t.y, t.z, t.vx, t.vy, t.vz, t.r};
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const double a0 = fabs(arrs[c][0]), ao = fabs(arrs[c][P]), aom1 = fabs(arrs[c][P - 1]);
                bad = bad || !finite_d(a0) || !finite_d(ao) || !finite_d(aom1);
                m0 = dmax(m0, a0);
                mo = dmax(mo, ao);
                mom1 = dmax(mom1, aom1);
            }
        }
        if (bad) {
            status = ST_NF;
            nf_last_t = tc_prev.hi;
            break;
        }
        const double num = m0 < 1. ? 1. : m0;
        const double rho_o = det_pow(num / mo, 1. / static_cast<double>(P));
        const double rho_om1 = det_pow(num / mom1, 1. / static_cast<double>(P - 1));
        double h = dmin(rho_o, rho_om1) * prm.rhofac;
        bool tlimit = false;
        if (!(h < max_dt)) {
            if (h != h) {
                status = ST_NF;
                nf_last_t = tc_prev.hi;
                break;
            }
            h = max_dt;
            tlimit = true;
        }

        // Terminal events in [0, h).
        bool ev_found = false;
        if constexpr (EVENTS) {
            if (h > 0.) {
                double ev_t = 0.;
                std::uint32_t idx = 0;
                // Where the object can be during the step, from the series of r: r0 -+ sum |r_n| h^n.
                // The event polynomials (700 multiply-adds each) are built only where that bound
                // reaches the surface, with a margin of 1e-9 for what separates the series of r
                // from the norm of the truncated position polynomials; a NaN keeps the search.
                double rlo, rhi;
                {
                    double sabs = fabs(t.r[P]);
#pragma unroll
                    for (int o = P - 1; o >= 1; --o) {
                        sabs = fma(sabs, h, fabs(t.r[o]));
                    }
                    const double dev = sabs * h;
                    rlo = t.r[0] - dev;
                    rhi = t.r[0] + dev;
                }
                const double re_kmin = dmin(prm.re_k[0], dmin(prm.re_k[1], prm.re_k[2]));
                const bool may_exit = !(rhi * rhi < prm.exit_c0 * (1. - 1e-9));
                const bool may_reenter = !(rlo > 0. && re_kmin > 0. && re_kmin * (rlo * rlo) > prm.re_c0 * (1. + 1e-9));
                if (prm.has_exit) {
                    double g[P + 1];
                    if (may_exit) {
                        ev_poly_order<P, 0>::run(t, g, 1., 1., 1.);
                        g[0] -= prm.exit_c0;
                    }
                    if (may_exit && !fex_excluded<P>(g, h) && !descartes_no_root<P>(g, h)) {
                        const double rt = event_first_root<P>(g, h, 1, &flags);
                        if (rt >= 0.) {
                            ev_found = true;
                            ev_t = rt;
                            ev_idx = 0;
                        }
                    }
                    idx = 1;
                }
                if (prm.has_reentry) {
                    double g[P + 1];
                    if (may_reenter) {
                        ev_poly_order<P, 0>::run(t, g, prm.re_k[0], prm.re_k[1], prm.re_k[2]);
                        g[0] -= prm.re_c0;
                    }
                    if (may_reenter && !fex_excluded<P>(g, h) && !descartes_no_root<P>(g, h)) {
                        const double rt = event_first_root<P>(g, h, -1, &flags);
                        if (rt >= 0. && (!ev_found || rt < ev_t)) {
                            ev_found = true;
                            ev_t = rt;
                            ev_idx = idx;
                        }
                    }
                }
                if (ev_found) {
                    h = ev_t;
                    tlimit = false;
                }
            }
        }

        // Advance state and time.
        const double nx = horner_fma<P>(t.x, h), ny = horner_fma<P>(t.y, h), nz = horner_fma<P>(t.z, h);
        const double nvx = horner_fma<P>(t.vx, h), nvy = horner_fma<P>(t.vy, h), nvz = horner_fma<P>(t.vz, h);
        const double nr = horner_fma<P>(t.r, h);
        const dfl nt = df_add(tm, df(h));
        const bool fin = finite_d(nx) && finite_d(ny) && finite_d(nz) && finite_d(nvx) && finite_d(nvy)
                         && finite_d(nvz) && finite_d(nr) && df_finite(nt);
        if (!fin) {
            status = ST_NF;
            nf_last_t = tc_prev.hi;
            break;
        }
        tm = nt;

        // The propagate callback (:710-755): record the substep unless h == 0.
        if (h != 0.) {
            const dfl tc = df_sub(tm, init_time);
            if (!df_finite(tc)) {
                flags |= ST_FLAG_BAD_TCOORD;
            }
            if constexpr (STORE) {
                // Never beyond the rows pass 1 counted for this particle (they are followed by the
                // next particle's): a pass that disagrees with pass 1 is reported below.
                if (ns < ns_owned) {
                    write_row(row + ns, tc);
                }
            }
            if constexpr (!STORE) {
                if (prm.row_cap != 0u) {
                    if (ns < prm.row_cap) {
                        write_row(row + ns, tc);
                    } else {
                        flags |= ST_FLAG_ROW_OVERFLOW;
                    }
                }
            }
            if constexpr (BOXES) {
                // Fused compute_particle_aabb (:217-395): chunks k with cb_k < tc and ce_k > tc_prev.
                while (kc < C) {
                    const double ce = (kc == C - 1u) ? prm.delta_t : prm.ct * static_cast<double>(kc + 1u);
                    if (df_lt(tc_prev, df(ce))) {
                        break;
                    }
                    ++kc;
                }
                for (std::uint32_t k = kc; k < C; ++k) {
                    const double cb = prm.ct * static_cast<double>(k);
                    if (!df_lt(df(cb), tc)) {
                        break;
                    }
                    const double ce = (k == C - 1u) ? prm.delta_t : prm.ct * static_cast<double>(k + 1u);
                    const dfl ev_lb = df_max(df(cb), tc_prev);
                    const dfl ev_ub = df_min(df(ce), tc);
                    const ivl hint{df_sub(ev_lb, tc_prev).hi, df_sub(ev_ub, tc_prev).hi};
                    if (k != kbox) {
                        flush_box();
                        kbox = k;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            blo[q] = finf;
                            bhi[q] = -finf;
                        }
                        smask = 0;
                        sbound = 0.;
                        sabs = 0.;
                    }
                    if (want_recs) {
                        const double hs = df_sub(tc, tc_prev).hi;
                        // Bound on |p''| over this substep: sum k (k - 1) |c_k| h^(k - 2), per coordinate.
                        double b2 = 0.;
                        {
                            const double *arrs3[3] = {t.x, t.y, t.z};
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                double acc = 0.;
#pragma unroll
                                for (int o = P; o >= 2; --o) {
                                    acc = fma(acc, hs, static_cast<double>(o * (o - 1)) * fabs(arrs3[q][o]));
                                }
                                b2 = fma(acc, acc, b2);
                            }
                        }
                        sbound = dmax(sbound, sqrt(b2));
                        const double dts = (ce - cb) * 0.25;
#pragma unroll
                        for (int sidx = 0; sidx < 5; ++sidx) {
                            const dfl ts = df(sidx == 4 ? ce : cb + dts * static_cast<double>(sidx));
                            if (!df_lt(ts, tc_prev) && !df_lt(tc, ts)) {
                                const double tau = df_sub(ts, tc_prev).hi;
                                const double sx = horner_fma<P>(t.x, tau), sy = horner_fma<P>(t.y, tau), sz = horner_fma<P>(t.z, tau);
                                smp[3 * sidx] = static_cast<float>(sx);
                                smp[3 * sidx + 1] = static_cast<float>(sy);
                                smp[3 * sidx + 2] = static_cast<float>(sz);
                                smask |= 1u << sidx;
                                sabs = dmax(sabs, dmax(fabs(sx), dmax(fabs(sy), fabs(sz))));
                            }
                        }
                    }
                    const double *arrs[4] = {t.x, t.y, t.z, t.r};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // (0 <= hint.lo < hint.hi: ev_lb >= tc_prev, and cb < tc, ce > tc_prev, cb < ce, tc_prev < tc)
                        ivl v = iv_poly_eval_nonneg<P>(arrs[q], hint);
                        v.lo -= infl;
                        v.hi += infl;
                        const float fl = nextafterf(static_cast<float>(v.lo), -finf);
                        const float fu = nextafterf(static_cast<float>(v.hi), finf);
                        if (!(fabsf(fl) <= FLT_MAX) || !(fabsf(fu) <= FLT_MAX)) {
                            flags |= ST_FLAG_BAD_AABB;
                        }
                        blo[q] = fminf_std(blo[q], fl);
                        bhi[q] = fmaxf_std(bhi[q], fu);
                    }
                }
            }
            tc_prev = tc;
            ++ns;
        }

        t.x[0] = nx;
        t.y[0] = ny;
        t.z[0] = nz;
        t.vx[0] = nvx;
        t.vy[0] = nvy;
        t.vz[0] = nvz;
        t.r[0] = nr;

        if (ev_found) {
            status = ST_STE;
            ev_time = df_sub(tm, init_time).hi;
            break;
        }
        if (tlimit) {
            break;
        }
        if (ns >= prm.max_substeps) {
            flags |= ST_FLAG_MAX_SUBSTEPS;
            break;
        }
    }

    if constexpr (STORE) {
        if (ns != ns_owned) {
            atomicOr(prm.store_err, 1);
        }
    }
    if constexpr (!STORE) {
        flush_box();
        prm.nsub[i] = ns;
        prm.status[i] = status | flags;
        prm.ev_idx[i] = ev_idx;
        prm.ev_time[i] = ev_time;
        prm.nf_last_t[i] = nf_last_t;
        double *fs = prm.final_state + static_cast<std::size_t>(i) * 6u;
        fs[0] = t.x[0];
        fs[1] = t.y[0];
        fs[2] = t.z[0];
        fs[3] = t.vx[0];
        fs[4] = t.vy[0];
        fs[5] = t.vz[0];
    }
}

// The per-chunk boxes and trajectory samples of pass 1 from its rows (k_propagate<.., DYN_NOBOX>
// wrote them into the fixed slots [i * row_cap, (i + 1) * row_cap) of particle i): the SAME
// operations on the same coefficients as the fused code of propagate_object, so the same bits
// (tests/test_gpu_parity.py::test_chunk_boxes_from_rows), but one lane per (particle, chunk): no
// lane waits for another particle's substeps, and the propagation kernel (512 registers, one wave
// per SIMD) no longer spends half of its instructions here. The lanes of a particle are
// neighbours (chunk fastest), so the chunks of one substep read its row together.
// A particle with more substeps than row_cap has no rows for the later ones: its boxes are
// incomplete and the caller, who sees ST_FLAG_ROW_OVERFLOW, integrates again without rows.
template <int P>
__global__ __launch_bounds__(256, 2) void k_chunk_boxes(const prop_params prm)
{
    const std::uint32_t C = prm.nchunks;
    const std::uint64_t g = static_cast<std::uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const std::uint32_t li = static_cast<std::uint32_t>(g / C), k = static_cast<std::uint32_t>(g % C);
    if (li >= prm.n) {
        return;
    }
    const std::uint32_t i = prm.first + li;
    const std::uint32_t ns_all = prm.nsub[i], ns = ns_all < prm.row_cap ? ns_all : prm.row_cap;
    const double p_radius = prm.state[static_cast<std::size_t>(i) * 7u + 6u];
    const double infl = dmax(p_radius, prm.conj_thresh / 2);
    const float finf = __builtin_huge_valf();
    const bool want_recs = prm.recs != nullptr;
    const double cb = prm.ct * static_cast<double>(k);
    const double ce = (k == C - 1u) ? prm.delta_t : prm.ct * static_cast<double>(k + 1u);
    constexpr int XS = tc_xyz_stride(P), VS = tc_vr_stride(P);
    // r starts at double 3 (P + 1) of the second array; read from the even index below it.
    constexpr int R0 = (3 * (P + 1)) & ~1, RN = VS - R0;

    float blo[4] = {finf, finf, finf, finf}, bhi[4] = {-finf, -finf, -finf, -finf};
    float smp[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) {
        smp[q] = 0.f;
    }
    std::uint32_t smask = 0;
    double sbound = 0., sabs = 0.;
    bool visited = false, bad = false;
    dfl tc_prev = df(0.);
    const std::size_t row0 = static_cast<std::size_t>(i - prm.row_first) * prm.row_cap;
    // The fused loop visits chunk k in the substeps with ce > tc_prev and cb < tc: the substeps from
    // the first one that ends after cb, for as long as they begin before ce. Finding the first one
    // takes the end times only; the loop below then runs once or twice per lane with (nearly) all
    // lanes of the wave in it (a loop over all substeps with the test inside ran at 40 % of the lanes).
    std::uint32_t s0 = 0;
    for (; s0 < ns; ++s0) {
        const double2 tcv = *reinterpret_cast<const double2 *>(prm.tcoords + (row0 + s0) * 2u);
        const dfl tc{tcv.x, tcv.y};
        if (df_lt(df(cb), tc)) {
            break;
        }
        tc_prev = tc;
    }
    for (std::uint32_t s = s0; s < ns && df_lt(tc_prev, df(ce)); ++s) {
        const double2 tcv = *reinterpret_cast<const double2 *>(prm.tcoords + (row0 + s) * 2u);
        const dfl tc{tcv.x, tcv.y};
        {
            visited = true;
            double cxyz[XS], cr[RN];
            const double2 *pxyz = reinterpret_cast<const double2 *>(prm.tcs_xyz + (row0 + s) * static_cast<std::size_t>(XS));
#pragma unroll
            for (int e = 0; e < XS / 2; ++e) {
                const double2 v = pxyz[e];
                cxyz[2 * e] = v.x;
                cxyz[2 * e + 1] = v.y;
            }
            const double2 *pr2 = reinterpret_cast<const double2 *>(prm.tcs_vr + (row0 + s) * static_cast<std::size_t>(VS) + R0);
#pragma unroll
            for (int e = 0; e < RN / 2; ++e) {
                const double2 v = pr2[e];
                cr[2 * e] = v.x;
                cr[2 * e + 1] = v.y;
            }
            const double *tx = cxyz, *ty = cxyz + (P + 1), *tz = cxyz + 2 * (P + 1), *tr = cr + (3 * (P + 1) - R0);
            const dfl ev_lb = df_max(df(cb), tc_prev);
            const dfl ev_ub = df_min(df(ce), tc);
            const ivl hint{df_sub(ev_lb, tc_prev).hi, df_sub(ev_ub, tc_prev).hi};
            if (want_recs) {
                const double hs = df_sub(tc, tc_prev).hi;
                double b2 = 0.;
                {
                    const double *arrs3[3] = {tx, ty, tz};
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        double acc = 0.;
#pragma unroll
                        for (int o = P; o >= 2; --o) {
                            acc = fma(acc, hs, static_cast<double>(o * (o - 1)) * fabs(arrs3[q][o]));
                        }
                        b2 = fma(acc, acc, b2);
                    }
                }
                sbound = dmax(sbound, sqrt(b2));
                const double dts = (ce - cb) * 0.25;
#pragma unroll
                for (int sidx = 0; sidx < 5; ++sidx) {
                    const dfl ts = df(sidx == 4 ? ce : cb + dts * static_cast<double>(sidx));
                    if (!df_lt(ts, tc_prev) && !df_lt(tc, ts)) {
                        const double tau = df_sub(ts, tc_prev).hi;
                        const double sx = horner_fma<P>(tx, tau), sy = horner_fma<P>(ty, tau), sz = horner_fma<P>(tz, tau);
                        smp[3 * sidx] = static_cast<float>(sx);
                        smp[3 * sidx + 1] = static_cast<float>(sy);
                        smp[3 * sidx + 2] = static_cast<float>(sz);
                        smask |= 1u << sidx;
                        sabs = dmax(sabs, dmax(fabs(sx), dmax(fabs(sy), fabs(sz))));
                    }
                }
            }
            const double *arrs[4] = {tx, ty, tz, tr};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ivl v = iv_poly_eval_nonneg<P>(arrs[q], hint);
                v.lo -= infl;
                v.hi += infl;
                const float fl = nextafterf(static_cast<float>(v.lo), -finf);
                const float fu = nextafterf(static_cast<float>(v.hi), finf);
                if (!(fabsf(fl) <= FLT_MAX) || !(fabsf(fu) <= FLT_MAX)) {
                    bad = true;
                }
                blo[q] = fminf_std(blo[q], fl);
                bhi[q] = fmaxf_std(bhi[q], fu);
            }
        }
        tc_prev = tc;
    }
    if (bad) {
        atomicOr(prm.status + i, ST_FLAG_BAD_AABB);
    }
    const std::size_t slot = static_cast<std::size_t>(k) * prm.n_total + i;
    reinterpret_cast<float4 *>(prm.lbs)[slot] = make_float4(blo[0], blo[1], blo[2], blo[3]);
    reinterpret_cast<float4 *>(prm.ubs)[slot] = make_float4(bhi[0], bhi[1], bhi[2], bhi[3]);
    if (want_recs) {
        float devf = finf; // (a chunk that is not completed keeps an invalid record: never excluded)
        if (visited) {
            const double dts = (ce - cb) * 0.25;
            const double dev = sbound * dts * dts * 0.125 + p_radius + 1e-6 * sabs;
            devf = (smask == 0x1fu && finite_d(dev)) ? nextafterf(static_cast<float>(dev), finf) : finf;
        }
        float4 *pr = reinterpret_cast<float4 *>(prm.recs) + slot * 4u;
        pr[0] = make_float4(smp[0], smp[1], smp[2], smp[3]);
        pr[1] = make_float4(smp[4], smp[5], smp[6], smp[7]);
        pr[2] = make_float4(smp[8], smp[9], smp[10], smp[11]);
        pr[3] = make_float4(smp[12], smp[13], smp[14], devf);
    }
}

// One object per lane. The built-in dynamics run one object per launched lane; tape dynamics
// loop over the groups of 64 objects with a bounded grid, so that the workspace is sized by the
// resident waves and not by the population.
template <int P, bool EVENTS, bool STORE, int DYN>
__global__ __launch_bounds__(64) void k_propagate(const prop_params prm)
{
    if constexpr ((DYN & DYN_TAPE) != 0) {
        double *ws = prm.tape_ws + static_cast<std::size_t>(blockIdx.x) * prm.tape_nslots * (P + 1) * 64u + threadIdx.x;
        for (std::uint32_t g = blockIdx.x; static_cast<std::uint64_t>(g) * 64u < prm.n; g += gridDim.x) {
            propagate_object<P, EVENTS, STORE, DYN>(prm, g * 64u + threadIdx.x, ws);
        }
    } else {
        propagate_object<P, EVENTS, STORE, DYN>(prm, blockIdx.x * blockDim.x + threadIdx.x, nullptr);
    }
}

} // namespace csc
