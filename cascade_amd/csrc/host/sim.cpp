// cascade_amd/csrc/host/sim.cpp
//
// Host side of cascade::sim (include/cascade/sim.hpp): argument validation, state ownership,
// outcome logic, conjunction bookkeeping, propagate_until -- the parts of the reference's
// src/sim_basic.cpp and src/sim_propagate.cpp that stay on the CPU -- and the orchestration of
// the per-superstep pipeline through the C ABI of libcascade_hip.so (include/cascade_hip.h).
// Plain C++17 compiled with g++: nothing here touches HIP directly.
//
//   g++ -O2 -std=c++17 -fPIC -shared -Iinclude cascade_amd/csrc/host/sim.cpp \
//       -Lcascade_amd -lcascade_hip -Wl,-rpath,'$ORIGIN' -o cascade_amd/libcascade.so

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <ctime>
#include <limits>
#include <map>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <future>
#include <thread>

#include <sys/mman.h>

#include <cascade/logging.hpp>
#include <cascade/sim.hpp>
#include <cascade_hip.h>

namespace cascade
{

namespace dynamics
{

namespace
{

constexpr double pi = 3.14159265358979323846;
constexpr double deg = pi / 180.;

struct state_vars {
    expression x{"x"}, y{"y"}, z{"z"}, vx{"vx"}, vy{"vy"}, vz{"vz"};
};

// x' = vx, ..., v' = -mu r / |r|^3 (src/sim_dynamics.cpp:25-42).
std::vector<dyn_t::eq_t> kepler_eqs(double mu)
{
    const state_vars v;
    const auto w = expression(-mu) * pow(v.x * v.x + v.y * v.y + v.z * v.z, -1.5);
    return {{v.x, v.vx}, {v.y, v.vy}, {v.z, v.vz}, {v.vx, v.x * w}, {v.vy, v.y * w}, {v.vz, v.z * w}};
}

void add_acc(std::vector<dyn_t::eq_t> &eqs, const expression &ax, const expression &ay, const expression &az)
{
    eqs[3].second += ax;
    eqs[4].second += ay;
    eqs[5].second += az;
}

// Acceleration -grad(u) of a potential-like term u.
void add_potential(std::vector<dyn_t::eq_t> &eqs, const expression &u, const state_vars &v)
{
    add_acc(eqs, -diff(u, v.x), -diff(u, v.y), -diff(u, v.z));
}

} // namespace

dyn_t kepler(double mu)
{
    dyn_t d;
    d.kind = dyn_t::kepler;
    d.mu = mu;
    d.npars = 0;
    d.eqs = kepler_eqs(mu);
    return d;
}

// cascade.py/dynamics/_simple_earth.py:86-411 (constants :136-153; all in SI units).
dyn_t simple_earth(bool J2, bool J3, bool J4, bool C22S22, bool sun, bool moon, bool SRP, bool drag, bool thermonets)
{
    if (thermonets) {
        throw std::invalid_argument("simple_earth(): the thermonets density model (a neural network evaluated by "
                                    "heyoka) is not available in this build; use the exponential density fit");
    }
    const double GMe = 3.986004407799724e5 * 1e9, GMsun = 1.32712440018e11 * 1e9, GMmoon = 4.9028e3 * 1e9;
    const double Re = 6378.1363 * 1000;
    const state_vars v;
    const auto &x = v.x, &y = v.y, &z = v.z;
    const auto t = time_expr;

    dyn_t d;
    d.mu = GMe;
    d.Re = Re;
    d.eqs = kepler_eqs(GMe);
    auto &eqs = d.eqs;
    const auto r2 = x * x + y * y + z * z;
    const auto z2 = z * z;

    // Zonal harmonics :169-200: potentials u_Jn, acceleration -grad u_Jn.
    if (J2) {
        const double k = 0.1082635854e-2 * GMe * Re * Re;
        add_potential(eqs, expression(k / 2) * (3. * z2 - r2) * pow(r2, -2.5), v);
    }
    if (J3) {
        const double k = -0.2532435346e-5 * GMe * Re * Re * Re;
        add_potential(eqs, expression(k / 2) * z * (5. * z2 - 3. * r2) * pow(r2, -3.5), v);
    }
    if (J4) {
        const double k = -0.1619331205e-5 * GMe * Re * Re * Re * Re;
        add_potential(eqs, expression(k / 8) * (35. * z2 * z2 - 30. * r2 * z2 + 3. * r2 * r2) * pow(r2, -4.5), v);
    }
    // Tesseral C22 / S22 in the Earth-fixed frame :202-242.
    if (C22S22) {
        const double C22 = 2.43914352398e-6, S22 = -1.40016683654e-6;
        const double theta_g = deg * 280.4606, nu_e = deg * 4.178074622024230e-3;
        const double k15 = GMe * Re * Re * std::sqrt(15.);
        const auto ang = theta_g + nu_e * t;
        const auto c = cos(ang), s = sin(ang);
        const auto X = x * c + y * s, Y = y * c - x * s;
        const auto &Z = z;
        const auto q7 = pow(r2, -3.5), q5 = pow(r2, -2.5);
        const auto c1 = expression(5 * k15 * C22 / 2) * q7, c2 = expression(k15 * C22) * q5;
        const auto s1 = expression(5 * k15 * S22) * q7, s2 = expression(k15 * S22) * q5;
        const auto dd = Y * Y - X * X;
        const auto FX = (c1 * X * dd + c2 * X) + (s2 * Y - s1 * (X * X) * Y);
        const auto FY = (c1 * Y * dd - c2 * Y) + (s2 * X - s1 * X * (Y * Y));
        const auto FZ = c1 * Z * dd - s1 * X * Y * Z;
        add_acc(eqs, FX * c - FY * s, FX * s + FY * c, FZ);
    }
    // Position of the Sun :244-258.
    expression Xo, Yo, Zo, dist_sun2;
    if (sun || SRP) {
        const double eps = deg * 23.4392911, nu_o = deg * 1.1407410259335311e-5;
        const auto lo = deg * 357.5256 + nu_o * t;
        const auto lam = deg * 282.94 + lo + deg * ((6892. / 3600) * sin(lo) + (72. / 3600) * sin(2. * lo));
        const auto ro = (149.619 - 2.499 * cos(lo) - 0.021 * cos(2. * lo)) * 1e9;
        Xo = ro * cos(lam);
        Yo = ro * sin(lam) * std::cos(eps);
        Zo = ro * sin(lam) * std::sin(eps);
        dist_sun2 = (x - Xo) * (x - Xo) + (y - Yo) * (y - Yo) + (z - Zo) * (z - Zo);
    }
    // Third-body gravity -GM ((r - R) / |r - R|^3 + R / |R|^3).
    auto third_body = [&](double GM, const expression &X, const expression &Y, const expression &Z, const expression &d2) {
        const auto a3 = pow(d2, -1.5), b3 = pow(X * X + Y * Y + Z * Z, -1.5);
        add_acc(eqs, expression(-GM) * ((x - X) * a3 + X * b3), expression(-GM) * ((y - Y) * a3 + Y * b3),
                expression(-GM) * ((z - Z) * a3 + Z * b3));
    };
    if (sun) {
        third_body(GMsun, Xo, Yo, Zo, dist_sun2);
    }
    if (moon) {
        // :277-363: truncated lunar theory; {coefficient, multiples of lm, llm, Fm, Dm}.
        const double eps = deg * 23.4392911;
        const double nu_o = deg * 1.1407410259335311e-5, nu_ma = deg * 1.512151961904581e-4,
                     nu_mp = deg * 1.2893925235125941e-6, nu_ms = deg * 6.128913003523574e-7;
        const auto L0 = (nu_mp + nu_ma) * t + deg * 218.31617;
        const auto lm = nu_ma * t + deg * 134.96292;
        const auto llm = nu_o * t + deg * 357.5256;
        const auto Fm = (nu_mp + nu_ma + nu_ms) * t + deg * 93.27283;
        const auto Dm = (nu_mp + nu_ma - nu_o) * t + deg * 297.85027;
        struct term {
            double c;
            int lm, llm, Fm, Dm;
        };
        auto arg = [&](const term &k) {
            return static_cast<double>(k.lm) * lm + static_cast<double>(k.llm) * llm + static_cast<double>(k.Fm) * Fm
                   + static_cast<double>(k.Dm) * Dm;
        };
        const term rterms[] = {{-20905, 1, 0, 0, 0}, {-3699, -1, 0, 0, 2}, {-2956, 0, 0, 0, 2}, {-570, 2, 0, 0, 0},
                               {246, 2, 0, 0, -2},   {-205, 0, 1, 0, -2},  {-171, 1, 0, 0, 2},  {-152, 1, 1, 0, -2}};
        expression rm(385000.);
        for (const auto &k : rterms) {
            rm += k.c * cos(arg(k));
        }
        rm = rm * 1000.;
        const term lterms[] = {{22640, 1, 0, 0, 0},  {769, 2, 0, 0, 0},    {-4856, 1, 0, 0, -2}, {2370, 0, 0, 0, 2},
                               {-668, 0, 1, 0, 0},   {-412, 0, 0, 2, 0},   {-212, 2, 0, 0, -2},  {-206, 1, 1, 0, -2},
                               {192, 1, 0, 0, 2},    {-165, 0, 1, 0, -2},  {148, 1, -1, 0, 0},   {-125, 0, 0, 0, 1},
                               {-110, 1, 1, 0, 0},   {-55, 0, 0, 2, -2}};
        expression lsum(0.);
        for (const auto &k : lterms) {
            lsum += (k.c / 3600) * sin(arg(k));
        }
        const auto lam = L0 + deg * lsum;
        const term bterms[] = {{-526, 0, 0, 1, -2}, {44, 1, 0, 1, -2}, {-31, -1, 0, 1, -2}, {-25, -2, 0, 1, 0},
                               {-23, 0, 1, 1, -2},  {21, -1, 0, 1, 0}, {11, 0, -1, 1, -2}};
        expression bsum = (18520. / 3600)
                          * sin(Fm + lam - L0 + deg * ((412. / 3600) * sin(2. * Fm) + (541. / 3600) * sin(llm)));
        for (const auto &k : bterms) {
            bsum += (k.c / 3600) * sin(arg(k));
        }
        const auto Bm = deg * bsum;
        const auto cB = cos(Bm), sB = sin(Bm), cl = cos(lam), sl = sin(lam);
        const auto Xm = cB * cl * rm;
        const auto Ym = std::cos(eps) * cB * sl * rm - std::sin(eps) * sB * rm;
        const auto Zm = std::cos(eps) * sB * rm + std::sin(eps) * cB * sl * rm;
        third_body(GMmoon, Xm, Ym, Zm, (x - Xm) * (x - Xm) + (y - Ym) * (y - Ym) + (z - Zm) * (z - Zm));
    }
    if (drag) {
        // :365-387 with the density fit :16-46 (beta per metre, gamma in metres); par[0] = BSTAR.
        d.npars = 1;
        d.dens_alpha = {1.01709935e-06, 7.86443375e-01, 7.50341883e-09, 8.63934252e-14};
        d.dens_beta = {4.63822910e-02 / 1000, 1.86080048e-01 / 1000, 2.48667176e-02 / 1000, 4.81080852e-03 / 1000};
        d.dens_gamma = {5.01594516e00 * 1000, 2.28429809e01 * 1000, 4.27957829e00 * 1000, 1.56291673e-01 * 1000};
        const auto h = sqrt(r2) - Re;
        expression rho(0.);
        for (int k = 0; k < 4; ++k) {
            rho += d.dens_alpha[k] * exp((d.dens_gamma[k] - h) * d.dens_beta[k]);
        }
        const auto vmag = sqrt(v.vx * v.vx + v.vy * v.vy + v.vz * v.vz);
        const auto fd = rho * (1. / (0.1570 / Re)) * par(0) * vmag;
        add_acc(eqs, -(fd * v.vx), -(fd * v.vy), -(fd * v.vz));
    }
    if (SRP) {
        // :390-409: par index of Cr A/m follows BSTAR when drag is on.
        const std::uint32_t ip = drag ? 1u : 0u;
        d.npars = ip + 1u;
        const double P_SI = 4.56e-3 / 1000, AU = 1.49619e8 * 1000;
        const auto k = par(ip) * (P_SI * AU * AU) * pow(dist_sun2, -1.5);
        add_acc(eqs, k * (x - Xo), k * (y - Yo), k * (z - Zo));
    }
    // The hand-unrolled kernels cover Kepler, + J2 and + J2 + drag; the rest runs from the tape.
    const bool others = J3 || J4 || C22S22 || sun || moon || SRP;
    if (others || (drag && !J2)) {
        d.kind = dyn_t::tape;
    } else {
        d.kind = dyn_t::kepler | (J2 ? dyn_t::j2 : 0) | (drag ? dyn_t::drag : 0);
        if (J2) {
            d.J2 = 0.1082635854e-2;
        }
    }
    return d;
}

} // namespace dynamics

namespace
{

// Shortest round-trip text of a double, like fmt's "{}" (the reference formats its messages
// with fmt: "... but it is -1 instead", "... [1, 2, 0]").
std::string fmt_d(double x)
{
    if (std::isnan(x)) {
        return "nan";
    }
    if (std::isinf(x)) {
        return x < 0 ? "-inf" : "inf";
    }
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof(buf), x);
    return std::string(buf, r.ptr);
}

template <typename V>
std::string fmt_list(const V &v)
{
    std::string s = "[";
    bool first = true;
    for (const auto &x : v) {
        if (!first) {
            s += ", ";
        }
        first = false;
        if constexpr (std::is_floating_point_v<std::decay_t<decltype(x)>>) {
            s += fmt_d(x);
        } else {
            s += std::to_string(x);
        }
    }
    return s + "]";
}

// heyoka::detail::dfloat<double> (EXTERNAL): normalised hi + lo.
struct dfloat {
    double hi = 0, lo = 0;
};

dfloat df_add(dfloat x, dfloat y)
{
    const double s = x.hi + y.hi;
    const double z = s - x.hi;
    double e = (x.hi - (s - z)) + (y.hi - z);
    e += x.lo + y.lo;
    const double h = s + e;
    return dfloat{h, e - (h - s)};
}

dfloat df_sub(dfloat x, dfloat y)
{
    return df_add(x, dfloat{-y.hi, -y.lo});
}

bool df_lt(dfloat a, dfloat b)
{
    return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
}

// CASCADE_HOST_TIMERS=1: wall time of the phases of step() on stderr (measurements only).
struct phase_timer {
    bool on = std::getenv("CASCADE_HOST_TIMERS") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (on) {
            const auto t1 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[step] %-12s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    }
};

// The "cascade" logger (include/cascade/logging.hpp): levels as spdlog's (trace = 0 ... critical = 5),
// default info, one line per message on stdout in spdlog's default format.
std::atomic<int> g_log_level{2};
enum : int { LOG_TRACE = 0, LOG_DEBUG = 1, LOG_INFO = 2, LOG_WARN = 3 };

bool log_on(int lvl)
{
    return lvl >= g_log_level.load(std::memory_order_relaxed);
}

void log_line(int lvl, const std::string &msg)
{
    if (!log_on(lvl)) {
        return;
    }
    static const char *names[] = {"trace", "debug", "info", "warning", "error", "critical"};
    const auto now = std::chrono::system_clock::now();
    const std::time_t tt = std::chrono::system_clock::to_time_t(now);
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count() % 1000;
    std::tm tmv{};
    localtime_r(&tt, &tmv);
    char stamp[32];
    std::strftime(stamp, sizeof(stamp), "%Y-%m-%d %H:%M:%S", &tmv);
    std::fprintf(stdout, "[%s.%03d] [cascade] [%s] %s\n", stamp, static_cast<int>(ms), names[lvl], msg.c_str());
    std::fflush(stdout);
}

// spdlog::stopwatch of the reference's phases; `sync` finishes the queued device work first, so
// that a phase's line holds the time of that phase (only called at trace level).
struct stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    [[nodiscard]] std::string elapsed() const
    {
        return fmt_d(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

// First touch of a large stretch of freshly reserved (not yet constructed) vector storage by
// several threads: a conjunction-heavy superstep appends gigabytes of records, and a single
// thread faulting the pages in while it copies runs at a fraction of the memory bandwidth.
void prefault(void *p, std::size_t bytes)
{
    constexpr std::size_t page = 4096;
    if (bytes < (64u << 20)) {
        return;
    }
    auto lo = (reinterpret_cast<std::uintptr_t>(p) + page - 1) / page * page;
    const auto hi = (reinterpret_cast<std::uintptr_t>(p) + bytes) / page * page;
    if (hi <= lo) {
        return;
    }
    (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const std::size_t per = ((hi - lo) / page + nt - 1) / nt * page;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        const auto b = lo + t * per, e = std::min<std::uintptr_t>(hi, b + per);
        if (b < e) {
            th.emplace_back([b, e] {
                for (auto a = b; a < e; a += page) {
                    *reinterpret_cast<volatile char *>(a) = 0;
                }
            });
        }
    }
    for (auto &t : th) {
        t.join();
    }
}

bool df_eq(dfloat a, dfloat b)
{
    return a.hi == b.hi && a.lo == b.lo;
}

std::uint32_t order_from_tol(double tol)
{
    // heyoka's order rule (EXTERNAL): max(2, ceil(-ln(tol) / 2 + 1)).
    const double o = std::ceil(-std::log(tol) / 2 + 1);
    return o < 2 ? 2u : static_cast<std::uint32_t>(o);
}

} // namespace

// Per-step scratch (the role of sim::sim_data, include/cascade/detail/sim_data.hpp:90-318):
// here the device context and the grow-only device buffers.
struct sim::sim_data {
    dynamics::dyn_t dyn;
    tape_t tape; // the lowered equations when dyn.kind == tape
    double tol = 0;
    bool high_accuracy = false, compact_mode = false;
    std::uint32_t order = 0;
    int device = 0;
    dfloat time;
    bool host_sync = true, np_prefilter = true;
    // With host_sync off the device copies of state and pars are the authoritative ones between
    // steps: dev_rows = number of particles they hold, -1 = stale (never uploaded, or the host
    // vectors were replaced by set_new_state_pars).
    std::int64_t dev_rows = -1;
    // Inside propagate_until() nobody can touch the host vectors between the supersteps: the
    // state goes up once (loop_uploaded) and comes down once, when the loop ends.
    bool in_loop = false, loop_uploaded = false;
    // Rows on demand (plain Kepler with the pre-filter): propagation writes no Taylor rows; after
    // the pre-filter only the particles of the candidate pairs are integrated again to produce
    // theirs (CASCADE_ROWS_ON_DEMAND=0 switches it off). sparse_likely: the candidates of the last
    // superstep involved few particles; rows_full: every particle has its rows (what
    // dense_propagate needs); want_full_rows: propagate_until announces the superstep whose rows
    // it is going to evaluate.
    bool rows_on_demand = std::getenv("CASCADE_ROWS_ON_DEMAND") == nullptr || std::atoi(std::getenv("CASCADE_ROWS_ON_DEMAND")) != 0;
    bool sparse_likely = true, rows_full = true, want_full_rows = false, state_prestep = false;
    csc_prop_cfg last_cfg{};
    // Dense-output rows in one pass: 0 never, 1 for everything but plain Kepler (default), 2 always
    // (environment CASCADE_SINGLE_PASS, for measurements); row_hint = most substeps of one particle
    // in the previous superstep (valid for the same ct and n_par_ct).
    int single_pass_mode = 2;
    std::uint32_t row_hint = 0, row_hint_nchunks = 0;
    double row_hint_ct = 0;
    std::uint32_t shard_rank = 0, shard_n = 1;
    exchange_t exchange;
    gather_t gather; // propagation sharded too when set
    bool shard_chunks = true;                 // ... and, with n_par_ct >= nranks, the chunks dealt to the processes
    alltoall_t alltoall;                      // ... receiving the records of the own chunks only
    std::uint32_t chunk_c0 = 0, chunk_cl = 0; // chunk block of the last step on this process
    // Hashes of the host state / parameter vectors as of the last transfer (csc_upload /
    // csc_download): when they still match at the next step(), the caller has not edited the
    // arrays and the device copies are current.
    std::uint64_t state_hash[2] = {0, 0}, pars_hash[2] = {0, 0};
    bool hash_valid = false;
    bool wave_env = true, fuse_env = true; // CASCADE_WAVE / CASCADE_BP_FUSE != 0
    bool verify = false;                  // CASCADE_VERIFY=1: the reference's debug verifiers after every stage
    std::uint64_t step_counter = 0; // step() calls so far (identical on the processes of a sharded simulation)
    std::vector<std::pair<std::uint64_t, std::uint64_t>> conj_segments;
    step_stats stats;
    std::vector<std::tuple<size_type, size_type, double>> last_coll;
    std::vector<conjunction> last_conj;
    const csc_conjunction *last_conj_view = nullptr; // in the context, until the next narrow phase
    std::uint64_t last_conj_n = 0;
    bool last_conj_on_device = false;  // ... or still on the device (csc_narrow_phase_lazy): fetched on demand
    std::uint64_t conj_pending = 0;    // records in the context's device-resident store, not yet in m_det_conj

    csc_ctx *ctx = nullptr;
    struct dbuf {
        void *p = nullptr;
        std::size_t bytes = 0;
    };
    std::map<std::string, dbuf> bufs;

    sim_data() = default;
    // A copy shares nothing on the device: it gets its own context on its first step().
    sim_data(const sim_data &o)
        : dyn(o.dyn), tape(o.tape), tol(o.tol), high_accuracy(o.high_accuracy), compact_mode(o.compact_mode), order(o.order),
          device(o.device), time(o.time), host_sync(o.host_sync), np_prefilter(o.np_prefilter),
          single_pass_mode(o.single_pass_mode),
          shard_rank(o.shard_rank), shard_n(o.shard_n), exchange(o.exchange), gather(o.gather), shard_chunks(o.shard_chunks), alltoall(o.alltoall),
          wave_env(o.wave_env), fuse_env(o.fuse_env), verify(o.verify)
    {
    }
    ~sim_data()
    {
        release();
    }
    void release()
    {
        if (ctx) {
            for (auto &kv : bufs) {
                if (kv.second.p) {
                    csc_free(ctx, kv.second.p);
                }
            }
            bufs.clear();
            csc_ctx_destroy(ctx);
            ctx = nullptr;
        }
    }
    void ensure_ctx()
    {
        if (!ctx) {
            const int rc = csc_ctx_create(device, &ctx);
            if (rc != CSC_OK) {
                ctx = nullptr;
                throw std::runtime_error(std::string("cascade: cannot create the HIP context (there is no CPU fallback): ")
                                         + csc_last_error(nullptr));
            }
            // Large conjunction lists stay on the device until somebody reads them (CASCADE_CONJ_LAZY=0: off).
            const char *e = std::getenv("CASCADE_CONJ_LAZY");
            csc_narrow_phase_lazy(ctx, (e && std::atoi(e) == 0) ? 0 : 1);
        }
    }
    void chk(int rc) const
    {
        if (rc != CSC_OK) {
            const std::string msg = csc_last_error(ctx);
            switch (rc) {
                case CSC_ERR_INVALID:
                case CSC_ERR_NONFINITE_AABB:
                case CSC_ERR_NONFINITE_TIME:
                case CSC_ERR_NO_SUBSTEPS:
                    throw std::invalid_argument(msg);
                case CSC_ERR_MAX_SUBSTEPS:
                case CSC_ERR_OVERFLOW:
                    throw std::overflow_error(msg);
                default:
                    throw std::runtime_error("libcascade_hip error " + std::to_string(rc) + ": " + msg);
            }
        }
    }
    // Grow-only named device buffer (like resize_if_needed, src/sim_propagate.cpp:561-569).
    template <typename T>
    T *buf(const char *name, std::size_t count)
    {
        auto &b = bufs[name];
        const std::size_t bytes = std::max<std::size_t>(count * sizeof(T), 256);
        if (b.bytes < bytes) {
            if (b.p) {
                chk(csc_free(ctx, b.p));
                b.p = nullptr;
                b.bytes = 0;
            }
            const std::size_t want = bytes + bytes / 4;
            chk(csc_malloc(ctx, want, &b.p));
            b.bytes = want;
        }
        return static_cast<T *>(b.p);
    }
    template <typename T>
    T *get(const char *name)
    {
        return static_cast<T *>(bufs[name].p);
    }
};

void set_logger_level_trace()
{
    g_log_level = 0;
}
void set_logger_level_debug()
{
    g_log_level = 1;
}
void set_logger_level_info()
{
    g_log_level = 2;
}
void set_logger_level_warn()
{
    g_log_level = 3;
}
void set_logger_level_err()
{
    g_log_level = 4;
}
void set_logger_level_critical()
{
    g_log_level = 5;
}

sim::sim(ptag_t, std::vector<double> state, double ct)
    : m_state(std::make_shared<std::vector<double>>(std::move(state))), m_pars(std::make_shared<std::vector<double>>()),
      m_ct(ct), m_det_conj(std::make_shared<std::vector<conjunction>>()), m_data(std::make_unique<sim_data>())
{
}

sim::sim() : sim(std::vector<double>{}, 1.) {}

sim::sim(const sim &o)
    : m_state(std::make_shared<std::vector<double>>(*o.m_state)), m_pars(std::make_shared<std::vector<double>>(*o.m_pars)),
      m_ct(o.m_ct), m_n_par_ct(o.m_n_par_ct), m_int_info(o.m_int_info), m_reentry_radius(o.m_reentry_radius),
      m_exit_radius(o.m_exit_radius), m_npars(o.m_npars), m_conj_thresh(o.m_conj_thresh),
      m_det_conj(std::make_shared<std::vector<conjunction>>(o.get_conjunctions())), m_min_coll_radius(o.m_min_coll_radius),
      m_coll_whitelist(o.m_coll_whitelist), m_conj_whitelist(o.m_conj_whitelist),
      m_data(std::make_unique<sim_data>(*o.m_data))
{
    // With host_sync off the device copy of the source is the authoritative one and its host
    // vector may be stale: the copy takes the coordinates from the device.
    const auto &od = *o.m_data;
    if (!od.host_sync && od.ctx != nullptr && od.dev_rows == static_cast<std::int64_t>(o.get_nparts()) && o.get_nparts() != 0u) {
        od.chk(csc_download(od.ctx, m_state->data(), od.bufs.at("state").p, m_state->size() * sizeof(double), nullptr));
    }
}

sim::sim(sim &&) noexcept = default;

sim &sim::operator=(const sim &o)
{
    if (this != &o) {
        *this = sim(o);
    }
    return *this;
}

sim &sim::operator=(sim &&) noexcept = default;

sim::~sim() = default;

bool sim::with_reentry_event() const
{
    if (const auto *p = std::get_if<double>(&m_reentry_radius)) {
        return *p > 0;
    }
    return true;
}

bool sim::with_exit_event() const
{
    return m_exit_radius > 0;
}

// src/sim_basic.cpp:359-615 (the heyoka integrator construction is replaced by the dynamics
// descriptor; everything else -- checks, messages, defaults -- follows the reference).
void sim::finalise_ctor(options o)
{
    if (m_state->size() % 7u != 0u) {
        throw std::invalid_argument("The size of the state vector is " + std::to_string(m_state->size())
                                    + ", which is not a multiple of 7");
    }
    if (!std::isfinite(m_ct) || m_ct <= 0) {
        throw std::invalid_argument("The collisional timestep must be finite and positive, but it is " + fmt_d(m_ct)
                                    + " instead");
    }
    set_n_par_ct(o.n_par_ct);
    set_conj_thresh(o.conj_thresh);
    set_min_coll_radius(o.min_coll_radius);
    m_coll_whitelist = std::move(o.coll_whitelist);
    m_conj_whitelist = std::move(o.conj_whitelist);

    // src/sim_basic.cpp:405-455: default dynamics, six equations, variable names, parameters.
    if (o.dyn.empty() && o.dyn.kind == dynamics::dyn_t::kepler) {
        o.dyn = dynamics::kepler(o.dyn.mu);
    }
    if (o.dyn.kind == dynamics::dyn_t::tape) {
        m_data->tape = lower_dynamics(o.dyn.eqs);
        o.dyn.npars = m_data->tape.npars;
    } else {
        if (o.dyn.kind != dynamics::dyn_t::kepler && o.dyn.kind != dynamics::dyn_t::j2
            && o.dyn.kind != (dynamics::dyn_t::j2 | dynamics::dyn_t::drag)) {
            throw std::invalid_argument("Unknown built-in dynamics kind " + std::to_string(o.dyn.kind));
        }
        if (!std::isfinite(o.dyn.mu) || o.dyn.mu <= 0) {
            throw std::invalid_argument("The mu parameter of the dynamics must be finite and positive, but it is "
                                        + fmt_d(o.dyn.mu) + " instead");
        }
        // J2 + drag keeps its equations' tape as well: below ~49 k objects the library runs it from
        // the tape's compiled form (8 objects per wave fill the GPU sooner than 64).
        if (o.dyn.kind == (dynamics::dyn_t::j2 | dynamics::dyn_t::drag) && !o.dyn.eqs.empty()) {
            m_data->tape = lower_dynamics(o.dyn.eqs);
        }
    }
    m_data->dyn = o.dyn;
    m_npars = o.dyn.npars;
    validate_pars_vector(o.pars, get_nparts());
    m_pars = std::make_shared<std::vector<double>>(std::move(o.pars));

    // src/sim_basic.cpp:471-501
    if (const auto *rr = std::get_if<double>(&o.reentry_radius)) {
        if (!std::isfinite(*rr) || *rr < 0) {
            throw std::invalid_argument("The reentry radius must be finite and non-negative, but it is " + fmt_d(*rr)
                                        + " instead");
        }
    } else {
        const auto &v = std::get<std::vector<double>>(o.reentry_radius);
        if (v.size() != 3u) {
            throw std::invalid_argument(
                "The reentry_radius argument must be either a scalar (for a spherical central body) "
                "or a vector of 3 elements (for a triaxial ellipsoid), but instead it is a vector of "
                + std::to_string(v.size()) + " element(s)");
        }
        if (std::any_of(v.begin(), v.end(), [](double x) { return !std::isfinite(x) || x <= 0; })) {
            throw std::invalid_argument(
                "A non-finite or non-positive value was detected among the 3 semiaxes of the central body: "
                + fmt_list(v));
        }
    }
    m_reentry_radius = std::move(o.reentry_radius);
    if (!std::isfinite(o.exit_radius) || o.exit_radius < 0) {
        throw std::invalid_argument("The exit radius must be finite and non-negative, but it is " + fmt_d(o.exit_radius)
                                    + " instead");
    }
    m_exit_radius = o.exit_radius;

    const double tol = o.tol == 0 ? std::numeric_limits<double>::epsilon() : o.tol;
    if (!std::isfinite(tol) || tol <= 0) {
        throw std::invalid_argument("The tolerance must be finite and positive, but it is " + fmt_d(tol) + " instead");
    }
    m_data->tol = tol;
    m_data->order = order_from_tol(tol);
    // The kernels are built for Taylor orders 6..24 (tol from ~5e-5 down to below machine epsilon;
    // src/sim_basic.cpp:553-577 hands any tol to heyoka). Order 20 -- the default tol = epsilon --
    // runs the hand-unrolled built-in dynamics; every other order runs from the dynamics tape.
    if (m_data->order < 6u || m_data->order > 24u) {
        throw std::invalid_argument("The tolerance " + fmt_d(tol) + " needs Taylor order " + std::to_string(m_data->order)
                                    + ", and the kernels are built for orders 6 to 24");
    }
    if (m_data->order != 20u && m_data->dyn.kind != dynamics::dyn_t::tape) {
        m_data->tape = lower_dynamics(m_data->dyn.eqs);
    }
    m_data->high_accuracy = o.high_accuracy;
    m_data->compact_mode = o.compact_mode;
    m_data->device = o.device;
    if (const char *e = std::getenv("CASCADE_SINGLE_PASS")) {
        m_data->single_pass_mode = std::atoi(e);
    }
    // (measurement / test switches, read when the simulation is constructed)
    m_data->wave_env = !(std::getenv("CASCADE_WAVE") && std::atoi(std::getenv("CASCADE_WAVE")) == 0);
    m_data->fuse_env = !(std::getenv("CASCADE_BP_FUSE") && std::atoi(std::getenv("CASCADE_BP_FUSE")) == 0);
    m_data->verify = std::getenv("CASCADE_VERIFY") && std::atoi(std::getenv("CASCADE_VERIFY")) != 0;

    verify_state_vector(*m_state);
}

// src/sim_basic.cpp:252-278
void sim::validate_pars_vector(std::vector<double> &pars, size_type nparts) const
{
    if (m_npars == 0u) {
        if (!pars.empty()) {
            throw std::invalid_argument("The input array of parameter values must be empty when the number of parameters "
                                        "in the dynamics is zero");
        }
    } else if (pars.empty()) {
        pars.resize(nparts * m_npars);
    } else if (pars.size() % m_npars != 0u || pars.size() / m_npars != nparts) {
        throw std::invalid_argument("The input array of parameter values must have shape (" + std::to_string(nparts) + ", "
                                    + std::to_string(m_npars) + "), but instead its flattened size is "
                                    + std::to_string(pars.size()));
    }
}

// src/sim_basic.cpp:686-760
void sim::verify_state_vector(const std::vector<double> &st) const
{
    if (st.size() % 7u != 0u) {
        throw std::invalid_argument("The size of the state vector is " + std::to_string(st.size())
                                    + ", which is not a multiple of 7");
    }
    const auto n = st.size() / 7u;
    const bool reentry = with_reentry_event(), exit_ev = with_exit_event();
    const auto *ell = std::get_if<std::vector<double>>(&m_reentry_radius);
    for (size_type i = 0; i < n; ++i) {
        const double *p = st.data() + i * 7u;
        const double x = p[0], y = p[1], z = p[2];
        if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) {
            throw std::invalid_argument("An non-finite position was detected in the state vector of the particle at index "
                                        + std::to_string(i));
        }
        if (reentry) {
            if (ell) {
                const double a = (*ell)[0], b = (*ell)[1], c = (*ell)[2];
                if (x * x / (a * a) + y * y / (b * b) + z * z / (c * c) < 1) {
                    throw std::invalid_argument("The particle at index " + std::to_string(i)
                                                + " is inside the ellipsoidal central body");
                }
            } else {
                const double r = std::get<double>(m_reentry_radius);
                if (x * x + y * y + z * z < r * r) {
                    throw std::invalid_argument("The particle at index " + std::to_string(i)
                                                + " is inside the spherical central body");
                }
            }
        }
        if (exit_ev && x * x + y * y + z * z >= m_exit_radius * m_exit_radius) {
            throw std::invalid_argument("The particle at index " + std::to_string(i) + " is outside the exit radius "
                                        + fmt_d(m_exit_radius));
        }
        if (!std::isfinite(p[3]) || !std::isfinite(p[4]) || !std::isfinite(p[5])) {
            throw std::invalid_argument("An non-finite velocity was detected in the state vector of the particle at index "
                                        + std::to_string(i));
        }
        if (!std::isfinite(p[6]) || p[6] < 0) {
            throw std::invalid_argument("An invalid particle size of " + fmt_d(p[6])
                                        + " was detected for the particle at index " + std::to_string(i));
        }
    }
}

// ---- getters / setters -------------------------------------------------------------------

double sim::get_time() const
{
    return m_data->time.hi;
}

void sim::set_time(double t)
{
    if (!std::isfinite(t)) {
        throw std::invalid_argument("Cannot set the simulation time to the non-finite value " + fmt_d(t));
    }
    m_data->time = dfloat{t, 0.};
}

double sim::get_ct() const
{
    return m_ct;
}

void sim::set_ct(double ct)
{
    if (!std::isfinite(ct) || ct <= 0) {
        throw std::invalid_argument("The collisional timestep must be finite and positive, but it is " + fmt_d(ct)
                                    + " instead");
    }
    m_ct = ct;
}

std::uint32_t sim::get_n_par_ct() const
{
    return m_n_par_ct;
}

void sim::set_n_par_ct(std::uint32_t n)
{
    if (n == 0u) {
        throw std::invalid_argument("The number of collisional timesteps to be processed in parallel cannot be zero");
    }
    m_n_par_ct = n;
}

double sim::get_tol() const
{
    return m_data->tol;
}

bool sim::get_high_accuracy() const
{
    return m_data->high_accuracy;
}

bool sim::get_compact_mode() const
{
    return m_data->compact_mode;
}

std::uint32_t sim::get_npars() const
{
    return m_npars;
}

std::variant<double, std::vector<double>> sim::get_reentry_radius() const
{
    return m_reentry_radius;
}

void sim::set_conj_thresh(double v)
{
    if (!std::isfinite(v) || v < 0) {
        throw std::invalid_argument("The conjunction threshold value " + fmt_d(v)
                                    + " is invalid: it must be finite and non-negative");
    }
    m_conj_thresh = v;
}

void sim::reset_conjunctions()
{
    if (m_data->ctx != nullptr && m_data->conj_pending != 0u) {
        m_data->chk(csc_conj_store_reset(m_data->ctx, 0));
    }
    m_data->conj_pending = 0;
    m_det_conj = std::make_shared<std::vector<conjunction>>();
    m_data->conj_segments.clear();
}

const std::vector<std::pair<std::uint64_t, std::uint64_t>> &sim::_get_conj_segments() const
{
    return m_data->conj_segments;
}

void sim::set_min_coll_radius(double v)
{
    if (std::isnan(v) || v < 0) {
        throw std::invalid_argument("The minimum collisional radius cannot be NaN or negative, but the invalid value "
                                    + fmt_d(v) + " was provided");
    }
    m_min_coll_radius = v;
}

void sim::set_coll_whitelist(whitelist_t wl)
{
    m_coll_whitelist = std::move(wl);
}

void sim::set_conj_whitelist(whitelist_t wl)
{
    m_conj_whitelist = std::move(wl);
}

// src/sim_basic.cpp:340-357
void sim::set_new_state_pars(std::vector<double> new_state, std::vector<double> new_pars)
{
    verify_state_vector(new_state);
    validate_pars_vector(new_pars, new_state.size() / 7u);
    auto st = std::make_shared<std::vector<double>>(std::move(new_state));
    auto ps = std::make_shared<std::vector<double>>(std::move(new_pars));
    m_state = std::move(st);
    m_pars = std::move(ps);
    m_data->dev_rows = -1;
    m_data->hash_valid = false;
}

// src/sim_basic.cpp:280-338. With the state resident on the device (host_sync off) the device
// copies are compacted there (csc_remove_particles); the host vectors are always edited, so
// their shapes, the sizes and the views handed out stay what the reference's are.
void sim::remove_particles(std::vector<size_type> idxs)
{
    std::sort(idxs.begin(), idxs.end());
    idxs.erase(std::unique(idxs.begin(), idxs.end()), idxs.end());
    const auto nparts = get_nparts();
    if (!idxs.empty() && idxs.back() >= nparts) {
        throw std::invalid_argument("An invalid vector of indices was passed to the function for particle removal: "
                                    + fmt_list(idxs));
    }
    if (auto &d = *m_data; !d.host_sync && d.ctx != nullptr && d.dev_rows == static_cast<std::int64_t>(nparts)) {
        const std::vector<std::uint64_t> idx64(idxs.begin(), idxs.end());
        std::uint32_t n_out = 0;
        d.chk(csc_remove_particles(d.ctx, static_cast<std::uint32_t>(nparts), idx64.data(), idx64.size(),
                                   d.get<double>("state"), m_npars, m_npars ? d.get<double>("pars") : nullptr, &n_out));
        d.dev_rows = n_out;
    }
    m_data->hash_valid = false;
    std::vector<double> new_state, new_pars;
    new_state.reserve((nparts - idxs.size()) * 7u);
    new_pars.reserve((nparts - idxs.size()) * m_npars);
    auto it = idxs.cbegin();
    for (size_type i = 0; i < nparts; ++i) {
        if (it != idxs.cend() && *it == i) {
            ++it;
            continue;
        }
        new_state.insert(new_state.end(), m_state->begin() + i * 7u, m_state->begin() + (i + 1u) * 7u);
        new_pars.insert(new_pars.end(), m_pars->begin() + i * m_npars, m_pars->begin() + (i + 1u) * m_npars);
    }
    auto st = std::make_shared<std::vector<double>>(std::move(new_state));
    auto ps = std::make_shared<std::vector<double>>(std::move(new_pars));
    m_state = std::move(st);
    m_pars = std::move(ps);
}

const dynamics::dyn_t &sim::get_dyn() const
{
    return m_data->dyn;
}

int sim::get_device() const
{
    return m_data->device;
}

void sim::set_host_sync(bool v)
{
    auto &d = *m_data;
    if (v && !d.host_sync && d.ctx != nullptr && d.dev_rows == static_cast<std::int64_t>(get_nparts())) {
        download_state(); // the host vectors become the authoritative copy again
    }
    d.host_sync = v;
}

bool sim::get_host_sync() const
{
    return m_data->host_sync;
}

void sim::set_shard(std::uint32_t rank, std::uint32_t nranks, exchange_t exchange)
{
    if (nranks == 0u || rank >= nranks) {
        throw std::invalid_argument("Invalid shard: rank " + std::to_string(rank) + " of " + std::to_string(nranks));
    }
    if (nranks > 1u && !exchange) {
        throw std::invalid_argument("A sharded simulation needs an event exchange function");
    }
    m_data->shard_rank = rank;
    m_data->shard_n = nranks;
    m_data->exchange = std::move(exchange);
}

void sim::set_shard_propagation(gather_t gather, bool chunks, alltoall_t alltoall)
{
    m_data->gather = std::move(gather);
    m_data->shard_chunks = chunks;
    m_data->alltoall = std::move(alltoall);
}

void sim::set_np_prefilter(bool v)
{
    m_data->np_prefilter = v;
}

bool sim::get_np_prefilter() const
{
    return m_data->np_prefilter;
}

const sim::step_stats &sim::get_step_stats() const
{
    return m_data->stats;
}

std::uint64_t sim::bp_nodes_visited()
{
    auto &d = *m_data;
    if (!d.ctx || d.stats.nchunks == 0u) {
        throw std::invalid_argument("bp_nodes_visited() needs a preceding step()");
    }
    std::uint64_t v = 0;
    d.chk(csc_broad_phase_stats(d.ctx, static_cast<std::uint32_t>(get_nparts()), d.stats.nchunks, d.get<float>("srt_lbs"),
                                d.get<float>("srt_ubs"), d.get<std::uint32_t>("vidx"), d.get<std::uint8_t>("active"), &v));
    return v;
}

const std::vector<std::tuple<sim::size_type, sim::size_type, double>> &sim::get_last_collisions() const
{
    return m_data->last_coll;
}

const std::vector<sim::conjunction> &sim::get_last_step_conjunctions() const
{
    auto &d = *m_data;
    if (d.last_conj_on_device && d.ctx != nullptr) {
        const csc_collision *cp = nullptr;
        std::uint64_t nc = 0, nj = 0;
        d.chk(csc_narrow_phase_view(d.ctx, &cp, &nc, &d.last_conj_view, &nj));
        d.last_conj_on_device = false;
    }
    if (d.last_conj_view != nullptr) {
        // Copied out of the context on first use.
        d.last_conj.resize(d.last_conj_n);
        if (d.last_conj_n) {
            std::memcpy(static_cast<void *>(d.last_conj.data()), d.last_conj_view, d.last_conj_n * sizeof(conjunction));
        }
        d.last_conj_view = nullptr;
    }
    return d.last_conj;
}

void *sim::_get_hip_context()
{
    m_data->ensure_ctx();
    return m_data->ctx;
}

void sim::upload_state()
{
    auto &d = *m_data;
    d.ensure_ctx();
    const auto n = get_nparts();
    double *d_state = d.buf<double>("state", n * 7u);
    d.state_hash[0] = d.state_hash[1] = d.pars_hash[0] = d.pars_hash[1] = 0;
    if (n) {
        d.chk(csc_upload(d.ctx, d_state, m_state->data(), n * 7u * sizeof(double), d.state_hash));
    }
    if (m_npars != 0u && n) {
        double *d_pars = d.buf<double>("pars", n * m_npars);
        d.chk(csc_upload(d.ctx, d_pars, m_pars->data(), n * m_npars * sizeof(double), d.pars_hash));
    }
    d.hash_valid = true;
    d.dev_rows = static_cast<std::int64_t>(n);
}

void sim::download_state()
{
    auto &d = *m_data;
    d.ensure_ctx();
    const auto n = get_nparts();
    if (d.dev_rows != static_cast<std::int64_t>(n)) {
        throw std::invalid_argument("download_state(): the device holds no state for the current particles");
    }
    // One copy of the whole [n][7] array: the sizes on the device are the host's as of the last
    // upload (only the six coordinates are outputs of a step, copy_from_final_state :1610-1630).
    if (n) {
        d.chk(csc_download(d.ctx, m_state->data(), d.buf<double>("state", n * 7u), n * 7u * sizeof(double), d.state_hash));
    }
}

// The records that supersteps left in the device-resident store (conjunction-heavy runs: 1.75 GB
// per superstep at 10M objects) join the host vector when somebody looks at it -- a NEW vector when
// the old one cannot take them, as in append_conj_data below.
void sim::_sync_conjunctions() const
{
    auto &d = *m_data;
    if (d.conj_pending == 0u || d.ctx == nullptr) {
        return;
    }
    const std::size_t old_n = m_det_conj->size(), n = d.conj_pending;
    if (old_n + n > m_det_conj->capacity()) {
        auto nv = std::make_shared<std::vector<conjunction>>();
        nv->reserve(2u * (old_n + n));
        prefault(nv->data(), (old_n + n) * sizeof(conjunction));
        nv->insert(nv->end(), m_det_conj->begin(), m_det_conj->end());
        m_det_conj = std::move(nv);
    } else {
        prefault(m_det_conj->data() + old_n, n * sizeof(conjunction));
    }
    m_det_conj->resize(old_n + n);
    static_assert(sizeof(conjunction) == sizeof(csc_conjunction), "conjunction layout");
    d.chk(csc_conj_store_download(d.ctx, reinterpret_cast<csc_conjunction *>(m_det_conj->data() + old_n)));
    d.conj_pending = 0;
}

// append_conj_data (src/sim_propagate.cpp:1644-1701): the records of this step, sorted by time,
// go behind the existing ones in a NEW vector (src/sim_narrow_phase.cpp:1000-1011: views of the
// old one stay valid).
void sim::append_conj_data(const void *recs, std::size_t n)
{
    static_assert(sizeof(conjunction) == sizeof(csc_conjunction), "conjunction layout");
    const stopwatch sw_append;
    _sync_conjunctions(); // (earlier supersteps' records come first)
    // src/sim_narrow_phase.cpp:999-1011: when the vector cannot take the new records, a NEW one
    // with twice the needed capacity replaces it (arrays handed out earlier keep the old one
    // alive); otherwise the records are appended in place.
    if (m_det_conj->size() + n > m_det_conj->capacity()) {
        auto nv = std::make_shared<std::vector<conjunction>>();
        nv->reserve(2u * (m_det_conj->size() + n));
        prefault(nv->data(), (m_det_conj->size() + n) * sizeof(conjunction));
        nv->insert(nv->end(), m_det_conj->begin(), m_det_conj->end());
        m_det_conj = std::move(nv);
    } else {
        prefault(m_det_conj->data() + m_det_conj->size(), n * sizeof(conjunction));
    }
    const auto *src = static_cast<const conjunction *>(recs);
    m_det_conj->insert(m_det_conj->end(), src, src + n);
    if (n) {
        m_data->conj_segments.emplace_back(m_data->step_counter, n);
    }
    log_line(LOG_TRACE, "Runtime for append_conj_data(): " + sw_append.elapsed() + "s");
}

// dense_propagate (src/sim_propagate.cpp:1386-1503) + copy_from_final_state.
void sim::dense_propagate(double t)
{
    auto &d = *m_data;
    const stopwatch sw_dense;
    const auto n = static_cast<std::uint32_t>(get_nparts());
    if (!d.rows_full) {
        // Rows on demand: only the particles of the candidate pairs have theirs. The state on the
        // device is still the one the superstep started from (step() calls this before it
        // advances the state; propagate_until announces the superstep it evaluates afterwards).
        if (!d.state_prestep) {
            throw std::logic_error("dense_propagate(): the Taylor rows of the last superstep are incomplete");
        }
        std::uint32_t xs = 0, vs = 0;
        d.chk(csc_tc_strides(d.order, &xs, &vs));
        const std::uint64_t tot = std::max<std::uint64_t>(d.stats.total_substeps, 1);
        // (row order: the Morton order of this process's first chunk, when it has one)
        const std::uint32_t *order = d.chunk_cl != 0u ? d.get<std::uint32_t>("vidx") + static_cast<std::size_t>(d.chunk_c0) * n : nullptr;
        d.chk(csc_propagate_store(d.ctx, &d.last_cfg, n, d.get<double>("state"), order,
                                  d.get<std::uint32_t>("nsub"), d.get<std::uint64_t>("tc_row"), d.buf<double>("tcs_xyz", tot * xs),
                                  d.buf<double>("tcs_vr", tot * vs), d.buf<double>("tcoords", tot * 2u)));
        d.rows_full = true;
    }
    d.chk(csc_dense_propagate(d.ctx, n, d.order, t, d.get<std::uint64_t>("tc_row"), d.get<std::uint32_t>("nsub"),
                              d.get<double>("tcs_xyz"), d.get<double>("tcs_vr"), d.get<double>("tcoords"),
                              d.get<double>("final_state")));
    const int rc = csc_check_flags(d.ctx);
    if (rc == CSC_ERR_NO_SUBSTEPS) {
        throw std::invalid_argument("The computation of dense_propagate() could not be performed because no timesteps "
                                    "have been taken for a particle");
    }
    d.chk(rc);
    d.chk(csc_copy_final_state(d.ctx, n, d.get<double>("final_state"), d.get<double>("state")));
    d.state_prestep = false;
    if (d.host_sync && !d.in_loop) {
        download_state();
    }
    if (log_on(LOG_TRACE)) {
        d.chk(csc_sync(d.ctx));
        log_line(LOG_TRACE, "Dense propagation time: " + sw_dense.elapsed() + "s");
    }
}

// ---- CASCADE_VERIFY=1 ------------------------------------------------------------------------
// The reference runs three verifiers inside step() in its debug builds: verify_global_aabbs
// (src/sim_propagate.cpp:1348-1380), verify_bvh_trees_parallel (src/sim_bvh.cpp:498-656) and
// verify_broad_phase_parallel (src/sim_broad_phase.cpp:221-321, up to 10 000 particles). Here they
// are a run-time mode of the product: the device outputs of every stage of a step() come back to
// the host and go through the same checks; a violation throws std::logic_error. Slow by design.
namespace
{

void vfail(const std::string &what)
{
    throw std::logic_error("CASCADE_VERIFY: " + what);
}

template <typename T>
std::vector<T> vfetch(csc_ctx *ctx, const T *d_src, std::size_t count)
{
    std::vector<T> v(count);
    if (count != 0u && csc_memcpy_d2h(ctx, v.data(), d_src, count * sizeof(T)) != CSC_OK) {
        vfail(std::string("cannot read back a device array: ") + csc_last_error(ctx));
    }
    return v;
}

struct verify_arrays {
    std::vector<float> lbs, ubs; // [C][n][4]
    std::vector<std::uint32_t> vidx;
};

// Global boxes, Morton order, sorted arrays and the trees of the chunks [0, cl).
verify_arrays verify_sort_and_trees(csc_ctx *ctx, std::uint32_t n, std::uint32_t cl, const float *d_lbs, const float *d_ubs,
                                    const float *d_glb, const float *d_gub, const std::uint64_t *d_codes, const std::uint32_t *d_vidx,
                                    const std::uint64_t *d_scodes, const float *d_slbs, const float *d_subs)
{
    const std::size_t cn = static_cast<std::size_t>(cl) * n;
    verify_arrays va{vfetch(ctx, d_lbs, cn * 4u), vfetch(ctx, d_ubs, cn * 4u), vfetch(ctx, d_vidx, cn)};
    const auto glb = vfetch(ctx, d_glb, cl * 4ull), gub = vfetch(ctx, d_gub, cl * 4ull);
    const auto codes = vfetch(ctx, d_codes, cn), scodes = vfetch(ctx, d_scodes, cn);
    const auto slbs = vfetch(ctx, d_slbs, cn * 4u), subs = vfetch(ctx, d_subs, cn * 4u);
    const float finf = std::numeric_limits<float>::infinity();
    std::vector<csc_bvh_node> tree(2ull * n);
    void *d_nodes = nullptr;
    if (csc_malloc(ctx, tree.size() * sizeof(csc_bvh_node), &d_nodes) != CSC_OK) {
        vfail("cannot allocate the tree export buffer");
    }
    struct free_nodes {
        csc_ctx *c;
        void *p;
        ~free_nodes()
        {
            csc_free(c, p);
        }
    } guard{ctx, d_nodes};
    for (std::uint32_t c = 0; c < cl; ++c) {
        const std::string at = " (chunk " + std::to_string(c) + ")";
        const float *lb = va.lbs.data() + static_cast<std::size_t>(c) * n * 4u, *ub = va.ubs.data() + static_cast<std::size_t>(c) * n * 4u;
        // verify_global_aabbs: exact equality with the serial minimum / maximum. (The upper bound is
        // checked before disc_single_coord's in-place nextafter, src/sim_propagate.cpp:448-449:
        // csc_morton_encode leaves the stored value alone.)
        float mn[4] = {finf, finf, finf, finf}, mx[4] = {-finf, -finf, -finf, -finf};
        for (std::uint32_t i = 0; i < n; ++i) {
            for (int j = 0; j < 4; ++j) {
                mn[j] = std::min(mn[j], lb[i * 4u + j]);
                mx[j] = std::max(mx[j], ub[i * 4u + j]);
            }
        }
        for (int j = 0; j < 4; ++j) {
            if (mn[j] != glb[c * 4u + j] || mx[j] != gub[c * 4u + j]) {
                vfail("the global AABB differs from the serial minimum / maximum" + at);
            }
        }
        // Sorted codes and gathered boxes (assert(is_sorted), src/sim_propagate.cpp:506; :493-504).
        const std::uint64_t *sc = scodes.data() + static_cast<std::size_t>(c) * n, *mc = codes.data() + static_cast<std::size_t>(c) * n;
        const std::uint32_t *vx = va.vidx.data() + static_cast<std::size_t>(c) * n;
        const float *slb = slbs.data() + static_cast<std::size_t>(c) * n * 4u, *sub = subs.data() + static_cast<std::size_t>(c) * n * 4u;
        std::vector<char> seen(n, 0);
        for (std::uint32_t k = 0; k < n; ++k) {
            if (vx[k] >= n || seen[vx[k]]) {
                vfail("the sort did not produce a permutation" + at);
            }
            seen[vx[k]] = 1;
            if ((k != 0u && sc[k] < sc[k - 1u]) || sc[k] != mc[vx[k]]) {
                vfail("the Morton codes are not sorted, or not those of the permuted particles" + at);
            }
            if (std::memcmp(slb + k * 4u, lb + vx[k] * 4u, 16) != 0 || std::memcmp(sub + k * 4u, ub + vx[k] * 4u, 16) != 0) {
                vfail("a sorted box is not the box of the permuted particle" + at);
            }
        }
        // verify_bvh_trees_parallel.
        std::uint64_t nn = 0;
        if (csc_bvh_export(ctx, c, static_cast<csc_bvh_node *>(d_nodes), tree.size(), &nn) != CSC_OK
            || csc_memcpy_d2h(ctx, tree.data(), d_nodes, nn * sizeof(csc_bvh_node)) != CSC_OK) {
            vfail(std::string("cannot export the tree: ") + csc_last_error(ctx) + at);
        }
        std::vector<char> pset(n, 0);
        for (std::uint64_t i = 0; i < nn; ++i) {
            const csc_bvh_node &nd = tree[i];
            bool ok = nd.end > nd.begin && nd.end <= n && (nd.left == -1) == (nd.right == -1);
            if (ok && nd.left == -1) {
                for (std::uint32_t j = nd.begin; j < nd.end && ok; ++j) {
                    ok = sc[j] == sc[nd.begin] && !pset[j];
                    pset[j] = 1;
                }
                ok = ok && nd.split_idx <= 64;
            } else if (ok) {
                const auto ul = static_cast<std::uint64_t>(nd.left), ur = static_cast<std::uint64_t>(nd.right);
                ok = nd.left > 0 && nd.right > 0 && ul > i && ul < nn && ur > i && ur < nn && tree[ul].begin == nd.begin && tree[ul].end < nd.end
                     && tree[ur].begin == tree[ul].end && tree[ur].end == nd.end && nd.split_idx <= 63;
                if (ok) {
                    const std::uint32_t sp = tree[ul].end - 1u;
                    // (first_diff_bit, src/sim_bvh.cpp:59-80: the children differ first at bit split_idx from the top)
                    ok = sp + 1u < n && sc[sp] != sc[sp + 1u] && __builtin_clzll(sc[sp] ^ sc[sp + 1u]) == nd.split_idx;
                }
            }
            if (ok && i == 0u) {
                ok = nd.parent == -1 && nd.begin == 0u && nd.end == n;
            } else if (ok) {
                const auto up = static_cast<std::uint64_t>(nd.parent);
                ok = nd.parent >= 0 && up < i && nd.begin >= tree[up].begin && nd.end <= tree[up].end
                     && (nd.begin == tree[up].begin || nd.end == tree[up].end);
            }
            ok = ok && nd.nn_level > 0u;
            if (ok) {
                float l4[4] = {finf, finf, finf, finf}, u4[4] = {-finf, -finf, -finf, -finf};
                for (std::uint32_t j = nd.begin; j < nd.end; ++j) {
                    for (int k = 0; k < 4; ++k) {
                        l4[k] = std::min(l4[k], slb[j * 4u + k]);
                        u4[k] = std::max(u4[k], sub[j * 4u + k]);
                    }
                }
                ok = std::memcmp(l4, nd.lb, 16) == 0 && std::memcmp(u4, nd.ub, 16) == 0;
            }
            if (!ok) {
                vfail("node " + std::to_string(i) + " of the BVH violates the tree invariants" + at);
            }
        }
        for (std::uint32_t j = 0; j < n; ++j) {
            if (!pset[j]) {
                vfail("a particle belongs to no leaf of the BVH" + at);
            }
        }
    }
    return va;
}

// verify_broad_phase_parallel: pairs unique with i < j; every overlapping, active pair present.
void verify_pairs(csc_ctx *ctx, std::uint32_t n, std::uint32_t cl, const verify_arrays &va, const std::uint8_t *d_active, std::uint64_t npairs)
{
    if (n > 10000u) {
        return;
    }
    const auto active = vfetch(ctx, d_active, n);
    std::vector<std::uint32_t> pairs(std::max<std::uint64_t>(npairs, 1) * 2u);
    std::uint64_t total = 0;
    for (std::uint32_t c = 0; c < cl; ++c) {
        const std::string at = " (chunk " + std::to_string(c) + ")";
        std::uint64_t cnt = 0;
        if (csc_broad_phase_get(ctx, c, pairs.data(), npairs, &cnt) != CSC_OK || cnt > npairs) {
            vfail(std::string("cannot read the pair list: ") + csc_last_error(ctx) + at);
        }
        total += cnt;
        std::vector<std::uint64_t> keys(cnt);
        for (std::uint64_t k = 0; k < cnt; ++k) {
            if (pairs[2u * k] >= pairs[2u * k + 1u] || pairs[2u * k + 1u] >= n) {
                vfail("a broad-phase pair is not (i < j)" + at);
            }
            keys[k] = static_cast<std::uint64_t>(pairs[2u * k]) << 32 | pairs[2u * k + 1u];
        }
        std::sort(keys.begin(), keys.end());
        if (std::adjacent_find(keys.begin(), keys.end()) != keys.end()) {
            vfail("a broad-phase pair is listed twice" + at);
        }
        const float *lb = va.lbs.data() + static_cast<std::size_t>(c) * n * 4u, *ub = va.ubs.data() + static_cast<std::size_t>(c) * n * 4u;
        for (std::uint32_t i = 0; i < n; ++i) {
            for (std::uint32_t j = i + 1u; j < n; ++j) {
                bool ov = (active[i] | active[j]) != 0;
                for (int k = 0; k < 4 && ov; ++k) {
                    ov = ub[i * 4u + k] >= lb[j * 4u + k] && lb[i * 4u + k] <= ub[j * 4u + k];
                }
                if (ov && !std::binary_search(keys.begin(), keys.end(), static_cast<std::uint64_t>(i) << 32 | j)) {
                    vfail("the overlapping pair (" + std::to_string(i) + ", " + std::to_string(j) + ") is missing from the broad phase" + at);
                }
            }
        }
    }
    if (total != npairs) {
        vfail("the per-chunk pair lists do not add up to the reported count");
    }
}

} // namespace

// step(): src/sim_propagate.cpp:516-1345.
outcome sim::step()
{
    auto &d = *m_data;
    const auto nparts = get_nparts();
    if (nparts == 0u) {
        throw std::invalid_argument("Cannot integrate a simulation with no particles");
    }
    if (nparts > 0xfffffff0ull) {
        throw std::overflow_error("Overflow detected: the number of particles does not fit 32 bits");
    }
    const double delta_t0 = m_ct * m_n_par_ct;
    if (!std::isfinite(delta_t0) || delta_t0 <= 0) {
        throw std::invalid_argument("An invalid superstep size of " + fmt_d(delta_t0) + " was inferred");
    }
    d.ensure_ctx();
    ++d.step_counter;

    // The host vectors are the source of truth at every step() entry (the bindings hand out
    // writable views of them, cascade.py/core.cpp:199-230). When the device still holds the state
    // of the last transfer, the step runs on it at once while a few host threads check, by hash,
    // that the caller has not edited the vectors since; nothing user-visible changes before that
    // check has been confirmed (step_impl calls confirm() before it commits or reports anything),
    // and an edit sends the step through the upload path instead.
    const bool want_host = d.host_sync && !(d.in_loop && d.loop_uploaded);
    const bool dev_valid = d.dev_rows == static_cast<std::int64_t>(nparts);
    if (want_host && dev_valid && d.hash_valid) {
        struct stale_state {
        };
        std::future<bool> same = std::async(std::launch::async, [&]() {
            std::uint64_t hs[2], hp[2];
            csc_host_hash(m_state->data(), m_state->size() * sizeof(double), hs);
            csc_host_hash(m_pars->data(), m_pars->size() * sizeof(double), hp);
            return hs[0] == d.state_hash[0] && hs[1] == d.state_hash[1] && hp[0] == d.pars_hash[0] && hp[1] == d.pars_hash[1];
        });
        bool checked = false, unchanged = false;
        const auto check = [&]() {
            if (!checked) {
                unchanged = same.get();
                checked = true;
            }
            return unchanged;
        };
        try {
            const outcome oc = step_impl(false, [&]() {
                if (!check()) {
                    throw stale_state{};
                }
            });
            d.loop_uploaded = true;
            return oc;
        } catch (const stale_state &) {
        } catch (...) {
            if (check()) {
                throw;
            }
        }
    }
    return step_impl(want_host || !dev_valid, []() {});
}

outcome sim::step_impl(bool upload, const std::function<void()> &confirm)
{
    auto &d = *m_data;
    const auto nparts = get_nparts();
    const auto n = static_cast<std::uint32_t>(nparts);
    const std::uint32_t nchunks0 = m_n_par_ct;
    const double delta_t0 = m_ct * nchunks0;
    csc_ctx *ctx = d.ctx;
    phase_timer pt;
    const bool tracing = log_on(LOG_TRACE);
    const stopwatch sw_total;
    log_line(LOG_TRACE, "---- STEP BEGIN ---");
    log_line(LOG_TRACE, "Number of chunks: " + std::to_string(nchunks0));
    stopwatch sw;

    if (upload) {
        // src/sim_propagate.cpp:250-253 (message pinned by test/invalid_state.cpp:48-49)
        for (size_type i = 0; i < nparts; ++i) {
            const double sz = (*m_state)[i * 7u + 6u];
            if (!std::isfinite(sz) || sz < 0) {
                throw std::invalid_argument("An invalid particle size of " + fmt_d(sz)
                                            + " was detected for the particle at index " + std::to_string(i));
            }
        }
        upload_state();
        d.loop_uploaded = true;
    }

    d.state_prestep = true;
    // (the view points into the context's pinned buffer, which the narrow phase below may free)
    d.last_conj_view = nullptr;
    d.last_conj_n = 0;
    d.last_conj_on_device = false;
    d.last_conj.clear();
    const std::size_t cn = static_cast<std::size_t>(nchunks0) * n;
    double *d_state = d.buf<double>("state", n * 7ull);
    auto *d_active = d.buf<std::uint8_t>("active", n);
    auto *d_nsub = d.buf<std::uint32_t>("nsub", n);
    auto *d_status = d.buf<std::int32_t>("status", n);
    auto *d_ev_idx = d.buf<std::uint32_t>("ev_idx", n);
    auto *d_ev_time = d.buf<double>("ev_time", n);
    auto *d_nf_t = d.buf<double>("nf_last_t", n);
    auto *d_final = d.buf<double>("final_state", n * 6ull);
    auto *d_lbs = d.buf<float>("lbs", cn * 4u);
    auto *d_ubs = d.buf<float>("ubs", cn * 4u);
    auto *d_glb = d.buf<float>("glb", nchunks0 * 4ull);
    auto *d_gub = d.buf<float>("gub", nchunks0 * 4ull);
    auto *d_codes = d.buf<std::uint64_t>("codes", cn);
    auto *d_vidx = d.buf<std::uint32_t>("vidx", cn);
    auto *d_scodes = d.buf<std::uint64_t>("srt_codes", cn);
    auto *d_slbs = d.buf<float>("srt_lbs", cn * 4u);
    auto *d_subs = d.buf<float>("srt_ubs", cn * 4u);
    auto *d_tc_row = d.buf<std::uint64_t>("tc_row", n);
    float *d_recs = d.np_prefilter ? d.buf<float>("recs", cn * 16u) : nullptr;
    float *d_srecs = d.np_prefilter ? d.buf<float>("srt_recs", cn * 16u) : nullptr;

    // Activity flags (:639-647).
    const bool with_conj = m_conj_thresh != 0;
    const std::vector<std::uint64_t> cwl(m_coll_whitelist.begin(), m_coll_whitelist.end());
    const std::vector<std::uint64_t> jwl(m_conj_whitelist.begin(), m_conj_whitelist.end());
    d.chk(csc_activity_flags(ctx, n, d_state, m_min_coll_radius, with_conj ? 1 : 0, cwl.data(), cwl.size(), jwl.data(),
                             jwl.size(), d_active));

    // Runtime parameters: uploaded with the state (upload_state).
    double *d_pars = m_npars != 0u ? d.buf<double>("pars", static_cast<std::size_t>(n) * m_npars) : nullptr;

    csc_prop_cfg cfg{};
    const bool from_tape = d.dyn.kind == dynamics::dyn_t::tape || d.order != 20u;
    cfg.dyn_kind = from_tape ? static_cast<int>(dynamics::dyn_t::tape) : d.dyn.kind;
    cfg.order = d.order;
    cfg.mu = d.dyn.mu;
    cfg.J2 = d.dyn.J2;
    cfg.Re = d.dyn.Re;
    for (int k = 0; k < 4; ++k) {
        cfg.dens_alpha[k] = d.dyn.dens_alpha[k];
        cfg.dens_beta[k] = d.dyn.dens_beta[k];
        cfg.dens_gamma[k] = d.dyn.dens_gamma[k];
    }
    cfg.d_pars = d_pars;
    cfg.npars = m_npars;
    // (the tagged built-ins bring their tape along too: small populations run from its compiled form)
    if (from_tape || !d.tape.ops.empty()) {
        static_assert(sizeof(tape_op) == sizeof(csc_tape_op), "tape layout");
        cfg.tape = reinterpret_cast<const csc_tape_op *>(d.tape.ops.data());
        cfg.tape_nops = static_cast<std::uint32_t>(d.tape.ops.size());
        cfg.tape_nslots = d.tape.nslots;
        for (int k = 0; k < 7; ++k) {
            cfg.tape_rhs[k] = d.tape.rhs[k];
        }
    }
    cfg.rhofac = std::exp(-0.7 / (d.order - 1)) / (std::exp(1.) * std::exp(1.));
    cfg.init_time_hi = d.time.hi;
    cfg.init_time_lo = d.time.lo;
    cfg.delta_t = delta_t0;
    cfg.ct = m_ct;
    cfg.nchunks = nchunks0;
    cfg.max_substeps = 0;
    cfg.conj_thresh = m_conj_thresh;
    cfg.has_exit = with_exit_event() ? 1 : 0;
    cfg.exit_c0 = m_exit_radius * m_exit_radius;
    cfg.has_reentry = with_reentry_event() ? 1 : 0;
    if (const auto *ell = std::get_if<std::vector<double>>(&m_reentry_radius)) {
        const double a = (*ell)[0], b = (*ell)[1], c = (*ell)[2];
        cfg.reentry_k[0] = (b * c) * (b * c);
        cfg.reentry_k[1] = (a * c) * (a * c);
        cfg.reentry_k[2] = (a * b) * (a * b);
        cfg.reentry_c0 = a * a * b * b * c * c;
    } else {
        const double r = std::get<double>(m_reentry_radius);
        cfg.reentry_k[0] = cfg.reentry_k[1] = cfg.reentry_k[2] = 1.;
        cfg.reentry_c0 = r * r;
    }

    // Dense-output rows (the propagate callback, :710-755). Plain Kepler is cheap enough to be
    // integrated twice: pass 1 for the boxes, pass 2 (after the sort) writes exact-fit rows in
    // Morton order. Every other right-hand side is integrated ONCE, writing its rows on the way
    // into fixed slots of row_cap rows per particle, row_cap following the previous superstep; a
    // particle that needs more falls back on the second pass for this superstep.
    std::uint32_t xs = 0, vs = 0;
    d.chk(csc_tc_strides(d.order, &xs, &vs));
    if (d.row_hint_ct != m_ct || d.row_hint_nchunks != nchunks0) {
        d.row_hint = 0;
    }
    // Propagation sharded over the processes: pass 1 without rows on this process's range, then
    // one all-gather; the rows are produced on demand whatever the dynamics.
    const bool shard_prop = d.shard_n > 1u && static_cast<bool>(d.gather) && d.np_prefilter;
    // ... and the chunks dealt to the processes when there is at least one for each.
    const bool chunk_mode = shard_prop && d.shard_chunks && nchunks0 >= d.shard_n;
    const bool sparse = shard_prop
                        || (d.rows_on_demand && d.np_prefilter && d.dyn.kind == dynamics::dyn_t::kepler && !from_tape && d.sparse_likely
                            && !d.want_full_rows);
    std::uint32_t row_cap = 0;
    // The several-lanes-per-object kernel (k_propagate_wave: tape dynamics at order 20; the
    // hand-unrolled Kepler / J2 / J2 + drag kernels are faster than it) always writes rows: in the first superstep of a configuration the slots are a guess, and a
    // particle that needs more has the pass repeated with what it asked for.
    const bool wave_ok = d.wave_env && d.order == 20u && !d.tape.ops.empty() && csc_wave_would_run(d.ctx, &cfg, n) != 0;
    const auto rows_fit = [&](std::uint32_t cap) { return static_cast<double>(n) * cap * (xs + vs + 2u) * 8. <= 96e9; };
    if (!sparse && d.single_pass_mode != 0 && wave_ok && d.row_hint == 0u) {
        row_cap = std::max(4u, nchunks0 / 2u + 3u);
        if (!rows_fit(row_cap)) {
            row_cap = 0;
        }
    } else if (!sparse && d.single_pass_mode != 0 && (d.single_pass_mode == 2 || d.dyn.kind != dynamics::dyn_t::kepler || from_tape)
        && d.row_hint != 0u) {
        row_cap = d.row_hint + std::max(2u, d.row_hint / 4u);
        if (const char *e = std::getenv("CASCADE_ROW_CAP")) {
            row_cap = std::max(1, std::atoi(e)); // (tests: slots that overflow)
        }
        if (!rows_fit(row_cap)) {
            row_cap = 0; // would not fit comfortably: two passes
        }
    }
    // Propagation of this process's range and the exchange of the per-particle records; rcap != 0:
    // through the several-lanes-per-object kernel, with rows of the range kept inside the context.
    const auto run_sharded = [&](std::uint32_t rcap) {
        const std::uint32_t per = (n + d.shard_n - 1u) / d.shard_n;
        const std::uint32_t first = std::min(n, d.shard_rank * per), count = std::min(per, n - first);
        if (rcap != 0u) {
            d.chk(csc_propagate_range_rows(ctx, &cfg, n, first, count, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t,
                                           d_final, d_lbs, d_ubs, d_recs, rcap));
        } else {
            d.chk(csc_propagate_range(ctx, &cfg, n, first, count, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final,
                                      d_lbs, d_ubs, d_recs));
        }
        if (chunk_mode && d.alltoall) {
            // Every process needs the per-chunk records of its own block of chunks only.
            std::vector<std::size_t> seg(d.shard_n);
            std::size_t total = 0;
            std::uint32_t my_c0 = 0, my_cl = 0;
            for (std::uint32_t r = 0; r < d.shard_n; ++r) {
                std::uint32_t bf = 0, bc = 0;
                d.chk(csc_shard_block(nchunks0, d.shard_n, r, &bf, &bc));
                seg[r] = static_cast<std::size_t>(per) * csc_shard_record_bytes(bc);
                total += seg[r];
                if (r == d.shard_rank) {
                    my_c0 = bf;
                    my_cl = bc;
                }
            }
            const std::size_t each = static_cast<std::size_t>(per) * csc_shard_record_bytes(my_cl);
            auto *d_send = d.buf<unsigned char>("shard_send", total);
            auto *d_recv = d.buf<unsigned char>("shard_recv", each * d.shard_n);
            d.chk(csc_shard_pack_blocks(ctx, n, first, count, per, nchunks0, d.shard_n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t,
                                        d_final, d_lbs, d_ubs, d_recs, d_send));
            d.chk(csc_sync(ctx));
            d.alltoall(reinterpret_cast<std::uintptr_t>(d_send), reinterpret_cast<std::uintptr_t>(d_recv), seg, each);
            d.chk(csc_shard_unpack_block(ctx, n, nchunks0, my_c0, my_cl, d_recv, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t,
                                         d_final, d_lbs, d_ubs, d_recs));
            d.stats.exchanged_bytes = each * d.shard_n;
        } else {
            const std::size_t rec = csc_shard_record_bytes(nchunks0);
            auto *d_send = d.buf<unsigned char>("shard_send", static_cast<std::size_t>(per) * rec);
            auto *d_recv = d.buf<unsigned char>("shard_recv", static_cast<std::size_t>(per) * d.shard_n * rec);
            d.chk(csc_shard_pack(ctx, n, first, count, nchunks0, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs,
                                 d_ubs, d_recs, d_send));
            d.chk(csc_sync(ctx));
            d.gather(reinterpret_cast<std::uintptr_t>(d_send), reinterpret_cast<std::uintptr_t>(d_recv),
                     static_cast<std::size_t>(per) * rec);
            d.chk(csc_shard_unpack(ctx, n, nchunks0, d_recv, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs,
                                   d_recs));
            d.stats.exchanged_bytes = static_cast<std::size_t>(per) * d.shard_n * rec;
        }
        d.stats.propagated = count;
    };
    std::uint32_t shard_rcap = 0;
    if (shard_prop && wave_ok) {
        shard_rcap = d.row_hint != 0u ? d.row_hint + std::max(2u, d.row_hint / 4u) : std::max(4u, nchunks0 / 2u + 3u);
    }
    double *d_txyz = nullptr, *d_tvr = nullptr, *d_tco = nullptr;
    if (row_cap != 0u) {
        const std::uint64_t rows = static_cast<std::uint64_t>(n) * row_cap;
        d_txyz = d.buf<double>("tcs_xyz", rows * xs);
        d_tvr = d.buf<double>("tcs_vr", rows * vs);
        d_tco = d.buf<double>("tcoords", rows * 2u);
        d.chk(csc_propagate_rows(ctx, &cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs,
                                 d_recs, row_cap, d_tc_row, d_txyz, d_tvr, d_tco));
    } else if (shard_prop) {
        run_sharded(shard_rcap);
    } else {
        d.chk(csc_propagate(ctx, &cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs, d_recs));
    }
    if (!shard_prop) {
        d.stats.propagated = n;
        d.stats.exchanged_bytes = 0;
    }
    csc_prop_summary summ{};
    d.chk(csc_propagate_summary(ctx, n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, &summ));
    if (shard_prop && shard_rcap != 0u && (summ.flags & CSC_ST_FLAG_ROW_OVERFLOW)) {
        // (the same decision on every process: the summary covers all particles)
        run_sharded(summ.max_substeps + 1u);
        d.chk(csc_propagate_summary(ctx, n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, &summ));
    }
    if (row_cap != 0u && (summ.flags & CSC_ST_FLAG_ROW_OVERFLOW) && wave_ok && rows_fit(summ.max_substeps + 1u)) {
        // (the substeps were counted in full: exactly enough slots this time)
        row_cap = summ.max_substeps + 1u;
        const std::uint64_t rows = static_cast<std::uint64_t>(n) * row_cap;
        d_txyz = d.buf<double>("tcs_xyz", rows * xs);
        d_tvr = d.buf<double>("tcs_vr", rows * vs);
        d_tco = d.buf<double>("tcoords", rows * 2u);
        d.chk(csc_propagate_rows(ctx, &cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs,
                                 d_recs, row_cap, d_tc_row, d_txyz, d_tvr, d_tco));
        d.chk(csc_propagate_summary(ctx, n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, &summ));
    }
    if (row_cap != 0u && (summ.flags & CSC_ST_FLAG_ROW_OVERFLOW)) {
        // A particle took more substeps than its row slots hold. The boxes of a pass with rows may
        // come from the rows (k_chunk_boxes) and are then incomplete for it: integrate again
        // without rows; the rows follow in the second pass below, as in a first superstep.
        d.chk(csc_propagate(ctx, &cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, d_final, d_lbs, d_ubs, d_recs));
        d.chk(csc_propagate_summary(ctx, n, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_t, &summ));
        row_cap = 0;
    }
    pt.lap("propagate");
    log_line(LOG_TRACE, "Propagation + AABB computation time: " + sw.elapsed() + "s");
    d.row_hint = summ.max_substeps;
    d.row_hint_ct = m_ct;
    d.row_hint_nchunks = nchunks0;
    const bool rows_done = row_cap != 0u && !(summ.flags & CSC_ST_FLAG_ROW_OVERFLOW);

    if (summ.n_nf) {
        // :1127-1145: nothing user-visible changes.
        confirm();
        m_int_info = std::tuple<size_type, double>{summ.nf_idx, summ.nf_time};
        log_line(LOG_DEBUG, "The step function was interrupted due to particle " + std::to_string(summ.nf_idx)
                                + " generating a non-finite state during the numerical integration");
        log_line(LOG_TRACE, "Total propagation time: " + sw_total.elapsed() + "s");
        log_line(LOG_TRACE, "---- STEP END ---");
        return outcome::err_nf_state;
    }
    if (summ.flags & CSC_ST_FLAG_BAD_AABB) {
        throw std::invalid_argument("The computation of the bounding box for a particle produced a non-finite bound");
    }
    if (summ.flags & CSC_ST_FLAG_BAD_TCOORD) {
        throw std::invalid_argument("A non-finite time coordinate was generated during the numerical integration of a particle");
    }
    if (summ.flags & CSC_ST_FLAG_MAX_SUBSTEPS) {
        throw std::overflow_error("Overflow detected during the numerical integration of a particle (substep cap reached)");
    }

    std::uint32_t nchunks = nchunks0;
    double delta_t = delta_t0;
    if (summ.n_ste) {
        // :1148-1171
        delta_t = std::min(summ.ste_time, delta_t0);
        const auto new_nchunks = static_cast<std::uint32_t>(std::ceil(delta_t / m_ct));
        if (new_nchunks == 0u) {
            throw std::invalid_argument(
                "The recomputed number of chunks after the triggering of a stopping terminal "
                "event cannot be zero (this likely indicates that the simulation was restarted "
                "without resolving a stopping terminal event)");
        }
        nchunks = std::min(new_nchunks, nchunks0);
    }

    // Chunks are independent once the trajectories are known (doc/collision_algorithm.rst:71-80):
    // with the propagation sharded and at least one chunk per process, every process sorts, builds
    // the tree of, traverses and narrow-phases ITS block of chunks [c0, c0 + cl) only -- nothing
    // but the integration's all-gather is shared, nothing is replicated. (Otherwise: all chunks
    // here, the pair tree of every chunk sharded over the processes.)
    std::uint32_t c0 = 0, cl = nchunks;
    if (chunk_mode) {
        // (csc_shard_block's rule on nchunks0, cut to the chunks a stopping event left)
        const std::uint32_t per_c = (nchunks0 + d.shard_n - 1u) / d.shard_n;
        c0 = std::min(nchunks, d.shard_rank * per_c);
        cl = std::min(per_c, nchunks - c0);
        const std::size_t o = static_cast<std::size_t>(c0) * n;
        d_lbs += o * 4u;
        d_ubs += o * 4u;
        d_glb += c0 * 4ull;
        d_gub += c0 * 4ull;
        d_codes += o;
        d_vidx += o;
        d_scodes += o;
        d_slbs += o * 4u;
        d_subs += o * 4u;
        if (d_recs != nullptr) {
            d_recs += o * 16u;
            d_srecs += o * 16u;
        }
    }
    d.chunk_c0 = c0;
    d.chunk_cl = cl;
    d.chk(csc_set_chunk_map(ctx, c0, chunk_mode ? nchunks : 0u));
    const bool have_chunks = cl != 0u;

    sw = stopwatch{};
    const std::uint64_t tot = summ.total_substeps;
    d.last_cfg = cfg;
    std::uint64_t npairs = 0;
    std::uint64_t ncoll = 0, nconj = 0;
    std::uint32_t warn = 0, np_attempts = 0;
    const csc_collision *coll_p = nullptr;
    const csc_conjunction *conj_p = nullptr;
    std::uint64_t ncoll_v = 0, nconj_v = 0;
    bool conj_on_device = false;
    // Everything between the integration and the summary exchange works on THIS process's chunks
    // when the chunks are dealt out, so an error in it is local: it is caught, carried through
    // the exchange as a code, and raised on every process afterwards (a process that threw here
    // would leave the others waiting in the all-gather forever).
    struct nonfinite_gaabb : std::invalid_argument {
        using std::invalid_argument::invalid_argument;
    };
    static const char *const gaabb_msg = "A global bounding box with non-finite boundaries and/or size was generated";
    const auto local_stages = [&]() {
    if (have_chunks) {
        d.chk(csc_global_aabb(ctx, n, cl, d_lbs, d_ubs, d_glb, d_gub));
        d.chk(csc_morton_encode(ctx, n, cl, d_lbs, d_ubs, d_glb, d_gub, d_codes));
        d.chk(csc_sort_gather(ctx, n, cl, d_codes, d_lbs, d_ubs, d_vidx, d_scodes, d_slbs, d_subs, d_recs, d_srecs));
    }

    if (tracing) {
        d.chk(csc_sync(ctx));
        log_line(LOG_TRACE, "Morton encoding and sorting time: " + sw.elapsed() + "s");
        sw = stopwatch{};
    }
    if (have_chunks) {
        d.chk(csc_bvh_build(ctx, n, cl, d_scodes, d_slbs, d_subs));
    }
    verify_arrays vdata;
    if (d.verify && have_chunks) {
        vdata = verify_sort_and_trees(ctx, n, cl, d_lbs, d_ubs, d_glb, d_gub, d_codes, d_vidx, d_scodes, d_slbs, d_subs);
    }
    if (tracing) {
        d.chk(csc_sync(ctx));
        log_line(LOG_TRACE, "BVH construction time: " + sw.elapsed() + "s");
        sw = stopwatch{};
    }
    if (chunk_mode) {
        d.chk(csc_set_shard(ctx, 0u, 1u));
    } else {
        d.chk(csc_set_shard(ctx, d.shard_rank, d.shard_n));
    }
    // With the pre-filter on, the box-overlap pairs are counted and filtered inside the tile kernel
    // of the traversal and only the candidates of the polynomial test are stored (csc_broad_phase_filtered).
    const bool fused = have_chunks && d.np_prefilter && d_srecs != nullptr && d.fuse_env && !d.verify; // (the verifier wants the pair list)
    if (fused) {
        std::uint64_t ncand = 0;
        d.chk(csc_broad_phase_filtered(ctx, n, cl, d_slbs, d_subs, d_vidx, d_active, m_conj_thresh, d_srecs, &npairs, &ncand));
    } else if (have_chunks) {
        d.chk(csc_broad_phase(ctx, n, cl, d_slbs, d_subs, d_vidx, d_active, &npairs));
        if (d.verify) {
            verify_pairs(ctx, n, cl, vdata, d_active, npairs);
        }
    }
    if (tracing) {
        // src/sim_broad_phase.cpp:209-213 (with a sharded simulation: this process's share)
        log_line(LOG_TRACE, "Broad phase collision detection time: " + sw.elapsed() + "s");
        log_line(LOG_TRACE, "Average number of AABB collisions per particle per chunk: "
                                + fmt_d(static_cast<double>(npairs) / static_cast<double>(std::max(cl, 1u)) / static_cast<double>(n)));
        sw = stopwatch{};
    }
    {
        const int rc = csc_check_flags(ctx);
        if (rc == CSC_ERR_NONFINITE_AABB) {
            throw nonfinite_gaabb(gaabb_msg);
        }
        d.chk(rc);
    }
    pt.lap("sort/bvh/bp");
    d.stats.stored_objects = n;
    // Taylor coefficients for dense output when the integration did not write them: exact-fit CSR
    // rows, of every particle in the Morton order of chunk 0 (two-pass mode), or only of the
    // particles that the candidate pairs involve.
    auto store_all = [&]() {
        d_txyz = d.buf<double>("tcs_xyz", std::max<std::uint64_t>(tot, 1) * xs);
        d_tvr = d.buf<double>("tcs_vr", std::max<std::uint64_t>(tot, 1) * vs);
        d_tco = d.buf<double>("tcoords", std::max<std::uint64_t>(tot, 1) * 2u);
        d.chk(csc_propagate_store(ctx, &cfg, n, d_state, have_chunks ? d_vidx : nullptr, d_nsub, d_tc_row, d_txyz, d_tvr, d_tco));
        d.rows_full = true;
        d.stats.stored_objects = n;
    };
    if (!have_chunks) {
        // (a terminal event cut the superstep short of this process's chunks: nothing to look for)
        d.rows_full = false;
    } else if (sparse) {
        std::uint64_t ncand = 0, nrows = 0;
        std::uint32_t nlist = 0;
        if (!fused) {
            d.chk(csc_narrow_phase_prefilter(ctx, n, cl, m_conj_thresh, d_srecs, &ncand));
            np_attempts += csc_narrow_phase_attempts(ctx) - 1u;
        }
        auto *d_list = d.buf<std::uint32_t>("cand_objects", n);
        d.chk(csc_candidate_objects(ctx, n, d_nsub, d_list, &nlist, nullptr));
        // (an upper bound is enough: the listed particles times the largest substep count, and
        // never more than the substeps of all particles)
        nrows = std::min<std::uint64_t>(static_cast<std::uint64_t>(nlist) * summ.max_substeps, tot);
        if (static_cast<std::uint64_t>(nlist) * 4u > n || d.want_full_rows) {
            store_all();
        } else {
            d.stats.stored_objects = nlist;
            d_txyz = d.buf<double>("tcs_xyz", std::max<std::uint64_t>(nrows, 1) * xs);
            d_tvr = d.buf<double>("tcs_vr", std::max<std::uint64_t>(nrows, 1) * vs);
            d_tco = d.buf<double>("tcoords", std::max<std::uint64_t>(nrows, 1) * 2u);
            if (nlist) {
                d.chk(csc_propagate_store(ctx, &cfg, nlist, d_state, d_list, d_nsub, d_tc_row, d_txyz, d_tvr, d_tco));
            }
            d.rows_full = false;
        }
        d.chk(csc_narrow_phase_finish(ctx, n, cl, m_ct, delta_t, d.time.hi, d.time.lo, m_conj_thresh, d.order, d_state,
                                      d_active, d_tc_row, d_nsub, d_txyz, d_tvr, d_tco, &ncoll, &nconj, &warn));
    } else {
        if (!rows_done) {
            store_all();
        } else {
            d.rows_full = true;
        }
        if (fused) {
            d.chk(csc_narrow_phase_finish(ctx, n, cl, m_ct, delta_t, d.time.hi, d.time.lo, m_conj_thresh, d.order, d_state,
                                          d_active, d_tc_row, d_nsub, d_txyz, d_tvr, d_tco, &ncoll, &nconj, &warn));
        } else {
            d.chk(csc_narrow_phase(ctx, n, cl, m_ct, delta_t, d.time.hi, d.time.lo, m_conj_thresh, d.order, d_state, d_active,
                                   d_tc_row, d_nsub, d_txyz, d_tvr, d_tco, d_srecs, &ncoll, &nconj, &warn));
        }
    }
    if (!rows_done && have_chunks) {
        // A second integration pass wrote the rows: it must have repeated the substep counts of the
        // first one (csc_propagate_store never writes past a particle's rows and reports a mismatch).
        d.chk(csc_check_flags(ctx));
    }
    // The event lists stay in the context (pinned host memory; a conjunction-heavy superstep
    // yields millions of 128-byte records): no copies besides the append to m_det_conj below.
    if (have_chunks) {
        // (a large conjunction list may have stayed on the device: counts now, records on demand)
        int on_dev = 0;
        d.chk(csc_narrow_phase_counts(ctx, &coll_p, &ncoll_v, &nconj_v, &on_dev));
        conj_on_device = on_dev != 0;
        if (!conj_on_device) {
            d.chk(csc_narrow_phase_view(ctx, &coll_p, &ncoll_v, &conj_p, &nconj_v));
        }
    }
    pt.lap("narrow");
    };
    std::exception_ptr local_err;
    std::uint64_t local_code = 0;
    if (d.shard_n > 1u) {
        try {
            local_stages();
        } catch (const nonfinite_gaabb &) {
            local_err = std::current_exception();
            local_code = 1;
        } catch (...) {
            local_err = std::current_exception();
            local_code = 2;
        }
        if (local_err) {
            ncoll_v = nconj_v = 0;
            coll_p = nullptr;
            conj_p = nullptr;
            conj_on_device = false;
        }
    } else {
        local_stages();
    }
    if (tracing) {
        // src/sim_narrow_phase.cpp:1017-1021
        log_line(LOG_TRACE, "Narrow phase collision detection time: " + sw.elapsed() + "s");
        log_line(LOG_TRACE, "Total number of collisions detected: " + std::to_string(ncoll_v));
        if (with_conj) {
            log_line(LOG_TRACE, "Total number of conjunctions detected: " + std::to_string(nconj_v));
        }
    }
    if (warn != 0u) {
        // src/sim_narrow_phase.cpp:373-381, 410-417: root-finder give-ups are warnings, the pair is skipped.
        log_line(LOG_WARN, "Polynomial root finding during collision/conjunction detection failed or produced non-finite "
                           "values for some pairs of this step, which were skipped (flags "
                               + std::to_string(warn) + ")");
    }

    d.stats.bp_pairs = npairs;
    d.stats.bp_node_pairs = have_chunks ? csc_broad_phase_node_pairs(ctx) : 0u;
    d.stats.bp_tiles = have_chunks ? csc_broad_phase_tiles(ctx) : 0u;
    d.stats.np_candidates = have_chunks ? csc_narrow_phase_candidates(ctx) : 0u;
    d.stats.np_attempts = have_chunks ? np_attempts + csc_narrow_phase_attempts(ctx) : 0u;
    d.stats.chunk_first = c0;
    d.stats.chunk_count = cl;
    d.stats.total_substeps = tot;
    d.stats.nchunks = nchunks;
    d.last_coll.clear();
    for (std::uint64_t k = 0; k < ncoll_v; ++k) {
        d.last_coll.emplace_back(coll_p[k].i, coll_p[k].j, coll_p[k].time);
    }
    // get_last_step_conjunctions() copies them on demand.
    d.last_conj.clear();
    d.last_conj_view = conj_p;
    d.last_conj_n = nconj_v;
    d.last_conj_on_device = conj_on_device;
    std::vector<csc_collision> coll(coll_p, coll_p + ncoll_v);
    std::uint64_t ncand_all = d.stats.np_candidates;
    if (d.shard_n > 1u) {
        // One 48-byte summary per process: the outcome logic below needs the earliest collision
        // of all processes (coll is sorted by (time, i, j)); the conjunction records stay here.
        shard_summary mine;
        mine.n_coll = ncoll_v;
        mine.n_conj = nconj_v;
        mine.n_cand = (ncand_all & 0x00ffffffffffffffull) | (local_code << 56); // (top byte: this process's error code)
        if (ncoll_v) {
            mine.coll_i = coll[0].i;
            mine.coll_j = coll[0].j;
            mine.coll_time = coll[0].time;
        }
        std::vector<shard_summary> all;
        d.exchange(mine, all);
        if (all.size() != d.shard_n) {
            throw std::runtime_error("sharded simulation: the exchange returned " + std::to_string(all.size())
                                     + " summaries for " + std::to_string(d.shard_n) + " processes");
        }
        for (std::size_t r = 0; r < all.size(); ++r) {
            const std::uint64_t code = all[r].n_cand >> 56;
            if (code == 0u) {
                continue;
            }
            // Some process failed in its share of the superstep: the same kind of exception on
            // every process, nothing user-visible has changed yet (strong guarantee as in the
            // unsharded step, src/sim_propagate.cpp:513-515).
            if (local_err) {
                std::rethrow_exception(local_err);
            }
            if (code == 1u) {
                throw std::invalid_argument(gaabb_msg);
            }
            throw std::runtime_error("sharded simulation: process " + std::to_string(r)
                                     + " failed in its share of the superstep (its own exception has the reason)");
        }
        coll.clear();
        ncand_all = 0;
        for (const auto &s : all) {
            ncand_all += s.n_cand & 0x00ffffffffffffffull;
            if (s.n_coll) {
                const csc_collision c{s.coll_i, s.coll_j, s.coll_time};
                if (coll.empty() || std::tie(c.time, c.i, c.j) < std::tie(coll[0].time, coll[0].i, coll[0].j)) {
                    coll.assign(1, c);
                }
            }
        }
    }
    // (the same decision on every process: ncand_all is the global count)
    constexpr std::uint64_t sparse_factor = 8;
    d.sparse_likely = ncand_all * sparse_factor <= n;

    // Outcome resolution (:1189-1345). coll is sorted by (time, i, j); a collision wins a tie
    // with a terminal event (:1225).
    outcome oc = outcome::success;
    double interrupt_time = 0;
    if (!coll.empty()) {
        interrupt_time = coll[0].time;
        oc = outcome::collision;
    }
    if (summ.n_ste && (oc == outcome::success || summ.ste_time < interrupt_time)) {
        interrupt_time = summ.ste_time;
        oc = (with_exit_event() && summ.ste_ev_idx == 0u) ? outcome::exit : outcome::reentry;
    }

    confirm();
    if (oc == outcome::success) {
        d.time = df_add(d.time, dfloat{delta_t0, 0.});
        d.chk(csc_copy_final_state(ctx, n, d_final, d_state));
        d.state_prestep = false;
        if (d.host_sync && !d.in_loop) {
            download_state();
        }
        m_int_info.reset();
        pt.lap("final state");
        if (with_conj && conj_on_device) {
            // The records stay on the device, behind those of the earlier supersteps
            // (src/sim_propagate.cpp:1644-1701 appends to the host vector: get_conjunctions() does it here).
            d.chk(csc_conj_store_append(ctx, nconj_v));
            d.conj_pending += nconj_v;
            if (nconj_v) {
                d.conj_segments.emplace_back(d.step_counter, nconj_v);
            }
            if (d.conj_pending * sizeof(conjunction) > (24ull << 30)) {
                _sync_conjunctions(); // (not more than 24 GB of HBM for records nobody has asked for)
            }
        } else if (with_conj) {
            append_conj_data(conj_p, nconj_v);
        }
        pt.lap("append conj");
    } else {
        pt.lap("outcome");
        dense_propagate(interrupt_time);
        d.time = df_add(d.time, dfloat{interrupt_time, 0.});
        if (oc == outcome::collision) {
            m_int_info = std::array<size_type, 2>{static_cast<size_type>(coll[0].i), static_cast<size_type>(coll[0].j)};
        } else {
            m_int_info = static_cast<size_type>(summ.ste_idx);
        }
        if (with_conj && conj_on_device) {
            d.chk(csc_narrow_phase_view(ctx, &coll_p, &ncoll_v, &conj_p, &nconj_v));
            d.last_conj_view = conj_p;
            d.last_conj_on_device = false;
        }
        if (with_conj) {
            // Keep the conjunctions that happen before the interruption (:1329-1336).
            const auto *last = std::lower_bound(conj_p, conj_p + nconj_v, d.time.hi,
                                                [](const csc_conjunction &c, double t) { return c.time < t; });
            append_conj_data(conj_p, static_cast<std::size_t>(last - conj_p));
        }
    }
    log_line(LOG_TRACE, "Total propagation time: " + sw_total.elapsed() + "s");
    log_line(LOG_TRACE, "---- STEP END ---");
    return oc;
}

// propagate_until (src/sim_propagate.cpp:1505-1603).
outcome sim::propagate_until(double t)
{
    auto &d = *m_data;
    if (!std::isfinite(t) || df_lt(dfloat{t, 0.}, d.time)) {
        throw std::invalid_argument("The final time passed to the propagate_until() function must be finite and not less "
                                    "than the current simulation time, but a value of "
                                    + fmt_d(t) + " was provided instead");
    }
    const dfloat final_t{t, 0.};
    if (df_eq(final_t, d.time)) {
        return outcome::time_limit;
    }
    // The host state is refreshed once, however the loop ends (an exception leaves the state of
    // the last completed superstep, as a sequence of step() calls would).
    struct loop_guard {
        sim *s;
        ~loop_guard()
        {
            auto &dd = *s->m_data;
            dd.in_loop = false;
            if (dd.host_sync && dd.loop_uploaded) {
                try {
                    s->download_state();
                } catch (...) {
                }
            }
        }
    } guard{this};
    d.in_loop = true;
    d.loop_uploaded = false;
    while (true) {
        const dfloat orig_t = d.time;
        // The superstep that reaches the final time is evaluated by dense_propagate() below: it
        // keeps the rows of every particle (the time a successful step() arrives at is known).
        d.want_full_rows = !df_lt(df_add(d.time, dfloat{m_ct * m_n_par_ct, 0.}), final_t);
        struct reset_flag {
            bool &f;
            ~reset_flag()
            {
                f = false;
            }
        } rf{d.want_full_rows};
        const outcome cur = step();
        if (cur == outcome::success) {
            if (!df_lt(d.time, final_t)) {
                dense_propagate(df_sub(final_t, orig_t).hi);
                d.time = final_t;
                return outcome::time_limit;
            }
        } else if (cur != outcome::err_nf_state) {
            if (df_lt(final_t, d.time)) {
                // Interrupted after the final time: roll back to it (:1545-1569).
                dense_propagate(df_sub(final_t, orig_t).hi);
                d.time = final_t;
                m_int_info.reset();
                return outcome::time_limit;
            }
            return cur;
        } else {
            return cur;
        }
    }
}

std::ostream &operator<<(std::ostream &os, const sim &s)
{
    os << "Total number of particles: " << s.get_nparts() << '\n';
    os << "Collisional timestep     : " << fmt_d(s.get_ct()) << '\n';
    return os;
}

std::ostream &operator<<(std::ostream &os, const sim::conjunction &c)
{
    return os << "{" << c.i << ", " << c.j << ", " << fmt_d(c.time) << ", " << fmt_d(c.dist) << "}";
}

} // namespace cascade
