// include/cascade/sim.hpp
//
// Public C++ API of the MI355X build: cascade::sim with the names, argument meaning and
// error behaviour of the reference class (esa/cascade include/cascade/sim.hpp:82-513), driving
// the gfx950 kernels of libcascade_hip.so through the C ABI of include/cascade_hip.h.
//
// Drop-in surface: outcome, sim::conjunction, sim::whitelist_t, the named constructor
// arguments kw::dyn / pars / reentry_radius / exit_radius / tol / high_accuracy /
// compact_mode / n_par_ct / conj_thresh / min_coll_radius / coll_whitelist / conj_whitelist,
// step(), propagate_until(), every getter and setter, set_new_state_pars(),
// remove_particles(), reset_conjunctions(), the _get_*_ptr() accessors used by the Python
// bindings, operator<<.
//
// The single intentional deviation: the equations of kw::dyn are cascade::expression objects
// (include/cascade/expression.hpp) instead of heyoka expressions; heyoka and its LLVM JIT are
// replaced by hand-unrolled Taylor kernels for Kepler / J2 / drag and by a tape interpreter on
// the GPU for every other right-hand side. kw::dyn takes what dynamics::kepler()
// and dynamics::simple_earth() return, modified or not, or any std::vector of six (variable,
// expression) pairs. One extra named argument, kw::device, selects the GPU.
//
// No CPU fallback: step() needs libcascade_hip.so and a HIP device and throws
// std::runtime_error otherwise. Construction, validation, getters and setters work anywhere.
#ifndef CASCADE_SIM_HPP
#define CASCADE_SIM_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iosfwd>
#include <iterator>
#include <memory>
#include <optional>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_set>
#include <utility>
#include <variant>
#include <vector>

#include "expression.hpp"

namespace cascade
{

// include/cascade/sim.hpp:82
enum class outcome { success, time_limit, collision, reentry, exit, err_nf_state };

namespace dynamics
{

// The dynamics of a simulation: the six equations (variable, right-hand side) of the reference's
// std::vector<std::pair<heyoka::expression, heyoka::expression>> (sim.hpp:213-222), with
// vector-like access, plus a tag that remembers when they are exactly one of the built-in
// right-hand sides (Kepler, Kepler + J2, Kepler + J2 + drag) for which the propagation kernel is
// hand-unrolled. Any write access (dyn[3].second += ...) drops the tag: the equations are then
// lowered to a tape and interpreted on the GPU.
struct dyn_t {
    using eq_t = std::pair<expression, expression>;
    enum kind_t : int { kepler = 0, j2 = 1, drag = 2, tape = 0x100 };
    int kind = kepler;
    double mu = 1.;
    std::uint32_t npars = 0; // runtime parameters per particle (drag: 1 = BSTAR)
    double J2 = 0.;
    double Re = 1.;
    // rho(h) = sum_k alpha_k exp(-(h - gamma_k) beta_k), h = r - Re (SI units)
    std::array<double, 4> dens_alpha = {}, dens_beta = {}, dens_gamma = {};
    std::vector<eq_t> eqs;

    dyn_t() = default;
    dyn_t(std::vector<eq_t> e) : kind(tape), eqs(std::move(e)) {} // NOLINT (implicit on purpose)

    [[nodiscard]] std::size_t size() const
    {
        return eqs.size();
    }
    [[nodiscard]] bool empty() const
    {
        return eqs.empty();
    }
    eq_t &operator[](std::size_t i)
    {
        kind = tape;
        return eqs[i];
    }
    const eq_t &operator[](std::size_t i) const
    {
        return eqs[i];
    }
    void push_back(eq_t e)
    {
        kind = tape;
        eqs.push_back(std::move(e));
    }
    [[nodiscard]] auto begin() const
    {
        return eqs.begin();
    }
    [[nodiscard]] auto end() const
    {
        return eqs.end();
    }

    friend bool operator==(const dyn_t &a, const dyn_t &b)
    {
        if (a.kind != b.kind) {
            return false;
        }
        if (a.kind == tape) {
            return a.eqs.size() == b.eqs.size() && (a.eqs.size() != 6u || lower_dynamics(a.eqs) == lower_dynamics(b.eqs));
        }
        return a.mu == b.mu && a.npars == b.npars && a.J2 == b.J2 && a.Re == b.Re && a.dens_alpha == b.dens_alpha
               && a.dens_beta == b.dens_beta && a.dens_gamma == b.dens_gamma;
    }
};

// src/sim_dynamics.cpp:25-42
dyn_t kepler(double mu = 1.);

// Perturbed dynamics around the Earth in SI units, same signature and defaults as the
// reference's cascade.dynamics.simple_earth (cascade.py/dynamics/_simple_earth.py:86-96): J2,
// J3, J4 (:169-200), C22/S22 (:202-242), Sun and Moon gravity (:244-363), drag with the
// exponential density fit (:365-387, :16-46; the BSTAR coefficient of every object is its first
// runtime parameter, kw::pars) and solar radiation pressure (:390-409; Cr A/m is the next
// parameter). J2 alone and J2 + drag run on the hand-unrolled kernels, every other combination
// on the tape interpreter. thermonets (a neural density model of heyoka) throws.
dyn_t simple_earth(bool J2 = true, bool J3 = false, bool J4 = false, bool C22S22 = true, bool sun = false,
                   bool moon = false, bool SRP = false, bool drag = true, bool thermonets = false);

} // namespace dynamics

namespace detail
{

template <typename Tag, typename T>
struct tagged_arg {
    T value;
};

template <typename Tag>
struct kwarg {
    template <typename T>
    constexpr tagged_arg<Tag, std::decay_t<T>> operator=(T &&v) const
    {
        return {std::forward<T>(v)};
    }
};

} // namespace detail

// Named constructor arguments (sim.hpp:56-72): sim(state, ct, kw::n_par_ct = 3, ...).
namespace kw
{

inline constexpr detail::kwarg<struct dyn_tag> dyn{};
inline constexpr detail::kwarg<struct pars_tag> pars{};
inline constexpr detail::kwarg<struct reentry_radius_tag> reentry_radius{};
inline constexpr detail::kwarg<struct exit_radius_tag> exit_radius{};
inline constexpr detail::kwarg<struct tol_tag> tol{};
inline constexpr detail::kwarg<struct high_accuracy_tag> high_accuracy{};
inline constexpr detail::kwarg<struct compact_mode_tag> compact_mode{};
inline constexpr detail::kwarg<struct n_par_ct_tag> n_par_ct{};
inline constexpr detail::kwarg<struct conj_thresh_tag> conj_thresh{};
inline constexpr detail::kwarg<struct min_coll_radius_tag> min_coll_radius{};
inline constexpr detail::kwarg<struct coll_whitelist_tag> coll_whitelist{};
inline constexpr detail::kwarg<struct conj_whitelist_tag> conj_whitelist{};
inline constexpr detail::kwarg<struct device_tag> device{};

} // namespace kw

class sim
{
public:
    using size_type = std::vector<double>::size_type;

    // sim.hpp:90-104 (numpy dtype: cascade.py/core.cpp:81)
    struct conjunction {
        size_type i = 0;
        size_type j = 0;
        double time = 0;
        double dist = 0;
        std::array<double, 6> state_i = {};
        std::array<double, 6> state_j = {};
    };

    using whitelist_t = std::unordered_set<size_type>;
    using interrupt_info_t
        = std::optional<std::variant<std::array<size_type, 2>, size_type, std::tuple<size_type, double>>>;

    // Everything the named arguments can set (defaults = the reference's, sim.hpp:200-379).
    struct options {
        dynamics::dyn_t dyn = dynamics::dyn_t{};
        std::vector<double> pars;
        std::variant<double, std::vector<double>> reentry_radius = 0.;
        double exit_radius = 0.;
        double tol = 0.; // 0: machine epsilon
        bool high_accuracy = false;
        bool compact_mode = false;
        std::uint32_t n_par_ct = 1;
        double conj_thresh = 0.;
        double min_coll_radius = 0.;
        whitelist_t coll_whitelist, conj_whitelist;
        int device = 0;
    };

    // Measurements of the last step() (no counterpart in the reference besides its trace log).
    struct step_stats {
        std::uint64_t bp_pairs = 0;       // AABB pairs found by the broad phase
        std::uint64_t bp_node_pairs = 0;  // node pairs expanded by the broad phase's dual traversal
        std::uint64_t bp_tiles = 0;       // ... and the tiles (small node pairs tested directly) it ended in
        std::uint64_t propagated = 0;     // particles this process integrated in the first pass
        std::uint64_t np_candidates = 0;  // pairs that reached the polynomial narrow phase
        std::uint64_t total_substeps = 0; // Taylor substeps of all particles
        std::uint32_t nchunks = 0;        // chunks actually processed (after an event shrink)
        std::uint32_t chunk_first = 0, chunk_count = 0; // the block of them this process handled (all, unless the chunks are sharded)
        std::uint64_t exchanged_bytes = 0;              // received by this process in the per-particle exchange (0: not sharded)
        std::uint64_t stored_objects = 0; // particles whose Taylor rows were written (all, or those of the candidate pairs)
        std::uint32_t np_attempts = 0;    // narrow-phase passes (more than one per call: an output list overflowed and was grown)
    };

private:
    struct sim_data;

    std::shared_ptr<std::vector<double>> m_state;
    std::shared_ptr<std::vector<double>> m_pars;
    double m_ct = 0;
    std::uint32_t m_n_par_ct = 0;
    interrupt_info_t m_int_info;
    std::variant<double, std::vector<double>> m_reentry_radius;
    double m_exit_radius = 0;
    std::uint32_t m_npars = 0;
    double m_conj_thresh = 0;
    mutable std::shared_ptr<std::vector<conjunction>> m_det_conj;
    double m_min_coll_radius = 0;
    whitelist_t m_coll_whitelist, m_conj_whitelist;
    std::unique_ptr<sim_data> m_data;

    struct ptag_t {
    };
    sim(ptag_t, std::vector<double>, double);
    void finalise_ctor(options);

    template <typename Tag, typename T>
    static void apply_kwarg(options &, detail::tagged_arg<Tag, T> &&);

    [[nodiscard]] bool with_reentry_event() const;
    [[nodiscard]] bool with_exit_event() const;
    void verify_state_vector(const std::vector<double> &) const;
    void validate_pars_vector(std::vector<double> &, size_type) const;
    outcome step_impl(bool, const std::function<void()> &);
    void dense_propagate(double);
    void append_conj_data(const void *, std::size_t);

public:
    sim();
    template <typename... KwArgs>
    explicit sim(std::vector<double> state, double ct, KwArgs &&...kw_args) : sim(ptag_t{}, std::move(state), ct)
    {
        options o;
        (apply_kwarg(o, std::forward<KwArgs>(kw_args)), ...);
        finalise_ctor(std::move(o));
    }
    explicit sim(std::vector<double> state, double ct, options o) : sim(ptag_t{}, std::move(state), ct)
    {
        finalise_ctor(std::move(o));
    }
    sim(const sim &);
    sim(sim &&) noexcept;
    ~sim();
    sim &operator=(const sim &);
    sim &operator=(sim &&) noexcept;

    [[nodiscard]] const interrupt_info_t &get_interrupt_info() const
    {
        return m_int_info;
    }
    [[nodiscard]] const std::vector<double> &get_state() const
    {
        return *m_state;
    }
    [[nodiscard]] const double *get_state_data() const
    {
        return m_state->data();
    }
    [[nodiscard]] double *get_state_data()
    {
        return m_state->data();
    }
    [[nodiscard]] const std::vector<double> &get_pars() const
    {
        return *m_pars;
    }
    [[nodiscard]] const double *get_pars_data() const
    {
        return m_pars->data();
    }
    [[nodiscard]] double *get_pars_data()
    {
        return m_pars->data();
    }
    [[nodiscard]] size_type get_nparts() const
    {
        return get_state().size() / 7u;
    }

    [[nodiscard]] double get_time() const;
    void set_time(double);
    [[nodiscard]] double get_ct() const;
    void set_ct(double);
    [[nodiscard]] std::uint32_t get_n_par_ct() const;
    void set_n_par_ct(std::uint32_t);
    [[nodiscard]] double get_tol() const;
    [[nodiscard]] bool get_high_accuracy() const;
    [[nodiscard]] bool get_compact_mode() const;
    [[nodiscard]] std::uint32_t get_npars() const;
    [[nodiscard]] std::variant<double, std::vector<double>> get_reentry_radius() const;
    [[nodiscard]] double get_exit_radius() const
    {
        return m_exit_radius;
    }
    [[nodiscard]] double get_conj_thresh() const
    {
        return m_conj_thresh;
    }
    void set_conj_thresh(double);
    // (This build keeps the records of conjunction-heavy supersteps on the device and brings them
    // to the host vector when somebody looks: _sync_conjunctions().)
    [[nodiscard]] const std::vector<conjunction> &get_conjunctions() const
    {
        _sync_conjunctions();
        return *m_det_conj;
    }
    void reset_conjunctions();
    [[nodiscard]] double get_min_coll_radius() const
    {
        return m_min_coll_radius;
    }
    void set_min_coll_radius(double);
    [[nodiscard]] const whitelist_t &get_coll_whitelist() const
    {
        return m_coll_whitelist;
    }
    void set_coll_whitelist(whitelist_t);
    [[nodiscard]] const whitelist_t &get_conj_whitelist() const
    {
        return m_conj_whitelist;
    }
    void set_conj_whitelist(whitelist_t);

    void set_new_state_pars(std::vector<double>, std::vector<double> = {});
    void remove_particles(std::vector<size_type>);

    outcome step();
    outcome propagate_until(double);

    [[nodiscard]] std::shared_ptr<std::vector<double>> _get_state_ptr() const
    {
        return m_state;
    }
    [[nodiscard]] std::shared_ptr<std::vector<double>> _get_pars_ptr() const
    {
        return m_pars;
    }
    [[nodiscard]] std::shared_ptr<std::vector<conjunction>> _get_conjunctions_ptr() const
    {
        _sync_conjunctions();
        return m_det_conj;
    }
    void _sync_conjunctions() const;

    // ---- additions of this build (not in the reference) -------------------------------
    [[nodiscard]] const dynamics::dyn_t &get_dyn() const;
    [[nodiscard]] int get_device() const;
    // The host state is the source of truth at every step() entry and exit (the Python bindings
    // hand out writable views of it, cascade.py/core.cpp:199-230). With host_sync off the state
    // and the parameters stay resident in HBM between steps: they are uploaded by the first
    // step() (or upload_state()) and again after set_new_state_pars(); remove_particles()
    // compacts the device copies in place (csc_remove_particles); download_state() refreshes the
    // host coordinates when they are needed; switching host_sync back on does it too. Edits of
    // the host views are not seen by the device while host_sync is off (call upload_state()).
    void set_host_sync(bool);
    [[nodiscard]] bool get_host_sync() const;
    void upload_state();
    void download_state();
    // Multi-GPU operation (no counterpart in the reference, which is single-process): `nranks`
    // processes, one per GPU, hold the SAME simulation (same state, same calls). Every process
    // sorts and builds the whole tree, expands only its share of the pair tree and runs the narrow
    // phase on the pairs it found. What the outcome logic (src/sim_propagate.cpp:1189-1345) needs
    // from the others is small: whether anybody found a collision and which one is the earliest.
    // `exchange(local, all)` all-gathers one 48-byte summary per process (all[r] = summary of rank
    // r, identical on every process; cascade_amd/parallel.py does it with one RCCL all-gather on
    // device tensors), after which every process applies the reference's outcome logic to the
    // same data and stays in lockstep. Conjunction records stay on the process that found them:
    // get_conjunctions() holds the local ones (cascade_amd.parallel.gather_conjunctions merges
    // them into the unsharded list when the caller wants it), and get_last_collisions() the local
    // collisions of the last step.
    using collision_t = std::tuple<size_type, size_type, double>;
    struct shard_summary {
        std::uint64_t n_coll = 0, n_conj = 0, n_cand = 0; // local collisions, conjunctions, narrow-phase candidates
        std::uint64_t coll_i = 0, coll_j = 0;             // earliest local collision (when n_coll != 0)
        double coll_time = 0;                             // relative to the superstep start
    };
    using exchange_t = std::function<void(const shard_summary &, std::vector<shard_summary> &)>;
    void set_shard(std::uint32_t rank, std::uint32_t nranks, exchange_t exchange);
    // Bookkeeping for merging the per-process conjunction lists: (superstep counter, records
    // appended by it) for every step() that appended records since the last reset_conjunctions().
    [[nodiscard]] const std::vector<std::pair<std::uint64_t, std::uint64_t>> &_get_conj_segments() const;
    // Propagation sharded too (the north star's exchange: every process integrates the particles
    // [rank * m, (rank + 1) * m), m = ceil(N / nranks), and the per-particle results -- substep
    // counts, events, final states, per-chunk boxes and trajectory records, 80 + 96 C bytes each --
    // are all-gathered; Taylor rows are produced on demand where the narrow phase needs them).
    // `gather(send, recv, bytes)` must all-gather `bytes` bytes from the device address `send` of
    // every process into the device address `recv` (rank-major) and return when the data is
    // there. Pays when the integration dominates the superstep (expensive right-hand sides,
    // many chunks); for 1M Kepler objects the replicated integration is cheaper than the gather.
    using gather_t = std::function<void(std::uintptr_t, std::uintptr_t, std::size_t)>;
    // chunks (default on): with n_par_ct >= nranks each process then also owns a block of the
    // n_par_ct chunks -- it sorts, builds, traverses and narrow-phases those chunks only (chunks are
    // independent once the trajectories are known), so nothing of the superstep is replicated.
    // alltoall (optional; used when the chunks are dealt out): alltoall(send, recv, send_bytes,
    // recv_bytes_each) sends send_bytes[r] consecutive bytes from the device address `send` to
    // process r and receives recv_bytes_each bytes from every process into `recv` (rank-major).
    // A process then receives the boxes and trajectory records of ITS chunks only: 80 + 96 *
    // (chunks of its block) bytes per particle instead of 80 + 96 C.
    using alltoall_t = std::function<void(std::uintptr_t, std::uintptr_t, const std::vector<std::size_t> &, std::size_t)>;
    void set_shard_propagation(gather_t gather, bool chunks = true, alltoall_t alltoall = {});
    // Geometric pre-filter in front of the polynomial narrow phase (on by default; it never
    // changes the detected events, see k_np_prefilter).
    void set_np_prefilter(bool);
    [[nodiscard]] bool get_np_prefilter() const;
    [[nodiscard]] const step_stats &get_step_stats() const;
    // Measurement aid: node AABB tests the reference's per-particle traversal would perform on
    // the trees of the last step() (the unit count behind the broad phase's algorithmic bytes).
    [[nodiscard]] std::uint64_t bp_nodes_visited();
    // Collisions (i, j, time relative to the superstep start) and conjunctions found by the
    // last step(), before the outcome logic discards any of them.
    [[nodiscard]] const std::vector<std::tuple<size_type, size_type, double>> &get_last_collisions() const;
    [[nodiscard]] const std::vector<conjunction> &get_last_step_conjunctions() const;
    // The C-ABI context (csc_ctx *) of this simulation, created by the first step() (or by this
    // call); for profiling and measurement.
    [[nodiscard]] void *_get_hip_context();
};

std::ostream &operator<<(std::ostream &, const sim &);
std::ostream &operator<<(std::ostream &, const sim::conjunction &);

// ---- named-argument plumbing ---------------------------------------------------------

template <typename Tag, typename T>
inline void sim::apply_kwarg(options &o, detail::tagged_arg<Tag, T> &&a)
{
    if constexpr (std::is_same_v<Tag, kw::dyn_tag>) {
        o.dyn = std::move(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::pars_tag>) {
        o.pars = std::vector<double>(std::begin(a.value), std::end(a.value));
    } else if constexpr (std::is_same_v<Tag, kw::reentry_radius_tag>) {
        if constexpr (std::is_arithmetic_v<T>) {
            o.reentry_radius = static_cast<double>(a.value);
        } else {
            o.reentry_radius = std::vector<double>(std::begin(a.value), std::end(a.value));
        }
    } else if constexpr (std::is_same_v<Tag, kw::exit_radius_tag>) {
        o.exit_radius = static_cast<double>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::tol_tag>) {
        o.tol = static_cast<double>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::high_accuracy_tag>) {
        o.high_accuracy = static_cast<bool>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::compact_mode_tag>) {
        o.compact_mode = static_cast<bool>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::n_par_ct_tag>) {
        o.n_par_ct = static_cast<std::uint32_t>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::conj_thresh_tag>) {
        o.conj_thresh = static_cast<double>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::min_coll_radius_tag>) {
        o.min_coll_radius = static_cast<double>(a.value);
    } else if constexpr (std::is_same_v<Tag, kw::coll_whitelist_tag>) {
        o.coll_whitelist = whitelist_t(std::begin(a.value), std::end(a.value));
    } else if constexpr (std::is_same_v<Tag, kw::conj_whitelist_tag>) {
        o.conj_whitelist = whitelist_t(std::begin(a.value), std::end(a.value));
    } else if constexpr (std::is_same_v<Tag, kw::device_tag>) {
        o.device = static_cast<int>(a.value);
    } else {
        static_assert(sizeof(T) == 0, "unknown named argument");
    }
}

} // namespace cascade

#endif
