/* include/cascade_hip.h
 *
 * C ABI of libcascade_hip.so: the MI355X (gfx950) kernels behind cascade::sim::step().
 *
 * The reference (esa/cascade) has no FFI boundary around step(): its stages are private
 * member functions of cascade::sim working on sim::sim_data. This header is the boundary
 * this build introduces between the C++ host orchestration (include/cascade/sim.hpp,
 * cascade_amd/csrc/sim.cpp) and the HIP kernels. Every entry point cites the reference
 * code it replaces (paths relative to the esa/cascade tree). Conventions:
 *   - plain pointers and sizes only; pointers named d_* are DEVICE pointers;
 *   - every function returns 0 on success or a negative csc_status; nothing throws;
 *   - launches are asynchronous on the context's stream unless the function has host
 *     outputs (documented per function), in which case it synchronises that stream;
 *   - scratch memory lives in the context (grow-only, like the reference's
 *     resize_if_needed, src/sim_propagate.cpp:561-569); callers own inputs and outputs;
 *   - one host thread per context; a context is bound to one device.
 * There is no CPU fallback: without a HIP device csc_ctx_create fails.
 */
#ifndef CASCADE_HIP_H
#define CASCADE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct csc_ctx csc_ctx;

typedef enum csc_status {
    CSC_OK = 0,
    CSC_ERR_HIP = -1,            /* a HIP runtime call failed (see csc_last_error) */
    CSC_ERR_INVALID = -2,        /* invalid argument */
    CSC_ERR_UNSUPPORTED = -3,    /* e.g. Taylor order other than 20, unknown dynamics */
    CSC_ERR_NONFINITE_AABB = -4, /* src/sim_propagate.cpp:367-383, :453-456 */
    CSC_ERR_NONFINITE_TIME = -5, /* src/sim_propagate.cpp:726-730 */
    CSC_ERR_MAX_SUBSTEPS = -6,   /* safety cap on substeps per particle per superstep */
    CSC_ERR_NO_SUBSTEPS = -7,    /* src/sim_propagate.cpp:1419-1440 */
    CSC_ERR_OVERFLOW = -8        /* an output buffer was too small (sizes returned) */
} csc_status;

/* Built-in dynamics (replace the heyoka expression vector of kw::dyn, sim.hpp:213-222): Kepler
 * (src/sim_dynamics.cpp:25-42) plus a bit mask of the perturbations of simple_earth()
 * (cascade.py/dynamics/_simple_earth.py): J2 (:169-178) and atmospheric drag with the 4-term
 * exponential density fit (:365-387, :16-46; runtime parameter 0 of a particle = BSTAR). */
enum { CSC_DYN_KEPLER = 0, CSC_DYN_J2 = 1, CSC_DYN_DRAG = 2, CSC_DYN_TAPE = 0x100 };

/* General dynamics (CSC_DYN_TAPE): the six user equations of kw::dyn (sim.hpp:213-222; what the
 * reference hands to heyoka's JIT, src/sim_basic.cpp:405-469, 553-577) plus the equation of r
 * (:461-469), lowered by include/cascade/expression.hpp to a tape of operations on truncated
 * Taylor series ("slots"). Slots 0..5 are x, y, z, vx, vy, vz, slot 6 is r; every operation
 * defines slot dst (CSC_TP_SINCOS also dst + 1 = cos). a is a slot; b is a slot (b >= 0), the
 * immediate c (b == -1) or runtime parameter -2 - b of the object (b <= -2). The propagation
 * kernel interprets the tape order by order, one object per lane. */
enum {
    CSC_TP_ADD = 1, /* a + b */
    CSC_TP_SUB,     /* a - b */
    CSC_TP_RSUB,    /* b - a (b scalar) */
    CSC_TP_MUL,     /* a * b */
    CSC_TP_DIV,     /* a / b */
    CSC_TP_RDIV,    /* b / a (b scalar) */
    CSC_TP_SQR,     /* a * a */
    CSC_TP_POW,     /* a ^ c */
    CSC_TP_SQRT,
    CSC_TP_EXP,
    CSC_TP_SINCOS,  /* dst = sin(a), dst + 1 = cos(a) */
    CSC_TP_TIME,    /* the simulation time */
    CSC_TP_CONST    /* the scalar b as a series */
};
typedef struct csc_tape_op {
    int32_t op, dst, a, b;
    double c;
} csc_tape_op;

/* Integrator + event configuration of one superstep. Replaces what sim::finalise_ctor bakes
 * into the heyoka integrators (src/sim_basic.cpp:461-578). */
typedef struct csc_prop_cfg {
    int32_t dyn_kind;      /* CSC_DYN_* */
    uint32_t order;        /* Taylor order p (20 for tol = DBL_EPSILON) */
    double mu;             /* gravitational parameter */
    double rhofac;         /* exp(-0.7 / (p - 1)) / e^2 (step-size safety factor) */
    double init_time_hi;   /* double-length simulation time at the superstep start */
    double init_time_lo;
    double delta_t;        /* superstep size = ct * nchunks (src/sim_propagate.cpp:538) */
    double ct;             /* collisional timestep */
    uint32_t nchunks;      /* n_par_ct */
    uint32_t max_substeps; /* safety cap (0 = default 1 << 20) */
    double conj_thresh;    /* AABB inflation uses max(size, conj_thresh / 2) (:354-359) */
    /* terminal events g = (kx x^2 + ky y^2) + kz z^2 - c0 (src/sim_basic.cpp:509-533) */
    int32_t has_exit;      /* event index 0 when present, direction + */
    double exit_c0;        /* R_exit^2 */
    int32_t has_reentry;   /* event index has_exit, direction - */
    double reentry_k[3];   /* {1,1,1} for a sphere; {(bc)^2, (ac)^2, (ab)^2} for an ellipsoid */
    double reentry_c0;     /* R^2 or (abc)^2 */
    /* perturbations (used when the matching bit of dyn_kind is set) */
    double J2;             /* 0.1082635854e-2 for the Earth */
    double Re;             /* equatorial radius (J2) / radius of the spherical Earth (drag altitude) */
    double dens_alpha[4];  /* rho(h) = sum_k alpha_k exp(-(h - gamma_k) beta_k), h = r - Re */
    double dens_beta[4];
    double dens_gamma[4];
    const double *d_pars;  /* DEVICE [n][npars] runtime parameters; BSTAR = d_pars[i * npars] */
    uint32_t npars;
    /* CSC_DYN_TAPE: HOST array of tape_nops operations (copied into the context when it
     * changes), the number of slots and the slot holding the right-hand side of x, y, z, vx, vy,
     * vz, r. mu, J2, Re and the density fit are not used. */
    const csc_tape_op *tape;
    uint32_t tape_nops;
    uint32_t tape_nslots;
    int32_t tape_rhs[7];
} csc_prop_cfg;

/* Per-particle status written by csc_propagate: low byte 0 time limit / 1 stopping terminal
 * event / 2 non-finite state; flag bits as below. */
enum {
    CSC_ST_TIME_LIMIT = 0,
    CSC_ST_STE = 1,
    CSC_ST_NF = 2,
    CSC_ST_FLAG_BAD_AABB = 0x100,
    CSC_ST_FLAG_BAD_TCOORD = 0x200,
    CSC_ST_FLAG_MAX_SUBSTEPS = 0x400,
    CSC_ST_FLAG_ROW_OVERFLOW = 0x1000 /* csc_propagate_rows: a particle took more than row_cap substeps */
};

/* Summary of a propagation pass (host struct filled by csc_propagate_summary). */
typedef struct csc_prop_summary {
    uint32_t flags;        /* OR of all per-particle flag bits */
    uint32_t n_nf;         /* particles with a non-finite state */
    uint32_t nf_idx;       /* earliest (lowest index on ties), src/sim_propagate.cpp:1127-1135 */
    double nf_time;
    uint32_t n_ste;        /* particles stopped by a terminal event */
    uint32_t ste_idx;      /* earliest (lowest index on ties), :1148-1153 */
    uint32_t ste_ev_idx;
    double ste_time;
    uint64_t total_substeps;
    uint32_t max_substeps; /* most substeps taken by one particle */
} csc_prop_summary;

/* Collision record (the tuple of sim_data::coll_vec, sim_data.hpp:304) and conjunction record
 * (sim::conjunction, include/cascade/sim.hpp:90-104; numpy dtype cascade.py/core.cpp:81). */
typedef struct csc_collision {
    uint64_t i, j;
    double time; /* relative to the superstep start */
} csc_collision;

typedef struct csc_conjunction {
    uint64_t i, j;
    double time; /* absolute */
    double dist;
    double state_i[6], state_j[6];
} csc_conjunction;

/* bvh_node of the reference (include/cascade/detail/sim_data.hpp:193-207), 60 bytes. */
typedef struct csc_bvh_node {
    uint32_t begin, end;
    int32_t parent, left, right;
    float lb[4], ub[4];
    uint32_t nn_level;
    int32_t split_idx;
} csc_bvh_node;

/* ---- context, memory, timing ------------------------------------------------------- */
int csc_ctx_create(int device, csc_ctx **out);
void csc_ctx_destroy(csc_ctx *ctx);
const char *csc_last_error(csc_ctx *ctx); /* ctx may be NULL: last creation error */
int csc_malloc(csc_ctx *ctx, size_t bytes, void **d_ptr);
int csc_free(csc_ctx *ctx, void *d_ptr);
int csc_memcpy_h2d(csc_ctx *ctx, void *d_dst, const void *src, size_t bytes); /* synchronous */
int csc_memcpy_d2h(csc_ctx *ctx, void *dst, const void *d_src, size_t bytes); /* synchronous */
int csc_memcpy_d2d(csc_ctx *ctx, void *d_dst, const void *d_src, size_t bytes); /* async */
/* Transfers of large PAGEABLE host arrays (the state / parameter vectors that cascade::sim shares
 * with its callers: include/cascade/sim.hpp:114-124, writable numpy views cascade.py/core.cpp:199-230)
 * through a pinned staging mirror in the context, chunk by chunk, several host threads next to the
 * DMA: ~3x the rate of a plain hipMemcpy of pageable memory. csc_upload returns once src has been
 * read (the device copy is ordered on the context's stream); csc_download is synchronous and
 * ordered after everything queued on the stream. hash (two words; may be NULL) receives
 * csc_host_hash of the host array, computed on the way: the simulation compares it at the next
 * step() to know whether its caller edited the array in between. */
int csc_upload(csc_ctx *ctx, void *d_dst, const void *src, size_t bytes, uint64_t hash[2]);
int csc_download(csc_ctx *ctx, void *dst, const void *d_src, size_t bytes, uint64_t hash[2]);
/* Multi-threaded 128-bit position-dependent checksum of a host array; no device involved. Every
 * 8-byte word is avalanched together with its position (two different 64-bit finalisers) before
 * it is summed into the two lanes: an edit is missed with probability 2^-128 whatever its
 * structure (sign flips, swapped rows, +d / -d). */
void csc_host_hash(const void *p, size_t bytes, uint64_t out[2]);
int csc_memset(csc_ctx *ctx, void *d_dst, int value, size_t bytes);           /* async */
int csc_sync(csc_ctx *ctx);
/* HIP-event stopwatch on the context's stream: ms between begin and end. */
int csc_timer_begin(csc_ctx *ctx);
int csc_timer_end(csc_ctx *ctx, float *ms);

/* Per-kernel timing with HIP events recorded on the context's stream around every kernel
 * launch (no counterpart in the reference, whose stages log spdlog stopwatches:
 * src/sim_propagate.cpp:510,1123,1340). csc_profile_get synchronises and writes one line per
 * kernel, "name launches total_ms", into buf. */
int csc_profile_enable(csc_ctx *ctx, int on);
/* Restricts the timing to one kernel (NULL or "": all of them). Two event records around every
 * launch cost ~8 us each way on the stream: with ~100 launches per superstep that is 0.8 ms, so
 * a measurement of whole supersteps times only the kernel it reports. */
int csc_profile_only(csc_ctx *ctx, const char *name);
int csc_profile_reset(csc_ctx *ctx);
int csc_profile_get(csc_ctx *ctx, char *buf, size_t cap);

/* ---- K1/K2: propagation + AABBs -----------------------------------------------------
 * Replaces batch_int_aabb / scalar_int_aabb (src/sim_propagate.cpp:650-1098), the
 * propagate callback (:710-755), compute_particle_aabb (:217-395) and heyoka's
 * propagate_for (EXTERNAL). d_state [n][7] f64. Outputs: d_nsub [n] u32 (substeps taken),
 * d_status [n] i32, d_ev_idx [n] u32, d_ev_time [n] f64 (terminal event time relative to
 * the superstep start), d_nf_last_t [n] f64, d_final_state [n][6] f64,
 * d_lbs / d_ubs [nchunks][n][4] f32. d_recs (may be NULL) [nchunks][n][16] f32: per particle and
 * chunk, the positions at the 5 times cb + s (ce - cb) / 4 (15 floats) and a radius that bounds
 * the distance of the trajectory from the chords through them plus the particle's size; input
 * of the narrow phase's geometric pre-filter (no counterpart in the reference). Asynchronous. */
int csc_propagate(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state, uint32_t *d_nsub,
                  int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t,
                  double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs);

/* Single-pass variant of csc_propagate + csc_propagate_store for dynamics whose integration is
 * too expensive to repeat (everything but plain Kepler): the propagate callback
 * (src/sim_propagate.cpp:710-755) writes the Taylor rows during the integration, as the reference
 * does, into fixed slots: particle i owns rows [i * row_cap, (i + 1) * row_cap) of d_tcs_xyz /
 * d_tcs_vr / d_tcoords (layouts as for csc_propagate_store, each array sized n * row_cap rows) and
 * d_tc_row[i] = i * row_cap. A particle that takes more than row_cap substeps sets
 * CSC_ST_FLAG_ROW_OVERFLOW in its status (csc_prop_summary.flags); the rows are then incomplete
 * and so may be that particle's boxes and records (at order 20 with the built-in dynamics they are
 * computed FROM the rows, by a kernel of their own, unless CASCADE_BOX_KERNEL=0): the caller falls
 * back on csc_propagate and csc_propagate_store. Asynchronous. */
int csc_propagate_rows(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state, uint32_t *d_nsub,
                       int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t,
                       double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs, uint32_t row_cap,
                       uint64_t *d_tc_row, double *d_tcs_xyz, double *d_tcs_vr, double *d_tcoords);

/* Reductions over the per-particle records (earliest nf state / terminal event, flags,
 * total substeps): src/sim_propagate.cpp:1127-1153. Synchronises; fills *out on the host. */
int csc_propagate_summary(csc_ctx *ctx, uint32_t n, const uint32_t *d_nsub, const int32_t *d_status,
                          const uint32_t *d_ev_idx, const double *d_ev_time, const double *d_nf_last_t,
                          csc_prop_summary *out);

/* Exclusive prefix sum u32 -> u64 with d_out[n] = total. Asynchronous. */
int csc_exclusive_scan(csc_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out);

/* Second pass: stores the Taylor coefficients (sim_data::step_data, sim_data.hpp:144-159)
 * as exact-fit CSR rows. Lane k handles particle d_order[k] (NULL = identity; the simulation
 * passes the Morton permutation of chunk 0 so that rows of spatial neighbours are neighbours
 * in memory) and owns rows [base, base + d_nsub[i]); d_tc_row[i] (out, [n] u64) = base.
 * Row r holds d_tcs_xyz [r][xyz_stride] (x, y, z coefficients back to back, zero padded),
 * d_tcs_vr [r][vr_stride] (vx, vy, vz, r) and d_tcoords [r][2] {hi, lo}; the strides (in
 * doubles) come from csc_tc_strides (64 and 84 for order 20). Asynchronous. */
int csc_tc_strides(uint32_t order, uint32_t *xyz_stride, uint32_t *vr_stride);
int csc_propagate_store(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state,
                        const uint32_t *d_order, const uint32_t *d_nsub, uint64_t *d_tc_row, double *d_tcs_xyz,
                        double *d_tcs_vr, double *d_tcoords);

/* Per-chunk global AABB (src/sim_propagate.cpp:883-920, src/detail/atomic_utils.cpp:20-55;
 * exact like verify_global_aabbs :1348-1380). d_glb / d_gub [nchunks][4] f32 (upper bound NOT
 * yet bumped). Asynchronous. */
int csc_global_aabb(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs, float *d_glb,
                    float *d_gub);

/* The ONE entry point that allocates nothing (no counterpart in the reference; for callers that
 * cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation). Same result, bit for bit, as csc_global_aabb; the partial boxes of the reduction
 * live in a static device array of 1 MiB (file scope of the library, one copy per GPU, resolved by
 * csc_ctx_create), so it calls no hipMalloc / hipFree / event creation. Limits: nchunks <=
 * csc_global_aabb_noalloc_max_chunks() (32768), otherwise CSC_ERR_INVALID with a message; n >= 1.
 * All contexts on one GPU share that array: calls that overlap on one GPU race. Asynchronous on the
 * context's stream; not timed by csc_profile_*. */
int csc_global_aabb_noalloc(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs,
                            float *d_glb, float *d_gub);
uint32_t csc_global_aabb_noalloc_max_chunks(void);

/* ---- K3/K4: Morton codes, sort, gather -------------------------------------------------
 * csc_morton_encode: src/sim_propagate.cpp:440-479 (+ disc_single_coord :94-124 and
 * MortonNDLutEncoder<4,16,8>, src/detail/mortonND_LUT.h:166-323). The ulp bump of the global
 * upper bound (:448-449) is applied internally. d_codes [nchunks][n] u64. Asynchronous; a
 * non-finite global box is reported by the next synchronising call as
 * CSC_ERR_NONFINITE_AABB through csc_check_flags. */
int csc_morton_encode(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs,
                      const float *d_glb, const float *d_gub, uint64_t *d_codes);
int csc_check_flags(csc_ctx *ctx); /* synchronises; returns and clears the pending error */

/* Stable indirect sort by code (src/sim_propagate.cpp:483-486; STABLE where oneTBB's
 * parallel_sort leaves ties unspecified) and gather (:493-504). d_vidx [nchunks][n] u32,
 * d_srt_codes [nchunks][n] u64, d_srt_lbs / d_srt_ubs [nchunks][n][4] f32. d_recs / d_srt_recs
 * (both NULL or both given): the records of csc_propagate, gathered the same way. Asynchronous. */
int csc_sort_gather(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const uint64_t *d_codes, const float *d_lbs,
                    const float *d_ubs, uint32_t *d_vidx, uint64_t *d_srt_codes, float *d_srt_lbs,
                    float *d_srt_ubs, const float *d_recs, float *d_srt_recs);

/* ---- K5: BVH ---------------------------------------------------------------------------
 * construct_bvh_trees_parallel (src/sim_bvh.cpp:89-496). The trees live in the context.
 * Asynchronous. csc_bvh_export copies the tree of one chunk in the reference's node layout
 * (canonical level order) to d_nodes (capacity cap nodes) and returns the node count in
 * *n_nodes (synchronises). */
int csc_bvh_build(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const uint64_t *d_srt_codes, const float *d_srt_lbs,
                  const float *d_srt_ubs);
int csc_bvh_export(csc_ctx *ctx, uint32_t chunk, csc_bvh_node *d_nodes, uint64_t cap, uint64_t *n_nodes);

/* ---- K6: broad phase -------------------------------------------------------------------
 * broad_phase_parallel (src/sim_broad_phase.cpp:44-218) on the trees of the last
 * csc_bvh_build. d_active [n] u8: bit 0 coll_active, bit 1 conj_active
 * (src/sim_propagate.cpp:639-647). The pairs live in the context as blocks of up to 256 {i, j}
 * u32 pairs (i < j), every block holding pairs of one group of 64 Morton-consecutive query
 * particles of one chunk, listed group by group for the narrow phase; within that the order
 * is unspecified (the reference's per-chunk vectors are unordered too, tbb::concurrent_vector).
 * Synchronises (the totals are read back); *n_pairs = total over chunks. */
int csc_broad_phase(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                    const uint32_t *d_vidx, const uint8_t *d_active, uint64_t *n_pairs);
/* The same traversal with the narrow phase's geometric exclusion test (csc_narrow_phase_prefilter;
 * no counterpart in the reference) applied to the hits of every tile while they are on chip: the
 * box-overlap pairs of src/sim_broad_phase.cpp:149-185 are counted (*n_pairs, the figure behind the
 * reference's "Average number of AABB collisions per particle per chunk", :211-213) but not stored;
 * only the pairs that the test cannot exclude leave the chip, as the candidate list that
 * csc_candidate_objects and csc_narrow_phase_finish take (*n_cand). What sim::step() calls;
 * csc_broad_phase + csc_broad_phase_get stay the stage that exports the bit-exact pair set. After
 * this call there is no pair list (csc_broad_phase_get / csc_narrow_phase return an error). d_srt_recs:
 * the sorted trajectory records of csc_sort_gather. Synchronises. */
int csc_broad_phase_filtered(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                             const uint32_t *d_vidx, const uint8_t *d_active, double conj_thresh, const float *d_srt_recs,
                             uint64_t *n_pairs, uint64_t *n_cand);
/* Node pairs expanded by the last csc_broad_phase (the work units of the dual traversal,
 * k_bp_dual.hpp: each reads two wide node records and tests up to four box pairs); 0 when the
 * per-particle walk was used. For the roofline accounting. */
uint64_t csc_broad_phase_node_pairs(csc_ctx *ctx);
/* Tiles of the last csc_broad_phase: node pairs of at most 8 x 8 particles at which the dual
 * traversal stopped and whose particle pairs k_bp_tiles tested directly (0: none). */
uint64_t csc_broad_phase_tiles(csc_ctx *ctx);
/* Multi-GPU query sharding (no counterpart in the reference, which is single-process): every
 * process holds the whole tree, and csc_broad_phase walks it only for the query groups dealt
 * to `rank` (runs of 16 groups of 64 Morton-consecutive particles, round-robin over `nranks`).
 * A pair (i < j) is reported by the walk of query i only, hence by exactly one process; the
 * union over the processes is the full pair set. The narrow phase then works on the local
 * pairs. Default (0, 1). */
int csc_set_shard(csc_ctx *ctx, uint32_t rank, uint32_t nranks);
/* Multi-GPU chunk sharding (no counterpart in the reference; its chunks are an outer loop of one
 * process, src/sim_propagate.cpp:533-538): the chunk arrays handed to the stages may be a block
 * [first, first + nchunks) of a superstep of nchunks_global chunks. Sort, tree and traversal do
 * not care; the narrow phase derives each chunk's time window (get_chunk_begin_end,
 * src/sim_basic.cpp:87-103) from the global index. nchunks_global = 0 (default): no mapping. */
int csc_set_chunk_map(csc_ctx *ctx, uint32_t first, uint32_t nchunks_global);
/* Diagnostic (measurement only): re-runs the traversal of the last csc_broad_phase without
 * writing pairs and returns the number of node AABB tests, i.e. the reference's "nodes
 * visited" (every node popped from the stack of src/sim_broad_phase.cpp:121-196), summed over
 * queries and chunks. Synchronises. */
int csc_broad_phase_stats(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                          const uint32_t *d_vidx, const uint8_t *d_active, uint64_t *nodes_visited);
/* Copies the pairs of one chunk to the host (for parity checks). *count = number of pairs;
 * at most cap pairs are written. Synchronises. */
int csc_broad_phase_get(csc_ctx *ctx, uint32_t chunk, uint32_t *pairs, uint64_t cap, uint64_t *count);

/* ---- K7: narrow phase ------------------------------------------------------------------
 * narrow_phase_parallel (src/sim_narrow_phase.cpp:495-1023) over the pair list of the last
 * csc_broad_phase. nchunks / delta_t are the values AFTER a possible terminal-event shrink
 * (:560). d_srt_recs: the sorted records of csc_sort_gather; when given, a rigorous geometric
 * pre-filter (chord distance of the sampled trajectories, see k_np_prefilter) discards the
 * pairs that cannot come within max(r_i + r_j, conj_thresh) of each other before the
 * reference's polynomial test; NULL runs that test on every pair (same events either way).
 * Results live in the context; synchronises; counts returned. csc_narrow_phase_candidates:
 * number of pairs that reached the polynomial test in the last call. */
int csc_narrow_phase(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                     double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                     const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub, const double *d_tcs_xyz,
                     const double *d_tcs_vr, const double *d_tcoords, const float *d_srt_recs, uint64_t *n_coll,
                     uint64_t *n_conj, uint32_t *warn_flags);
uint64_t csc_narrow_phase_candidates(csc_ctx *ctx);
/* Diagnostic: passes the last narrow-phase call made over its lists. The reference grows
 * concurrent vectors (src/sim_narrow_phase.cpp:985-1011); here the output lists have capacities
 * that follow the previous superstep, and a list that overflows is grown and the pass repeated
 * (results are recomputed from scratch). 1 = nothing overflowed. */
uint32_t csc_narrow_phase_attempts(csc_ctx *ctx);
/* The same narrow phase in two calls, for callers that produce the Taylor rows on demand: the
 * reference stores the rows of every particle (the propagate callback, src/sim_propagate.cpp:710-755)
 * although narrow_phase_parallel (src/sim_narrow_phase.cpp:495-1023) reads only those of the pairs
 * it tests, and the pre-filter leaves ~0.1 % of the pairs. csc_narrow_phase_prefilter runs the
 * pre-filter over the pairs of the last broad phase and keeps the candidates on the device;
 * csc_candidate_objects lists the particles they involve (d_list [n] receives n_list ascending
 * indices; n_rows, when not NULL, the sum of their substep counts), so that csc_propagate_store on that list
 * writes just their rows; csc_narrow_phase_finish then runs the polynomial tests on the
 * candidates. Same events as csc_narrow_phase. */
int csc_narrow_phase_prefilter(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double conj_thresh, const float *d_srt_recs,
                               uint64_t *n_cand);
int csc_candidate_objects(csc_ctx *ctx, uint32_t n, const uint32_t *d_nsub, uint32_t *d_list, uint32_t *n_list,
                          uint64_t *n_rows);
int csc_narrow_phase_finish(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                            double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                            const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub,
                            const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords, uint64_t *n_coll,
                            uint64_t *n_conj, uint32_t *warn_flags);
/* Same, on a caller-provided pair list d_pairs [npairs][2] with per-chunk row offsets
 * pair_off [nchunks + 1] (host array). For parity checks on injected pairs. */
int csc_narrow_phase_pairs(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                           double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                           const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub,
                           const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords,
                           const uint32_t *d_pairs, const uint64_t *pair_off, uint64_t *n_coll, uint64_t *n_conj,
                           uint32_t *warn_flags);
int csc_narrow_phase_get(csc_ctx *ctx, csc_collision *coll, uint64_t coll_cap, csc_conjunction *conj,
                         uint64_t conj_cap); /* host copies, sorted by (time, i, j) */

/* The same lists without a copy: pointers into the context's host arrays (pinned memory for the
 * conjunctions), valid until the next narrow phase on this context. */
int csc_narrow_phase_view(csc_ctx *ctx, const csc_collision **coll, uint64_t *n_coll, const csc_conjunction **conj,
                          uint64_t *n_conj);

/* Conjunction-heavy supersteps (10M objects in shells: 1.4e7 records = 1.75 GB per superstep) are
 * bound by what happens to the records on the way out (src/sim_narrow_phase.cpp:985-1011,
 * src/sim_propagate.cpp:1644-1701 append them to a host vector). Lazy mode: a list that was sorted
 * on the device (more than 32768 records, no two with the same time) stays there;
 * csc_narrow_phase_view / _get fetch the host copy when somebody asks. csc_narrow_phase_counts gives
 * the collisions and the number of conjunctions without fetching anything. csc_conj_store_append
 * adds the first `count` records of that device list to a device-resident store that grows by
 * doubling; csc_conj_store_download copies the whole store to `dst` (pageable memory is fine:
 * pipelined like csc_download) and empties it; csc_conj_store_reset empties (and optionally
 * frees) it. cascade::sim keeps the conjunctions of successive steps there and materialises
 * get_conjunctions() on access. */
int csc_narrow_phase_lazy(csc_ctx *ctx, int on);
int csc_narrow_phase_counts(csc_ctx *ctx, const csc_collision **coll, uint64_t *n_coll, uint64_t *n_conj, int *conj_on_device);
int csc_conj_store_append(csc_ctx *ctx, uint64_t count);
uint64_t csc_conj_store_size(csc_ctx *ctx);
int csc_conj_store_download(csc_ctx *ctx, csc_conjunction *dst);
int csc_conj_store_reset(csc_ctx *ctx, int release);

/* ---- propagation sharded over GPUs ("particles are sharded for propagation, chunk
 * AABBs and Morton codes are gathered with RCCL all-gather") -------------------------------
 * No counterpart in the reference, which is single-process. csc_propagate_range is csc_propagate
 * for the particles [first, first + count) only (the arrays keep their full size n). What it
 * leaves per particle -- substep count, status, event, final state, the per-chunk boxes and
 * trajectory records -- is packed into one record of csc_shard_record_bytes(nchunks) bytes
 * (csc_shard_pack: particle first + k -> record k), so that ONE all-gather of equal-sized blocks
 * (per_rank = ceil(n / nranks) records from every process, rank r owning the particles from
 * r * per_rank) moves it; csc_shard_unpack scatters the gathered records (record i = particle i)
 * back into the arrays. The Taylor rows do not travel: they are produced on demand
 * (csc_narrow_phase_prefilter ... csc_narrow_phase_finish). Asynchronous. */
int csc_propagate_range(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, uint32_t first, uint32_t count,
                        const double *d_state, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                        double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs);
/* The same with dense-output rows written on the way into row_cap fixed slots per particle of the
 * range, kept inside the context: the pass then runs the several-lanes-per-object kernel where
 * the dynamics come with their tape (every right-hand side except plain Kepler) and takes the
 * boxes and trajectory records from the rows. A particle with more substeps than row_cap raises
 * CSC_ST_FLAG_ROW_OVERFLOW in its status (the caller repeats the pass with the substep count the
 * summary reports). */
int csc_propagate_range_rows(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, uint32_t first, uint32_t count,
                             const double *d_state, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                             double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs, uint32_t row_cap);
/* Diagnostic, no GPU work and no context: the several-lanes-per-object kernel does not run the
 * tape as given but a FUSED one (sums, differences, scalings and products folded into
 * sum-of-products operations, scheduled into rows of independent operations for `ng` = 1, 2, 4, 8
 * groups of 8 lanes per object; 0: the layout the library would pick). This builds that schedule
 * and executes it on the host for one state (x, y, z, vx, vy, vz, r), operation by operation as
 * the kernel does: out_coeffs[7][order + 1] are the Taylor coefficients of the state variables,
 * to be compared with the plain tape's (heyoka's taylor_step in the reference,
 * src/sim_basic.cpp:553-577 EXTERNAL; oracle/orc.hpp::tape_compute_tcs). out_stats (8 values, may
 * be null): fused operations, series, dependency depth, rows, groups, LDS bytes per workgroup,
 * waves per workgroup, lanes per operation. Returns 0, 1 (invalid arguments), 2 (the tape is
 * not in single-assignment form or does not fit: the kernel is not used for it) or 3 (pars is
 * null and the schedule reads a runtime parameter somewhere, an unused place of a row included). */
int csc_wave_plan_eval(const void *tape /* csc_tape_op[nops] */, uint32_t nops, uint32_t nslots, const int32_t *rhs7, int ng,
                       uint32_t order, const double *state7, const double *pars, double time, double *out_coeffs,
                       uint32_t *out_stats);
/* What a pass with rows (csc_propagate_rows, csc_propagate_range_rows) over n particles would run
 * for this configuration: 0 the one-lane-per-object kernels, 1 k_propagate_wave interpreting the
 * fused schedule of cfg->tape, 2 one of its compiled tapes, 3 a tape compiled at run time (hipRTC, at
 * the first launch; environment CASCADE_WAVE_JIT=0 switches that off). A caller that guesses row slots for the
 * first superstep, or shards the propagation, asks here so that every pass of a superstep (and
 * every process) runs the same arithmetic. No counterpart in the reference. */
int csc_wave_would_run(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n);
/* Build tool, no GPU work: the fused schedule of the tape with one operation per row as
 * straight-line HIP source (a specialisation wave_compiled_rows<id>, see csrc/k_propagate_wave.hpp),
 * from which profiles/gen_compiled_tapes.py writes csrc/gen/compiled_tapes.hpp for the dynamics
 * known when the library is built -- the counterpart of heyoka's LLVM JIT in the reference
 * (src/sim_basic.cpp:553-577 EXTERNAL), ahead of time. Returns the bytes the text needs (it is
 * written when it fits `cap`; out may be null), 0 when the tape has no such schedule; *hash: what
 * identifies the schedule at run time. */
uint64_t csc_wave_emit_source(const void *tape /* csc_tape_op[nops] */, uint32_t nops, uint32_t nslots, const int32_t *rhs7,
                              uint32_t order, int id, char *out, uint64_t cap, uint64_t *hash);
size_t csc_shard_record_bytes(uint32_t nchunks);
int csc_shard_pack(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t nchunks, const uint32_t *d_nsub,
                   const int32_t *d_status, const uint32_t *d_ev_idx, const double *d_ev_time, const double *d_nf_last_t,
                   const double *d_final_state, const float *d_lbs, const float *d_ubs, const float *d_recs, void *d_records);
int csc_shard_unpack(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const void *d_records, uint32_t *d_nsub, int32_t *d_status,
                     uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t, double *d_final_state, float *d_lbs,
                     float *d_ubs, float *d_recs);
/* The same exchange when the chunks are dealt to the processes as well (csc_set_chunk_map): the
 * process that owns a block of chunks needs the boxes and trajectory records of THOSE chunks
 * only, so an all-to-all replaces the all-gather and moves 80 + 96 * (chunks of the block) bytes
 * per particle instead of 80 + 96 C. csc_shard_block: the block of `rank` -- blocks of
 * ceil(nchunks / nranks) consecutive chunks in rank order until the chunks run out (count may
 * be 0). csc_shard_pack_blocks writes nranks segments into d_records, segment r =
 * per_rank records of csc_shard_record_bytes(count_r) bytes holding the chunks of block r
 * (particle first + k -> record k of every segment; segment r starts at
 * per_rank * (80 r + 96 first_r) bytes); after an all-to-all in which every process sends
 * segment r to process r, csc_shard_unpack_block scatters the nranks equal segments received
 * (record i = particle i, chunks [chunk_first, chunk_first + chunk_count)). Asynchronous. */
int csc_shard_block(uint32_t nchunks, uint32_t nranks, uint32_t rank, uint32_t *first, uint32_t *count);
int csc_shard_pack_blocks(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t per_rank, uint32_t nchunks,
                          uint32_t nranks, const uint32_t *d_nsub, const int32_t *d_status, const uint32_t *d_ev_idx,
                          const double *d_ev_time, const double *d_nf_last_t, const double *d_final_state, const float *d_lbs,
                          const float *d_ubs, const float *d_recs, void *d_records);
int csc_shard_unpack_block(csc_ctx *ctx, uint32_t n, uint32_t nchunks, uint32_t chunk_first, uint32_t chunk_count,
                           const void *d_records, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                           double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs);

/* ---- K8: dense output ------------------------------------------------------------------
 * dense_propagate (src/sim_propagate.cpp:1386-1503): state of every particle at time t
 * (relative to the superstep start) into d_out [n][6]. Asynchronous; a particle without
 * substeps raises CSC_ERR_NO_SUBSTEPS through csc_check_flags. */
int csc_dense_propagate(csc_ctx *ctx, uint32_t n, uint32_t order, double t, const uint64_t *d_tc_row,
                        const uint32_t *d_nsub, const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords,
                        double *d_out);

/* state[:, 0:6] <- final_state (copy_from_final_state, src/sim_propagate.cpp:1610-1630). */
int csc_copy_final_state(csc_ctx *ctx, uint32_t n, const double *d_final_state, double *d_state);

/* Activity flags (src/sim_propagate.cpp:639-647): d_active[i] = coll | conj << 1. Whitelists
 * are host arrays of particle indices (NULL/0 = empty = everybody). */
int csc_activity_flags(csc_ctx *ctx, uint32_t n, const double *d_state, double min_coll_radius, int with_conj,
                       const uint64_t *coll_wl, uint64_t n_coll_wl, const uint64_t *conj_wl, uint64_t n_conj_wl,
                       uint8_t *d_active);

/* remove_particles (src/sim_basic.cpp:280-338) on the device copies: drops the rows idx[0..n_idx)
 * (host array, sorted, unique, < n) from d_state [n][7] and, when npars != 0, d_pars [n][npars],
 * keeping the order of the others; the compacted arrays hold *n_out = n - n_idx rows. For
 * callers that keep the state resident between supersteps (the removal after each reentry or
 * collision, benchmark/2022_leo_pop.cpp:311-337). Synchronous. */
int csc_remove_particles(csc_ctx *ctx, uint32_t n, const uint64_t *idx, uint64_t n_idx, double *d_state, uint32_t npars,
                         double *d_pars, uint32_t *n_out);

#ifdef __cplusplus
}
#endif

#endif
