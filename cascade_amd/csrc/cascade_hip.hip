// cascade_amd/csrc/cascade_hip.hip
//
// libcascade_hip.so: C ABI (include/cascade_hip.h) over the gfx950 kernels. Unity build:
// all kernels are compiled in this translation unit.
//
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared \
//         -Iinclude cascade_amd/csrc/cascade_hip.hip -o cascade_amd/libcascade_hip.so

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "cascade_hip.h"

#include "dev_math.hpp"
#include "k_bvh.hpp"
#include "k_bp_dual.hpp"
#include "k_morton.hpp"
#include "k_narrow.hpp"
#include "k_propagate.hpp"
#include "k_scan_sort.hpp"
#include "order_api.hpp"
#include "wave_api.hpp"

// Scratch of csc_global_aabb_noalloc (the one entry point that allocates nothing): partial
// (min, max) boxes of the block reduction. A file-scope device array: every GPU the library runs
// on holds its own copy at its own address until the process ends. In host code the NAME is only
// a stand-in for hipGetSymbolAddress (csc_ctx_create); kernels get the address it returned.
constexpr size_t CSC_NA_SCRATCH_F4 = 65536; // float4 slots = 1 MiB
static __device__ __align__(256) float4 scratch[CSC_NA_SCRATCH_F4];

// Taylor orders other than 20: one translation unit each (cascade_order.hip, order_api.hpp).
#define CSC_FOR_EACH_ORDER(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(21) X(22) X(23) X(24)
#define CSC_DECLARE_ORDER(p) extern "C" const csc::order_api *csc_order_api_##p();
CSC_FOR_EACH_ORDER(CSC_DECLARE_ORDER)
#undef CSC_DECLARE_ORDER
extern "C" __attribute__((visibility("hidden"))) const csc::order_api *csc_order_api(int order)
{
    switch (order) {
#define CSC_ORDER_CASE(p)                                                                                              \
    case p:                                                                                                            \
        return csc_order_api_##p();
        CSC_FOR_EACH_ORDER(CSC_ORDER_CASE)
#undef CSC_ORDER_CASE
        default:
            return nullptr;
    }
}

namespace
{

std::string g_create_error;

struct dbuf {
    void *p = nullptr;
    size_t cap = 0;
};

} // namespace

struct csc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::map<std::string, dbuf> bufs;
    int *d_flags = nullptr; // [0] morton non-finite box, [1] dense: no substeps, [2] broad-phase stack overflow, [3] store pass != pass 1
    // The flags were read clean (by the dual broad phase, together with its counters) and no kernel
    // that can raise one has been launched since: csc_check_flags need not synchronise.
    bool flags_fresh = false;
    // BVH state of the last csc_bvh_build
    csc::bvh_bufs bvh{};
    uint32_t bvh_n = 0, bvh_chunks = 0;
    bool bvh_packed = false; // wide traversal nodes built for the current trees
    // broad phase state
    uint64_t bp_total = 0;
    uint32_t bp_blocks = 0, bp_cap_per_shard = 0; // pair blocks used / pool capacity
    csc::pair_lists bp_pl{};
    const uint2 *bp_order = nullptr; // group-major {block, time chunk} list of the last broad phase
    const uint32_t *bp_vidx = nullptr; // the permutation the pair positions refer to
    uint32_t bp_n = 0, bp_chunks = 0;
    uint32_t shard_rank = 0, shard_n = 1; // query sharding of the broad phase (csc_set_shard)
    uint32_t chunk_first = 0, chunk_global = 0; // csc_set_chunk_map: the narrow phase's chunk c is chunk chunk_first + c of chunk_global
    // dual traversal (k_bp_dual.hpp): flat pair lists of the last broad phase, capacities that
    // follow the last superstep
    bool bp_flat = false;
    bool bp_fused = false; // the last broad phase filtered its pairs on chip and kept no pair list
    const unsigned long long *bp_flat_pairs = nullptr;
    uint32_t bp_flat_cap = 0;
    uint32_t bp_flat_cnt[csc::DT_SHARDS] = {};
    bool bvh_canonical = false; // the nodes of ctx->bvh are in the reference's order (csc_bvh_export)
    uint32_t dt_fr_cap = 0, dt_pair_cap = 0, dt_tile_cap = 0;
    uint64_t bp_tiles = 0; // tiles of the last dual traversal
    uint64_t bp_node_pairs = 0;
    int bp_mode = -1; // -1: read CASCADE_BP_MODE once; 0 per-particle walks, 1 dual traversal
    unsigned long long *d_bp_counter = nullptr;
    // narrow phase results (host copies)
    std::vector<csc_collision> coll;
    // Conjunction records of the last narrow phase: pinned host memory (there can be millions
    // of 128-byte records per superstep), grow-only.
    csc_conjunction *conj = nullptr;
    size_t conj_cap = 0, conj_count = 0;
    // Lazy mode (csc_narrow_phase_lazy): a large, device-sorted conjunction list stays on the device
    // (conj_dev, in "np_conj_sorted") until somebody asks for the host copy; csc_conj_store_* keep the
    // records of successive supersteps in a device-resident store.
    bool conj_lazy = false, conj_host_valid = true;
    const csc_conjunction *conj_dev = nullptr;
    csc_conjunction *cstore = nullptr;
    size_t cstore_cap = 0, cstore_n = 0;
    uint64_t np_candidates = 0; // pairs that reached the polynomial test in the last narrow phase
    uint32_t np_attempts = 0;   // passes the last narrow-phase call needed (1 = no list overflowed)
    int dt_levels = 0;          // levels with a non-empty frontier in the last dual traversal
    std::vector<uint64_t> dt_level_items; // frontier size of every level of the last dual traversal (grid sizing)
    // csc_narrow_phase_prefilter -> csc_narrow_phase_finish: the candidate list left on the device
    const uint4 *np_cand_list = nullptr;
    uint64_t np_cand_count = 0;
    bool np_cand_valid = false;
    // CASCADE_NP_CONJ_FAST=0: conjunction searches through the general root isolation only (tests).
    bool np_conj_fast = std::getenv("CASCADE_NP_CONJ_FAST") == nullptr || std::atoi(std::getenv("CASCADE_NP_CONJ_FAST")) != 0;
    // per-kernel HIP-event timing (csc_profile_*)
    struct ktimer {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
        size_t used = 0;
        double total_ms = 0;
        uint64_t launches = 0;
    };
    std::map<std::string, ktimer> kt;
    bool profiling = false;
    std::string profile_only; // when not empty: only this kernel is timed
    std::vector<csc_tape_op> tape_host; // the tape currently in the "tape_ops" buffer
    // k_propagate_wave: the schedule of the tape it was built from (wave_tape), on the device in
    // "wave_ops" / "wave_ops0". wave_mode: -1 read CASCADE_WAVE once; 0 off; 1 on.
    std::vector<csc_tape_op> wave_tape;
    csc::wave_plan wave_plan;
    bool wave_fits = false;
    // ... and, when the schedule is one of the compiled tapes (cascade_tapes.hip), its plan with one operation per row.
    csc::wave_plan wave_plan1;
    int wave_compiled = 0;
    bool wave_uploaded = false; // (the interpreted schedule is on the device)
    // A schedule without an ahead-of-time tape is compiled at first use (hipRTC; CASCADE_WAVE_JIT=0: never);
    // wave_jit: 0 no, 1 the plan with one operation per row fits, -1 the compilation failed (interpreted rows from then on).
    int wave_jit = 0;
    int wave_mode = -1;
    bool pass1_wave = false; // the last pass 1 ran k_propagate_wave: the store pass then does too (same substeps)
    float4 *na_scratch = nullptr;       // device address of `scratch` on this context's device
    // csc_upload / csc_download: pinned staging mirror (grow-only), copy stream, one event per chunk
    void *stage = nullptr;
    size_t stage_cap = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> stage_ev;
    void kt_begin(const char *name)
    {
        if (!profiling || (!profile_only.empty() && profile_only != name)) {
            return;
        }
        auto &t = kt[name];
        if (t.used == t.evs.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            t.evs.emplace_back(a, b);
        }
        (void)hipEventRecord(t.evs[t.used].first, stream);
    }
    void kt_end(const char *name)
    {
        if (!profiling || (!profile_only.empty() && profile_only != name)) {
            return;
        }
        auto &t = kt[name];
        (void)hipEventRecord(t.evs[t.used].second, stream);
        ++t.used;
    }
    void kt_collect()
    {
        (void)hipStreamSynchronize(stream);
        for (auto &kv : kt) {
            auto &t = kv.second;
            for (size_t i = 0; i < t.used; ++i) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, t.evs[i].first, t.evs[i].second) == hipSuccess) {
                    t.total_ms += ms;
                    ++t.launches;
                }
            }
            t.used = 0;
        }
    }

    int fail(hipError_t e, const char *what)
    {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return CSC_ERR_HIP;
    }
    int invalid(const char *what)
    {
        err = what;
        return CSC_ERR_INVALID;
    }
    // Grow-only named scratch buffer.
    template <typename T>
    T *buf(const char *name, size_t count)
    {
        auto &b = bufs[name];
        const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
        if (b.cap < bytes) {
            if (b.p) {
                (void)hipStreamSynchronize(stream);
                (void)hipFree(b.p);
                b.p = nullptr;
                b.cap = 0;
            }
            const size_t want = bytes + bytes / 4;
            const hipError_t e = hipMalloc(&b.p, want);
            if (e != hipSuccess) {
                err = std::string("hipMalloc(") + name + ", " + std::to_string(want) + "): " + hipGetErrorString(e);
                return nullptr;
            }
            b.cap = want;
        }
        return static_cast<T *>(b.p);
    }
};

#define CSC_HIP(call)                                                                                                  \
    do {                                                                                                               \
        const hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                        \
            return ctx->fail(e_, #call);                                                                               \
        }                                                                                                              \
    } while (0)

#define CSC_LAUNCH_CHECK()                                                                                             \
    do {                                                                                                               \
        const hipError_t e_ = hipGetLastError();                                                                       \
        if (e_ != hipSuccess) {                                                                                        \
            return ctx->fail(e_, "kernel launch");                                                                     \
        }                                                                                                              \
    } while (0)

#define KT(name, ...)                                                                                                  \
    do {                                                                                                               \
        ctx->kt_begin(name);                                                                                           \
        __VA_ARGS__;                                                                                                   \
        ctx->kt_end(name);                                                                                             \
    } while (0)

#define CSC_BUF(var, T, name, count)                                                                                   \
    T *var = ctx->buf<T>(name, count);                                                                                 \
    if (!var) {                                                                                                        \
        return CSC_ERR_HIP;                                                                                            \
    }

namespace
{

inline unsigned cdiv(uint64_t a, uint64_t b)
{
    return static_cast<unsigned>((a + b - 1) / b);
}

// ---- small kernels that do not deserve a header ---------------------------------------

struct sum_partial {
    uint32_t flags, n_nf, nf_idx, n_ste;
    double nf_time;
    uint32_t ste_idx, ste_ev;
    double ste_time;
    uint64_t nsub;
    uint32_t max_nsub, pad;
};

constexpr int SUM_BLOCK = 256;

__global__ __launch_bounds__(SUM_BLOCK) void k_prop_summary(uint32_t n, const uint32_t *nsub, const int32_t *status,
                                                            const uint32_t *ev_idx, const double *ev_time,
                                                            const double *nf_last_t, sum_partial *out)
{
    __shared__ sum_partial sh[SUM_BLOCK];
    sum_partial p;
    p.flags = 0;
    p.n_nf = 0;
    p.nf_idx = 0xffffffffu;
    p.n_ste = 0;
    p.nf_time = 0;
    p.ste_idx = 0xffffffffu;
    p.ste_ev = 0;
    p.ste_time = 0;
    p.nsub = 0;
    p.max_nsub = 0;
    p.pad = 0;
    for (uint32_t i = blockIdx.x * SUM_BLOCK + threadIdx.x; i < n; i += gridDim.x * SUM_BLOCK) {
        const int32_t st = status[i];
        p.flags |= static_cast<uint32_t>(st) & 0xffffff00u;
        p.nsub += nsub[i];
        p.max_nsub = p.max_nsub > nsub[i] ? p.max_nsub : nsub[i];
        const int code = st & 0xff;
        if (code == csc::ST_NF) {
            const double t = nf_last_t[i];
            if (p.n_nf == 0 || t < p.nf_time || (t == p.nf_time && i < p.nf_idx)) {
                p.nf_time = t;
                p.nf_idx = i;
            }
            ++p.n_nf;
        } else if (code == csc::ST_STE) {
            const double t = ev_time[i];
            if (p.n_ste == 0 || t < p.ste_time || (t == p.ste_time && i < p.ste_idx)) {
                p.ste_time = t;
                p.ste_idx = i;
                p.ste_ev = ev_idx[i];
            }
            ++p.n_ste;
        }
    }
    sh[threadIdx.x] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < SUM_BLOCK; ++k) {
            const sum_partial &o = sh[k];
            p.flags |= o.flags;
            p.nsub += o.nsub;
            p.max_nsub = p.max_nsub > o.max_nsub ? p.max_nsub : o.max_nsub;
            if (o.n_nf) {
                if (p.n_nf == 0 || o.nf_time < p.nf_time || (o.nf_time == p.nf_time && o.nf_idx < p.nf_idx)) {
                    p.nf_time = o.nf_time;
                    p.nf_idx = o.nf_idx;
                }
                p.n_nf += o.n_nf;
            }
            if (o.n_ste) {
                if (p.n_ste == 0 || o.ste_time < p.ste_time || (o.ste_time == p.ste_time && o.ste_idx < p.ste_idx)) {
                    p.ste_time = o.ste_time;
                    p.ste_idx = o.ste_idx;
                    p.ste_ev = o.ste_ev;
                }
                p.n_ste += o.n_ste;
            }
        }
        out[blockIdx.x] = p;
    }
}

// Conjunction records -> sortable keys: the bit pattern of the (absolute) time, mapped so that
// unsigned order = numeric order.
__global__ __launch_bounds__(256) void k_conj_keys(uint32_t n, const csc::conj_rec *c, uint64_t *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(c[i].time));
    keys[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Number of adjacent equal keys in a sorted list (records with equal times still need their
// (i, j) order from the host; almost always there are none).
__global__ __launch_bounds__(256) void k_count_ties(uint32_t n, const uint64_t *keys, uint32_t *count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1u < n && keys[i] == keys[i + 1u]) {
        atomicAdd(count, 1u);
    }
}

// dst[k] = src[idx[k]] for 128-byte records, 8 lanes per record (one 16-byte piece each).
__global__ __launch_bounds__(256) void k_conj_gather(uint32_t n, const uint4 *src, const uint32_t *idx, uint4 *dst)
{
    const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t rec = t >> 3;
    if (rec >= n) {
        return;
    }
    dst[t] = src[static_cast<uint64_t>(idx[rec]) * 8u + (t & 7u)];
}

// Propagation sharded over GPUs: everything pass 1 leaves per particle, as ONE record of
// 80 + 96 C bytes, so that one all-gather moves it: {nsub, status, ev_idx, pad, ev_time, nf_last_t,
// final_state[6]} then per chunk {lb, ub, trajectory record[4]} (float4 each).
struct shard_arrays {
    uint32_t *nsub;
    int32_t *status;
    uint32_t *ev_idx;
    double *ev_time, *nf_last_t, *final_state;
    float4 *lbs, *ubs, *recs; // recs may be null
};

__host__ __device__ inline size_t shard_record_bytes(uint32_t nchunks)
{
    return 80u + 96u * static_cast<size_t>(nchunks);
}

// The block of chunks a process owns when the chunks are dealt out (csc_shard_block): blocks of
// per_c = ceil(nchunks / nranks) consecutive chunks, in rank order, until the chunks run out.
__host__ __device__ inline void shard_block(uint32_t nchunks, uint32_t per_c, uint32_t rank, uint32_t &cb, uint32_t &ce)
{
    const uint64_t b = static_cast<uint64_t>(rank) * per_c;
    cb = b < nchunks ? static_cast<uint32_t>(b) : nchunks;
    ce = nchunks - cb < per_c ? nchunks : cb + per_c;
}

// UNPACK = false: particle first + k -> record k of segment blockIdx.y of buf; segment y holds
// records with the chunks of block y only (per slots each; one segment with every chunk when
// per_c = nchunks: the all-gather form). UNPACK = true: record k of buf, which holds the chunks
// [cb_u, ce_u) -> particle first + k (equal segments of per records in rank order: record k is
// particle k).
// One thread per 16-byte piece of a record (5 of the header, 6 per chunk), ordered so that the
// WRITES of consecutive threads are consecutive: the pieces of one record when packing, the same
// piece of consecutive particles when unpacking (a record per thread wrote 16 bytes into 64
// different lines per instruction).
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_shard_records(uint32_t first, uint32_t count, uint32_t n, uint32_t per, uint32_t nchunks,
                                                       uint32_t per_c, uint32_t cb_u, uint32_t ce_u, shard_arrays a, unsigned char *buf)
{
    uint32_t cb = cb_u, ce = ce_u;
    if constexpr (!UNPACK) {
        shard_block(nchunks, per_c, blockIdx.y, cb, ce);
        buf += static_cast<size_t>(per) * (80u * static_cast<size_t>(blockIdx.y) + 96u * static_cast<size_t>(cb));
    }
    const uint32_t npieces = 5u + 6u * (ce - cb);
    const size_t rec_bytes = shard_record_bytes(ce - cb);
    const uint64_t total = static_cast<uint64_t>(count) * npieces;
    for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        uint32_t k, j;
        if constexpr (UNPACK) {
            j = static_cast<uint32_t>(idx / count);
            k = static_cast<uint32_t>(idx % count);
        } else {
            k = static_cast<uint32_t>(idx / npieces);
            j = static_cast<uint32_t>(idx % npieces);
        }
        const uint32_t i = first + k;
        uint4 *piece = reinterpret_cast<uint4 *>(buf + static_cast<size_t>(k) * rec_bytes + static_cast<size_t>(j) * 16u);
        if (j >= 5u) {
            const uint32_t c = cb + (j - 5u) / 6u, t = (j - 5u) % 6u;
            const size_t e = static_cast<size_t>(c) * n + i;
            float4 *arr = t == 0u ? a.lbs + e : (t == 1u ? a.ubs + e : (a.recs ? a.recs + e * 4u + (t - 2u) : nullptr));
            if constexpr (!UNPACK) {
                const float4 v = arr ? *arr : make_float4(0.f, 0.f, 0.f, 0.f);
                *piece = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
            } else if (arr) {
                const uint4 v = *piece;
                *arr = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        } else if (j == 0u) {
            if constexpr (!UNPACK) {
                *piece = make_uint4(a.nsub[i], static_cast<uint32_t>(a.status[i]), a.ev_idx[i], 0u);
            } else {
                const uint4 v = *piece;
                a.nsub[i] = v.x;
                a.status[i] = static_cast<int32_t>(v.y);
                a.ev_idx[i] = v.z;
            }
        } else {
            // two doubles: {ev_time, nf_last_t} or a pair of the final state
            double *d0 = j == 1u ? a.ev_time + i : a.final_state + static_cast<size_t>(i) * 6u + 2u * (j - 2u);
            double *d1 = j == 1u ? a.nf_last_t + i : d0 + 1;
            double2 *pd = reinterpret_cast<double2 *>(piece);
            if constexpr (!UNPACK) {
                *pd = make_double2(*d0, *d1);
            } else {
                const double2 v = *pd;
                *d0 = v.x;
                *d1 = v.y;
            }
        }
    }
}

// Unpacking through LDS: a workgroup copies the records of `pb` consecutive particles into LDS
// with coalesced 16-byte loads, then writes them out piece by piece with the particle index
// running fastest, so that both the reads and the writes of consecutive threads are consecutive.
__global__ __launch_bounds__(256) void k_shard_unpack_staged(uint32_t count, uint32_t n, uint32_t cb, uint32_t ce, uint32_t pb,
                                                             shard_arrays a, const unsigned char *buf)
{
    extern __shared__ uint4 s_rec[];
    const uint32_t npieces = 5u + 6u * (ce - cb);
    for (uint32_t k0 = blockIdx.x * pb; k0 < count; k0 += gridDim.x * pb) {
        const uint32_t m = count - k0 < pb ? count - k0 : pb;
        const uint4 *src = reinterpret_cast<const uint4 *>(buf + static_cast<size_t>(k0) * npieces * 16u);
        for (uint32_t t = threadIdx.x; t < m * npieces; t += blockDim.x) {
            s_rec[t] = src[t];
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < m * npieces; t += blockDim.x) {
            const uint32_t j = t / m, kk = t % m;
            const uint32_t i = k0 + kk;
            const uint4 v = s_rec[kk * npieces + j];
            if (j >= 5u) {
                const uint32_t c = cb + (j - 5u) / 6u, tt = (j - 5u) % 6u;
                const size_t e = static_cast<size_t>(c) * n + i;
                float4 *arr = tt == 0u ? a.lbs + e : (tt == 1u ? a.ubs + e : (a.recs ? a.recs + e * 4u + (tt - 2u) : nullptr));
                if (arr) {
                    *arr = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            } else if (j == 0u) {
                a.nsub[i] = v.x;
                a.status[i] = static_cast<int32_t>(v.y);
                a.ev_idx[i] = v.z;
            } else {
                const double d0 = __hiloint2double(static_cast<int>(v.y), static_cast<int>(v.x));
                const double d1 = __hiloint2double(static_cast<int>(v.w), static_cast<int>(v.z));
                if (j == 1u) {
                    a.ev_time[i] = d0;
                    a.nf_last_t[i] = d1;
                } else {
                    double *d = a.final_state + static_cast<size_t>(i) * 6u + 2u * (j - 2u);
                    d[0] = d0;
                    d[1] = d1;
                }
            }
        }
        __syncthreads();
    }
}

// Packing through LDS, the mirror image: the arrays are read with the particle index running
// fastest, the records leave LDS as one contiguous run. blockIdx.y = destination segment.
__global__ __launch_bounds__(256) void k_shard_pack_staged(uint32_t first, uint32_t count, uint32_t n, uint32_t per, uint32_t nchunks,
                                                           uint32_t per_c, uint32_t pb, shard_arrays a, unsigned char *buf)
{
    extern __shared__ uint4 s_rec[];
    uint32_t cb, ce;
    shard_block(nchunks, per_c, blockIdx.y, cb, ce);
    buf += static_cast<size_t>(per) * (80u * static_cast<size_t>(blockIdx.y) + 96u * static_cast<size_t>(cb));
    const uint32_t npieces = 5u + 6u * (ce - cb);
    for (uint32_t k0 = blockIdx.x * pb; k0 < count; k0 += gridDim.x * pb) {
        const uint32_t m = count - k0 < pb ? count - k0 : pb;
        for (uint32_t t = threadIdx.x; t < m * npieces; t += blockDim.x) {
            const uint32_t j = t / m, kk = t % m;
            const uint32_t i = first + k0 + kk;
            uint4 v;
            if (j >= 5u) {
                const uint32_t c = cb + (j - 5u) / 6u, tt = (j - 5u) % 6u;
                const size_t e = static_cast<size_t>(c) * n + i;
                const float4 *arr = tt == 0u ? a.lbs + e : (tt == 1u ? a.ubs + e : (a.recs ? a.recs + e * 4u + (tt - 2u) : nullptr));
                const float4 f = arr ? *arr : make_float4(0.f, 0.f, 0.f, 0.f);
                v = make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
            } else if (j == 0u) {
                v = make_uint4(a.nsub[i], static_cast<uint32_t>(a.status[i]), a.ev_idx[i], 0u);
            } else {
                const double d0 = j == 1u ? a.ev_time[i] : a.final_state[static_cast<size_t>(i) * 6u + 2u * (j - 2u)];
                const double d1 = j == 1u ? a.nf_last_t[i] : a.final_state[static_cast<size_t>(i) * 6u + 2u * (j - 2u) + 1u];
                v = make_uint4(static_cast<uint32_t>(__double2loint(d0)), static_cast<uint32_t>(__double2hiint(d0)),
                               static_cast<uint32_t>(__double2loint(d1)), static_cast<uint32_t>(__double2hiint(d1)));
            }
            s_rec[kk * npieces + j] = v;
        }
        __syncthreads();
        uint4 *dst = reinterpret_cast<uint4 *>(buf + static_cast<size_t>(k0) * npieces * 16u);
        for (uint32_t t = threadIdx.x; t < m * npieces; t += blockDim.x) {
            dst[t] = s_rec[t];
        }
        __syncthreads();
    }
}

// Rows on demand: the objects that appear in a candidate list, and the rows they need.
__global__ __launch_bounds__(256) void k_cand_mark(uint32_t ncand, const uint4 *cand, uint32_t *need)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncand) {
        const uint4 c = cand[k];
        need[c.x] = 1u;
        need[c.y] = 1u;
    }
}

__global__ __launch_bounds__(256) void k_cand_weights(uint32_t n, const uint32_t *need, const uint32_t *nsub, uint32_t *w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        w[i] = need[i] ? nsub[i] : 0u;
    }
}

__global__ __launch_bounds__(256) void k_cand_compact(uint32_t n, const uint32_t *need, const uint64_t *off, uint32_t *list)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && need[i]) {
        list[off[i]] = i;
    }
}

// Row compaction for the removal of particles: the row of particle i moves up by the number of
// removed indices below i (binary search in the sorted list), removed rows are dropped. One
// thread per element, src and dst distinct.
__global__ __launch_bounds__(256) void k_remove_rows(uint32_t n, uint32_t width, const uint32_t *idx, uint32_t nidx,
                                                      const double *src, double *dst)
{
    const uint64_t t = blockIdx.x * 256ull + threadIdx.x;
    if (t >= static_cast<uint64_t>(n) * width) {
        return;
    }
    const uint32_t i = static_cast<uint32_t>(t / width), c = static_cast<uint32_t>(t % width);
    uint32_t lo = 0, hi = nidx;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (idx[mid] < i) {
            lo = mid + 1u;
        } else {
            hi = mid;
        }
    }
    if (lo < nidx && idx[lo] == i) {
        return;
    }
    dst[static_cast<uint64_t>(i - lo) * width + c] = src[t];
}

__global__ __launch_bounds__(256) void k_copy_final_state(uint32_t n, const double *fs, double *state)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        state[static_cast<size_t>(i) * 7u + j] = fs[static_cast<size_t>(i) * 6u + j];
    }
}

// wl_mask may be null (both whitelists empty); bit 0: in coll whitelist, bit 1: in conj whitelist.
__global__ __launch_bounds__(256) void k_activity(uint32_t n, const double *state, double min_coll_radius, int with_conj,
                                                  int coll_wl_empty, int conj_wl_empty, const uint8_t *wl_mask,
                                                  uint8_t *active)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t m = wl_mask ? wl_mask[i] : 0;
    const bool coll = (state[static_cast<size_t>(i) * 7u + 6u] > min_coll_radius) && (coll_wl_empty || (m & 1u));
    const bool conj = with_conj && (conj_wl_empty || (m & 2u));
    active[i] = static_cast<uint8_t>((coll ? 1u : 0u) | (conj ? 2u : 0u));
}

__global__ __launch_bounds__(256) void k_gather_u32(uint32_t n, const uint32_t *in, const uint32_t *idx, uint32_t *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        out[k] = in[idx ? idx[k] : k];
    }
}

// Caller-provided [npairs][2] pairs with per-chunk offsets -> {i, j, chunk, 0} candidates.
__global__ __launch_bounds__(256) void k_tag_pairs(uint64_t npairs, const uint32_t *pairs, const uint64_t *off,
                                                   uint32_t nchunks, uint4 *out)
{
    const uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= npairs) {
        return;
    }
    uint32_t c = 0;
    while (c + 1u < nchunks && off[c + 1u] <= k) {
        ++c;
    }
    out[k] = make_uint4(pairs[k * 2u], pairs[k * 2u + 1u], c, 0u);
}

csc::prop_params make_prop_params(const csc_prop_cfg *cfg, uint32_t n, const double *d_state)
{
    csc::prop_params p{};
    p.state = d_state;
    p.n = n;
    p.first = 0;
    p.n_total = n;
    p.init_hi = cfg->init_time_hi;
    p.init_lo = cfg->init_time_lo;
    p.delta_t = cfg->delta_t;
    p.ct = cfg->ct;
    p.nchunks = cfg->nchunks;
    p.conj_thresh = cfg->conj_thresh;
    p.mu = cfg->mu;
    p.rhofac = cfg->rhofac;
    p.has_exit = cfg->has_exit;
    p.exit_c0 = cfg->exit_c0;
    p.has_reentry = cfg->has_reentry;
    for (int i = 0; i < 3; ++i) {
        p.re_k[i] = cfg->reentry_k[i];
    }
    p.re_c0 = cfg->reentry_c0;
    p.max_substeps = cfg->max_substeps ? cfg->max_substeps : (1u << 20);
    p.kj2 = 1.5 * cfg->J2 * cfg->mu * cfg->Re * cfg->Re;
    p.Re = cfg->Re;
    for (int i = 0; i < 4; ++i) {
        p.dens_alpha[i] = cfg->dens_alpha[i];
        p.dens_beta[i] = cfg->dens_beta[i];
        p.dens_gamma[i] = cfg->dens_gamma[i];
    }
    p.pars = cfg->d_pars;
    p.npars = cfg->npars;
    p.row_cap = 0;
    p.row_first = 0;
    p.tape = nullptr;
    p.tape_nops = p.tape_nslots = 0;
    p.tape_ws = nullptr;
    for (int i = 0; i < 7; ++i) {
        p.tape_rhs[i] = cfg->tape_rhs[i];
    }
    return p;
}

// nobox (pass 1 with rows, order 20, built-in dynamics): the variant that leaves the per-chunk boxes
// and samples to k_chunk_boxes.
template <bool STORE>
void launch_propagate(uint32_t order, int dyn, bool events, dim3 grid, dim3 block, hipStream_t stream, const csc::prop_params &p,
                      bool nobox = false)
{
    if constexpr (!STORE) {
        if (nobox && order == 20u && dyn != csc::DYN_TAPE) {
#define CSC_PROP_NOBOX_CASE(D)                                                                                         \
    if (events) {                                                                                                      \
        hipLaunchKernelGGL((csc::k_propagate<20, true, false, (D) | csc::DYN_NOBOX>), grid, block, 0, stream, p);      \
    } else {                                                                                                           \
        hipLaunchKernelGGL((csc::k_propagate<20, false, false, (D) | csc::DYN_NOBOX>), grid, block, 0, stream, p);     \
    }
            switch (dyn) {
                case 0:
                    CSC_PROP_NOBOX_CASE(0)
                    break;
                case 1:
                    CSC_PROP_NOBOX_CASE(1)
                    break;
                default:
                    CSC_PROP_NOBOX_CASE(3)
                    break;
            }
#undef CSC_PROP_NOBOX_CASE
            return;
        }
    }
    if (order != 20u) {
        // check_cfg has made sure that the order is built and that the dynamics come as a tape.
        csc_order_api(static_cast<int>(order))->propagate(STORE, events, grid, block, stream, p);
        return;
    }
#define CSC_PROP_CASE(D)                                                                                               \
    if (events) {                                                                                                      \
        hipLaunchKernelGGL((csc::k_propagate<20, true, STORE, D>), grid, block, 0, stream, p);                         \
    } else {                                                                                                           \
        hipLaunchKernelGGL((csc::k_propagate<20, false, STORE, D>), grid, block, 0, stream, p);                        \
    }
    switch (dyn) {
        case 0:
            CSC_PROP_CASE(0)
            break;
        case 1:
            CSC_PROP_CASE(1)
            break;
        case csc::DYN_TAPE:
            CSC_PROP_CASE(csc::DYN_TAPE)
            break;
        default:
            CSC_PROP_CASE(3)
            break;
    }
#undef CSC_PROP_CASE
}

int check_cfg(csc_ctx *ctx, const csc_prop_cfg *cfg)
{
    if (!cfg) {
        return ctx->invalid("null configuration");
    }
    if (cfg->order != 20u) {
        if (csc_order_api(static_cast<int>(cfg->order)) == nullptr) {
            ctx->err = "Taylor order " + std::to_string(cfg->order) + " is not built (built: " + std::to_string(csc::ORDER_MIN)
                       + " to " + std::to_string(csc::ORDER_MAX) + ")";
            return CSC_ERR_UNSUPPORTED;
        }
        if (cfg->dyn_kind != CSC_DYN_TAPE) {
            ctx->err = "Taylor orders other than 20 run from the dynamics tape (CSC_DYN_TAPE)";
            return CSC_ERR_UNSUPPORTED;
        }
    }
    if (cfg->dyn_kind != CSC_DYN_KEPLER && cfg->dyn_kind != CSC_DYN_J2 && cfg->dyn_kind != (CSC_DYN_J2 | CSC_DYN_DRAG)
        && cfg->dyn_kind != CSC_DYN_TAPE) {
        ctx->err = "unknown dynamics kind (built: Kepler, Kepler + J2, Kepler + J2 + drag, tape)";
        return CSC_ERR_UNSUPPORTED;
    }
    if (cfg->dyn_kind == CSC_DYN_TAPE || cfg->tape != nullptr) {
        // (a tape may come with the built-in dynamics too: the several-lanes-per-object kernel runs from it)
        // Every operation defines a new slot from earlier ones; anything else would read
        // coefficients that do not exist yet (or memory outside the workspace).
        if (cfg->tape == nullptr || cfg->tape_nops == 0u || cfg->tape_nslots < 8u || cfg->tape_nslots >= 0xffffu) {
            return ctx->invalid("tape dynamics without a tape");
        }
        for (uint32_t k = 0; k < cfg->tape_nops; ++k) {
            const csc_tape_op &o = cfg->tape[k];
            const uint32_t last = static_cast<uint32_t>(o.dst) + (o.op == CSC_TP_SINCOS ? 1u : 0u);
            if (o.op < CSC_TP_ADD || o.op > CSC_TP_CONST || o.dst < 7 || last >= cfg->tape_nslots || o.a < 0 || o.a >= o.dst
                || o.b >= o.dst || (o.b <= -2 && static_cast<uint32_t>(-2 - o.b) >= cfg->npars)
                || ((o.op == CSC_TP_RSUB || o.op == CSC_TP_RDIV || o.op == CSC_TP_CONST) && o.b >= 0)) {
                return ctx->invalid("malformed dynamics tape");
            }
        }
        for (int c = 0; c < 7; ++c) {
            if (cfg->tape_rhs[c] < 0 || static_cast<uint32_t>(cfg->tape_rhs[c]) >= cfg->tape_nslots) {
                return ctx->invalid("malformed dynamics tape (right-hand sides)");
            }
        }
        if (cfg->npars != 0u && cfg->d_pars == nullptr) {
            return ctx->invalid("the dynamics use runtime parameters but no parameter array was given");
        }
    }
    if ((cfg->dyn_kind & CSC_DYN_DRAG) && (cfg->d_pars == nullptr || cfg->npars == 0u)) {
        return ctx->invalid("drag needs the runtime parameter array (BSTAR)");
    }
    if (cfg->nchunks == 0u || !(cfg->delta_t > 0) || !(cfg->ct > 0)) {
        return ctx->invalid("invalid superstep configuration");
    }
    return CSC_OK;
}

} // namespace

static int scan_impl(csc_ctx *ctx, const uint32_t *d_in, uint64_t n, uint32_t nseg, uint64_t *d_out64,
                     uint32_t *d_out32);

// Tape dynamics: uploads the tape when it differs from the cached copy and sizes the workspace
// for `grid` resident waves. Fills the tape fields of p.
static int bind_tape(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, csc::prop_params &p, dim3 &grid)
{
    static_assert(sizeof(csc_tape_op) == sizeof(csc::tape_op), "tape layout");
    const size_t bytes = static_cast<size_t>(cfg->tape_nops) * sizeof(csc_tape_op);
    CSC_BUF(d_tape, csc::tape_op, "tape_ops", cfg->tape_nops);
    if (ctx->tape_host.size() != cfg->tape_nops || std::memcmp(ctx->tape_host.data(), cfg->tape, bytes) != 0) {
        ctx->tape_host.assign(cfg->tape, cfg->tape + cfg->tape_nops);
        CSC_HIP(hipMemcpyAsync(d_tape, ctx->tape_host.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        CSC_HIP(hipStreamSynchronize(ctx->stream));
    }
    // At most 8 waves per CU resident (256 CUs): more would only add workspace.
    const uint32_t ngroups = cdiv(n, 64);
    grid = dim3(std::min<uint32_t>(ngroups, 2048u));
    const size_t per_wave = static_cast<size_t>(cfg->tape_nslots) * (cfg->order + 1u) * 64u;
    CSC_BUF(d_ws, double, "tape_ws", per_wave * grid.x);
    p.tape = d_tape;
    p.tape_nops = cfg->tape_nops;
    p.tape_nslots = cfg->tape_nslots;
    p.tape_ws = d_ws;
    return CSC_OK;
}

extern "C" {

int csc_ctx_create(int device, csc_ctx **out)
{
    if (!out) {
        return CSC_ERR_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return CSC_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "invalid device index";
        return CSC_ERR_INVALID;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return CSC_ERR_HIP;
    }
    auto *ctx = new csc_ctx;
    ctx->device = device;
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess || (e = hipEventCreate(&ctx->ev0)) != hipSuccess
        || (e = hipEventCreate(&ctx->ev1)) != hipSuccess || (e = hipMalloc(&ctx->d_flags, 256)) != hipSuccess
        || (e = hipMemset(ctx->d_flags, 0, 256)) != hipSuccess) {
        g_create_error = std::string("context creation: ") + hipGetErrorString(e);
        delete ctx;
        return CSC_ERR_HIP;
    }
    // [0..1] error flags (int), broad-phase pair counter: u64 at byte offset 64.
    ctx->d_bp_counter = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->d_flags) + 64);
    // Address of the static scratch array on THIS device (the device is current, see above).
    if ((e = hipGetSymbolAddress(reinterpret_cast<void **>(&ctx->na_scratch), HIP_SYMBOL(scratch))) != hipSuccess
        || ctx->na_scratch == nullptr) {
        g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        csc_ctx_destroy(ctx);
        return CSC_ERR_HIP;
    }
    *out = ctx;
    return CSC_OK;
}

void csc_ctx_destroy(csc_ctx *ctx)
{
    if (!ctx) {
        return;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->bufs) {
        if (kv.second.p) {
            (void)hipFree(kv.second.p);
        }
    }
    if (ctx->d_flags) {
        (void)hipFree(ctx->d_flags);
    }
    if (ctx->cstore) {
        (void)hipFree(ctx->cstore);
        ctx->cstore = nullptr;
    }
    if (ctx->conj) {
        (void)hipHostFree(ctx->conj);
    }
    if (ctx->stage) {
        (void)hipHostFree(ctx->stage);
    }
    for (hipEvent_t e : ctx->stage_ev) {
        (void)hipEventDestroy(e);
    }
    if (ctx->copy_stream) {
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *csc_last_error(csc_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int csc_malloc(csc_ctx *ctx, size_t bytes, void **d_ptr)
{
    CSC_HIP(hipSetDevice(ctx->device));
    CSC_HIP(hipMalloc(d_ptr, bytes ? bytes : 256));
    return CSC_OK;
}

int csc_free(csc_ctx *ctx, void *d_ptr)
{
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    CSC_HIP(hipFree(d_ptr));
    return CSC_OK;
}

int csc_memcpy_h2d(csc_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    CSC_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    return CSC_OK;
}

int csc_memcpy_d2h(csc_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    CSC_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    return CSC_OK;
}

// ---- pipelined transfers of pageable host arrays ------------------------------------------
// hipMemcpy from / to pageable memory runs at ~17 GB/s (one staging thread inside the runtime);
// a pinned mirror filled / drained by several threads, chunk by chunk, next to the DMA of the
// neighbouring chunks reaches the PCIe rate.
namespace
{

constexpr unsigned XFER_CHUNKS = 16, XFER_THREADS = 8;

// 128-bit position-dependent checksum of 8-byte words. The words are taken in pairs (2j, 2j + 1);
// each pair, xored with two position-dependent keys, goes through a full 64 x 64 -> 128 bit
// multiplication whose halves are folded together (the "mum" mixer of wyhash), once per lane with
// different keys, BEFORE it is added to the lane. A product is non-linear in every bit of both
// words, so a structured edit -- an even number of sign flips, +d / -d on two words, two rows
// swapped -- changes each lane by a sum of pseudo-random 64-bit terms that cancels with probability
// 2^-64 per lane (2^-128 for both). The sums are order-independent, so pieces that start at even
// word indices combine by addition. (Round 2 added (w ^ k) * odd without any mixing: linear mod
// 2^64, blind to any even number of bit-63 flips.)
struct hash128 {
    uint64_t a = 0, b = 0;
    void operator+=(const hash128 &o)
    {
        a += o.a;
        b += o.b;
    }
};

extern "C++" inline uint64_t hash_mum(uint64_t x, uint64_t y)
{
    const unsigned __int128 m = static_cast<unsigned __int128>(x) * y;
    return static_cast<uint64_t>(m) ^ static_cast<uint64_t>(m >> 64);
}

// first_word must be even; an odd word count pairs the last word with a zero.
extern "C++" inline hash128 hash_words(const unsigned char *p, size_t first_word, size_t nwords)
{
    hash128 h;
    for (size_t i = 0; i < nwords; i += 2u) {
        uint64_t w0, w1 = 0;
        std::memcpy(&w0, p + i * 8u, 8);
        if (i + 1u < nwords) {
            std::memcpy(&w1, p + (i + 1u) * 8u, 8);
        }
        const uint64_t k = (first_word + i + 1u) * 0x9e3779b97f4a7c15ull; // (odd multiples: never 0)
        h.a += hash_mum(w0 ^ k, w1 ^ (k + 0xa0761d6478bd642full));
        h.b += hash_mum(w0 ^ ((k >> 7 | k << 57) + 0xe7037ed1a0b428dbull), w1 ^ (~k * 0x8ebc6af09c88c6e3ull));
    }
    return h;
}

int xfer_prepare(csc_ctx *ctx, size_t bytes)
{
    CSC_HIP(hipSetDevice(ctx->device));
    if (ctx->copy_stream == nullptr) {
        CSC_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    }
    while (ctx->stage_ev.size() < XFER_CHUNKS) {
        hipEvent_t e;
        CSC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->stage_ev.push_back(e);
    }
    if (ctx->stage_cap < bytes) {
        if (ctx->stage) {
            CSC_HIP(hipStreamSynchronize(ctx->copy_stream));
            CSC_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipHostFree(ctx->stage);
            ctx->stage = nullptr;
            ctx->stage_cap = 0;
        }
        const size_t want = bytes + bytes / 8u + 4096u;
        CSC_HIP(hipHostMalloc(&ctx->stage, want, hipHostMallocDefault));
        ctx->stage_cap = want;
    }
    return CSC_OK;
}

} // namespace

void csc_host_hash(const void *p, size_t bytes, uint64_t out[2])
{
    const size_t nwords = bytes / 8u;
    const unsigned nt = nwords < (1u << 16) ? 1u : std::max(1u, std::min(XFER_THREADS, std::thread::hardware_concurrency()));
    std::vector<hash128> part(nt);
    const size_t per = ((nwords + nt - 1) / nt + 1u) & ~static_cast<size_t>(1); // (even: whole pairs)
    auto work = [&](unsigned t) {
        const size_t b = std::min(nwords, t * per), e = std::min(nwords, b + per);
        part[t] = hash_words(static_cast<const unsigned char *>(p) + b * 8u, b, e - b);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
        th.emplace_back(work, t);
    }
    work(0);
    for (auto &t : th) {
        t.join();
    }
    hash128 h;
    for (const hash128 &v : part) {
        h += v;
    }
    // (the trailing bytes of an array that is not a multiple of 8 long: one zero-padded word)
    if (bytes % 8u != 0u) {
        unsigned char tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(tail, static_cast<const unsigned char *>(p) + nwords * 8u, bytes % 8u);
        h += hash_words(tail, (nwords + 2u) & ~static_cast<size_t>(1), 1);
    }
    out[0] = h.a;
    out[1] = h.b;
}

int csc_upload(csc_ctx *ctx, void *d_dst, const void *src, size_t bytes, uint64_t hash[2])
{
    if (hash) {
        hash[0] = hash[1] = 0;
    }
    if (bytes == 0) {
        return CSC_OK;
    }
    if (bytes < (4u << 20) || bytes % 8u != 0u) {
        if (hash) {
            csc_host_hash(src, bytes, hash);
        }
        return csc_memcpy_h2d(ctx, d_dst, src, bytes);
    }
    if (const int rc = xfer_prepare(ctx, bytes)) {
        return rc;
    }
    // The staging mirror may still be the source of an earlier upload's DMA.
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    const size_t nwords = bytes / 8u, per = ((nwords + XFER_CHUNKS - 1) / XFER_CHUNKS + 1u) & ~static_cast<size_t>(1); // (even: whole pairs)
    const unsigned nt = std::max(1u, std::min(XFER_THREADS, std::thread::hardware_concurrency()));
    std::vector<hash128> part(XFER_CHUNKS);
    std::vector<hipError_t> err(nt, hipSuccess);
    auto work = [&](unsigned t) {
        (void)hipSetDevice(ctx->device);
        for (unsigned c = t; c < XFER_CHUNKS; c += nt) {
            const size_t b = std::min(nwords, c * per), e = std::min(nwords, b + per);
            if (b == e) {
                continue;
            }
            const auto *s = static_cast<const unsigned char *>(src) + b * 8u;
            auto *m = static_cast<unsigned char *>(ctx->stage) + b * 8u;
            std::memcpy(m, s, (e - b) * 8u);
            if (hash) {
                part[c] = hash_words(m, b, e - b);
            }
            const hipError_t rc = hipMemcpyAsync(static_cast<unsigned char *>(d_dst) + b * 8u, m, (e - b) * 8u, hipMemcpyHostToDevice,
                                                 ctx->stream);
            if (rc != hipSuccess) {
                err[t] = rc;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
        th.emplace_back(work, t);
    }
    work(0);
    for (auto &t : th) {
        t.join();
    }
    for (hipError_t e : err) {
        if (e != hipSuccess) {
            return ctx->fail(e, "csc_upload: hipMemcpyAsync");
        }
    }
    if (hash) {
        hash128 h;
        for (const hash128 &v : part) {
            h += v;
        }
        hash[0] = h.a;
        hash[1] = h.b;
    }
    return CSC_OK;
}

int csc_download(csc_ctx *ctx, void *dst, const void *d_src, size_t bytes, uint64_t hash[2])
{
    if (hash) {
        hash[0] = hash[1] = 0;
    }
    if (bytes == 0) {
        return CSC_OK;
    }
    if (bytes < (4u << 20) || bytes % 8u != 0u) {
        const int rc = csc_memcpy_d2h(ctx, dst, d_src, bytes);
        if (rc == CSC_OK && hash) {
            csc_host_hash(dst, bytes, hash);
        }
        return rc;
    }
    if (const int rc = xfer_prepare(ctx, bytes)) {
        return rc;
    }
    // Everything queued on the context's stream has to be done before the copies start.
    CSC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    CSC_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev0, 0));
    const size_t nwords = bytes / 8u, per = ((nwords + XFER_CHUNKS - 1) / XFER_CHUNKS + 1u) & ~static_cast<size_t>(1); // (even: whole pairs)
    for (unsigned c = 0; c < XFER_CHUNKS; ++c) {
        const size_t b = std::min(nwords, c * per), e = std::min(nwords, b + per);
        if (b < e) {
            CSC_HIP(hipMemcpyAsync(static_cast<unsigned char *>(ctx->stage) + b * 8u, static_cast<const unsigned char *>(d_src) + b * 8u,
                                   (e - b) * 8u, hipMemcpyDeviceToHost, ctx->copy_stream));
        }
        CSC_HIP(hipEventRecord(ctx->stage_ev[c], ctx->copy_stream));
    }
    const unsigned nt = std::max(1u, std::min(XFER_THREADS, std::thread::hardware_concurrency()));
    std::vector<hash128> part(XFER_CHUNKS);
    std::vector<hipError_t> err(nt, hipSuccess);
    auto work = [&](unsigned t) {
        (void)hipSetDevice(ctx->device);
        for (unsigned c = t; c < XFER_CHUNKS; c += nt) {
            const size_t b = std::min(nwords, c * per), e = std::min(nwords, b + per);
            const hipError_t rc = hipEventSynchronize(ctx->stage_ev[c]);
            if (rc != hipSuccess) {
                err[t] = rc;
                continue;
            }
            if (b == e) {
                continue;
            }
            const auto *m = static_cast<const unsigned char *>(ctx->stage) + b * 8u;
            std::memcpy(static_cast<unsigned char *>(dst) + b * 8u, m, (e - b) * 8u);
            if (hash) {
                part[c] = hash_words(m, b, e - b);
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) {
        th.emplace_back(work, t);
    }
    work(0);
    for (auto &t : th) {
        t.join();
    }
    for (hipError_t e : err) {
        if (e != hipSuccess) {
            return ctx->fail(e, "csc_download: hipEventSynchronize");
        }
    }
    if (hash) {
        hash128 h;
        for (const hash128 &v : part) {
            h += v;
        }
        hash[0] = h.a;
        hash[1] = h.b;
    }
    return CSC_OK;
}

int csc_memcpy_d2d(csc_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    CSC_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CSC_OK;
}

int csc_memset(csc_ctx *ctx, void *d_dst, int value, size_t bytes)
{
    CSC_HIP(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return CSC_OK;
}

int csc_sync(csc_ctx *ctx)
{
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    return CSC_OK;
}

int csc_timer_begin(csc_ctx *ctx)
{
    CSC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return CSC_OK;
}

int csc_timer_end(csc_ctx *ctx, float *ms)
{
    CSC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    CSC_HIP(hipEventSynchronize(ctx->ev1));
    CSC_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return CSC_OK;
}

int csc_profile_enable(csc_ctx *ctx, int on)
{
    ctx->kt_collect();
    ctx->profiling = on != 0;
    return CSC_OK;
}

int csc_profile_only(csc_ctx *ctx, const char *name)
{
    ctx->kt_collect();
    ctx->profile_only = name ? name : "";
    return CSC_OK;
}

int csc_profile_reset(csc_ctx *ctx)
{
    ctx->kt_collect();
    for (auto &kv : ctx->kt) {
        kv.second.total_ms = 0;
        kv.second.launches = 0;
    }
    return CSC_OK;
}

int csc_profile_get(csc_ctx *ctx, char *buf, size_t cap)
{
    ctx->kt_collect();
    std::string out;
    for (const auto &kv : ctx->kt) {
        if (kv.second.launches) {
            char line[256];
            std::snprintf(line, sizeof(line), "%s %llu %.6f\n", kv.first.c_str(),
                          static_cast<unsigned long long>(kv.second.launches), kv.second.total_ms);
            out += line;
        }
    }
    if (cap) {
        std::strncpy(buf, out.c_str(), cap - 1);
        buf[cap - 1] = 0;
    }
    return CSC_OK;
}

int csc_check_flags(csc_ctx *ctx)
{
    if (ctx->flags_fresh) {
        return CSC_OK;
    }
    int h[4] = {0, 0, 0, 0};
    CSC_HIP(hipMemcpyAsync(h, ctx->d_flags, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    if (h[0] || h[1] || h[2] || h[3]) {
        CSC_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(h), ctx->stream));
    }
    if (h[0]) {
        ctx->err = "A global bounding box with non-finite boundaries and/or size was generated";
        return CSC_ERR_NONFINITE_AABB;
    }
    if (h[1]) {
        ctx->err = "dense_propagate() could not be performed because no timesteps have been taken for a particle";
        return CSC_ERR_NO_SUBSTEPS;
    }
    if (h[2]) {
        ctx->err = "broad phase traversal stack overflow (BVH deeper than 64 levels?)";
        return CSC_ERR_OVERFLOW;
    }
    if (h[3]) {
        ctx->err = "csc_propagate_store: the second integration pass did not repeat the substep count of the first "
                   "(the Taylor rows of the superstep are not valid)";
        return CSC_ERR_HIP;
    }
    return CSC_OK;
}

// ---- K1 / K2 ---------------------------------------------------------------------------

// One object over several lanes, series in LDS (k_propagate_wave.hpp): every right-hand side that
// comes with its tape, except plain Kepler (order 20). Schedules the tape when it is new, launches
// the kernel and sets `launched`; leaves it false when the kernel does not apply.
// The tagged built-ins (Kepler + J2 + drag ...) have hand-unrolled kernels, one object per lane, that
// win once the population fills the GPU with them (64 objects per wave); below that a COMPILED tape
// -- 8 objects per wave -- is the faster one (J2 + drag, 19 647 objects: 3.6 against 5.4 ms; 1M: 44.8 against 39.7).
constexpr uint32_t WAVE_TAGGED_MAX_N = 49152;

// Schedules the tape when it is new and says what would run it: 0 the one-lane kernels, 1 the
// interpreted rows of k_propagate_wave, 2 one of its compiled tapes.
static int wave_select(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n_total, bool *fresh_out = nullptr)
{
    if (fresh_out) {
        *fresh_out = false;
    }
    if (ctx->wave_mode < 0) {
        const char *e = std::getenv("CASCADE_WAVE");
        ctx->wave_mode = (e && std::atoi(e) == 0) ? 0 : 1;
    }
    if (ctx->wave_mode != 1 || cfg->order != 20u || cfg->tape == nullptr || cfg->tape_nops == 0u) {
        return 0;
    }
    static_assert(sizeof(csc_tape_op) == sizeof(csc::tape_op), "tape layout");
    const size_t bytes = static_cast<size_t>(cfg->tape_nops) * sizeof(csc_tape_op);
    const bool fresh = ctx->wave_tape.size() != cfg->tape_nops || std::memcmp(ctx->wave_tape.data(), cfg->tape, bytes) != 0
                       || ctx->wave_plan.tape_nslots != cfg->tape_nslots
                       || std::memcmp(ctx->wave_plan.tape_rhs, cfg->tape_rhs, sizeof(cfg->tape_rhs)) != 0;
    if (fresh) {
        ctx->wave_tape.assign(cfg->tape, cfg->tape + cfg->tape_nops);
        ctx->wave_fits = csc::wave_build_plan(reinterpret_cast<const csc::tape_op *>(cfg->tape), cfg->tape_nops, cfg->tape_nslots,
                                              cfg->tape_rhs, 0, cfg->order, ctx->wave_plan);
        ctx->wave_compiled = 0;
        ctx->wave_uploaded = false;
        ctx->wave_jit = 0;
        const char *ce = std::getenv("CASCADE_WAVE_COMPILED");
        if (!(ce && std::atoi(ce) == 0) && !std::getenv("CASCADE_WAVE_NG")
            && csc::wave_build_plan(reinterpret_cast<const csc::tape_op *>(cfg->tape), cfg->tape_nops, cfg->tape_nslots, cfg->tape_rhs, 1,
                                    cfg->order, ctx->wave_plan1)) {
            // (two waves per compute unit at least, counted in 2 KB blocks of LDS as in wave_plan::waves_per_cu)
            const size_t granted = (ctx->wave_plan1.wave_bytes() + 2047u) / 2048u * 2048u;
            if (csc::WAVE_LDS_MAX / granted >= 2u) {
                ctx->wave_compiled = csc::wave_compiled_lookup(csc::wave_plan_hash(ctx->wave_plan1));
                // (2: also for the dynamics that have an ahead-of-time tape, to compare the two)
                const char *je = std::getenv("CASCADE_WAVE_JIT");
                const int jv = je ? std::atoi(je) : 1;
                if (jv == 2 && cfg->dyn_kind == CSC_DYN_TAPE) {
                    ctx->wave_compiled = 0;
                }
                if (ctx->wave_compiled == 0 && jv != 0) {
                    ctx->wave_jit = 1;
                }
            }
        }
        if (std::getenv("CASCADE_WAVE_DEBUG")) {
            std::fprintf(stderr, "[wave] compiled tape: %d\n", ctx->wave_compiled);
        }
        if (fresh_out) {
            *fresh_out = true;
        }
    }
    if (cfg->dyn_kind != CSC_DYN_TAPE) {
        return (ctx->wave_compiled != 0 && cfg->dyn_kind == (CSC_DYN_J2 | CSC_DYN_DRAG) && n_total <= WAVE_TAGGED_MAX_N) ? 2 : 0;
    }
    if (ctx->wave_compiled != 0) {
        return 2;
    }
    if (ctx->wave_jit == 1) {
        return 3;
    }
    // Two waves per compute unit at least: with less, the one-lane-per-object interpreter with its
    // global workspace is the faster one.
    if (!ctx->wave_fits || (ctx->wave_plan.waves_per_cu() < 2u && !std::getenv("CASCADE_WAVE_NG"))) {
        if (fresh && std::getenv("CASCADE_WAVE_DEBUG")) {
            std::fprintf(stderr, "[wave] not used: %s\n", ctx->wave_fits ? "fewer than two waves per compute unit" : "the tape has no schedule");
        }
        return 0;
    }
    return 1;
}

static int wave_try_launch(csc_ctx *ctx, const csc_prop_cfg *cfg, const csc::prop_params &p, bool events, bool &launched)
{
    launched = false;
    int how = wave_select(ctx, cfg, p.n_total != 0u ? p.n_total : p.n);
    if (how == 3) {
        std::string jerr;
        hipError_t jrc = hipSuccess;
        KT(p.nsub_in ? "k_propagate_wave_store" : "k_propagate_wave",
           jrc = csc::wave_launch_jit(ctx->wave_plan1, cfg->order, events, p, ctx->stream, jerr));
        if (jrc == hipSuccess) {
            launched = true;
            return CSC_OK;
        }
        if (jrc != hipErrorNotSupported) {
            return ctx->fail(jrc, "k_propagate_wave (run-time compiled tape) launch");
        }
        // (no compiler, no headers, a source it rejects: said once, then the interpreted rows)
        std::fprintf(stderr, "cascade: run-time compilation of the dynamics failed, the interpreted schedule runs instead: %s\n", jerr.c_str());
        ctx->wave_jit = -1;
        how = wave_select(ctx, cfg, p.n_total != 0u ? p.n_total : p.n);
    }
    if (how == 0) {
        return CSC_OK;
    }
    if (how == 2) {
        hipError_t crc = hipSuccess;
        KT(p.nsub_in ? "k_propagate_wave_store" : "k_propagate_wave",
           crc = csc::wave_launch_compiled(ctx->wave_compiled, ctx->wave_plan1, events, p, ctx->stream));
        if (crc != hipSuccess) {
            return ctx->fail(crc, "k_propagate_wave (compiled tape) launch");
        }
        launched = true;
        return CSC_OK;
    }
    const csc::wave_plan &wp = ctx->wave_plan;
    CSC_BUF(d_wblk, uint32_t, "wave_blk", wp.blk.size());
    CSC_BUF(d_wops0, csc::wave_op0, "wave_ops0", wp.ops0.size());
    if (!ctx->wave_uploaded) {
        CSC_HIP(hipMemcpyAsync(d_wblk, wp.blk.data(), wp.blk_bytes(), hipMemcpyHostToDevice, ctx->stream));
        CSC_HIP(hipMemcpyAsync(d_wops0, wp.ops0.data(), wp.ops0.size() * sizeof(csc::wave_op0), hipMemcpyHostToDevice, ctx->stream));
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        ctx->wave_uploaded = true;
    }
    hipError_t lrc = hipSuccess;
    KT(p.nsub_in ? "k_propagate_wave_store" : "k_propagate_wave", lrc = csc::wave_launch(wp, d_wblk, d_wops0, events, p, ctx->stream));
    if (lrc != hipSuccess) {
        return ctx->fail(lrc, "k_propagate_wave launch");
    }
    launched = true;
    return CSC_OK;
}

static int propagate_impl(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state, uint32_t *d_nsub,
                          int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t,
                          double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs, uint32_t row_cap,
                          uint64_t *d_tc_row, double *d_tcs_xyz, double *d_tcs_vr, double *d_tcoords, uint32_t first = 0,
                          uint32_t count = 0xffffffffu)
{
    if (const int rc = check_cfg(ctx, cfg)) {
        return rc;
    }
    if (n == 0u) {
        return ctx->invalid("Cannot integrate a simulation with no particles");
    }
    const uint32_t n_total = n;
    if (count != 0xffffffffu) {
        if (first > n_total || count > n_total - first) {
            return ctx->invalid("csc_propagate_range: the range does not lie within the particles");
        }
        if (count == 0u) {
            return CSC_OK;
        }
        n = count;
    }
    csc::prop_params p = make_prop_params(cfg, n, d_state);
    p.first = count != 0xffffffffu ? first : 0u;
    p.n_total = n_total;
    p.nsub = d_nsub;
    p.status = d_status;
    p.ev_idx = d_ev_idx;
    p.ev_time = d_ev_time;
    p.nf_last_t = d_nf_last_t;
    p.final_state = d_final_state;
    p.lbs = d_lbs;
    p.ubs = d_ubs;
    p.recs = d_recs;
    p.row_cap = row_cap;
    p.row_first = (count != 0xffffffffu && row_cap != 0u) ? first : 0u; // (a range with rows: the rows are its own)
    p.tc_row = d_tc_row;
    p.tcs_xyz = d_tcs_xyz;
    p.tcs_vr = d_tcs_vr;
    p.tcoords = d_tcoords;
    const bool events = cfg->has_exit || cfg->has_reentry;
    dim3 grid(cdiv(n, 64)), block(64);
    // With rows: the boxes and samples come from the rows, in a kernel of their own (order 20,
    // built-in dynamics; CASCADE_BOX_KERNEL=0: fused into the integration as without rows).
    // (k_propagate_wave whenever the pass writes rows and the dynamics come with their tape)
    bool waved = false;
    if (row_cap != 0u) {
        if (const int rc = wave_try_launch(ctx, cfg, p, events, waved)) {
            return rc;
        }
    }
    ctx->pass1_wave = waved;
    const int wave_ng = waved ? 1 : 0;
    const bool nobox = wave_ng != 0 || (row_cap != 0u && cfg->order == 20u && cfg->dyn_kind != CSC_DYN_TAPE);
    if (wave_ng == 0) {
        if (cfg->dyn_kind == CSC_DYN_TAPE) {
            if (const int rc = bind_tape(ctx, cfg, n, p, grid)) {
                return rc;
            }
        }
        KT(row_cap ? "k_propagate_rows" : "k_propagate",
           launch_propagate<false>(cfg->order, cfg->dyn_kind, events, grid, block, ctx->stream, p, nobox));
    }
    CSC_LAUNCH_CHECK();
    if (nobox) {
        const uint64_t lanes = static_cast<uint64_t>(n) * p.nchunks;
        KT("k_chunk_boxes", hipLaunchKernelGGL((csc::k_chunk_boxes<20>), dim3(static_cast<uint32_t>((lanes + 255u) / 256u)), dim3(256), 0,
                                                ctx->stream, p));
        CSC_LAUNCH_CHECK();
    }
    return CSC_OK;
}

int csc_propagate(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state, uint32_t *d_nsub,
                  int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t,
                  double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs)
{
    return propagate_impl(ctx, cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, d_lbs,
                          d_ubs, d_recs, 0, nullptr, nullptr, nullptr, nullptr);
}

size_t csc_shard_record_bytes(uint32_t nchunks)
{
    return shard_record_bytes(nchunks);
}

// Particles [first, first + count) -> nsegs segments of records (segment y: the chunks of block y).
static void launch_shard_pack(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t per, uint32_t nchunks, uint32_t nsegs,
                              const shard_arrays &a, void *d_records);
// Records of particles [0, n) with the chunks [cb, ce) -> the arrays.
static void launch_shard_unpack(csc_ctx *ctx, uint32_t n, uint32_t nchunks, uint32_t cb, uint32_t ce, const shard_arrays &a,
                                const void *d_records);

// Workgroups for `count` records of 5 + 6 * chunks pieces (the kernel strides over the rest).
static unsigned shard_grid(uint32_t count, uint32_t chunks)
{
    const uint64_t pieces = static_cast<uint64_t>(count) * (5u + 6u * static_cast<uint64_t>(chunks));
    return static_cast<unsigned>(std::min<uint64_t>(std::max<uint64_t>(cdiv(pieces, static_cast<uint64_t>(256)), 1u), 65536u));
}

int csc_shard_pack(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t nchunks, const uint32_t *d_nsub,
                   const int32_t *d_status, const uint32_t *d_ev_idx, const double *d_ev_time, const double *d_nf_last_t,
                   const double *d_final_state, const float *d_lbs, const float *d_ubs, const float *d_recs, void *d_records)
{
    if (first > n || count > n - first) {
        return ctx->invalid("csc_shard_pack: the range does not lie within the particles");
    }
    if (count == 0u) {
        return CSC_OK;
    }
    shard_arrays a{const_cast<uint32_t *>(d_nsub), const_cast<int32_t *>(d_status), const_cast<uint32_t *>(d_ev_idx),
                   const_cast<double *>(d_ev_time), const_cast<double *>(d_nf_last_t), const_cast<double *>(d_final_state),
                   reinterpret_cast<float4 *>(const_cast<float *>(d_lbs)), reinterpret_cast<float4 *>(const_cast<float *>(d_ubs)),
                   reinterpret_cast<float4 *>(const_cast<float *>(d_recs))};
    launch_shard_pack(ctx, n, first, count, count, nchunks, 1u, a, d_records);
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_shard_block(uint32_t nchunks, uint32_t nranks, uint32_t rank, uint32_t *first, uint32_t *count)
{
    if (nranks == 0u) {
        return CSC_ERR_INVALID;
    }
    uint32_t cb = 0, ce = 0;
    shard_block(nchunks, cdiv(nchunks, nranks), rank, cb, ce);
    *first = cb;
    *count = ce - cb;
    return CSC_OK;
}

int csc_shard_pack_blocks(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t per_rank, uint32_t nchunks,
                          uint32_t nranks, const uint32_t *d_nsub, const int32_t *d_status, const uint32_t *d_ev_idx,
                          const double *d_ev_time, const double *d_nf_last_t, const double *d_final_state, const float *d_lbs,
                          const float *d_ubs, const float *d_recs, void *d_records)
{
    if (first > n || count > n - first || count > per_rank) {
        return ctx->invalid("csc_shard_pack_blocks: the range does not lie within the particles (or exceeds per_rank)");
    }
    if (nranks == 0u || nranks > 65535u || nchunks == 0u) {
        return ctx->invalid("csc_shard_pack_blocks: needs 1 <= nranks <= 65535 and at least one chunk");
    }
    if (count == 0u) {
        return CSC_OK;
    }
    shard_arrays a{const_cast<uint32_t *>(d_nsub), const_cast<int32_t *>(d_status), const_cast<uint32_t *>(d_ev_idx),
                   const_cast<double *>(d_ev_time), const_cast<double *>(d_nf_last_t), const_cast<double *>(d_final_state),
                   reinterpret_cast<float4 *>(const_cast<float *>(d_lbs)), reinterpret_cast<float4 *>(const_cast<float *>(d_ubs)),
                   reinterpret_cast<float4 *>(const_cast<float *>(d_recs))};
    launch_shard_pack(ctx, n, first, count, per_rank, nchunks, nranks, a, d_records);
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_shard_unpack_block(csc_ctx *ctx, uint32_t n, uint32_t nchunks, uint32_t chunk_first, uint32_t chunk_count,
                           const void *d_records, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                           double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs)
{
    if (chunk_first > nchunks || chunk_count > nchunks - chunk_first) {
        return ctx->invalid("csc_shard_unpack_block: the block does not lie within the chunks");
    }
    if (n == 0u) {
        return CSC_OK;
    }
    shard_arrays a{d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, reinterpret_cast<float4 *>(d_lbs),
                   reinterpret_cast<float4 *>(d_ubs), reinterpret_cast<float4 *>(d_recs)};
    launch_shard_unpack(ctx, n, nchunks, chunk_first, chunk_first + chunk_count, a, d_records);
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_shard_unpack(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const void *d_records, uint32_t *d_nsub, int32_t *d_status,
                     uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t, double *d_final_state, float *d_lbs,
                     float *d_ubs, float *d_recs)
{
    if (n == 0u) {
        return CSC_OK;
    }
    shard_arrays a{d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, reinterpret_cast<float4 *>(d_lbs),
                   reinterpret_cast<float4 *>(d_ubs), reinterpret_cast<float4 *>(d_recs)};
    launch_shard_unpack(ctx, n, nchunks, 0u, nchunks, a, d_records);
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

static void launch_shard_pack(csc_ctx *ctx, uint32_t n, uint32_t first, uint32_t count, uint32_t per, uint32_t nchunks, uint32_t nsegs,
                              const shard_arrays &a, void *d_records)
{
    const uint32_t per_c = cdiv(nchunks, nsegs);
    const size_t rec = shard_record_bytes(per_c);
    const uint32_t pb = static_cast<uint32_t>(std::min<size_t>(64u, 49152u / rec));
    if (pb >= 8u) {
        const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(cdiv(count, pb), 8192u));
        KT("k_shard_pack", hipLaunchKernelGGL(k_shard_pack_staged, dim3(grid, nsegs), dim3(256), pb * rec, ctx->stream, first, count, n, per,
                                               nchunks, per_c, pb, a, static_cast<unsigned char *>(d_records)));
    } else {
        KT("k_shard_pack", hipLaunchKernelGGL((k_shard_records<false>), dim3(shard_grid(count, per_c), nsegs), dim3(256), 0, ctx->stream, first,
                                               count, n, per, nchunks, per_c, 0u, 0u, a, static_cast<unsigned char *>(d_records)));
    }
}

static void launch_shard_unpack(csc_ctx *ctx, uint32_t n, uint32_t nchunks, uint32_t cb, uint32_t ce, const shard_arrays &a,
                                const void *d_records)
{
    const size_t rec = shard_record_bytes(ce - cb);
    const uint32_t pb = static_cast<uint32_t>(std::min<size_t>(64u, 49152u / rec));
    if (pb >= 8u) {
        const unsigned grid = static_cast<unsigned>(std::min<uint64_t>(cdiv(n, pb), 8192u));
        KT("k_shard_unpack", hipLaunchKernelGGL(k_shard_unpack_staged, dim3(grid), dim3(256), pb * rec, ctx->stream, n, n, cb, ce, pb, a,
                                                 static_cast<const unsigned char *>(d_records)));
    } else {
        KT("k_shard_unpack", hipLaunchKernelGGL((k_shard_records<true>), dim3(shard_grid(n, ce - cb)), dim3(256), 0, ctx->stream, 0u, n, n, n,
                                                 nchunks, nchunks, cb, ce, a, static_cast<unsigned char *>(const_cast<void *>(d_records))));
    }
}

int csc_propagate_range(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, uint32_t first, uint32_t count,
                        const double *d_state, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                        double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs)
{
    return propagate_impl(ctx, cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, d_lbs,
                          d_ubs, d_recs, 0, nullptr, nullptr, nullptr, nullptr, first, count);
}

int csc_propagate_range_rows(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, uint32_t first, uint32_t count,
                             const double *d_state, uint32_t *d_nsub, int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time,
                             double *d_nf_last_t, double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs, uint32_t row_cap)
{
    if (row_cap == 0u) {
        return ctx->invalid("csc_propagate_range_rows needs row_cap > 0");
    }
    if (first > n || count > n - first) {
        return ctx->invalid("csc_propagate_range: the range does not lie within the particles");
    }
    // The rows of the range live in the context: they feed k_chunk_boxes and are not handed out
    // (the narrow phase of a sharded superstep gets its rows from csc_propagate_store).
    uint32_t xs = 0, vs = 0;
    csc_tc_strides(cfg ? cfg->order : 20u, &xs, &vs);
    const size_t rows = std::max<size_t>(static_cast<size_t>(count) * row_cap, 1);
    CSC_BUF(d_row, uint64_t, "range_tc_row", n);
    CSC_BUF(d_xyz, double, "range_tcs_xyz", rows * xs);
    CSC_BUF(d_vr, double, "range_tcs_vr", rows * vs);
    CSC_BUF(d_tco, double, "range_tcoords", rows * 2u);
    return propagate_impl(ctx, cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, d_lbs,
                          d_ubs, d_recs, row_cap, d_row, d_xyz, d_vr, d_tco, first, count);
}

int csc_wave_would_run(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n)
{
    if (ctx == nullptr || cfg == nullptr) {
        return 0;
    }
    if (check_cfg(ctx, cfg) != CSC_OK) {
        return 0;
    }
    return wave_select(ctx, cfg, n);
}

int csc_propagate_rows(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state, uint32_t *d_nsub,
                       int32_t *d_status, uint32_t *d_ev_idx, double *d_ev_time, double *d_nf_last_t,
                       double *d_final_state, float *d_lbs, float *d_ubs, float *d_recs, uint32_t row_cap,
                       uint64_t *d_tc_row, double *d_tcs_xyz, double *d_tcs_vr, double *d_tcoords)
{
    if (row_cap == 0u || !d_tc_row || !d_tcs_xyz || !d_tcs_vr || !d_tcoords) {
        return ctx->invalid("csc_propagate_rows needs a row capacity and the row arrays");
    }
    return propagate_impl(ctx, cfg, n, d_state, d_nsub, d_status, d_ev_idx, d_ev_time, d_nf_last_t, d_final_state, d_lbs,
                          d_ubs, d_recs, row_cap, d_tc_row, d_tcs_xyz, d_tcs_vr, d_tcoords);
}

int csc_tc_strides(uint32_t order, uint32_t *xyz_stride, uint32_t *vr_stride)
{
    *xyz_stride = static_cast<uint32_t>(csc::tc_xyz_stride(static_cast<int>(order)));
    *vr_stride = static_cast<uint32_t>(csc::tc_vr_stride(static_cast<int>(order)));
    return CSC_OK;
}

int csc_propagate_store(csc_ctx *ctx, const csc_prop_cfg *cfg, uint32_t n, const double *d_state,
                        const uint32_t *d_order, const uint32_t *d_nsub, uint64_t *d_tc_row, double *d_tcs_xyz,
                        double *d_tcs_vr, double *d_tcoords)
{
    if (const int rc = check_cfg(ctx, cfg)) {
        return rc;
    }
    // Row starts in lane (= Morton) order: exclusive scan of nsub[order[k]].
    CSC_BUF(d_ns, uint32_t, "tc_nsub_ordered", n);
    CSC_BUF(d_base, uint64_t, "tc_row_base", n + 1u);
    KT("k_gather_u32", hipLaunchKernelGGL(k_gather_u32, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, d_nsub, d_order, d_ns));
    if (const int rc = scan_impl(ctx, d_ns, n, 1, d_base, nullptr)) {
        return rc;
    }
    csc::prop_params p = make_prop_params(cfg, n, d_state);
    p.order = d_order;
    p.nsub_in = d_nsub;
    p.store_err = ctx->d_flags + 3;
    ctx->flags_fresh = false;
    p.row_base = d_base;
    p.tc_row = d_tc_row;
    p.tcs_xyz = d_tcs_xyz;
    p.tcs_vr = d_tcs_vr;
    p.tcoords = d_tcoords;
    const bool events = cfg->has_exit || cfg->has_reentry;
    dim3 grid(cdiv(n, 64)), block(64);
    if (ctx->pass1_wave) {
        // (the integration that counted the substeps: the same arithmetic repeats them)
        bool waved = false;
        if (const int rc = wave_try_launch(ctx, cfg, p, events, waved)) {
            return rc;
        }
        if (waved) {
            return CSC_OK;
        }
    }
    if (cfg->dyn_kind == CSC_DYN_TAPE) {
        if (const int rc = bind_tape(ctx, cfg, n, p, grid)) {
            return rc;
        }
    }
    KT("k_propagate_store", launch_propagate<true>(cfg->order, cfg->dyn_kind, events, grid, block, ctx->stream, p));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_propagate_summary(csc_ctx *ctx, uint32_t n, const uint32_t *d_nsub, const int32_t *d_status,
                          const uint32_t *d_ev_idx, const double *d_ev_time, const double *d_nf_last_t,
                          csc_prop_summary *out)
{
    const unsigned nb = std::min(cdiv(n, SUM_BLOCK), 1024u);
    CSC_BUF(d_part, sum_partial, "sum_partial", nb);
    KT("k_prop_summary", hipLaunchKernelGGL(k_prop_summary, dim3(nb), dim3(SUM_BLOCK), 0, ctx->stream, n, d_nsub, d_status, d_ev_idx,
                       d_ev_time, d_nf_last_t, d_part));
    CSC_LAUNCH_CHECK();
    std::vector<sum_partial> h(nb);
    CSC_HIP(hipMemcpyAsync(h.data(), d_part, nb * sizeof(sum_partial), hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    csc_prop_summary s{};
    s.nf_idx = 0xffffffffu;
    s.ste_idx = 0xffffffffu;
    for (const auto &o : h) {
        s.flags |= o.flags;
        s.total_substeps += o.nsub;
        s.max_substeps = std::max(s.max_substeps, o.max_nsub);
        if (o.n_nf) {
            if (s.n_nf == 0 || o.nf_time < s.nf_time || (o.nf_time == s.nf_time && o.nf_idx < s.nf_idx)) {
                s.nf_time = o.nf_time;
                s.nf_idx = o.nf_idx;
            }
            s.n_nf += o.n_nf;
        }
        if (o.n_ste) {
            if (s.n_ste == 0 || o.ste_time < s.ste_time || (o.ste_time == s.ste_time && o.ste_idx < s.ste_idx)) {
                s.ste_time = o.ste_time;
                s.ste_idx = o.ste_idx;
                s.ste_ev_idx = o.ste_ev;
            }
            s.n_ste += o.n_ste;
        }
    }
    *out = s;
    return CSC_OK;
}

static int scan_impl(csc_ctx *ctx, const uint32_t *d_in, uint64_t n, uint32_t nseg, uint64_t *d_out64,
                     uint32_t *d_out32)
{
    const unsigned nb = std::max(1u, cdiv(n, csc::SCAN_TILE));
    CSC_BUF(d_bs, uint64_t, "scan_block_sums", static_cast<size_t>(nb) * nseg);
    CSC_BUF(d_tot, uint64_t, "scan_totals", nseg);
    KT("k_scan_block_sums", hipLaunchKernelGGL(csc::k_scan_block_sums, dim3(nb, nseg), dim3(csc::SCAN_BLOCK), 0, ctx->stream, d_in, n, d_bs));
    KT("k_scan_sums", hipLaunchKernelGGL(csc::k_scan_sums, dim3(nseg), dim3(1024), 0, ctx->stream, d_bs, static_cast<uint64_t>(nb),
                       d_tot));
    if (d_out64) {
        KT("k_scan_final", hipLaunchKernelGGL((csc::k_scan_final<uint64_t>), dim3(nb, nseg), dim3(csc::SCAN_BLOCK), 0, ctx->stream, d_in,
                           n, d_bs, d_out64, d_tot));
    } else {
        KT("k_scan_final", hipLaunchKernelGGL((csc::k_scan_final<uint32_t>), dim3(nb, nseg), dim3(csc::SCAN_BLOCK), 0, ctx->stream, d_in,
                           n, d_bs, d_out32, d_tot));
    }
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_exclusive_scan(csc_ctx *ctx, const uint32_t *d_in, uint64_t n, uint64_t *d_out)
{
    return scan_impl(ctx, d_in, n, 1, d_out, nullptr);
}

int csc_global_aabb(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs, float *d_glb,
                    float *d_gub)
{
    const unsigned nb = std::min(cdiv(n, csc::GA_BLOCK * 4), static_cast<unsigned>(csc::GA_MAX_PARTIAL));
    CSC_BUF(d_part, float4, "gaabb_partial", static_cast<size_t>(nb) * nchunks * 2u);
    KT("k_gaabb_partial", hipLaunchKernelGGL(csc::k_gaabb_partial, dim3(nb, nchunks), dim3(csc::GA_BLOCK), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_lbs), reinterpret_cast<const float4 *>(d_ubs), n, d_part));
    KT("k_gaabb_final", hipLaunchKernelGGL(csc::k_gaabb_final, dim3(nchunks), dim3(64), 0, ctx->stream, d_part, nb,
                       reinterpret_cast<float4 *>(d_glb), reinterpret_cast<float4 *>(d_gub)));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

uint32_t csc_global_aabb_noalloc_max_chunks(void)
{
    return static_cast<uint32_t>(CSC_NA_SCRATCH_F4 / 2u);
}

// The same two kernels as csc_global_aabb, the partial boxes in the static device array: no
// hipMalloc / hipFree / event creation on this path. The number of partial boxes per chunk is
// cut to what the array holds; min / max are exact, so the split does not change a bit of the
// result. Every partial slot that stage 2 reads was written by stage 1 of the same call (nothing
// has to start from zero); calls that overlap on one GPU (two contexts, two streams) race.
int csc_global_aabb_noalloc(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs,
                            float *d_glb, float *d_gub)
{
    if (ctx->na_scratch == nullptr) {
        return ctx->invalid("csc_global_aabb_noalloc: the static scratch array was not resolved");
    }
    if (n == 0u || nchunks == 0u || !d_lbs || !d_ubs || !d_glb || !d_gub) {
        return ctx->invalid("csc_global_aabb_noalloc: empty input");
    }
    const size_t max_chunks = CSC_NA_SCRATCH_F4 / 2u;
    if (nchunks > max_chunks) {
        ctx->err = "csc_global_aabb_noalloc: " + std::to_string(nchunks) + " chunks exceed the maximum of "
                   + std::to_string(max_chunks) + " that the static scratch array holds";
        return CSC_ERR_INVALID;
    }
    const unsigned nb = std::min<unsigned>({cdiv(n, csc::GA_BLOCK * 4), static_cast<unsigned>(csc::GA_MAX_PARTIAL),
                                            static_cast<unsigned>(max_chunks / nchunks)});
    float4 *d_part = ctx->na_scratch;
    hipLaunchKernelGGL(csc::k_gaabb_partial, dim3(nb, nchunks), dim3(csc::GA_BLOCK), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_lbs), reinterpret_cast<const float4 *>(d_ubs), n, d_part);
    hipLaunchKernelGGL(csc::k_gaabb_final, dim3(nchunks), dim3(64), 0, ctx->stream, static_cast<const float4 *>(d_part), nb,
                       reinterpret_cast<float4 *>(d_glb), reinterpret_cast<float4 *>(d_gub));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

// ---- K3 / K4 ---------------------------------------------------------------------------

int csc_morton_encode(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_lbs, const float *d_ubs,
                      const float *d_glb, const float *d_gub, uint64_t *d_codes)
{
    ctx->flags_fresh = false;
    KT("k_morton", hipLaunchKernelGGL(csc::k_morton, dim3(cdiv(n, 256), nchunks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_lbs), reinterpret_cast<const float4 *>(d_ubs), n,
                       reinterpret_cast<const float4 *>(d_glb), reinterpret_cast<const float4 *>(d_gub), d_codes,
                       ctx->d_flags));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

// Stable LSD radix sort of (key, value) over bits [0, nbits), nseg segments of n keys each.
// keys_a/vals_a hold the input (vals_a may be null on input = iota) and are clobbered; the
// sorted result is left in *keys_res / *vals_res (one of the a/b buffers).
static int radix_sort(csc_ctx *ctx, uint32_t n, uint32_t nseg, int nbits, uint64_t *keys_a, uint32_t *vals_a,
                      bool vals_iota, uint64_t *keys_b, uint32_t *vals_b, uint64_t **keys_res, uint32_t **vals_res)
{
    // 2048 keys per block (8192 was measured in round 2 and is slower even at 8M keys: the ranking
    // chain of a wave is four times as long and fewer waves hide it).
    const unsigned nblocks = cdiv(n, csc::RS_BLOCK * csc::RS_ITEMS);
    CSC_BUF(d_hist, uint32_t, "rs_hist", static_cast<size_t>(nseg) * csc::RS_BINS * nblocks);
    CSC_BUF(d_btot, uint32_t, "rs_bin_tot", static_cast<size_t>(nseg) * csc::RS_BINS);
    uint64_t *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    bool iota = vals_iota;
    for (int shift = 0; shift < nbits; shift += 8) {
        KT("k_rs_hist", hipLaunchKernelGGL(csc::k_rs_hist<csc::RS_ITEMS>, dim3(nblocks, nseg), dim3(csc::RS_BLOCK), 0, ctx->stream, kin,
                                            n, shift, d_hist, nblocks));
        KT("k_rs_scan", hipLaunchKernelGGL(csc::k_rs_scan, dim3(csc::RS_BINS, nseg), dim3(256), 0, ctx->stream, d_hist, nblocks, d_btot));
        KT("k_rs_scatter", hipLaunchKernelGGL(csc::k_rs_scatter<csc::RS_ITEMS>, dim3(nblocks, nseg), dim3(csc::RS_BLOCK), 0, ctx->stream,
                                               kin, iota ? nullptr : vin, kout, vout, n, shift, d_hist, nblocks, d_btot));
        CSC_LAUNCH_CHECK();
        std::swap(kin, kout);
        std::swap(vin, vout);
        iota = false;
    }
    *keys_res = kin;
    *vals_res = vin;
    return CSC_OK;
}

int csc_sort_gather(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const uint64_t *d_codes, const float *d_lbs,
                    const float *d_ubs, uint32_t *d_vidx, uint64_t *d_srt_codes, float *d_srt_lbs,
                    float *d_srt_ubs, const float *d_recs, float *d_srt_recs)
{
    if ((d_recs == nullptr) != (d_srt_recs == nullptr)) {
        return ctx->invalid("d_recs and d_srt_recs must be given together");
    }
    const size_t tot = static_cast<size_t>(n) * nchunks;
    CSC_BUF(d_ka, uint64_t, "sort_keys_a", tot);
    CSC_BUF(d_va, uint32_t, "sort_vals_a", tot);
    CSC_BUF(d_vb, uint32_t, "sort_vals_b", tot);
    CSC_HIP(hipMemcpyAsync(d_ka, d_codes, tot * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    uint64_t *kres;
    uint32_t *vres;
    // 8 passes: a -> srt -> a ... ends in a; use d_srt_codes as the b buffer and copy at the end.
    if (const int rc = radix_sort(ctx, n, nchunks, 64, d_ka, d_va, true, d_srt_codes, d_vb, &kres, &vres)) {
        return rc;
    }
    if (kres != d_srt_codes) {
        CSC_HIP(hipMemcpyAsync(d_srt_codes, kres, tot * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    }
    CSC_HIP(hipMemcpyAsync(d_vidx, vres, tot * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    KT("k_gather_aabb", hipLaunchKernelGGL(csc::k_gather_aabb, dim3(cdiv(n, 256), nchunks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float4 *>(d_lbs), reinterpret_cast<const float4 *>(d_ubs), d_vidx, n,
                       reinterpret_cast<float4 *>(d_srt_lbs), reinterpret_cast<float4 *>(d_srt_ubs),
                       reinterpret_cast<const float4 *>(d_recs), reinterpret_cast<float4 *>(d_srt_recs)));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

// ---- K5 --------------------------------------------------------------------------------

// canonical: the reference's node order (needed by csc_bvh_export only); otherwise the nodes are
// numbered straight from the Karras ids (k_bvh_emit_fast) and the depth / sort / renumber passes
// are skipped. CASCADE_BVH_CANONICAL=1 forces the former (measurements, tests).
static int bvh_build_impl(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const uint64_t *d_srt_codes, const float *d_srt_lbs,
                          const float *d_srt_ubs, bool canonical)
{
    if (n == 0u || nchunks == 0u) {
        return ctx->invalid("empty BVH input");
    }
    const uint32_t nn_max = 2u * n - 1u;
    const size_t cn = static_cast<size_t>(nchunks) * n, cn1 = static_cast<size_t>(nchunks) * (n + 1u),
                 cnn = static_cast<size_t>(nchunks) * nn_max;
    csc::bvh_bufs b{};
    b.n = n;
    b.nn_max = nn_max;
    b.codes = d_srt_codes;
    b.slb = reinterpret_cast<const float4 *>(d_srt_lbs);
    b.sub = reinterpret_cast<const float4 *>(d_srt_ubs);
    b.begin_bits = (n <= (1u << 24)) ? 24 : 32;
#define BB(field, T, count)                                                                                            \
    {                                                                                                                  \
        T *p_ = ctx->buf<T>("bvh_" #field, count);                                                                     \
        if (!p_) {                                                                                                     \
            return CSC_ERR_HIP;                                                                                        \
        }                                                                                                              \
        b.field = p_;                                                                                                  \
    }
    BB(head, uint32_t, cn)
    BB(excl, uint32_t, cn1)
    BB(run_start, uint32_t, cn1)
    BB(k_parent, int32_t, cnn)
    BB(k_left, int32_t, cn)
    BB(k_right, int32_t, cn)
    BB(k_first, uint32_t, cn)
    BB(k_last, uint32_t, cn)
    BB(k_delta, int32_t, cn)
    BB(k_depth, uint32_t, cnn)
    BB(skey_a, uint64_t, cnn)
    BB(skey_b, uint64_t, cnn)
    BB(sval_a, uint32_t, cnn)
    BB(sval_b, uint32_t, cnn)
    BB(new_of, uint32_t, cnn)
    BB(level_cnt, uint32_t, static_cast<size_t>(nchunks) * csc::BVH_MAX_LEVELS)
    BB(level_off, uint32_t, static_cast<size_t>(nchunks) * (csc::BVH_MAX_LEVELS + 1))
    BB(visit, uint32_t, cnn)
    BB(nb, uint32_t, cnn)
    BB(ne, uint32_t, cnn)
    BB(npar, int32_t, cnn)
    BB(nleft, int32_t, cnn)
    BB(nlb, float4, cnn)
    BB(nub, float4, cnn)
    BB(ndepth, uint32_t, cnn)
    BB(nsplit, int32_t, cnn)
    BB(n_nodes, uint32_t, nchunks)
#undef BB
    CSC_BUF(d_tn, csc::tnode, "bvh_tnodes", cnn);
    const dim3 gp(cdiv(n, 256), nchunks), gn(cdiv(nn_max, 256), nchunks), blk(256);
    // Per-level node counters start from zero (hipMemsetAsync on the same stream).
    CSC_HIP(hipMemsetAsync(b.level_cnt, 0, static_cast<size_t>(nchunks) * csc::BVH_MAX_LEVELS * sizeof(uint32_t), ctx->stream));
    KT("k_bvh_heads", hipLaunchKernelGGL(csc::k_bvh_heads, gp, blk, 0, ctx->stream, b));
    if (const int rc = scan_impl(ctx, b.head, n, nchunks, nullptr, b.excl)) {
        return rc;
    }
    KT("k_bvh_run_start", hipLaunchKernelGGL(csc::k_bvh_run_start, gp, blk, 0, ctx->stream, b));
    KT("k_bvh_karras", hipLaunchKernelGGL(csc::k_bvh_karras, gp, blk, 0, ctx->stream, b));
    if (canonical) {
        KT("k_bvh_depth", hipLaunchKernelGGL(csc::k_bvh_depth, gn, blk, 0, ctx->stream, b));
        KT("k_bvh_level_off", hipLaunchKernelGGL(csc::k_bvh_level_off, dim3(nchunks), dim3(64), 0, ctx->stream, b));
        CSC_LAUNCH_CHECK();
        uint64_t *kres;
        uint32_t *vres;
        if (const int rc = radix_sort(ctx, nn_max, nchunks, b.begin_bits + 8, b.skey_a, b.sval_a, true, b.skey_b,
                                      b.sval_b, &kres, &vres)) {
            return rc;
        }
        KT("k_bvh_newidx", hipLaunchKernelGGL(csc::k_bvh_newidx, gn, blk, 0, ctx->stream, b, vres));
        KT("k_bvh_emit", hipLaunchKernelGGL(csc::k_bvh_emit, gn, blk, 0, ctx->stream, b, vres));
    } else {
        KT("k_bvh_emit", hipLaunchKernelGGL(csc::k_bvh_emit_fast, gn, blk, 0, ctx->stream, b));
    }
    // Bottom-up refit (arrival counters start from zero): the subtrees inside a workgroup's leaves
    // first (Karras numbering only), then one launch across workgroups (the reference's node
    // order: the second kernel alone, from the leaves).
    CSC_HIP(hipMemsetAsync(b.visit, 0, cnn * sizeof(uint32_t), ctx->stream));
    const bool local_first = !canonical;
    if (local_first) {
        KT("k_bvh_refit_local", hipLaunchKernelGGL(csc::k_bvh_refit_local, gp, blk, 0, ctx->stream, b));
    }
    KT("k_bvh_refit", hipLaunchKernelGGL(csc::k_bvh_refit, gn, blk, 0, ctx->stream, b, local_first ? 1 : 0));
    CSC_LAUNCH_CHECK();
    ctx->bvh = b;
    ctx->bvh_packed = false;
    ctx->bvh_n = n;
    ctx->bvh_chunks = nchunks;
    ctx->bvh_canonical = canonical;
    return CSC_OK;
}

int csc_bvh_build(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const uint64_t *d_srt_codes, const float *d_srt_lbs,
                  const float *d_srt_ubs)
{
    const char *e = std::getenv("CASCADE_BVH_CANONICAL");
    return bvh_build_impl(ctx, n, nchunks, d_srt_codes, d_srt_lbs, d_srt_ubs, e && std::atoi(e) != 0);
}

int csc_bvh_export(csc_ctx *ctx, uint32_t chunk, csc_bvh_node *d_nodes, uint64_t cap, uint64_t *n_nodes)
{
    if (chunk >= ctx->bvh_chunks) {
        return ctx->invalid("invalid chunk index");
    }
    if (!ctx->bvh_canonical) {
        // The reference's node order: built now, from the inputs of the last csc_bvh_build (which
        // the caller must not have overwritten since).
        if (const int rc = bvh_build_impl(ctx, ctx->bvh_n, ctx->bvh_chunks, ctx->bvh.codes, reinterpret_cast<const float *>(ctx->bvh.slb),
                                          reinterpret_cast<const float *>(ctx->bvh.sub), true)) {
            return rc;
        }
    }
    uint32_t nn = 0;
    CSC_HIP(hipMemcpyAsync(&nn, ctx->bvh.n_nodes + chunk, sizeof(nn), hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    *n_nodes = nn;
    if (nn > cap) {
        return CSC_ERR_OVERFLOW;
    }
    static_assert(sizeof(csc_bvh_node) == sizeof(csc::bvh_node_ref), "node layout");
    KT("k_bvh_export", hipLaunchKernelGGL(csc::k_bvh_export, dim3(cdiv(nn, 256)), dim3(256), 0, ctx->stream, ctx->bvh, chunk,
                       reinterpret_cast<csc::bvh_node_ref *>(d_nodes)));
    CSC_LAUNCH_CHECK();
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    return CSC_OK;
}

// ---- K6 --------------------------------------------------------------------------------

uint64_t csc_broad_phase_node_pairs(csc_ctx *ctx)
{
    return ctx->bp_node_pairs;
}

uint64_t csc_broad_phase_tiles(csc_ctx *ctx)
{
    return ctx->bp_tiles;
}

int csc_set_chunk_map(csc_ctx *ctx, uint32_t first, uint32_t nchunks_global)
{
    ctx->chunk_first = nchunks_global ? first : 0u;
    ctx->chunk_global = nchunks_global;
    return CSC_OK;
}

int csc_set_shard(csc_ctx *ctx, uint32_t rank, uint32_t nranks)
{
    if (nranks == 0u || rank >= nranks) {
        return ctx->invalid("invalid shard (rank, nranks)");
    }
    ctx->shard_rank = rank;
    ctx->shard_n = nranks;
    return CSC_OK;
}

// Dual traversal (k_bp_dual.hpp): one launch per level of the pair tree, the frontier sizes stay
// on the device; the host looks at them every few levels to stop.
static_assert(csc::NP_FLAT_SHARDS == csc::DT_SHARDS, "flat pair lists");
// csc_broad_phase_filtered: the tiles' hits go through the narrow phase's exclusion test on chip.
struct bp_fuse {
    const float *srt_recs;
    double conj_thresh;
};
static int broad_phase_dual(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                            const uint32_t *d_vidx, const uint8_t *d_active, const csc::tnode *d_tn, const uint8_t *d_sact,
                            const uint32_t *d_sinfo, const uint32_t *d_sleaf, const uint32_t *d_multi, uint32_t tile_max,
                            uint64_t *n_pairs, const bp_fuse *fuse = nullptr)
{
    using namespace csc;
    const size_t ctr_per_level = static_cast<size_t>(DT_SHARDS) * DT_CTR_STRIDE;
    CSC_BUF(d_cnt, uint32_t, "dt_counters", ctr_per_level * (DT_MAX_LEVELS + 5));
    uint32_t *d_pair_cnt = d_cnt + ctr_per_level * (DT_MAX_LEVELS + 1);
    uint32_t *d_tile_cnt = d_cnt + ctr_per_level * (DT_MAX_LEVELS + 2);
    uint32_t *d_fuse_cnt = d_cnt + ctr_per_level * (DT_MAX_LEVELS + 3);
    uint32_t *d_cand_cnt = d_cnt + ctr_per_level * (DT_MAX_LEVELS + 4);
    // (fused: the candidates that survive the on-chip exclusion test; what the last superstep needed)
    uint32_t cand_cap = fuse ? static_cast<uint32_t>(std::max<size_t>(ctx->bufs["np_cand"].cap / sizeof(uint4), 262144)) : 0u;
    // Tiles: about 6 per particle at 1M objects with tiles of 16.
    uint32_t tile_cap = tile_max ? std::max<uint32_t>({ctx->dt_tile_cap, cdiv(static_cast<uint64_t>(n) * nchunks, DT_SHARDS) * 8u, 4096u}) : 1u;
    // First guesses: 4 frontier items and 8 pairs per particle and chunk; afterwards what the last
    // superstep needed (populations change slowly), and an overflow redoes the traversal.
    uint32_t fr_cap = std::max<uint32_t>({ctx->dt_fr_cap, cdiv(static_cast<uint64_t>(n) * nchunks, DT_SHARDS) * 4u, 4096u});
    uint32_t pair_cap = std::max<uint32_t>({ctx->dt_pair_cap, cdiv(static_cast<uint64_t>(n) * nchunks, DT_SHARDS) * 8u, 4096u});
    std::vector<uint32_t> hc(ctr_per_level * (DT_MAX_LEVELS + 5));
    // Workgroups per shard (each strides over the DT_BLOCK-item tiles of its shard): 512 of 128
    // threads, i.e. 16384 workgroups in all, measured best at 1M objects (per traversal: 1.65 ms
    // with 64 x 256 threads, 1.47 with 256 x 256, 1.42 with 512 x 128, 1.45 with 1024 x 128, 1.47
    // with 1024 x 64: more, smaller workgroups balance the load and hide the gather latency better
    // until the empty ones of the small levels cost more); 128 for small populations.
    const unsigned per_shard = static_cast<uint64_t>(n) * nchunks >= 200000u ? 16384u / DT_SHARDS : 4096u / DT_SHARDS;
    const unsigned grid = DT_SHARDS * per_shard;
    uint64_t total = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        CSC_BUF(d_fa, unsigned long long, "dt_frontier_a", static_cast<size_t>(fr_cap) * DT_SHARDS);
        CSC_BUF(d_fb, unsigned long long, "dt_frontier_b", static_cast<size_t>(fr_cap) * DT_SHARDS);
        CSC_BUF(d_pairs, unsigned long long, "dt_pairs", static_cast<size_t>(pair_cap) * DT_SHARDS);
        CSC_BUF(d_tiles, unsigned long long, "dt_tiles", static_cast<size_t>(tile_cap) * DT_SHARDS);
        uint4 *d_cand = nullptr;
        if (fuse) {
            CSC_BUF(d_cand_, uint4, "np_cand", cand_cap);
            d_cand = d_cand_;
        }
        CSC_HIP(hipMemsetAsync(d_cnt, 0, hc.size() * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(k_bp_dual_init, dim3(cdiv(nchunks, 64)), dim3(64), 0, ctx->stream, nchunks, ctx->bvh.nn_max, ctx->bvh.nleft,
                           d_fa, d_cnt);
        dual_params q{};
        q.tn = d_tn;
        q.nn_max = ctx->bvh.nn_max;
        q.n = n;
        q.nb = ctx->bvh.nb;
        q.ne = ctx->bvh.ne;
        q.nlb = ctx->bvh.nlb;
        q.nub = ctx->bvh.nub;
        q.slb = reinterpret_cast<const float4 *>(d_srt_lbs);
        q.sub = reinterpret_cast<const float4 *>(d_srt_ubs);
        q.vidx = d_vidx;
        q.active = d_active;
        q.sact = d_sact;
        q.fr_cap = fr_cap;
        q.pairs = d_pairs;
        q.pair_cnt = d_pair_cnt;
        q.pair_cap = pair_cap;
        q.shard_level = 12;
        q.shard_rank = ctx->shard_rank;
        q.shard_n = ctx->shard_n;
        q.tile_max = tile_max;
        q.tiles = d_tiles;
        q.tile_cnt = d_tile_cnt;
        q.tile_cap = tile_cap;
        q.sinfo = d_sinfo;
        q.sleaf = d_sleaf;
        q.multi_flag = d_multi;
        if (fuse) {
            q.srecs = reinterpret_cast<const float4 *>(fuse->srt_recs);
            q.thr_extra = static_cast<float>(fuse->conj_thresh);
            q.cand = d_cand;
            q.cand_cnt = d_cand_cnt;
            q.cand_cap = cand_cap;
            q.fuse_cnt = d_fuse_cnt;
        }
        int level = 0;
        bool done = false;
        // Tiles are handed to k_bp_tiles as soon as the levels the last traversal needed (+ 1) have
        // been launched, without waiting to see that the frontier is empty: ONE read-back of all
        // counters (and of the sticky flags) per traversal. If the tree got deeper, the remaining
        // levels run and a second tile launch takes the tiles added since.
        std::vector<uint32_t> tiles_done(DT_SHARDS, 0u);
        bool tiles_seen = false;
        int flags_h[4] = {0, 0, 0, 0};
        auto launch_tiles = [&]() {
            for (int s = 0; s < DT_SHARDS; ++s) {
                q.tile_begin[s] = tiles_done[s];
            }
            // Workgroups stride over the tile batches of their shard: 8 per CU fill the chip.
            const unsigned tg = static_cast<uint64_t>(n) * nchunks >= 200000u ? 2048u : 256u;
            if (fuse) {
                KT("k_bp_tiles", hipLaunchKernelGGL((k_bp_tiles<8, true>), dim3(tg), dim3(64 * DT_TILE_WAVES), 0, ctx->stream, q));
            } else {
                KT("k_bp_tiles", hipLaunchKernelGGL(k_bp_tiles<8>, dim3(tg), dim3(64 * DT_TILE_WAVES), 0, ctx->stream, q));
            }
        };
        bool tiles_launched = false;
        while (!done && level < DT_MAX_LEVELS) {
            // A batch of levels, then a look at the size of the frontier they left.
            // (first batch: as many levels as the last traversal needed, plus one)
            const int batch_end = std::min(DT_MAX_LEVELS, level + (level == 0 ? (ctx->dt_levels ? ctx->dt_levels + 1 : 32) : 8));
            ctx->kt_begin("k_bp_dual");
            for (; level < batch_end; ++level) {
                q.level = static_cast<uint32_t>(level);
                q.in = (level % 2 == 0) ? d_fa : d_fb;
                q.out = (level % 2 == 0) ? d_fb : d_fa;
                q.in_cnt = d_cnt + ctr_per_level * level;
                q.out_cnt = d_cnt + ctr_per_level * (level + 1);
                // Workgroups for this level: what its frontier needed last time (+50 %), at most
                // `grid`; the tiles are strided over, so a frontier that grew is still covered.
                // (16384 workgroups that find nothing to do cost 6 us per level, and half of the
                // ~32 levels hold a few thousand items.)
                unsigned g = grid;
                if (static_cast<size_t>(level) < ctx->dt_level_items.size()) {
                    const uint64_t want = cdiv(ctx->dt_level_items[level] * 3u / 2u + 1u, static_cast<uint64_t>(DT_SHARDS) * DT_BLOCK);
                    g = DT_SHARDS * static_cast<unsigned>(std::min<uint64_t>(std::max<uint64_t>(want, 1u), per_shard));
                }
                hipLaunchKernelGGL(k_bp_dual, dim3(g), dim3(DT_BLOCK), 0, ctx->stream, q);
            }
            ctx->kt_end("k_bp_dual");
            CSC_LAUNCH_CHECK();
            if (tile_max != 0u && !tiles_launched) {
                launch_tiles();
                CSC_LAUNCH_CHECK();
                tiles_launched = true;
            }
            // Everything the host wants to know, in one go.
            CSC_HIP(hipMemcpyAsync(hc.data(), d_cnt, ctr_per_level * (level + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            CSC_HIP(hipMemcpyAsync(hc.data() + ctr_per_level * (DT_MAX_LEVELS + 1), d_pair_cnt, 4 * ctr_per_level * sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, ctx->stream));
            CSC_HIP(hipMemcpyAsync(flags_h, ctx->d_flags, sizeof(flags_h), hipMemcpyDeviceToHost, ctx->stream));
            CSC_HIP(hipStreamSynchronize(ctx->stream));
            done = true;
            for (int s = 0; s < DT_SHARDS; ++s) {
                done = done && hc[ctr_per_level * level + static_cast<size_t>(s) * DT_CTR_STRIDE] == 0u;
            }
            if (tile_max != 0u && !tiles_seen) {
                // (what the first tile launch saw: the tile counts at that point)
                for (int s = 0; s < DT_SHARDS; ++s) {
                    tiles_done[s] = std::min(hc[ctr_per_level * (DT_MAX_LEVELS + 2) + static_cast<size_t>(s) * DT_CTR_STRIDE], tile_cap);
                }
                tiles_seen = true;
            }
        }
        if (!done) {
            ctx->err = "broad phase: the dual traversal did not terminate";
            return CSC_ERR_OVERFLOW;
        }
        if (tile_max != 0u) {
            // Tiles that arrived after the first tile launch (the tree got deeper than last time)?
            bool more = false;
            for (int s = 0; s < DT_SHARDS; ++s) {
                more = more || std::min(hc[ctr_per_level * (DT_MAX_LEVELS + 2) + static_cast<size_t>(s) * DT_CTR_STRIDE], tile_cap) > tiles_done[s];
            }
            if (more) {
                launch_tiles();
                CSC_LAUNCH_CHECK();
                CSC_HIP(hipMemcpyAsync(hc.data() + ctr_per_level * (DT_MAX_LEVELS + 1), d_pair_cnt, 4 * ctr_per_level * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
                CSC_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
        ctx->flags_fresh = flags_h[0] == 0 && flags_h[1] == 0 && flags_h[2] == 0 && flags_h[3] == 0;
        uint32_t max_fr = 0, max_pr = 0, max_tl = 0;
        uint64_t ntiles = 0;
        for (int s = 0; s < DT_SHARDS; ++s) {
            const uint32_t v = hc[ctr_per_level * (DT_MAX_LEVELS + 2) + static_cast<size_t>(s) * DT_CTR_STRIDE];
            max_tl = std::max(max_tl, v);
            ntiles += v;
        }
        ctx->bp_tiles = ntiles;
        ctx->dt_tile_cap = std::max(ctx->dt_tile_cap, max_tl + max_tl / 4u + 1024u);
        uint64_t items = 0;
        int used_levels = 0;
        ctx->dt_level_items.assign(static_cast<size_t>(level) + 1u, 0);
        for (int l = 0; l <= level; ++l) {
            uint64_t at_level = 0;
            uint32_t max_shard = 0;
            for (int s = 0; s < DT_SHARDS; ++s) {
                const uint32_t v = hc[ctr_per_level * l + static_cast<size_t>(s) * DT_CTR_STRIDE];
                max_fr = std::max(max_fr, v);
                max_shard = std::max(max_shard, v);
                at_level += v;
            }
            items += at_level;
            // (for the grid of the next traversal: every shard gets the workgroups its fullest one
            // needs -- the upper levels sit in a few shards, and a workgroup that walks 16 tiles of
            // its shard one after the other costs 16 memory round trips)
            ctx->dt_level_items[l] = static_cast<uint64_t>(max_shard) * DT_SHARDS;
            if (at_level) {
                used_levels = l + 1;
            }
        }
        ctx->dt_levels = used_levels;
        ctx->bp_node_pairs = items;
        total = 0;
        for (int s = 0; s < DT_SHARDS; ++s) {
            const uint32_t v = hc[ctr_per_level * (DT_MAX_LEVELS + 1) + static_cast<size_t>(s) * DT_CTR_STRIDE];
            max_pr = std::max(max_pr, v);
            total += v;
            ctx->bp_flat_cnt[s] = v;
        }
        ctx->dt_fr_cap = std::max(ctx->dt_fr_cap, max_fr + max_fr / 4u + 1024u);
        ctx->dt_pair_cap = std::max(ctx->dt_pair_cap, max_pr + max_pr / 4u + 1024u);
        uint32_t ncand = 0;
        if (fuse && max_fr <= fr_cap && max_pr <= pair_cap && max_tl <= tile_cap) {
            // The pairs that the traversal wrote itself (members of oversized multi-particle leaves:
            // rare, usually none) take the stored-list version of the same test, into the same list.
            if (total != 0u) {
                np_params f{};
                f.n = n;
                f.nchunks = nchunks;
                f.flat_pairs = d_pairs;
                f.flat_cap = pair_cap;
                uint32_t flat_tiles = 0;
                for (int s = 0; s < NP_FLAT_SHARDS; ++s) {
                    f.flat_cnt[s] = ctx->bp_flat_cnt[s];
                    f.flat_tile_off[s] = flat_tiles;
                    flat_tiles += cdiv(ctx->bp_flat_cnt[s], NP_FLAT_TILE);
                }
                f.flat_tile_off[NP_FLAT_SHARDS] = flat_tiles;
                f.vidx = d_vidx;
                f.srt_recs = reinterpret_cast<const float4 *>(fuse->srt_recs);
                f.thr_extra = fuse->conj_thresh;
                f.cand = d_cand;
                f.cand_cap = cand_cap;
                f.counters = d_cand_cnt - 5; // (the kernel touches counters[5] only)
                KT("k_np_prefilter", hipLaunchKernelGGL((k_np_prefilter_flat<true>), dim3(std::min<uint32_t>(flat_tiles, 2048u)),
                                                         dim3(NP_FLAT_BLOCK), 0, ctx->stream, f));
                CSC_LAUNCH_CHECK();
                CSC_HIP(hipMemcpyAsync(hc.data() + ctr_per_level * (DT_MAX_LEVELS + 4), d_cand_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                       ctx->stream));
                CSC_HIP(hipStreamSynchronize(ctx->stream));
            }
            ncand = hc[ctr_per_level * (DT_MAX_LEVELS + 4)];
            for (int s = 0; s < DT_SHARDS; ++s) {
                total += hc[ctr_per_level * (DT_MAX_LEVELS + 3) + static_cast<size_t>(s) * DT_CTR_STRIDE];
            }
        }
        if (fuse && ncand > cand_cap) {
            cand_cap = ncand + ncand / 4u + 16u; // (counted in full: one more pass is enough)
            continue;
        }
        if (max_fr <= fr_cap && max_pr <= pair_cap && max_tl <= tile_cap) {
            ctx->bp_fused = fuse != nullptr;
            if (fuse) {
                ctx->np_cand_list = d_cand;
                ctx->np_cand_count = ncand;
                ctx->np_candidates = ncand;
                ctx->np_cand_valid = true;
            }
            ctx->bp_flat = true;
            ctx->bp_flat_pairs = d_pairs;
            ctx->bp_flat_cap = pair_cap;
            ctx->bp_total = total;
            ctx->bp_vidx = d_vidx;
            ctx->bp_n = n;
            ctx->bp_chunks = nchunks;
            *n_pairs = total;
            return CSC_OK;
        }
        // A frontier or the pair list overflowed (the counters kept counting): grow and redo. A
        // truncated frontier hides part of what follows, hence the generous factor.
        const bool fr_over = max_fr > fr_cap;
        const bool truncated = fr_over || max_tl > tile_cap;
        if (fr_over) {
            fr_cap = max_fr * 4u;
        }
        if (fr_over || max_tl > tile_cap) {
            tile_cap = std::max(tile_cap, max_tl * (fr_over ? 4u : 1u) + max_tl / 4u + 1024u);
        }
        if (truncated || max_pr > pair_cap) {
            pair_cap = std::max(pair_cap, max_pr * (truncated ? 4u : 1u) + max_pr / 4u + 1024u);
        }
    }
    ctx->err = "broad phase pair list overflow";
    return CSC_ERR_OVERFLOW;
}

static int broad_phase_impl(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                            const uint32_t *d_vidx, const uint8_t *d_active, uint64_t *n_pairs, const bp_fuse *fuse)
{
    ctx->bp_fused = false;
    if (n != ctx->bvh_n || nchunks > ctx->bvh_chunks) {
        return ctx->invalid("broad phase does not match the last BVH build");
    }
    const size_t cnn = static_cast<size_t>(ctx->bvh_chunks) * ctx->bvh.nn_max;
    CSC_BUF(d_tn, csc::tnode, "bvh_tnodes", cnn);
    // Wide traversal nodes (need the Morton permutation and the activity flags for the leaf
    // descriptors, so they are rebuilt on every call).
    CSC_BUF(d_sact, uint8_t, "bvh_sorted_active", static_cast<size_t>(ctx->bvh_chunks) * n);
    if (ctx->bp_mode < 0) {
        const char *e = std::getenv("CASCADE_BP_MODE");
        ctx->bp_mode = (e && std::string(e) == "walk") ? 0 : 1;
    }
    // Tiles at the bottom of the dual traversal (k_bp_tiles): ranges of up to 8 particles
    // (measured best at 1M objects: traversal + tiles 0.57 + 0.55 ms per chunk, against 0.41 + 0.93
    // with 16 and 1.56 + 0 without tiles), positions below 2^24. CASCADE_DT_TILE overrides (0:
    // none, at most 16; measurements).
    const bool dual = ctx->bp_mode == 1 && ctx->bvh.nn_max < csc::DT_MAX_NODES && nchunks <= csc::DT_MAX_CHUNKS;
    uint32_t tile_max = 0;
    if (dual && n < csc::DT_TILE_MAX_N) {
        const char *e = std::getenv("CASCADE_DT_TILE"); // (0: the traversal down to the leaves; tests)
        tile_max = (e && std::atoi(e) == 0) ? 0u : 8u;
    }
    uint32_t *d_sinfo = nullptr, *d_sleaf = nullptr;
    CSC_BUF(d_multi, uint32_t, "bvh_multi_flag", 1);
    CSC_HIP(hipMemsetAsync(d_multi, 0, sizeof(uint32_t), ctx->stream));
    if (tile_max != 0u) {
        CSC_BUF(d_si, uint32_t, "bvh_sorted_info", static_cast<size_t>(ctx->bvh_chunks) * n);
        CSC_BUF(d_sl, uint32_t, "bvh_sorted_leaf", static_cast<size_t>(ctx->bvh_chunks) * n);
        d_sinfo = d_si;
        d_sleaf = d_sl;
    }
    KT("k_bvh_pack", hipLaunchKernelGGL(csc::k_bvh_pack, dim3(cdiv(ctx->bvh.nn_max, 256), ctx->bvh_chunks), dim3(256), 0,
                                         ctx->stream, ctx->bvh, d_vidx, d_active, d_tn, d_sact, d_sinfo, d_sleaf, d_multi));
    CSC_LAUNCH_CHECK();
    ctx->bvh_packed = true;
    ctx->bp_flat = false;
    ctx->bp_node_pairs = 0;
    ctx->bp_tiles = 0;
    if (dual) {
        return broad_phase_dual(ctx, n, nchunks, d_srt_lbs, d_srt_ubs, d_vidx, d_active, d_tn, d_sact, d_sinfo, d_sleaf, d_multi, tile_max, n_pairs,
                                tile_max != 0u ? fuse : nullptr);
    }
    csc::bp_params q{};
    q.n = n;
    q.nn_max = ctx->bvh.nn_max;
    q.slb = reinterpret_cast<const float4 *>(d_srt_lbs);
    q.sub = reinterpret_cast<const float4 *>(d_srt_ubs);
    q.vidx = d_vidx;
    q.tn = d_tn;
    q.nlb = ctx->bvh.nlb;
    q.nub = ctx->bvh.nub;
    q.nleft = ctx->bvh.nleft;
    q.active = d_active;
    q.counter = ctx->d_bp_counter;
    q.err_flag = ctx->d_flags + 2;
    ctx->flags_fresh = false;
    // Query groups of this process (all of them when there is one process).
    const uint32_t ngroups_all = cdiv(n, 64);
    const uint32_t ngroups = cdiv(ngroups_all, csc::BP_SHARD_RUN * ctx->shard_n) * csc::BP_SHARD_RUN;
    q.shard_rank = ctx->shard_rank;
    q.shard_n = ctx->shard_n;
    q.ngroups = ngroups_all;
    const uint32_t nlists = ngroups * nchunks;
    const dim3 grid(ngroups, nchunks), blk(64);
    CSC_BUF(d_head, int32_t, "bp_head", nlists);
    CSC_BUF(d_nblk, uint32_t, "bp_nblk", nlists);
    CSC_BUF(d_tchunk, uint32_t, "bp_tchunk", nlists);
    CSC_BUF(d_off, uint32_t, "bp_list_off", nlists + 1u);
    CSC_BUF(d_ctr, unsigned long long, "bp_shard_ctr", csc::PL_SHARDS * csc::PL_CTR_STRIDE);
#ifdef CSC_BP_GROUP_STATS
    CSC_BUF(d_gstats, uint32_t, "bp_gstats", 2 * static_cast<size_t>(nlists));
    q.gstats = d_gstats;
#endif
    // Pair blocks come from per-shard pools; the pool capacity follows the last superstep's use
    // (populations change slowly), and an overflow re-runs the traversal.
    uint32_t cap_ps = std::max<uint32_t>({ctx->bp_cap_per_shard, cdiv(nlists * 2u, csc::PL_SHARDS), 64u});
    std::vector<unsigned long long> hc(csc::PL_SHARDS * csc::PL_CTR_STRIDE);
    unsigned long long total_pairs = 0, total_blocks = 0, max_local = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        const size_t cap_blocks = static_cast<size_t>(cap_ps) * csc::PL_SHARDS;
        CSC_BUF(d_pairs, uint2, "bp_pairs", cap_blocks * csc::PB_SIZE);
        CSC_BUF(d_pbc, uint32_t, "bp_pb_count", cap_blocks);
        CSC_BUF(d_pbn, int32_t, "bp_pb_next", cap_blocks);
        q.pl.pairs = d_pairs;
        q.pl.pb_count = d_pbc;
        q.pl.pb_next = d_pbn;
        q.pl.head = d_head;
        q.pl.nblk = d_nblk;
        q.pl.tchunk = d_tchunk;
        q.pl.shard_ctr = d_ctr;
        q.pl.cap_per_shard = cap_ps;
        q.pl.nlists = nlists;
        CSC_HIP(hipMemsetAsync(d_ctr, 0, hc.size() * sizeof(unsigned long long), ctx->stream));
        CSC_HIP(hipMemsetAsync(d_nblk, 0, nlists * sizeof(uint32_t), ctx->stream));
        KT("k_broad_phase", hipLaunchKernelGGL(csc::k_broad_phase, grid, blk, 0, ctx->stream, static_cast<const csc::tnode *>(d_tn), q));
        CSC_LAUNCH_CHECK();
        CSC_HIP(hipMemcpyAsync(hc.data(), d_ctr, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        total_pairs = total_blocks = max_local = 0;
        for (int sh = 0; sh < csc::PL_SHARDS; ++sh) {
            const unsigned long long v = hc[static_cast<size_t>(sh) * csc::PL_CTR_STRIDE];
            total_pairs += v & ((1ull << 40) - 1u);
            total_blocks += v >> 40;
            max_local = std::max(max_local, v >> 40);
        }
        if (max_local <= cap_ps) {
            break;
        }
        cap_ps = static_cast<uint32_t>(max_local + max_local / 8u + 16u);
    }
    if (max_local > cap_ps) {
        ctx->err = "broad phase pair list overflow";
        return CSC_ERR_OVERFLOW;
    }
    // Group-major order of the blocks for the narrow phase.
    CSC_BUF(d_order, uint2, "bp_order", std::max<unsigned long long>(total_blocks, 1));
    if (const int rc = scan_impl(ctx, d_nblk, nlists, 1, nullptr, d_off)) {
        return rc;
    }
    KT("k_pl_order", hipLaunchKernelGGL(csc::k_pl_order, dim3(cdiv(nlists, 256)), dim3(256), 0, ctx->stream, q.pl, d_off, d_order));
    CSC_LAUNCH_CHECK();
    ctx->bp_cap_per_shard = std::max<uint32_t>(cap_ps, static_cast<uint32_t>(max_local + max_local / 4u + 16u));
    ctx->bp_total = total_pairs;
    ctx->bp_blocks = static_cast<uint32_t>(total_blocks);
    ctx->bp_pl = q.pl;
    ctx->bp_order = d_order;
    ctx->bp_vidx = d_vidx;
    ctx->bp_n = n;
    ctx->bp_chunks = nchunks;
    *n_pairs = total_pairs;
    return CSC_OK;
}

int csc_broad_phase(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                    const uint32_t *d_vidx, const uint8_t *d_active, uint64_t *n_pairs)
{
    return broad_phase_impl(ctx, n, nchunks, d_srt_lbs, d_srt_ubs, d_vidx, d_active, n_pairs, nullptr);
}

int csc_broad_phase_filtered(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                             const uint32_t *d_vidx, const uint8_t *d_active, double conj_thresh, const float *d_srt_recs,
                             uint64_t *n_pairs, uint64_t *n_cand)
{
    if (d_srt_recs == nullptr) {
        return ctx->invalid("csc_broad_phase_filtered needs the sorted trajectory records");
    }
    const bp_fuse fuse{d_srt_recs, conj_thresh};
    if (const int rc = broad_phase_impl(ctx, n, nchunks, d_srt_lbs, d_srt_ubs, d_vidx, d_active, n_pairs, &fuse)) {
        return rc;
    }
    if (!ctx->bp_fused) {
        // (no tiles in this configuration: the stored pair list, then the same test over it)
        return csc_narrow_phase_prefilter(ctx, n, nchunks, conj_thresh, d_srt_recs, n_cand);
    }
    if (n_cand) {
        *n_cand = ctx->np_cand_count;
    }
    return CSC_OK;
}

#ifdef CSC_BP_GROUP_STATS
int csc_debug_gstats(csc_ctx *ctx, uint32_t *out, uint32_t nlists)
{
    CSC_HIP(hipMemcpy(out, ctx->bufs["bp_gstats"].p, 2 * static_cast<size_t>(nlists) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CSC_OK;
}
#endif

int csc_broad_phase_stats(csc_ctx *ctx, uint32_t n, uint32_t nchunks, const float *d_srt_lbs, const float *d_srt_ubs,
                          const uint32_t *d_vidx, const uint8_t *d_active, uint64_t *nodes_visited)
{
    if (n != ctx->bp_n || nchunks > ctx->bp_chunks || !ctx->bvh_packed) {
        return ctx->invalid("broad phase statistics need a preceding csc_broad_phase on the same trees");
    }
    csc::bp_params q{};
    q.n = n;
    q.nn_max = ctx->bvh.nn_max;
    q.slb = reinterpret_cast<const float4 *>(d_srt_lbs);
    q.sub = reinterpret_cast<const float4 *>(d_srt_ubs);
    q.vidx = d_vidx;
    q.tn = static_cast<const csc::tnode *>(ctx->bufs["bvh_tnodes"].p);
    q.nlb = ctx->bvh.nlb;
    q.nub = ctx->bvh.nub;
    q.nleft = ctx->bvh.nleft;
    q.active = d_active;
    q.counter = ctx->d_bp_counter;
    q.err_flag = ctx->d_flags + 2;
    ctx->flags_fresh = false;
    CSC_HIP(hipMemsetAsync(ctx->d_bp_counter, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(csc::k_broad_phase_stats, dim3(cdiv(n, csc::BP_BLOCK), nchunks), dim3(csc::BP_BLOCK), 0, ctx->stream, q);
    CSC_LAUNCH_CHECK();
    unsigned long long v = 0;
    CSC_HIP(hipMemcpyAsync(&v, ctx->d_bp_counter + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    *nodes_visited = v;
    return CSC_OK;
}

int csc_broad_phase_get(csc_ctx *ctx, uint32_t chunk, uint32_t *pairs, uint64_t cap, uint64_t *count)
{
    if (chunk >= ctx->bp_chunks) {
        return ctx->invalid("invalid chunk index");
    }
    if (ctx->bp_fused) {
        return ctx->invalid("the last broad phase (csc_broad_phase_filtered) kept no pair list");
    }
    if (ctx->bp_flat) {
        std::vector<uint32_t> vx(ctx->bp_n);
        CSC_HIP(hipMemcpy(vx.data(), ctx->bp_vidx + static_cast<size_t>(chunk) * ctx->bp_n, vx.size() * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
        uint64_t k = 0;
        std::vector<unsigned long long> h;
        for (int s = 0; s < csc::DT_SHARDS; ++s) {
            h.resize(ctx->bp_flat_cnt[s]);
            if (h.empty()) {
                continue;
            }
            CSC_HIP(hipMemcpy(h.data(), ctx->bp_flat_pairs + static_cast<size_t>(s) * ctx->bp_flat_cap,
                              h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (const unsigned long long rec : h) {
                if (static_cast<uint32_t>(rec >> 56) != chunk) {
                    continue;
                }
                if (k < cap) {
                    // (the tile kernel records positions in no particular order)
                    const uint32_t oa = vx[static_cast<uint32_t>(rec) & 0xfffffffu], ob = vx[static_cast<uint32_t>(rec >> 28) & 0xfffffffu];
                    pairs[k * 2u] = std::min(oa, ob);
                    pairs[k * 2u + 1u] = std::max(oa, ob);
                }
                ++k;
            }
        }
        *count = k;
        return CSC_OK;
    }
    const csc::pair_lists &pl = ctx->bp_pl;
    const uint32_t nb = ctx->bp_blocks;
    std::vector<uint2> order(nb);
    if (nb) {
        CSC_HIP(hipMemcpyAsync(order.data(), ctx->bp_order, nb * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
    }
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> vx(ctx->bp_n);
    CSC_HIP(hipMemcpy(vx.data(), ctx->bp_vidx + static_cast<size_t>(chunk) * ctx->bp_n, vx.size() * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    uint64_t k = 0;
    std::vector<uint2> h(csc::PB_SIZE);
    for (uint32_t e = 0; e < nb; ++e) {
        if (order[e].y != chunk) {
            continue;
        }
        uint32_t cnt = 0;
        CSC_HIP(hipMemcpy(&cnt, pl.pb_count + order[e].x, sizeof(cnt), hipMemcpyDeviceToHost));
        CSC_HIP(hipMemcpy(h.data(), pl.pairs + static_cast<size_t>(order[e].x) * csc::PB_SIZE, cnt * sizeof(uint2),
                          hipMemcpyDeviceToHost));
        for (uint32_t t = 0; t < cnt; ++t, ++k) {
            if (k < cap) {
                pairs[k * 2u] = vx[h[t].x];
                pairs[k * 2u + 1u] = vx[h[t].y];
            }
        }
    }
    *count = k;
    return CSC_OK;
}

// ---- K7 --------------------------------------------------------------------------------

// Where the narrow phase takes its candidate pairs from.
struct np_source {
    // (a) the pair blocks of the last broad phase (sorted positions), optionally pre-filtered
    const uint2 *order = nullptr;
    uint32_t norder = 0;
    const uint32_t *pb_count = nullptr;
    const uint2 *pb_pairs = nullptr;
    const uint32_t *vidx = nullptr;
    const float *srt_recs = nullptr; // null: no pre-filter, every pair is a candidate
    // (a') the flat pair lists of the dual traversal
    const unsigned long long *flat_pairs = nullptr;
    uint32_t flat_cap = 0;
    const uint32_t *flat_cnt = nullptr;
    // (b) a ready-made candidate list {pi, pj, chunk, 0} (original indices)
    const uint4 *cand = nullptr;
    uint64_t npairs = 0; // number of pairs of the source (capacity bound for the candidates)
};

// The pinned host array of the last narrow phase's conjunction records, for at least `count` of them.
static int conj_pinned(csc_ctx *ctx, size_t count)
{
    if (count > ctx->conj_cap) {
        if (ctx->conj) {
            (void)hipHostFree(ctx->conj);
            ctx->conj = nullptr;
            ctx->conj_cap = 0;
        }
        const size_t want = count + count / 2u + 1024u;
        CSC_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->conj), want * sizeof(csc_conjunction), hipHostMallocDefault));
        ctx->conj_cap = want;
    }
    return CSC_OK;
}

// Lazy mode: the host copy of a list that was left on the device.
static int conj_materialise(csc_ctx *ctx)
{
    if (ctx->conj_host_valid || ctx->conj_dev == nullptr) {
        return CSC_OK;
    }
    if (const int rc = conj_pinned(ctx, ctx->conj_count)) {
        return rc;
    }
    CSC_HIP(hipMemcpyAsync(ctx->conj, ctx->conj_dev, ctx->conj_count * sizeof(csc_conjunction), hipMemcpyDeviceToHost, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    ctx->conj_host_valid = true;
    return CSC_OK;
}

static int narrow_impl(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                       double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                       const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub, const double *d_tcs_xyz,
                       const double *d_tcs_vr, const double *d_tcoords, const np_source &src, uint64_t *n_coll,
                       uint64_t *n_conj, uint32_t *warn_flags, bool prefilter_only = false)
{
    const csc::order_api *oapi = order == 20u ? nullptr : csc_order_api(static_cast<int>(order));
    if (order != 20u && oapi == nullptr) {
        ctx->err = "Taylor order " + std::to_string(order) + " is not built";
        return CSC_ERR_UNSUPPORTED;
    }
    ctx->coll.clear();
    ctx->conj_count = 0;
    ctx->conj_dev = nullptr;
    ctx->conj_host_valid = true;
    ctx->np_candidates = 0;
    *n_coll = 0;
    *n_conj = 0;
    if (warn_flags) {
        *warn_flags = 0;
    }
    ctx->np_cand_valid = false;
    if (src.npairs == 0) {
        if (prefilter_only) {
            ctx->np_cand_list = nullptr;
            ctx->np_cand_count = 0;
            ctx->np_cand_valid = true;
        }
        return CSC_OK;
    }
    if (src.npairs > 0xffffff00ull && src.cand == nullptr && src.srt_recs == nullptr) {
        return ctx->invalid("too many pairs for an unfiltered narrow phase");
    }
    CSC_BUF(d_cnt, uint32_t, "np_counters", 8);
    uint32_t coll_cap = static_cast<uint32_t>(std::max<size_t>(ctx->bufs["np_coll"].cap / sizeof(csc::coll_rec), 4096));
    uint32_t conj_cap = static_cast<uint32_t>(std::max<size_t>(ctx->bufs["np_conj"].cap / sizeof(csc::conj_rec), 4096));
    uint32_t surv_cap = static_cast<uint32_t>(std::max<size_t>(ctx->bufs["np_surv"].cap / (2 * sizeof(uint4)), 65536));
    uint32_t defer_cap = static_cast<uint32_t>(std::max<size_t>(ctx->bufs["np_defer"].cap / (2 * sizeof(uint4)), 65536));
    // Candidates: everything when there is no pre-filter, else what the last superstep needed.
    uint32_t cand_cap;
    if (src.cand != nullptr || src.srt_recs == nullptr) {
        cand_cap = static_cast<uint32_t>(src.npairs);
    } else {
        cand_cap = static_cast<uint32_t>(std::min<uint64_t>(
            src.npairs, std::max<size_t>(ctx->bufs["np_cand"].cap / sizeof(uint4), 262144)));
    }
    uint32_t h[8];
    // The lists feed each other (candidates -> survivors / deferred -> events), so an overflow
    // upstream hides the true size downstream: up to one retry per list, plus slack.
    ctx->np_attempts = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        ctx->np_attempts = static_cast<uint32_t>(attempt) + 1u;
        CSC_BUF(d_coll, csc::coll_rec, "np_coll", coll_cap);
        CSC_BUF(d_conj, csc::conj_rec, "np_conj", conj_cap);
        CSC_BUF(d_surv, uint4, "np_surv", static_cast<size_t>(surv_cap) * 2u);
        CSC_BUF(d_defer, uint4, "np_defer", static_cast<size_t>(defer_cap) * 2u);
        uint4 *d_cand = nullptr;
        if (src.cand == nullptr) {
            CSC_BUF(d_cand_, uint4, "np_cand", cand_cap);
            d_cand = d_cand_;
        }
        CSC_HIP(hipMemsetAsync(d_cnt, 0, 8 * sizeof(uint32_t), ctx->stream));
        csc::np_params q{};
        q.n = n;
        q.nchunks = nchunks;
        q.chunk_first = ctx->chunk_first;
        q.nchunks_global = ctx->chunk_global ? ctx->chunk_global : nchunks;
        q.ct = ct;
        q.delta_t = delta_t;
        q.init_hi = init_time_hi;
        q.init_lo = init_time_lo;
        q.conj_thresh2 = conj_thresh * conj_thresh;
        q.thr_extra = conj_thresh;
        q.conj_fast = ctx->np_conj_fast ? 1 : 0;
        q.state = d_state;
        q.active = d_active;
        q.tc_row = d_tc_row;
        q.nsub = d_nsub;
        q.tcs_xyz = d_tcs_xyz;
        q.tcs_vr = d_tcs_vr;
        q.tcoords = d_tcoords;
        q.order = src.order;
        q.norder = src.norder;
        q.pb_count = src.pb_count;
        q.pairs = src.pb_pairs;
        q.vidx = src.vidx;
        uint32_t flat_tiles = 0;
        if (src.flat_pairs != nullptr) {
            q.flat_pairs = src.flat_pairs;
            q.flat_cap = src.flat_cap;
            for (int s = 0; s < csc::NP_FLAT_SHARDS; ++s) {
                q.flat_cnt[s] = src.flat_cnt[s];
                q.flat_tile_off[s] = flat_tiles;
                flat_tiles += cdiv(src.flat_cnt[s], csc::NP_FLAT_TILE);
            }
            q.flat_tile_off[csc::NP_FLAT_SHARDS] = flat_tiles;
            // Runs of 256 consecutive workgroups (of 1024 pairs) per XCD (measured in round 2: 4.46 ms in
            // dispatch order, 4.41 / 4.31 / 4.33 with runs of 64 / 1024 / 16384).
            q.flat_xcd_run = 256u;
        }
        q.srt_recs = reinterpret_cast<const float4 *>(src.srt_recs);
        q.cand = src.cand ? const_cast<uint4 *>(src.cand) : d_cand;
        q.cand_cap = cand_cap;
        q.coll = d_coll;
        q.conj = d_conj;
        q.coll_cap = coll_cap;
        q.conj_cap = conj_cap;
        q.surv = d_surv;
        q.surv_cap = surv_cap;
        q.defer = d_defer;
        q.defer_cap = defer_cap;
        q.counters = d_cnt;
        // Pre-filter over all pairs -> candidates; fast exclusion over the candidates (windows
        // that need a polynomial translation in a second kernel); root finding over the
        // survivors. The grids of the follow-up kernels cover the list capacities; lanes beyond
        // the device-side counts exit at once.
        if (src.cand != nullptr) {
            const uint32_t nc = static_cast<uint32_t>(src.npairs);
            CSC_HIP(hipMemcpyAsync(d_cnt + 5, &nc, sizeof(nc), hipMemcpyHostToDevice, ctx->stream));
        } else if (src.flat_pairs != nullptr) {
            // Workgroups loop over the tiles: 8 per CU (each buffers its candidates, one atomic per flush).
            const uint32_t flat_grid = std::max<uint32_t>(1u, std::min<uint32_t>(flat_tiles, 2048u));
            if (src.srt_recs != nullptr) {
                KT("k_np_prefilter", hipLaunchKernelGGL((csc::k_np_prefilter_flat<true>), dim3(flat_grid), dim3(csc::NP_FLAT_BLOCK), 0, ctx->stream, q));
            } else {
                KT("k_np_prefilter", hipLaunchKernelGGL((csc::k_np_prefilter_flat<false>), dim3(flat_grid), dim3(csc::NP_FLAT_BLOCK), 0, ctx->stream, q));
            }
        } else if (src.srt_recs != nullptr) {
            KT("k_np_prefilter", hipLaunchKernelGGL((csc::k_np_prefilter<true>), dim3(src.norder * (csc::PB_SIZE / 64)), dim3(64), 0,
                                                     ctx->stream, q));
        } else {
            KT("k_np_prefilter", hipLaunchKernelGGL((csc::k_np_prefilter<false>), dim3(src.norder * (csc::PB_SIZE / 64)), dim3(64), 0,
                                                     ctx->stream, q));
        }
        if (prefilter_only) {
            // The candidates stay on the device for csc_narrow_phase_finish.
            CSC_LAUNCH_CHECK();
            CSC_HIP(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            CSC_HIP(hipStreamSynchronize(ctx->stream));
            if (h[5] <= cand_cap) {
                ctx->np_candidates = h[5];
                ctx->np_cand_list = q.cand;
                ctx->np_cand_count = h[5];
                ctx->np_cand_valid = true;
                return CSC_OK;
            }
            cand_cap = std::max(cand_cap, h[5] + h[5] / 4u + 16u);
            continue;
        }
        if (oapi == nullptr) {
            KT("k_np_fex", hipLaunchKernelGGL((csc::k_np_fex<20>), dim3(cdiv(cand_cap, 64)), dim3(64), 0, ctx->stream, q));
            KT("k_np_fex_tr", hipLaunchKernelGGL((csc::k_np_fex_tr<20>), dim3(cdiv(defer_cap, 64)), dim3(64), 0, ctx->stream, q));
            KT("k_np_roots", hipLaunchKernelGGL((csc::k_np_roots<20>), dim3(cdiv(surv_cap, 64)), dim3(64), 0, ctx->stream, q));
        } else {
            KT("k_np_fex", oapi->np_fex(dim3(cdiv(cand_cap, 64)), ctx->stream, q));
            KT("k_np_fex_tr", oapi->np_fex_tr(dim3(cdiv(defer_cap, 64)), ctx->stream, q));
            KT("k_np_roots", oapi->np_roots(dim3(cdiv(surv_cap, 64)), ctx->stream, q));
        }
        CSC_LAUNCH_CHECK();
        CSC_HIP(hipMemcpyAsync(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        const bool tm = std::getenv("CASCADE_HOST_TIMERS") != nullptr;
        auto tm0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (tm) {
                const auto t1 = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[narrow] %-12s %9.3f ms (attempt %d)\n", what,
                             std::chrono::duration<double, std::milli>(t1 - tm0).count(), attempt);
                tm0 = t1;
            }
        };
        lap("kernels");
        if (tm) {
            std::fprintf(stderr, "[narrow] counters: coll %u conj %u warn %u surv %u defer %u cand %u c6 %u c7 %u\n", h[0], h[1], h[2],
                         h[3], h[4], h[5], h[6], h[7]);
        }
        if (h[0] <= coll_cap && h[1] <= conj_cap && h[3] <= surv_cap && h[4] <= defer_cap && h[5] <= cand_cap) {
            ctx->np_candidates = h[5];
            ctx->coll.resize(h[0]);
            ctx->conj_count = h[1];
            // Large conjunction lists are put in time order on the device (radix sort of the time
            // keys, gather of the 128-byte records); ties and small lists are ordered on the host.
            const bool dev_sorted = h[1] > 32768u;
            bool lazy = ctx->conj_lazy && dev_sorted;
            ctx->conj_dev = nullptr;
            ctx->conj_host_valid = true;
            if (!lazy) {
                if (const int rc = conj_pinned(ctx, h[1])) {
                    return rc;
                }
            }
            lap("pinned alloc");
            if (h[0]) {
                CSC_HIP(hipMemcpyAsync(ctx->coll.data(), d_coll, h[0] * sizeof(csc::coll_rec), hipMemcpyDeviceToHost,
                                       ctx->stream));
            }
            uint32_t n_ties = 0;
            if (dev_sorted) {
                const uint32_t nc = h[1];
                CSC_BUF(d_ka, uint64_t, "conj_keys_a", nc);
                CSC_BUF(d_kb, uint64_t, "conj_keys_b", nc);
                CSC_BUF(d_va, uint32_t, "conj_vals_a", nc);
                CSC_BUF(d_vb, uint32_t, "conj_vals_b", nc);
                CSC_BUF(d_sorted, csc::conj_rec, "np_conj_sorted", nc);
                KT("k_conj_keys", hipLaunchKernelGGL(k_conj_keys, dim3(cdiv(nc, 256)), dim3(256), 0, ctx->stream, nc, d_conj, d_ka));
                uint64_t *kres;
                uint32_t *vres;
                if (const int rc = radix_sort(ctx, nc, 1, 64, d_ka, d_va, true, d_kb, d_vb, &kres, &vres)) {
                    return rc;
                }
                KT("k_conj_gather", hipLaunchKernelGGL(k_conj_gather, dim3(cdiv(static_cast<uint64_t>(nc) * 8u, 256)), dim3(256), 0,
                                                        ctx->stream, nc, reinterpret_cast<const uint4 *>(d_conj), vres,
                                                        reinterpret_cast<uint4 *>(d_sorted)));
                KT("k_count_ties", hipLaunchKernelGGL(k_count_ties, dim3(cdiv(nc, 256)), dim3(256), 0, ctx->stream, nc, kres, d_cnt + 7));
                CSC_LAUNCH_CHECK();
                CSC_HIP(hipMemcpyAsync(&n_ties, d_cnt + 7, sizeof(n_ties), hipMemcpyDeviceToHost, ctx->stream));
                if (lazy) {
                    // The records stay where they are; equal times (rare) are put in (i, j) order on
                    // the host, so a list with ties comes down after all.
                    CSC_HIP(hipStreamSynchronize(ctx->stream));
                    if (n_ties != 0u) {
                        lazy = false;
                        if (const int rc = conj_pinned(ctx, h[1])) {
                            return rc;
                        }
                    } else {
                        ctx->conj_dev = reinterpret_cast<const csc_conjunction *>(d_sorted);
                        ctx->conj_host_valid = false;
                    }
                }
                if (!lazy) {
                    CSC_HIP(hipMemcpyAsync(ctx->conj, d_sorted, nc * sizeof(csc::conj_rec), hipMemcpyDeviceToHost, ctx->stream));
                }
            } else if (h[1]) {
                CSC_HIP(hipMemcpyAsync(ctx->conj, d_conj, h[1] * sizeof(csc::conj_rec), hipMemcpyDeviceToHost,
                                       ctx->stream));
            }
            CSC_HIP(hipStreamSynchronize(ctx->stream));
            lap("sort + d2h");
            // Deterministic container order: (time, i, j).
            std::sort(ctx->coll.begin(), ctx->coll.end(), [](const csc_collision &a, const csc_collision &b) {
                return std::tie(a.time, a.i, a.j) < std::tie(b.time, b.i, b.j);
            });
            const auto conj_less = [](const csc_conjunction &a, const csc_conjunction &b) {
                return std::tie(a.time, a.i, a.j) < std::tie(b.time, b.i, b.j);
            };
            if (lazy || (dev_sorted && n_ties == 0u)) {
                // in time order, all times distinct: nothing left to do
            } else if (dev_sorted) {
                // Already in time order: only runs of equal times need (i, j) order.
                for (size_t b0 = 0; b0 < ctx->conj_count;) {
                    size_t e0 = b0 + 1;
                    while (e0 < ctx->conj_count && ctx->conj[e0].time == ctx->conj[b0].time) {
                        ++e0;
                    }
                    if (e0 - b0 > 1) {
                        std::sort(ctx->conj + b0, ctx->conj + e0, conj_less);
                    }
                    b0 = e0;
                }
            } else if (ctx->conj_count) {
                std::sort(ctx->conj, ctx->conj + ctx->conj_count, conj_less);
            }
            lap("tie order");
            *n_coll = h[0];
            *n_conj = h[1];
            if (warn_flags) {
                *warn_flags = h[2];
            }
            return CSC_OK;
        }
        // Output overflow: grow and run again (results are recomputed from scratch).
        // When the candidate list overflowed, the downstream counts are those of the truncated
        // list: scale them by the same factor.
        const double up = h[5] > cand_cap ? static_cast<double>(h[5]) / cand_cap : 1.;
        auto grow = [up](uint32_t cap, uint32_t seen) {
            const double want = static_cast<double>(seen) * up * 1.25 + 16.;
            return std::max(cap, static_cast<uint32_t>(std::min(want, 4.0e9)));
        };
        coll_cap = grow(coll_cap, h[0]);
        conj_cap = grow(conj_cap, h[1]);
        surv_cap = grow(surv_cap, h[3]);
        defer_cap = grow(defer_cap, h[4]);
        cand_cap = std::max(cand_cap, h[5] + h[5] / 4u + 16u);
    }
    ctx->err = "narrow phase output overflow";
    return CSC_ERR_OVERFLOW;
}

int csc_narrow_phase(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                     double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                     const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub, const double *d_tcs_xyz,
                     const double *d_tcs_vr, const double *d_tcoords, const float *d_srt_recs, uint64_t *n_coll,
                     uint64_t *n_conj, uint32_t *warn_flags)
{
    if (n != ctx->bp_n || nchunks > ctx->bp_chunks || ctx->bp_fused) {
        return ctx->invalid("narrow phase does not match the last broad phase");
    }
    // After a terminal-event shrink the kernels skip the pairs of the dropped chunks.
    np_source src;
    if (ctx->bp_flat) {
        src.flat_pairs = ctx->bp_flat_pairs;
        src.flat_cap = ctx->bp_flat_cap;
        src.flat_cnt = ctx->bp_flat_cnt;
    }
    src.order = ctx->bp_order;
    src.norder = ctx->bp_blocks;
    src.pb_count = ctx->bp_pl.pb_count;
    src.pb_pairs = ctx->bp_pl.pairs;
    src.vidx = ctx->bp_vidx;
    src.srt_recs = d_srt_recs;
    src.npairs = ctx->bp_total;
    return narrow_impl(ctx, n, nchunks, ct, delta_t, init_time_hi, init_time_lo, conj_thresh, order, d_state, d_active,
                       d_tc_row, d_nsub, d_tcs_xyz, d_tcs_vr, d_tcoords, src, n_coll, n_conj, warn_flags);
}

int csc_narrow_phase_pairs(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                           double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                           const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub,
                           const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords,
                           const uint32_t *d_pairs, const uint64_t *pair_off, uint64_t *n_coll, uint64_t *n_conj,
                           uint32_t *warn_flags)
{
    const uint64_t npairs = pair_off[nchunks];
    if (npairs > 0xffffff00ull) {
        return ctx->invalid("too many injected pairs");
    }
    CSC_BUF(d_off, uint64_t, "np_pair_off", nchunks + 1u);
    CSC_BUF(d_rec, uint4, "np_pair_rec", std::max<uint64_t>(npairs, 1));
    CSC_HIP(hipMemcpyAsync(d_off, pair_off, (nchunks + 1u) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    if (npairs) {
        KT("k_tag_pairs", hipLaunchKernelGGL(k_tag_pairs, dim3(cdiv(npairs, 256)), dim3(256), 0, ctx->stream, npairs, d_pairs, d_off,
                                              nchunks, d_rec));
        CSC_LAUNCH_CHECK();
    }
    np_source src;
    src.cand = d_rec;
    src.npairs = npairs;
    return narrow_impl(ctx, n, nchunks, ct, delta_t, init_time_hi, init_time_lo, conj_thresh, order, d_state, d_active,
                       d_tc_row, d_nsub, d_tcs_xyz, d_tcs_vr, d_tcoords, src, n_coll, n_conj, warn_flags);
}

int csc_narrow_phase_prefilter(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double conj_thresh, const float *d_srt_recs,
                               uint64_t *n_cand)
{
    if (n != ctx->bp_n || nchunks > ctx->bp_chunks || ctx->bp_fused) {
        return ctx->invalid("narrow phase does not match the last broad phase");
    }
    if (d_srt_recs == nullptr) {
        return ctx->invalid("csc_narrow_phase_prefilter needs the sorted trajectory records");
    }
    np_source src;
    if (ctx->bp_flat) {
        src.flat_pairs = ctx->bp_flat_pairs;
        src.flat_cap = ctx->bp_flat_cap;
        src.flat_cnt = ctx->bp_flat_cnt;
    }
    src.order = ctx->bp_order;
    src.norder = ctx->bp_blocks;
    src.pb_count = ctx->bp_pl.pb_count;
    src.pb_pairs = ctx->bp_pl.pairs;
    src.vidx = ctx->bp_vidx;
    src.srt_recs = d_srt_recs;
    src.npairs = ctx->bp_total;
    uint64_t nc = 0, nj = 0;
    const int rc = narrow_impl(ctx, n, nchunks, 0., 0., 0., 0., conj_thresh, 20u, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr, src, &nc, &nj, nullptr, true);
    if (rc == CSC_OK && n_cand) {
        *n_cand = ctx->np_cand_count;
    }
    return rc;
}

int csc_candidate_objects(csc_ctx *ctx, uint32_t n, const uint32_t *d_nsub, uint32_t *d_list, uint32_t *n_list,
                          uint64_t *n_rows)
{
    if (!ctx->np_cand_valid) {
        return ctx->invalid("csc_candidate_objects needs a preceding csc_narrow_phase_prefilter");
    }
    *n_list = 0;
    if (n_rows) {
        *n_rows = 0;
    }
    if (ctx->np_cand_count == 0 || n == 0) {
        return CSC_OK;
    }
    const uint32_t nc = static_cast<uint32_t>(ctx->np_cand_count);
    CSC_BUF(d_need, uint32_t, "cand_need", n);
    CSC_BUF(d_off, uint64_t, "cand_off", n + 1ull);
    CSC_HIP(hipMemsetAsync(d_need, 0, n * sizeof(uint32_t), ctx->stream));
    KT("k_cand_mark", hipLaunchKernelGGL(k_cand_mark, dim3(cdiv(nc, 256)), dim3(256), 0, ctx->stream, nc, ctx->np_cand_list, d_need));
    CSC_LAUNCH_CHECK();
    if (const int rc = scan_impl(ctx, d_need, n, 1, d_off, nullptr)) {
        return rc;
    }
    KT("k_cand_compact", hipLaunchKernelGGL(k_cand_compact, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, d_need, d_off, d_list));
    CSC_LAUNCH_CHECK();
    uint64_t tot[2] = {0, 0};
    CSC_HIP(hipMemcpyAsync(&tot[0], d_off + n, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (n_rows) {
        CSC_BUF(d_w, uint32_t, "cand_weights", n);
        CSC_BUF(d_roff, uint64_t, "cand_row_off", n + 1ull);
        KT("k_cand_weights", hipLaunchKernelGGL(k_cand_weights, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, d_need, d_nsub, d_w));
        CSC_LAUNCH_CHECK();
        if (const int rc = scan_impl(ctx, d_w, n, 1, d_roff, nullptr)) {
            return rc;
        }
        CSC_HIP(hipMemcpyAsync(&tot[1], d_roff + n, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    CSC_HIP(hipStreamSynchronize(ctx->stream));
    *n_list = static_cast<uint32_t>(tot[0]);
    if (n_rows) {
        *n_rows = tot[1];
    }
    return CSC_OK;
}

int csc_narrow_phase_finish(csc_ctx *ctx, uint32_t n, uint32_t nchunks, double ct, double delta_t, double init_time_hi,
                            double init_time_lo, double conj_thresh, uint32_t order, const double *d_state,
                            const uint8_t *d_active, const uint64_t *d_tc_row, const uint32_t *d_nsub,
                            const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords, uint64_t *n_coll,
                            uint64_t *n_conj, uint32_t *warn_flags)
{
    if (!ctx->np_cand_valid) {
        return ctx->invalid("csc_narrow_phase_finish needs a preceding csc_narrow_phase_prefilter");
    }
    np_source src;
    src.cand = ctx->np_cand_list;
    src.npairs = ctx->np_cand_count;
    return narrow_impl(ctx, n, nchunks, ct, delta_t, init_time_hi, init_time_lo, conj_thresh, order, d_state, d_active,
                       d_tc_row, d_nsub, d_tcs_xyz, d_tcs_vr, d_tcoords, src, n_coll, n_conj, warn_flags);
}

uint64_t csc_narrow_phase_candidates(csc_ctx *ctx)
{
    return ctx->np_candidates;
}

uint32_t csc_narrow_phase_attempts(csc_ctx *ctx)
{
    return ctx->np_attempts;
}

int csc_narrow_phase_get(csc_ctx *ctx, csc_collision *coll, uint64_t coll_cap, csc_conjunction *conj,
                         uint64_t conj_cap)
{
    if (coll && !ctx->coll.empty()) {
        std::memcpy(coll, ctx->coll.data(), std::min<uint64_t>(coll_cap, ctx->coll.size()) * sizeof(csc_collision));
    }
    if (const int rc = conj_materialise(ctx)) {
        return rc;
    }
    if (conj && ctx->conj_count) {
        std::memcpy(conj, ctx->conj, std::min<uint64_t>(conj_cap, ctx->conj_count) * sizeof(csc_conjunction));
    }
    return CSC_OK;
}

int csc_narrow_phase_view(csc_ctx *ctx, const csc_collision **coll, uint64_t *n_coll, const csc_conjunction **conj,
                          uint64_t *n_conj)
{
    if (const int rc = conj_materialise(ctx)) {
        return rc;
    }
    *coll = ctx->coll.data();
    *n_coll = ctx->coll.size();
    *conj = ctx->conj;
    *n_conj = ctx->conj_count;
    return CSC_OK;
}

int csc_narrow_phase_lazy(csc_ctx *ctx, int on)
{
    ctx->conj_lazy = on != 0;
    return CSC_OK;
}

int csc_narrow_phase_counts(csc_ctx *ctx, const csc_collision **coll, uint64_t *n_coll, uint64_t *n_conj, int *conj_on_device)
{
    *coll = ctx->coll.data();
    *n_coll = ctx->coll.size();
    *n_conj = ctx->conj_count;
    *conj_on_device = (ctx->conj_dev != nullptr && !ctx->conj_host_valid) ? 1 : 0;
    return CSC_OK;
}

int csc_conj_store_append(csc_ctx *ctx, uint64_t count)
{
    if (ctx->conj_dev == nullptr || count > ctx->conj_count) {
        return ctx->invalid("csc_conj_store_append: the last narrow phase left no such records on the device");
    }
    if (count == 0u) {
        return CSC_OK;
    }
    CSC_HIP(hipSetDevice(ctx->device));
    if (ctx->cstore_n + count > ctx->cstore_cap) {
        const size_t want = std::max<size_t>(2u * ctx->cstore_cap, ctx->cstore_n + count);
        csc_conjunction *nw = nullptr;
        CSC_HIP(hipMalloc(reinterpret_cast<void **>(&nw), want * sizeof(csc_conjunction)));
        if (ctx->cstore_n) {
            CSC_HIP(hipMemcpyAsync(nw, ctx->cstore, ctx->cstore_n * sizeof(csc_conjunction), hipMemcpyDeviceToDevice, ctx->stream));
        }
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->cstore) {
            (void)hipFree(ctx->cstore);
        }
        ctx->cstore = nw;
        ctx->cstore_cap = want;
    }
    CSC_HIP(hipMemcpyAsync(ctx->cstore + ctx->cstore_n, ctx->conj_dev, count * sizeof(csc_conjunction), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->cstore_n += count;
    return CSC_OK;
}

uint64_t csc_conj_store_size(csc_ctx *ctx)
{
    return ctx->cstore_n;
}

int csc_conj_store_download(csc_ctx *ctx, csc_conjunction *dst)
{
    if (ctx->cstore_n != 0u) {
        if (const int rc = csc_download(ctx, dst, ctx->cstore, ctx->cstore_n * sizeof(csc_conjunction), nullptr)) {
            return rc;
        }
    }
    ctx->cstore_n = 0;
    return CSC_OK;
}

int csc_conj_store_reset(csc_ctx *ctx, int release)
{
    ctx->cstore_n = 0;
    if (release && ctx->cstore) {
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->cstore);
        ctx->cstore = nullptr;
        ctx->cstore_cap = 0;
    }
    return CSC_OK;
}

// ---- K8 and helpers --------------------------------------------------------------------

int csc_dense_propagate(csc_ctx *ctx, uint32_t n, uint32_t order, double t, const uint64_t *d_tc_row,
                        const uint32_t *d_nsub, const double *d_tcs_xyz, const double *d_tcs_vr, const double *d_tcoords,
                        double *d_out)
{
    ctx->flags_fresh = false;
    if (order != 20u) {
        const csc::order_api *oapi = csc_order_api(static_cast<int>(order));
        if (oapi == nullptr) {
            ctx->err = "Taylor order " + std::to_string(order) + " is not built";
            return CSC_ERR_UNSUPPORTED;
        }
        KT("k_dense", oapi->dense(dim3(cdiv(n, 256)), ctx->stream, n, t, d_tc_row, d_nsub, d_tcs_xyz, d_tcs_vr, d_tcoords, d_out,
                                  ctx->d_flags + 1));
        CSC_LAUNCH_CHECK();
        return CSC_OK;
    }
    KT("k_dense", hipLaunchKernelGGL((csc::k_dense<20>), dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, t, d_tc_row, d_nsub,
                       d_tcs_xyz, d_tcs_vr, d_tcoords, d_out, ctx->d_flags + 1));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_copy_final_state(csc_ctx *ctx, uint32_t n, const double *d_final_state, double *d_state)
{
    KT("k_copy_final_state", hipLaunchKernelGGL(k_copy_final_state, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, d_final_state, d_state));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_activity_flags(csc_ctx *ctx, uint32_t n, const double *d_state, double min_coll_radius, int with_conj,
                       const uint64_t *coll_wl, uint64_t n_coll_wl, const uint64_t *conj_wl, uint64_t n_conj_wl,
                       uint8_t *d_active)
{
    const uint8_t *d_mask = nullptr;
    if (n_coll_wl || n_conj_wl) {
        std::vector<uint8_t> mask(n, 0);
        for (uint64_t k = 0; k < n_coll_wl; ++k) {
            if (coll_wl[k] < n) {
                mask[coll_wl[k]] |= 1u;
            }
        }
        for (uint64_t k = 0; k < n_conj_wl; ++k) {
            if (conj_wl[k] < n) {
                mask[conj_wl[k]] |= 2u;
            }
        }
        CSC_BUF(d_m, uint8_t, "wl_mask", n);
        CSC_HIP(hipMemcpyAsync(d_m, mask.data(), n, hipMemcpyHostToDevice, ctx->stream));
        CSC_HIP(hipStreamSynchronize(ctx->stream));
        d_mask = d_m;
    }
    KT("k_activity", hipLaunchKernelGGL(k_activity, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, n, d_state, min_coll_radius,
                       with_conj, n_coll_wl == 0 ? 1 : 0, n_conj_wl == 0 ? 1 : 0, d_mask, d_active));
    CSC_LAUNCH_CHECK();
    return CSC_OK;
}

int csc_remove_particles(csc_ctx *ctx, uint32_t n, const uint64_t *idx, uint64_t n_idx, double *d_state, uint32_t npars,
                         double *d_pars, uint32_t *n_out)
{
    for (uint64_t k = 0; k < n_idx; ++k) {
        if (idx[k] >= n || (k && idx[k] <= idx[k - 1])) {
            ctx->err = "csc_remove_particles: the indices must be sorted, unique and smaller than n";
            return CSC_ERR_INVALID;
        }
    }
    if (n_out) {
        *n_out = n - static_cast<uint32_t>(n_idx);
    }
    if (n_idx == 0 || n == 0) {
        return CSC_OK;
    }
    const uint32_t keep = n - static_cast<uint32_t>(n_idx);
    std::vector<uint32_t> idx32(idx, idx + n_idx);
    CSC_BUF(d_idx, uint32_t, "rm_idx", n_idx);
    CSC_HIP(hipMemcpyAsync(d_idx, idx32.data(), n_idx * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t wmax = std::max(7u, npars);
    CSC_BUF(d_tmp, double, "rm_tmp", static_cast<uint64_t>(std::max(keep, 1u)) * wmax);
    auto compact = [&](double *d_rows, uint32_t width) -> int {
        KT("k_remove_rows", hipLaunchKernelGGL(k_remove_rows, dim3(cdiv(static_cast<uint64_t>(n) * width, 256)), dim3(256), 0,
                           ctx->stream, n, width, d_idx, static_cast<uint32_t>(n_idx), d_rows, d_tmp));
        CSC_LAUNCH_CHECK();
        CSC_HIP(hipMemcpyAsync(d_rows, d_tmp, static_cast<uint64_t>(keep) * width * sizeof(double), hipMemcpyDeviceToDevice,
                               ctx->stream));
        return CSC_OK;
    };
    if (int rc = compact(d_state, 7u)) {
        return rc;
    }
    if (npars != 0u && d_pars != nullptr) {
        if (int rc = compact(d_pars, npars)) {
            return rc;
        }
    }
    CSC_HIP(hipStreamSynchronize(ctx->stream)); // idx32 is pageable host memory
    return CSC_OK;
}

} // extern "C"
