// cascade_amd/csrc/k_bp_dual.hpp
//
// K6, second formulation: simultaneous (dual) traversal of the BVH against itself.
//
// Replaces src/sim_broad_phase.cpp:92-202 like k_broad_phase (k_bvh.hpp), with the same output
// set. The reference walks the tree once per particle; its pair set is
//   {(a, b) : orig_a < orig_b, box(a) overlaps box(leaf(b)), some activity flag set}
// (see k_bvh.hpp). A parent box is the exact union of its children's boxes, so if box(a)
// overlaps box(leaf(b)) then every pair of ancestors (X of a, Y of b) has overlapping boxes:
// descending from (root, root) into the child pairs whose boxes overlap reaches every such
// leaf pair exactly once, and at the leaves the reference's own test decides. This visits
// ~2.4 node pairs per output pair (7 box tests) where the per-particle walks visit ~20 nodes
// per output pair (1M-object LEO population: 6.0e7 node pairs against 4.8e8 node visits).
//
// One launch per level of the pair tree. A frontier item is a node pair of one time chunk,
//   a | a_kind << 26 | b << 28 | b_kind << 54 | chunk << 56,
// kind 0: internal node (value = canonical index), 1: single-particle leaf (value = its sorted
// position: box, original index and activity come from the sorted particle arrays with no
// dependent loads), 2: multi-particle leaf (value = canonical index; rare: equal Morton codes),
// a thread expands one item: the (up to four) child pairs with overlapping boxes go to the
// next frontier, leaf-leaf pairs to the pair list. Both outputs are compacted per workgroup
// (ballot-free block scan, one atomic per DT_BLOCK items and output kind) into DT_SHARDS independent
// regions, each with its own counter on its own cache line: same-address atomics from
// thousands of workgroups serialise at ~13 ns each.
//
// A pair is recorded as pos_a | pos_b << 28 | chunk << 56 (sorted positions within the chunk; the
// readers of the list order the two by original index). Limits of this encoding: 2^26 nodes per chunk, 256 chunks; the
// caller falls back on k_broad_phase beyond them.
//
// Multi-GPU: every process expands the first levels completely (a few thousand items), at
// level q.shard_level it keeps the items whose hash falls on its rank and from then on only
// their descendants; pairs found before that level are dealt by the same hash. The subtrees of
// distinct items are disjoint, so the processes' pair sets partition the global set.
#pragma once

#include <type_traits>
#include "k_bvh.hpp"
#include "k_chord.hpp"

namespace csc
{

constexpr int DT_BLOCK = 128;
constexpr int DT_SHARDS = 32;
constexpr int DT_CTR_STRIDE = 16; // u32 counters 64 bytes apart
constexpr int DT_MAX_LEVELS = 2 * BVH_MAX_LEVELS + 2;
constexpr std::uint32_t DT_MAX_NODES = 1u << 26;
constexpr std::uint32_t DT_MAX_CHUNKS = 256;

struct dual_params {
    const tnode *tn;           // [C][nn_max] wide nodes (internal nodes only)
    std::uint32_t nn_max, n;
    const std::uint32_t *nb, *ne; // [C][nn_max] particle range of a node
    const float4 *nlb, *nub;      // [C][nn_max] node boxes
    const float4 *slb, *sub;      // [C][n] particle boxes (sorted)
    const std::uint32_t *vidx;    // [C][n]
    const std::uint8_t *active;   // [n]
    const std::uint8_t *sact;     // [C][n] activity flags by sorted position (single-particle leaves)
    const unsigned long long *in; // [DT_SHARDS][fr_cap]
    unsigned long long *out;
    const std::uint32_t *in_cnt;  // [DT_SHARDS * DT_CTR_STRIDE]
    std::uint32_t *out_cnt;
    std::uint32_t fr_cap;         // frontier capacity per shard
    unsigned long long *pairs;    // [DT_SHARDS][pair_cap]
    std::uint32_t *pair_cnt;      // [DT_SHARDS * DT_CTR_STRIDE]
    std::uint32_t pair_cap;
    std::uint32_t level, shard_level, shard_rank, shard_n;
    // Tiles (see k_bp_tiles): a child pair whose two particle ranges hold at most tile_max
    // particles each is not descended into but recorded; 0 = no tiles.
    std::uint32_t tile_max;
    unsigned long long *tiles;    // [DT_SHARDS][tile_cap]
    std::uint32_t *tile_cnt;      // [DT_SHARDS * DT_CTR_STRIDE]
    std::uint32_t tile_cap;
    const std::uint32_t *multi_flag; // != 0: some leaf of the build holds several particles (k_bvh_pack)
    std::uint32_t tile_begin[DT_SHARDS]; // k_bp_tiles starts here in every shard's list (a second launch: the tiles added since)
    const std::uint32_t *sinfo;   // [C][n] orig | act << 24 | multi-leaf member << 26 (k_bvh_pack)
    const std::uint32_t *sleaf;   // [C][n] leaf of a multi-leaf member
    // k_bp_tiles<.., FUSE>: the hits of the tiles are not stored. They are counted (fuse_cnt, per
    // shard) and put through the geometric exclusion test of the narrow phase (k_chord.hpp) while
    // they sit in the wave's buffer; only the survivors leave the chip, as narrow-phase candidates
    // {orig_i < orig_j, chunk, 0}.
    const float4 *srecs;          // [C][n][4] sampled-trajectory records in sorted order
    float thr_extra;              // conjunction threshold (0: collisions only)
    uint4 *cand;                  // [cand_cap]
    std::uint32_t *cand_cnt;      // one counter
    std::uint32_t cand_cap;
    std::uint32_t *fuse_cnt;      // [DT_SHARDS * DT_CTR_STRIDE] box-overlap pairs found by the tiles
};

// DT_FROZEN: an internal node that is small enough for a tile (<= tile_max particles) paired
// with one that is not: only the other side is descended into, this one is carried whole (its
// own box and range come from the node arrays), so that the tiles at the end are ~6 x 5 instead
// of ~6 x 3 particles and a quarter fewer (1M objects: 8.4e6 tiles, 1.16e7 node pairs).
enum : std::uint32_t { DT_INTERNAL = 0, DT_LEAF1 = 1, DT_LEAFN = 2, DT_FROZEN = 3 };
CSC_DEV bool dt_is_leaf(std::uint32_t kind)
{
    return kind == DT_LEAF1 || kind == DT_LEAFN;
}

CSC_DEV unsigned long long dt_item(std::uint32_t a, std::uint32_t ak, std::uint32_t b, std::uint32_t bk, std::uint32_t c)
{
    return static_cast<unsigned long long>(a) | (static_cast<unsigned long long>(ak) << 26)
           | (static_cast<unsigned long long>(b) << 28) | (static_cast<unsigned long long>(bk) << 54)
           | (static_cast<unsigned long long>(c) << 56);
}
CSC_DEV unsigned long long dt_pair(std::uint32_t pa, std::uint32_t pb, std::uint32_t c)
{
    return static_cast<unsigned long long>(pa) | (static_cast<unsigned long long>(pb) << 28)
           | (static_cast<unsigned long long>(c) << 56);
}
// A tile: the particle ranges [xb, xb + xn) x [yb, yb + yn) of one chunk, 1 <= xn, yn <= 16,
// positions below 2^24; xb == yb: a range against itself.
constexpr std::uint32_t DT_TILE_LIMIT = 16;
constexpr std::uint32_t DT_TILE_MAX_N = 1u << 24;
CSC_DEV unsigned long long dt_tile(std::uint32_t xb, std::uint32_t xn, std::uint32_t yb, std::uint32_t yn, std::uint32_t c)
{
    return static_cast<unsigned long long>(xb) | (static_cast<unsigned long long>(yb) << 24)
           | (static_cast<unsigned long long>(xn - 1u) << 48) | (static_cast<unsigned long long>(yn - 1u) << 52)
           | (static_cast<unsigned long long>(c) << 56);
}
CSC_DEV std::uint32_t dt_hash(unsigned long long v)
{
    v ^= v >> 33;
    v *= 0xff51afd7ed558ccdull;
    v ^= v >> 33;
    v *= 0xc4ceb9fe1a85ec53ull;
    v ^= v >> 33;
    return static_cast<std::uint32_t>(v);
}

// One side of a node pair: a child of an internal node, or a leaf carried over from the level above.
struct dt_side {
    float4 lb, ub;
    std::uint32_t idx;   // canonical node index
    bool leaf;
    [[nodiscard]] CSC_DEV std::uint32_t kind() const
    {
        return !leaf ? DT_INTERNAL : (end - begin == 1u ? DT_LEAF1 : DT_LEAFN);
    }
    // What a frontier item stores for this side.
    [[nodiscard]] CSC_DEV std::uint32_t value() const
    {
        return (leaf && end - begin == 1u) ? begin : idx;
    }
    std::uint32_t begin, end; // particle range (leaves)
    std::uint32_t orig;       // original index of the first particle (leaves)
    std::uint32_t act;        // activity flags of the first particle (leaves)
};

CSC_DEV dt_side dt_child(const float4 &lb, const float4 &ub, const int4 &c)
{
    dt_side s;
    s.lb = lb;
    s.ub = ub;
    s.leaf = c.x < 0;
    const std::uint32_t code = static_cast<std::uint32_t>(-1 - c.x);
    s.idx = s.leaf ? (code >> 2) : static_cast<std::uint32_t>(c.x);
    s.act = code & 3u;
    s.begin = static_cast<std::uint32_t>(c.y);
    s.end = static_cast<std::uint32_t>(c.z);
    s.orig = static_cast<std::uint32_t>(c.w);
    return s;
}

CSC_DEV std::uint32_t dt_wave_incl_scan(std::uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const std::uint32_t t = __shfl_up(v, d);
        if (lane >= d) {
            v += t;
        }
    }
    return v;
}

// Seeds the frontier of level 0 with (root, root) of every chunk (shard 0).
__global__ void k_bp_dual_init(std::uint32_t nchunks, std::uint32_t nn_max, const std::int32_t *nleft, unsigned long long *out,
                               std::uint32_t *out_cnt)
{
    const std::uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchunks) {
        // A root that is a leaf: one particle (nothing to do, but harmless) or equal codes.
        const std::uint32_t kd = nleft[static_cast<std::size_t>(c) * nn_max] < 0 ? DT_LEAFN : DT_INTERNAL;
        out[c] = dt_item(0u, kd, 0u, kd, c);
    }
    if (c == 0u) {
        out_cnt[0] = nchunks;
    }
}

// A child of one side of an item (or the carried leaf itself): box, what an item stores for it
// (value, kind), original index and activity flags of its first particle. Plain scalars: the four
// kids of an item must stay in registers (no arrays, no addresses taken).
struct dt_kid {
    float4 lb, ub;
    std::uint32_t val, kind, orig, act;
    std::uint32_t beg, cnt; // particle range
};

CSC_DEV dt_kid dt_kid_from_desc(const float4 &lb, const float4 &ub, const int4 &c)
{
    dt_kid s;
    s.lb = lb;
    s.ub = ub;
    const bool leaf = c.x < 0;
    const std::uint32_t code = static_cast<std::uint32_t>(-1 - c.x);
    const std::uint32_t begin = static_cast<std::uint32_t>(c.y), end = static_cast<std::uint32_t>(c.z);
    s.kind = !leaf ? DT_INTERNAL : (end - begin == 1u ? DT_LEAF1 : DT_LEAFN);
    s.val = !leaf ? static_cast<std::uint32_t>(c.x) : (end - begin == 1u ? begin : (code >> 2));
    s.act = code & 3u;
    s.orig = static_cast<std::uint32_t>(c.w);
    s.beg = begin;
    s.cnt = end - begin;
    return s;
}

// Every lane runs the SAME code whatever the kinds of its item (internal x internal, carried leaf
// x internal, self pair): only the loads of a side differ. Up to four child pairs (i, j), slot
// 2 i + j; outputs are compacted slot by slot with wave ballots (position = outputs of the
// earlier slots of the workgroup + lanes before me in this slot), one atomic per workgroup tile
// and output kind.
__global__ __launch_bounds__(DT_BLOCK) void k_bp_dual(const dual_params q)
{
    constexpr int NW = DT_BLOCK / 64;
    __shared__ std::uint32_t s_cnt[2][NW][4];
    __shared__ std::uint32_t s_base[2];
    const std::uint32_t shard = blockIdx.x % DT_SHARDS;
    const std::uint32_t tile_stride = gridDim.x / DT_SHARDS;
    const std::uint32_t cnt_raw = q.in_cnt[shard * DT_CTR_STRIDE];
    const std::uint32_t cnt = cnt_raw < q.fr_cap ? cnt_raw : q.fr_cap;
    const unsigned long long *in = q.in + static_cast<std::size_t>(shard) * q.fr_cap;
    const int wave = threadIdx.x >> 6;
    const bool deal_pairs = q.shard_n > 1u && q.level < q.shard_level;
    const bool deal_items = q.shard_n > 1u && q.level + 1u == q.shard_level;

    for (std::uint32_t tile = blockIdx.x / DT_SHARDS; static_cast<std::uint64_t>(tile) * DT_BLOCK < cnt; tile += tile_stride) {
        const std::uint32_t k = tile * DT_BLOCK + threadIdx.x;
        // The outputs of successive tiles go to successive shards: a frontier that starts in one
        // shard (the roots) is spread over all of them within a few levels.
        const std::uint32_t oshard = (shard + tile) % DT_SHARDS;
        unsigned long long *out = q.out + static_cast<std::size_t>(oshard) * q.fr_cap;
        unsigned long long *pout = q.pairs + static_cast<std::size_t>(oshard) * q.pair_cap;
        std::uint32_t *out_cnt = q.out_cnt + oshard * DT_CTR_STRIDE;
        std::uint32_t *pair_cnt = q.pair_cnt + oshard * DT_CTR_STRIDE;
        // Second output channel: the pair list, or the tile list when tiles are on (then the only
        // pairs this kernel writes itself are those of oversized multi-particle leaves).
        unsigned long long *out2 = q.tile_max ? q.tiles + static_cast<std::size_t>(oshard) * q.tile_cap : pout;
        std::uint32_t *out2_cnt = q.tile_max ? q.tile_cnt + oshard * DT_CTR_STRIDE : pair_cnt;
        const std::uint32_t out2_cap = q.tile_max ? q.tile_cap : q.pair_cap;
        // One output per (statically numbered) child pair: kind 1 = frontier item, 2 = pair / tile.
        unsigned long long rec_out[4] = {0, 0, 0, 0};
        std::uint32_t kind[4] = {0, 0, 0, 0};
        if (k < cnt) {
            const unsigned long long it = in[k];
            const std::uint32_t a = static_cast<std::uint32_t>(it) & (DT_MAX_NODES - 1u);
            const std::uint32_t ak = static_cast<std::uint32_t>(it >> 26) & 3u;
            const std::uint32_t b = static_cast<std::uint32_t>(it >> 28) & (DT_MAX_NODES - 1u);
            const std::uint32_t bk = static_cast<std::uint32_t>(it >> 54) & 3u;
            const std::uint32_t c = static_cast<std::uint32_t>(it >> 56);
            const std::size_t no = static_cast<std::size_t>(c) * q.nn_max, po = static_cast<std::size_t>(c) * q.n;
            const tnode *tn = q.tn + no;
            const bool self = a == b && ak == bk;

            // Particle range of a leaf child (multi-particle leaves: rare, equal Morton codes).
            auto range_of = [&](std::uint32_t v, std::uint32_t kd, std::uint32_t &rb, std::uint32_t &re) {
                if (kd == DT_LEAF1) {
                    rb = v;
                    re = v + 1u;
                } else {
                    rb = q.nb[no + v];
                    re = q.ne[no + v];
                }
            };
            // Exact leaf-leaf test of the reference for every member pair (multi-particle leaves:
            // the particle with the smaller original index queries the LEAF box of the other,
            // src/sim_broad_phase.cpp:161-185). Rare: appended one by one.
            auto slow_pairs = [&](const dt_kid x, const dt_kid y, bool same) {
                std::uint32_t xb, xe, yb, ye;
                range_of(x.val, x.kind, xb, xe);
                range_of(y.val, y.kind, yb, ye);
                for (std::uint32_t p = xb; p != xe; ++p) {
                    for (std::uint32_t r = same ? p + 1u : yb; r != ye; ++r) {
                        const std::uint32_t op = q.vidx[po + p], orr = q.vidx[po + r];
                        if ((q.active[op] | q.active[orr]) == 0) {
                            continue;
                        }
                        const bool p_first = op < orr;
                        const std::uint32_t qa = p_first ? p : r, qb = p_first ? r : p;
                        const float4 llb = p_first ? y.lb : x.lb, lub = p_first ? y.ub : x.ub;
                        if (!same && !bp_overlap(q.slb[po + qa], q.sub[po + qa], llb, lub)) {
                            continue;
                        }
                        const unsigned long long rec = dt_pair(qa, qb, c);
                        if (deal_pairs && dt_hash(rec) % q.shard_n != q.shard_rank) {
                            continue;
                        }
                        const std::uint32_t slot = atomicAdd(pair_cnt, 1u);
                        if (slot < q.pair_cap) {
                            pout[slot] = rec;
                        }
                    }
                }
            };
            // The two kids of a side: x1 valid only for internal nodes.
            auto load_side = [&](std::uint32_t v, std::uint32_t kd, dt_kid &x0, dt_kid &x1) {
                if (kd == DT_INTERNAL) {
                    const tnode t = tn[v];
                    x0 = dt_kid_from_desc(t.lb0, t.ub0, t.c0);
                    x1 = dt_kid_from_desc(t.lb1, t.ub1, t.c1);
                } else if (kd == DT_FROZEN) {
                    x0.kind = DT_FROZEN;
                    x0.val = v;
                    x0.lb = q.nlb[no + v];
                    x0.ub = q.nub[no + v];
                    x0.beg = q.nb[no + v];
                    x0.cnt = q.ne[no + v] - x0.beg;
                    x0.orig = 0u;
                    x0.act = 3u;
                    x1 = x0;
                } else {
                    x0.kind = kd;
                    x0.val = v;
                    if (kd == DT_LEAF1) {
                        // v = sorted position: the leaf box is the particle's box; independent loads.
                        x0.lb = q.slb[po + v];
                        x0.ub = q.sub[po + v];
                        x0.orig = q.vidx[po + v];
                        x0.act = q.sact[po + v];
                        x0.beg = v;
                        x0.cnt = 1u;
                    } else {
                        x0.lb = q.nlb[no + v];
                        x0.ub = q.nub[no + v];
                        x0.beg = q.nb[no + v];
                        x0.cnt = q.ne[no + v] - x0.beg;
                        x0.orig = q.vidx[po + x0.beg];
                        x0.act = q.active[x0.orig];
                    }
                    x1 = x0;
                }
            };
            // One child pair; slot numbers are compile-time constants at every call.
            auto pair_of = [&](const dt_kid &x, const dt_kid &y, bool go, bool self_kid, const int slot) {
                go = go && (self_kid || bp_overlap(x.lb, x.ub, y.lb, y.ub));
                if (!go) {
                    return;
                }
                if (self_kid) {
                    // (a single-particle leaf has no pair with itself)
                    if (x.kind != DT_LEAF1) {
                        if (x.cnt <= q.tile_max) {
                            const unsigned long long rec = dt_tile(x.beg, x.cnt, x.beg, x.cnt, c);
                            if (!deal_pairs || dt_hash(rec) % q.shard_n == q.shard_rank) {
                                rec_out[slot] = rec;
                                kind[slot] = 2u;
                            }
                        } else {
                            const unsigned long long rec = dt_item(x.val, x.kind, x.val, x.kind, c);
                            if (!deal_items || dt_hash(rec) % q.shard_n == q.shard_rank) {
                                rec_out[slot] = rec;
                                kind[slot] = 1u;
                            }
                        }
                    }
                } else if (x.cnt <= q.tile_max && y.cnt <= q.tile_max) {
                    // (tile_max != 0 here; covers leaf x leaf as a 1 x 1 tile)
                    const unsigned long long rec = dt_tile(x.beg, x.cnt, y.beg, y.cnt, c);
                    if (!deal_pairs || dt_hash(rec) % q.shard_n == q.shard_rank) {
                        rec_out[slot] = rec;
                        kind[slot] = 2u;
                    }
                } else if (x.kind == DT_LEAF1 && y.kind == DT_LEAF1) {
                    if ((x.act | y.act) != 0u) {
                        const bool x_first = x.orig < y.orig;
                        const unsigned long long rec = dt_pair(x_first ? x.val : y.val, x_first ? y.val : x.val, c);
                        if (!deal_pairs || dt_hash(rec) % q.shard_n == q.shard_rank) {
                            rec_out[slot] = rec;
                            kind[slot] = 2u;
                        }
                    }
                } else if (dt_is_leaf(x.kind) && dt_is_leaf(y.kind)) {
                    slow_pairs(x, y, false);
                } else {
                    // One side small, the other an internal node: the small one stops descending.
                    // (Against an oversized multi-particle leaf both keep their plain kinds: the
                    // one-by-one path wants leaves on both sides.)
                    const bool xi = !dt_is_leaf(x.kind), yi = !dt_is_leaf(y.kind);
                    const std::uint32_t xk = xi ? ((yi && x.cnt <= q.tile_max) ? DT_FROZEN : DT_INTERNAL) : x.kind;
                    const std::uint32_t yk = yi ? ((xi && y.cnt <= q.tile_max) ? DT_FROZEN : DT_INTERNAL) : y.kind;
                    const unsigned long long rec = dt_item(x.val, xk, y.val, yk, c);
                    if (!deal_items || dt_hash(rec) % q.shard_n == q.shard_rank) {
                        rec_out[slot] = rec;
                        kind[slot] = 1u;
                    }
                }
            };

            dt_kid A0, A1, B0, B1;
            load_side(a, ak, A0, A1);
            if (self && ak != DT_INTERNAL) {
                // A multi-particle leaf against itself.
                slow_pairs(A0, A0, true);
            } else {
                if (self) {
                    B0 = A0;
                    B1 = A1;
                } else {
                    load_side(b, bk, B0, B1);
                }
                const bool a2 = ak == DT_INTERNAL, b2 = bk == DT_INTERNAL;
                pair_of(A0, B0, true, self, 0);
                pair_of(A0, B1, b2, false, 1);
                pair_of(A1, B0, a2 && !self, false, 2); // (self items: (A1, A0) is (A0, A1) again)
                pair_of(A1, B1, a2 && b2, self, 3);
            }
        }

        // Compaction in lane order (the children of one item stay next to each other: the lanes of
        // the next level then share node records). A lane has 0..4 outputs of a kind: its offset
        // within the wave is the sum over the three bits of that count of (1 << bit) x the number
        // of earlier lanes with the bit set -- three ballots per kind, no shuffles.
        std::uint32_t nf = 0, np = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            nf += kind[j] == 1u;
            np += kind[j] == 2u;
        }
        auto lane_prefix = [](std::uint32_t v, std::uint32_t &wave_total) {
            std::uint32_t pre = 0;
            wave_total = 0;
#pragma unroll
            for (int bit = 0; bit < 3; ++bit) {
                const unsigned long long m = __ballot(((v >> bit) & 1u) != 0u);
                pre += __builtin_amdgcn_mbcnt_hi(static_cast<std::uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<std::uint32_t>(m), 0u)) << bit;
                wave_total += static_cast<std::uint32_t>(__popcll(m)) << bit;
            }
            return pre;
        };
        std::uint32_t wt_f, wt_p;
        std::uint32_t off_f = lane_prefix(nf, wt_f), off_p = lane_prefix(np, wt_p);
        if ((threadIdx.x & 63) == 0) {
            s_cnt[0][wave][0] = wt_f;
            s_cnt[1][wave][0] = wt_p;
        }
        __syncthreads();
        std::uint32_t tot_f = 0, tot_p = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const std::uint32_t wf = s_cnt[0][w][0], wp = s_cnt[1][w][0];
            if (w < wave) {
                off_f += wf;
                off_p += wp;
            }
            tot_f += wf;
            tot_p += wp;
        }
        if (threadIdx.x == 0) {
            s_base[0] = tot_f ? atomicAdd(out_cnt, tot_f) : 0u;
            s_base[1] = tot_p ? atomicAdd(out2_cnt, tot_p) : 0u;
        }
        __syncthreads();
        std::uint32_t pos_f = s_base[0] + off_f, pos_p = s_base[1] + off_p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kind[j] == 1u) {
                if (pos_f < q.fr_cap) {
                    out[pos_f] = rec_out[j];
                }
                ++pos_f;
            } else if (kind[j] == 2u) {
                if (pos_p < out2_cap) {
                    out2[pos_p] = rec_out[j];
                }
                ++pos_p;
            }
        }
        __syncthreads();
    }
}

// The tiles of the traversal above: every particle pair of a tile gets the reference's own leaf
// test directly -- the particle with the smaller original index queries the box of the other
// one's LEAF (src/sim_broad_phase.cpp:149-185; the particle's own box unless it shares its Morton
// code with others) -- without descending the last levels of the pair tree, where nine tenths of
// its node pairs are (1M-object LEO population, tiles of <= 16 x 16: 6.2e6 node pairs + 3.8e8
// box tests instead of 6.0e7 node pairs, each of which costs two 96-byte gathers, four box tests
// and a compaction). TL lanes per tile, lane i holds particle i of the x range and loops over the
// y range; the hits of a lane are kept as a bit mask and appended, tile by tile, to a buffer in
// LDS that is flushed to the pair list with one atomic per ~1000 pairs: the pairs of a tile stay
// together, so the narrow phase's pre-filter finds both trajectory records of consecutive pairs
// in the same few cache lines.
constexpr int DT_TILE_WAVES = 4;

// What a lane holds of its tile: the tile, particle i of the longer range ("x") and particle i of
// the shorter one ("y"; the test is symmetric: the original indices decide who queries whom).
struct dt_tile_in {
    std::uint32_t xb, yb, xn, yn, c; // xn >= yn; xn = 0: no tile
    float4 lbx, ubx, lby, uby;
    std::uint32_t ix, iy;
};

CSC_DEV void dt_tile_decode(unsigned long long rec, bool valid, dt_tile_in &t)
{
    std::uint32_t xb = static_cast<std::uint32_t>(rec) & (DT_TILE_MAX_N - 1u);
    std::uint32_t yb = static_cast<std::uint32_t>(rec >> 24) & (DT_TILE_MAX_N - 1u);
    std::uint32_t xn = (static_cast<std::uint32_t>(rec >> 48) & 15u) + 1u;
    std::uint32_t yn = (static_cast<std::uint32_t>(rec >> 52) & 15u) + 1u;
    if (yn > xn) {
        const std::uint32_t tb = xb, tn = xn;
        xb = yb;
        xn = yn;
        yb = tb;
        yn = tn;
    }
    t.xb = xb;
    t.yb = yb;
    t.xn = valid ? xn : 0u;
    t.yn = valid ? yn : 0u;
    t.c = valid ? static_cast<std::uint32_t>(rec >> 56) : 0u;
}

// (four waves per SIMD: 128 registers. The kernel waits on dependent instructions more than on
// anything else -- see DESIGN.md 3.4 -- and the fused variant with two groups of records in flight
// per trip of its filter needed 152: 6.0 ms per launch of the bench against 5.4 with one group.)
template <int TL, bool FUSE = false>
__global__ __launch_bounds__(64 * DT_TILE_WAVES, 4) void k_bp_tiles(const dual_params q)
{
    constexpr int TPW = 64 / TL;
    constexpr std::uint32_t BUF = 64u * TL; // what one batch can yield at most
    // Staging area of the y particles: the TL slots of a tile, then one slot of padding, so that
    // the TPW broadcast reads of one instruction (slot j of every tile) fall on distinct banks
    // (unpadded, the tiles are 128 or 256 bytes apart: the same four banks, a TPW-way conflict;
    // measured: 41 % of the LDS cycles).
    constexpr int YS = 64 + TPW;
    __shared__ float4 s_lb[DT_TILE_WAVES][YS], s_ub[DT_TILE_WAVES][YS];
    __shared__ std::uint32_t s_info[DT_TILE_WAVES][YS];
    __shared__ unsigned long long s_buf[DT_TILE_WAVES][BUF];
    __shared__ std::uint32_t s_fill[DT_TILE_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const std::uint32_t xi = lane % TL, gbase = lane - xi;
    const std::uint32_t ybase = gbase + lane / TL, yslot = ybase + xi; // (padded staging slots)
    const std::uint32_t shard = blockIdx.x % DT_SHARDS;
    const std::uint32_t cnt_raw = q.tile_cnt[shard * DT_CTR_STRIDE];
    const std::uint32_t cnt_all = cnt_raw < q.tile_cap ? cnt_raw : q.tile_cap;
    const std::uint32_t tbegin = q.tile_begin[shard] < cnt_all ? q.tile_begin[shard] : cnt_all;
    const std::uint32_t cnt = cnt_all - tbegin; // tiles [tbegin, cnt_all) of the shard's list
    const unsigned long long *tiles = q.tiles + static_cast<std::size_t>(shard) * q.tile_cap + tbegin;
    unsigned long long *pout = q.pairs + static_cast<std::size_t>(shard) * q.pair_cap;
    std::uint32_t *pair_cnt = q.pair_cnt + shard * DT_CTR_STRIDE;
    unsigned long long *buf = s_buf[wave];
    float4 *ylb = s_lb[wave], *yub = s_ub[wave];
    std::uint32_t *yinfo = s_info[wave];
    std::uint32_t *fillp = &s_fill[wave];

    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (lane == 0) {
        *fillp = 0u;
    }
    wave_sync();
    // The wave's pair buffer -> the pair list of the shard: one atomic, coalesced stores.
    auto flush = [&](std::uint32_t fill) {
        if constexpr (FUSE) {
            // Count the pairs, test them (64 at a time, every lane busy: the records of a tile's
            // particles are neighbours in the sorted array, so the two 64-byte gathers per pair hit
            // the lines that the other pairs of the tile fetch), append the survivors.
            if (lane == 0) {
                atomicAdd(q.fuse_cnt + shard * DT_CTR_STRIDE, fill);
                *fillp = 0u;
            }
            // (FG groups of 64 pairs per trip, their record gathers in flight together: two cost a wave
            // per SIMD in registers, which costs more than the gathers' latency -- see above.)
            // The survivors are compacted to the front of the buffer -- never ahead of the read
            // position -- and leave with ONE atomic per flush: with conjunction tracking a twentieth
            // of the pairs survives, and one atomic per group of 64 on the list's counter is 5e6
            // same-address atomics per superstep of 10M objects, ~10 ns each, serialised.
            std::uint32_t nsurv = 0;
            constexpr int FG = 1;
            for (std::uint32_t k0 = 0; k0 < fill; k0 += 64u * FG) {
                bool keep[FG];
                unsigned long long rec[FG];
                float4 ra[FG][4], rb[FG][4];
#pragma unroll
                for (int g = 0; g < FG; ++g) {
                    const std::uint32_t k = k0 + 64u * g + lane;
                    keep[g] = k < fill;
                    rec[g] = buf[keep[g] ? k : k0];
                    const std::uint32_t pa = static_cast<std::uint32_t>(rec[g]) & 0xfffffffu;
                    const std::uint32_t pb = static_cast<std::uint32_t>(rec[g] >> 28) & 0xfffffffu;
                    const std::size_t co = static_cast<std::size_t>(rec[g] >> 56) * q.n;
                    const float4 *qa = q.srecs + (co + pa) * 4u;
                    const float4 *qb = q.srecs + (co + pb) * 4u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ra[g][e] = qa[e];
                        rb[g][e] = qb[e];
                    }
                }
                wave_sync(); // (every lane has read its two records of the buffer)
#pragma unroll
                for (int g = 0; g < FG; ++g) {
                    keep[g] = keep[g] && chord_keep_recs(ra[g], rb[g], q.thr_extra);
                    const unsigned long long km = __ballot(keep[g]);
                    if (keep[g]) {
                        buf[nsurv + __builtin_amdgcn_mbcnt_hi(static_cast<std::uint32_t>(km >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<std::uint32_t>(km), 0u))] = rec[g];
                    }
                    nsurv += static_cast<std::uint32_t>(__popcll(km));
                }
            }
            wave_sync();
            if (nsurv != 0u) {
                std::uint32_t base = 0;
                if (lane == 0) {
                    base = atomicAdd(q.cand_cnt, nsurv);
                }
                base = __shfl(base, 0);
                for (std::uint32_t k = lane; k < nsurv; k += 64u) {
                    if (base + k < q.cand_cap) {
                        const unsigned long long rec = buf[k];
                        const std::uint32_t pa = static_cast<std::uint32_t>(rec) & 0xfffffffu;
                        const std::uint32_t pb = static_cast<std::uint32_t>(rec >> 28) & 0xfffffffu;
                        const std::uint32_t c = static_cast<std::uint32_t>(rec >> 56);
                        const std::size_t co = static_cast<std::size_t>(c) * q.n;
                        const std::uint32_t oa = q.vidx[co + pa], ob = q.vidx[co + pb];
                        q.cand[base + k] = make_uint4(oa < ob ? oa : ob, oa < ob ? ob : oa, c, 0u);
                    }
                }
            }
            wave_sync();
            return;
        }
        std::uint32_t base = 0;
        if (lane == 0) {
            base = atomicAdd(pair_cnt, fill);
            *fillp = 0u;
        }
        base = __shfl(base, 0);
        for (std::uint32_t k = lane; k < fill; k += 64u) {
            if (base + k < q.pair_cap) {
                pout[base + k] = buf[k];
            }
        }
        wave_sync();
    };
    // Batches of TPW tiles: a contiguous run of the shard's list per workgroup (consecutive tiles
    // come from neighbouring node pairs: their boxes share cache lines), its waves interleaved.
    const std::uint64_t step = DT_TILE_WAVES;
    const std::uint64_t nb_all = (static_cast<std::uint64_t>(cnt) + TPW - 1u) / TPW;
    const std::uint64_t per_wg = (nb_all + gridDim.x / DT_SHARDS - 1u) / (gridDim.x / DT_SHARDS);
    const std::uint64_t b0 = static_cast<std::uint64_t>(blockIdx.x / DT_SHARDS) * per_wg;
    const std::uint64_t nbatch = b0 + per_wg < nb_all ? b0 + per_wg : nb_all;
    std::uint64_t batch = b0 + wave;
    if (batch >= nbatch) {
        return; // (nothing for this wave; in particular no tiles at all)
    }
    // (Every load below is unconditional, on a clamped index: loads under a branch would make
    // the compiler wait for ALL outstanding loads, the prefetched ones included, at the first use.)
    auto rec_of = [&](std::uint64_t b, bool &valid) {
        const std::uint64_t t = b * TPW + lane / TL;
        valid = t < cnt; // (cnt != 0 here)
        return tiles[valid ? t : cnt - 1u];
    };
    auto load_boxes = [&](dt_tile_in &t) {
        const std::size_t po = static_cast<std::size_t>(t.c) * q.n;
        const bool xon = xi < t.xn, yon = xi < t.yn;
        const std::size_t px = po + t.xb + (xon ? xi : 0u), py = po + t.yb + (yon ? xi : 0u);
        t.lbx = q.slb[px];
        t.ubx = q.sub[px];
        t.ix = q.sinfo[px];
        t.lby = q.slb[py];
        t.uby = q.sub[py];
        t.iy = q.sinfo[py];
    };
    // One batch: stage the y particles, test, append the hits.
    auto process = [&](const dt_tile_in &t, auto multi_c) {
        constexpr bool MULTI = decltype(multi_c)::value;
        const bool on = xi < t.xn;
        const bool self = t.xb == t.yb;
        const std::uint32_t tix = on ? t.ix : 0u, tiy = xi < t.yn ? t.iy : 0u;
        ylb[yslot] = t.lby;
        yub[yslot] = t.uby;
        // The y particles a lane has to look at: [jlo, jhi); of those, the ones that make an
        // active pair with its x particle (activity flags of either, src/sim_broad_phase.cpp:175-185).
        const std::uint32_t jlo = self ? xi + 1u : 0u, jhi = on ? t.yn : 0u;
        std::uint32_t want = (jhi > jlo) ? (((1u << jhi) - 1u) & ~((1u << jlo) - 1u)) : 0u;
        const unsigned long long yact = __ballot((tiy & (3u << 24)) != 0u);
        if ((tix & (3u << 24)) == 0u) {
            want &= static_cast<std::uint32_t>(yact >> gbase);
        }
        // A member of a multi-particle leaf anywhere in this batch: the generic test (leaf boxes).
        bool anym = false;
        if constexpr (MULTI) {
            anym = __any(((tix | tiy) & SINFO_MULTI) != 0u);
            if (anym) {
                yinfo[yslot] = tiy;
            }
        }
        wave_sync();
        std::uint32_t hits = 0;
        if (!anym) {
            // Every lane runs every test (the staging area has 64 valid slots); the comparisons
            // are wave ballots, their conjunction is scalar work.
            // (the y boxes of step j + 1 are read from LDS while step j is tested: the slot after a
            // tile's last one exists -- the padding, or the next tile's first)
            float4 nl = ylb[ybase], nu = yub[ybase];
            for (std::uint32_t j = 0; j < static_cast<std::uint32_t>(TL); ++j) {
                if (!__any(j < jhi)) {
                    break;
                }
                const float4 cl = nl, cu = nu;
                nl = ylb[ybase + j + 1u];
                nu = yub[ybase + j + 1u];
                const unsigned long long m = bp_overlap_mask(t.lbx, t.ubx, cl, cu);
                hits |= __builtin_amdgcn_inverse_ballot_w64(m) ? (1u << j) : 0u; // (one v_cndmask on the scalar mask)
            }
            hits &= want;
        } else if constexpr (MULTI) {
            const std::size_t po = static_cast<std::size_t>(t.c) * q.n, no = static_cast<std::size_t>(t.c) * q.nn_max;
            float4 llx = t.lbx, lux = t.ubx;
            if (on && (tix & SINFO_MULTI)) {
                const std::uint32_t lf = q.sleaf[po + t.xb + xi];
                llx = q.nlb[no + lf];
                lux = q.nub[no + lf];
            }
            const std::uint32_t ox = tix & 0xffffffu;
#pragma unroll 1
            for (std::uint32_t j = 0; j < static_cast<std::uint32_t>(TL); ++j) {
                if ((want >> j) & 1u) {
                    const float4 lby = ylb[ybase + j], uby = yub[ybase + j];
                    const std::uint32_t iy = yinfo[ybase + j];
                    float4 lly = lby, luy = uby;
                    if (iy & SINFO_MULTI) {
                        const std::uint32_t lf = q.sleaf[po + t.yb + j];
                        lly = q.nlb[no + lf];
                        luy = q.nub[no + lf];
                    }
                    const bool hit = ox < (iy & 0xffffffu) ? bp_overlap(t.lbx, t.ubx, lly, luy) : bp_overlap(lby, uby, llx, lux);
                    hits |= static_cast<std::uint32_t>(hit) << j;
                }
            }
        }
        // Append to the wave's buffer (positions from an LDS counter; pairs as (x, y) positions:
        // the readers of the list order them by original index).
        const unsigned long long any_hit = __ballot(hits != 0u);
        if (any_hit != 0ull) {
            const std::uint32_t fill = *fillp;
            if (fill + static_cast<std::uint32_t>(__popcll(any_hit)) * TL > BUF) {
                flush(fill);
            }
            if (hits != 0u) {
                std::uint32_t pos = atomicAdd(fillp, static_cast<std::uint32_t>(__popc(hits)));
                const unsigned long long rec0 = dt_pair(t.xb + xi, t.yb, t.c);
                std::uint32_t m = hits;
                while (m != 0u) {
                    const std::uint32_t j = static_cast<std::uint32_t>(__ffs(m)) - 1u;
                    m &= m - 1u;
                    buf[pos++] = rec0 + (static_cast<unsigned long long>(j) << 28);
                }
            }
        }
        wave_sync(); // (the staging area is rewritten by the next batch)
    };
    // The boxes of the batch about to be tested arrived long ago (they were requested a batch
    // earlier), but the compiler's wait-count bookkeeping loses track of that across the loop's
    // branches and put s_waitcnt vmcnt(0) in front of the test loop -- AFTER the next batch's
    // loads had been issued, so that every batch waited for its successor's boxes and the
    // prefetch hid nothing. An empty asm statement that "rewrites" the registers of the current
    // batch BEFORE those loads are issued moves the wait to where it costs nothing; `next` (an
    // address operand of the loads to come) keeps the scheduler from hoisting them above it.
    auto settle = [](dt_tile_in &t, std::uint32_t &next) {
        asm volatile(""
                     : "+v"(t.lbx.x), "+v"(t.lbx.y), "+v"(t.lbx.z), "+v"(t.lbx.w), "+v"(t.ubx.x), "+v"(t.ubx.y), "+v"(t.ubx.z),
                       "+v"(t.ubx.w), "+v"(t.lby.x), "+v"(t.lby.y), "+v"(t.lby.z), "+v"(t.lby.w), "+v"(t.uby.x), "+v"(t.uby.y),
                       "+v"(t.uby.z), "+v"(t.uby.w), "+v"(t.ix), "+v"(t.iy), "+v"(next));
    };
    // Software pipeline, unrolled by two so that no register moves are needed: the tile of batch
    // k + 2 and the boxes of batch k + 1 are in flight while batch k is tested.
    // Two copies of the loop: the one that runs when no leaf of this build holds more than one
    // particle (k_bvh_pack's flag; the normal case) has no generic test in it -- whose loads,
    // under branches, also cost the fast path its wait-count precision.
    auto run = [&](auto multi_c) {
        dt_tile_in ta{}, tb{};
        bool va, vb;
        unsigned long long ra = rec_of(batch, va);
        dt_tile_decode(ra, va, ta);
        load_boxes(ta);
        unsigned long long rb = rec_of(batch + step, vb);
        while (batch < nbatch) {
            dt_tile_decode(rb, vb, tb);
            settle(ta, tb.xb);
            load_boxes(tb);
            ra = rec_of(batch + 2u * step, va);
            process(ta, multi_c);
            batch += step;
            if (batch >= nbatch) {
                break;
            }
            dt_tile_decode(ra, va, ta);
            settle(tb, ta.xb);
            load_boxes(ta);
            rb = rec_of(batch + 2u * step, vb);
            process(tb, multi_c);
            batch += step;
        }
    };
    if (*q.multi_flag != 0u) {
        run(std::true_type{});
    } else {
        run(std::false_type{});
    }
    wave_sync();
    const std::uint32_t fill = *fillp;
    if (fill != 0u) {
        flush(fill);
    }
}

} // namespace csc
