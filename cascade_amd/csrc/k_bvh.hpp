// cascade_amd/csrc/k_bvh.hpp
//
// K5: LBVH over the sorted 64-bit Morton codes with EXACTLY the topology, node order and
// AABBs of the reference's level-synchronous builder, but built in a fixed number of
// data-parallel passes instead of one pass per tree level.
//
// Replaces src/sim_bvh.cpp:89-496 (construct_bvh_trees_parallel). Why the result is the
// same tree: a reference node covers a sorted range whose codes share all bits above its
// split_idx; it splits at the first bit >= split_idx where the range is not constant, i.e.
// at clz(code[first] ^ code[last]), at the first index where that bit is set (:207-240);
// ranges of equal codes are (multi-particle) leaves (:224-231). That is the binary radix
// tree over the DISTINCT codes (Karras 2012), with one leaf per run of equal codes. The
// reference stores nodes level by level, each level ordered by `begin` (children are
// placed by a prefix sum over the parents in order, :311-327,:360-368), so the canonical
// index of a node is its rank under the key (depth, begin). Passes:
//   heads -> scan -> run starts -> Karras internal nodes -> depths (parent walk) ->
//   stable radix sort by (depth, begin) -> emit nodes (+ leaf AABBs) -> bottom-up refit.
// Everything is HBM/L2 bound integer work: ~250 B per particle per chunk.
//
// Node arrays (structure of arrays, canonical order), per chunk with stride 2N - 1:
//   nb/ne (u32 begin/end), npar/nleft (i32; right = left + 1), nlb/nub (float4),
//   ndepth (u32), nsplit (i32: final value of the reference's construction scratch).
// k_bvh_export packs them into the reference's 60-byte bvh_node
// (include/cascade/detail/sim_data.hpp:193-207) for parity checks.
#pragma once

#include "dev_math.hpp"

namespace csc
{

constexpr int BVH_MAX_LEVELS = 66; // depth <= 64

struct bvh_bufs {
    std::uint32_t n;        // particles per chunk
    std::uint32_t nn_max;   // 2n - 1
    const std::uint64_t *codes; // [C][n] sorted
    const float4 *slb, *sub;    // [C][n] sorted AABBs
    // scratch
    std::uint32_t *head;      // [C][n]
    std::uint32_t *excl;      // [C][n + 1] (excl[n] = M)
    std::uint32_t *run_start; // [C][n + 1]
    std::int32_t *k_parent;   // [C][nn_max] Karras ids: internal i, leaf (M - 1) + r
    std::int32_t *k_left, *k_right; // [C][n]
    std::uint32_t *k_first, *k_last; // [C][n] run indices
    std::int32_t *k_delta;    // [C][n]
    std::uint32_t *k_depth;   // [C][nn_max]
    std::uint64_t *skey_a, *skey_b; // [C][nn_max]
    std::uint32_t *sval_a, *sval_b; // [C][nn_max]
    std::uint32_t *new_of;    // [C][nn_max]
    std::uint32_t *level_cnt; // [C][BVH_MAX_LEVELS]
    std::uint32_t *level_off; // [C][BVH_MAX_LEVELS + 1]
    std::uint32_t *visit;     // [C][nn_max] refit arrival counters (zeroed per build)
    // output
    std::uint32_t *nb, *ne;
    std::int32_t *npar, *nleft;
    float4 *nlb, *nub;
    std::uint32_t *ndepth;
    std::int32_t *nsplit;
    std::uint32_t *n_nodes; // [C]
    int begin_bits;         // 24 or 32
};

CSC_GLOBAL __launch_bounds__(256) void k_bvh_heads(bvh_bufs b)
{
    const std::uint32_t c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) {
        return;
    }
    const std::uint64_t *codes = b.codes + static_cast<std::size_t>(c) * b.n;
    b.head[static_cast<std::size_t>(c) * b.n + i] = (i == 0 || codes[i] != codes[i - 1]) ? 1u : 0u;
}

CSC_GLOBAL __launch_bounds__(256) void k_bvh_run_start(bvh_bufs b)
{
    const std::uint32_t c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) {
        return;
    }
    const std::uint32_t *excl = b.excl + static_cast<std::size_t>(c) * (b.n + 1u);
    std::uint32_t *rs = b.run_start + static_cast<std::size_t>(c) * (b.n + 1u);
    if (b.head[static_cast<std::size_t>(c) * b.n + i]) {
        rs[excl[i]] = i;
    }
    if (i == 0) {
        const std::uint32_t M = excl[b.n];
        rs[M] = b.n;
        b.n_nodes[c] = 2u * M - 1u;
    }
}

// delta(i, j) over the distinct keys; -1 out of range.
CSC_DEV int bvh_delta(const std::uint64_t *codes, const std::uint32_t *rs, std::int64_t M, std::uint64_t ki,
                      std::int64_t j)
{
    if (j < 0 || j >= M) {
        return -1;
    }
    const std::uint64_t kj = codes[rs[j]];
    return __clzll(static_cast<long long>(ki ^ kj));
}

CSC_GLOBAL __launch_bounds__(256) void k_bvh_karras(bvh_bufs b)
{
    const std::uint32_t c = blockIdx.y;
    const std::int64_t i = static_cast<std::int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const std::uint32_t *excl = b.excl + static_cast<std::size_t>(c) * (b.n + 1u);
    const std::int64_t M = excl[b.n];
    const std::uint64_t *codes = b.codes + static_cast<std::size_t>(c) * b.n;
    const std::uint32_t *rs = b.run_start + static_cast<std::size_t>(c) * (b.n + 1u);
    std::int32_t *kpar = b.k_parent + static_cast<std::size_t>(c) * b.nn_max;
    if (M == 1) {
        if (i == 0) {
            kpar[0] = -1;
        }
        return;
    }
    if (i >= M - 1) {
        return;
    }
    const std::size_t o = static_cast<std::size_t>(c) * b.n + i;
    const std::uint64_t ki = codes[rs[i]];
    const int dl = bvh_delta(codes, rs, M, ki, i - 1), dr = bvh_delta(codes, rs, M, ki, i + 1);
    const std::int64_t d = (dr - dl) > 0 ? 1 : -1;
    const int dmin = (d > 0) ? dl : dr;
    std::int64_t lmax = 2;
    while (bvh_delta(codes, rs, M, ki, i + lmax * d) > dmin) {
        lmax *= 2;
    }
    std::int64_t l = 0;
    for (std::int64_t t = lmax / 2; t >= 1; t /= 2) {
        if (bvh_delta(codes, rs, M, ki, i + (l + t) * d) > dmin) {
            l += t;
        }
    }
    const std::int64_t j = i + l * d;
    const int dnode = bvh_delta(codes, rs, M, ki, j);
    std::int64_t s = 0;
    {
        std::int64_t t = l;
        do {
            t = (t + 1) / 2;
            if (bvh_delta(codes, rs, M, ki, i + (s + t) * d) > dnode) {
                s += t;
            }
        } while (t > 1);
    }
    const std::int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const std::int64_t first = i < j ? i : j, last = i < j ? j : i;
    const std::int32_t left = (first == gamma) ? static_cast<std::int32_t>((M - 1) + gamma) : static_cast<std::int32_t>(gamma);
    const std::int32_t right
        = (last == gamma + 1) ? static_cast<std::int32_t>((M - 1) + gamma + 1) : static_cast<std::int32_t>(gamma + 1);
    b.k_left[o] = left;
    b.k_right[o] = right;
    b.k_first[o] = static_cast<std::uint32_t>(first);
    b.k_last[o] = static_cast<std::uint32_t>(last);
    b.k_delta[o] = dnode;
    kpar[left] = static_cast<std::int32_t>(i);
    kpar[right] = static_cast<std::int32_t>(i);
    if (i == 0) {
        kpar[0] = -1;
    }
}

// Depth by walking the parent chain; sort key (depth, begin); per-level histogram.
CSC_GLOBAL __launch_bounds__(256) void k_bvh_depth(bvh_bufs b)
{
    __shared__ std::uint32_t hist[BVH_MAX_LEVELS];
    const std::uint32_t c = blockIdx.y;
    const std::uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < BVH_MAX_LEVELS) {
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    const std::uint32_t *excl = b.excl + static_cast<std::size_t>(c) * (b.n + 1u);
    const std::uint32_t M = excl[b.n];
    const std::uint32_t nn = 2u * M - 1u;
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    if (id < b.nn_max) {
        std::uint64_t key = ~0ull;
        if (id < nn) {
            const std::int32_t *kpar = b.k_parent + no;
            std::uint32_t depth = 0;
            std::int32_t p = kpar[id];
            while (p >= 0) {
                ++depth;
                p = kpar[p];
            }
            b.k_depth[no + id] = depth;
            const std::uint32_t *rs = b.run_start + static_cast<std::size_t>(c) * (b.n + 1u);
            std::uint32_t begin;
            if (id < M - 1u) {
                begin = rs[b.k_first[static_cast<std::size_t>(c) * b.n + id]];
            } else {
                begin = rs[id - (M - 1u)];
            }
            key = (static_cast<std::uint64_t>(depth) << b.begin_bits) | begin;
            atomicAdd(&hist[depth < BVH_MAX_LEVELS ? depth : BVH_MAX_LEVELS - 1], 1u);
        }
        b.skey_a[no + id] = key;
    }
    __syncthreads();
    if (threadIdx.x < BVH_MAX_LEVELS && hist[threadIdx.x]) {
        atomicAdd(&b.level_cnt[c * BVH_MAX_LEVELS + threadIdx.x], hist[threadIdx.x]);
    }
}

CSC_GLOBAL void k_bvh_level_off(bvh_bufs b)
{
    const std::uint32_t c = blockIdx.x;
    if (threadIdx.x == 0) {
        std::uint32_t run = 0;
        for (int l = 0; l < BVH_MAX_LEVELS; ++l) {
            b.level_off[c * (BVH_MAX_LEVELS + 1) + l] = run;
            run += b.level_cnt[c * BVH_MAX_LEVELS + l];
        }
        b.level_off[c * (BVH_MAX_LEVELS + 1) + BVH_MAX_LEVELS] = run;
    }
}

// sval: sorted Karras ids.
CSC_GLOBAL __launch_bounds__(256) void k_bvh_newidx(bvh_bufs b, const std::uint32_t *sval)
{
    const std::uint32_t c = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    const std::uint32_t nn = b.n_nodes[c];
    if (p >= nn) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    b.new_of[no + sval[no + p]] = p;
}

CSC_GLOBAL __launch_bounds__(256) void k_bvh_emit(bvh_bufs b, const std::uint32_t *sval)
{
    const std::uint32_t c = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    const std::uint32_t nn = b.n_nodes[c];
    if (p >= nn) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    const std::size_t po = static_cast<std::size_t>(c) * b.n;
    const std::uint32_t M = (nn + 1u) / 2u;
    const std::uint32_t id = sval[no + p];
    const std::uint32_t *rs = b.run_start + static_cast<std::size_t>(c) * (b.n + 1u);
    const std::int32_t kp = b.k_parent[no + id];
    const std::int32_t parent = kp < 0 ? -1 : static_cast<std::int32_t>(b.new_of[no + kp]);
    // Initial split_idx of the node: parent's final split bit + 1 (root: 0).
    const int init_split = kp < 0 ? 0 : b.k_delta[po + kp] + 1;
    const float finf = __builtin_huge_valf();
    std::uint32_t begin, end;
    std::int32_t left;
    int split;
    float4 lb = make_float4(finf, finf, finf, finf), ub = make_float4(-finf, -finf, -finf, -finf);
    if (id < M - 1u) {
        begin = rs[b.k_first[po + id]];
        end = rs[b.k_last[po + id] + 1u];
        left = static_cast<std::int32_t>(b.new_of[no + b.k_left[po + id]]);
        split = b.k_delta[po + id];
    } else {
        const std::uint32_t r = id - (M - 1u);
        begin = rs[r];
        end = rs[r + 1u];
        left = -1;
        // A multi-particle leaf scans the remaining bits up to 63 (src/sim_bvh.cpp:220-240).
        split = (end - begin > 1u && init_split <= 63) ? 63 : init_split;
        const float4 *slb = b.slb + po, *sub = b.sub + po;
        for (std::uint32_t k = begin; k < end; ++k) {
            const float4 l4 = slb[k], u4 = sub[k];
            lb = make_float4(fminf_std(lb.x, l4.x), fminf_std(lb.y, l4.y), fminf_std(lb.z, l4.z), fminf_std(lb.w, l4.w));
            ub = make_float4(fmaxf_std(ub.x, u4.x), fmaxf_std(ub.y, u4.y), fmaxf_std(ub.z, u4.z), fmaxf_std(ub.w, u4.w));
        }
    }
    b.nb[no + p] = begin;
    b.ne[no + p] = end;
    b.npar[no + p] = parent;
    b.nleft[no + p] = left;
    b.nlb[no + p] = lb;
    b.nub[no + p] = ub;
    b.ndepth[no + p] = b.k_depth[no + id];
    b.nsplit[no + p] = split;
}

// Traversal order. The reference's node order (level by level, each level by `begin`) is what
// csc_bvh_export must produce, but nothing on the path from the tree to the pairs depends on
// it: the traversals need the topology, the boxes and siblings next to each other. The depths,
// the radix sort by (depth, begin) and the renumbering are therefore left to the export, and
// the build numbers the nodes straight from the Karras ids: the root is 0 and the children of
// internal Karras node i are 2 i + 1 and 2 i + 2 (a bijection onto [0, 2 M - 1)). Same tree,
// same boxes, same pair set (tests/test_gpu_parity.py runs every stage-parity case both ways).
CSC_GLOBAL __launch_bounds__(256) void k_bvh_emit_fast(bvh_bufs b)
{
    const std::uint32_t c = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    const std::uint32_t nn = b.n_nodes[c];
    if (p >= nn) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    const std::size_t po = static_cast<std::size_t>(c) * b.n;
    const std::uint32_t M = (nn + 1u) / 2u;
    const std::uint32_t *rs = b.run_start + static_cast<std::size_t>(c) * (b.n + 1u);
    // Karras id of node p and of its parent.
    const std::int32_t kp = p == 0u ? -1 : static_cast<std::int32_t>((p - 1u) >> 1);
    const std::uint32_t id = p == 0u ? 0u : static_cast<std::uint32_t>(((p - 1u) & 1u) ? b.k_right[po + kp] : b.k_left[po + kp]);
    std::int32_t parent = -1;
    if (kp > 0) {
        const std::int32_t kpp = b.k_parent[no + kp];
        parent = 2 * kpp + 1 + (b.k_right[po + kpp] == kp ? 1 : 0);
    } else if (kp == 0) {
        parent = 0;
    }
    const int init_split = kp < 0 ? 0 : b.k_delta[po + kp] + 1;
    const float finf = __builtin_huge_valf();
    std::uint32_t begin, end;
    std::int32_t left;
    int split;
    float4 lb = make_float4(finf, finf, finf, finf), ub = make_float4(-finf, -finf, -finf, -finf);
    if (id < M - 1u) {
        begin = rs[b.k_first[po + id]];
        end = rs[b.k_last[po + id] + 1u];
        left = static_cast<std::int32_t>(2u * id + 1u);
        split = b.k_delta[po + id];
    } else {
        const std::uint32_t r = id - (M - 1u);
        begin = rs[r];
        end = rs[r + 1u];
        left = -1;
        split = (end - begin > 1u && init_split <= 63) ? 63 : init_split;
        const float4 *slb = b.slb + po, *sub = b.sub + po;
        for (std::uint32_t k = begin; k < end; ++k) {
            const float4 l4 = slb[k], u4 = sub[k];
            lb = make_float4(fminf_std(lb.x, l4.x), fminf_std(lb.y, l4.y), fminf_std(lb.z, l4.z), fminf_std(lb.w, l4.w));
            ub = make_float4(fmaxf_std(ub.x, u4.x), fmaxf_std(ub.y, u4.y), fmaxf_std(ub.z, u4.z), fmaxf_std(ub.w, u4.w));
        }
    }
    b.nb[no + p] = begin;
    b.ne[no + p] = end;
    b.npar[no + p] = parent;
    b.nleft[no + p] = left;
    b.nlb[no + p] = lb;
    b.nub[no + p] = ub;
    b.ndepth[no + p] = 0u;
    b.nsplit[no + p] = split;
}

// Internal-node AABBs, bottom-up in ONE launch (replaces the level-by-level loop of
// src/sim_bvh.cpp:411-485). One thread per leaf climbs towards the root; at every internal
// node the first arriving child stops, the second one (both subtrees final) forms the union of
// the two child boxes and goes on. min/max are exact and order-independent, so the boxes are
// bit-identical to the level-synchronous result. Hand-off between the two children (they may
// run on different XCDs): the box is written with agent-scope (sc1, write-through) stores, the
// storing wave waits for them (s_waitcnt vmcnt(0)) and only then bumps the node's counter with
// an agent-scope atomic; the thread whose add came last reads the child boxes with agent-scope
// (sc1) loads issued after its add has returned. No cache-wide fence is involved: a
// __threadfence() here costs an L2 write-back per call (measured: 6 ms instead of 0.05 ms), and
// so would a release / acquire pair on visit[] (an agent-scope release is that same L2
// write-back on gfx950, whose L2s are per XCD).
// ISA this relies on (outside the HIP memory model, which has no edge between relaxed atomics):
//   global_store_dword ... sc1   the box goes through to memory, past the XCD's own L2;
//   s_waitcnt vmcnt(0)           gfx9 counts stores in vmcnt: the stores have been acknowledged
//                                before the add is issued;
//   global_atomic_add ... sc0    performed at the memory side, returns the arrival order;
//   global_load_dword ... sc1    issued after the add has returned, misses the XCD's L2.
// tests/test_capi_symbols.py::test_refit_handoff_isa disassembles the built kernel and checks
// exactly these properties, so a toolchain that lowers the atomics differently fails the CPU
// suite; tests/test_gpu_fullsize.py and test_gpu_parity.py check the boxes themselves (exact
// unions, byte-identical trees) on many sizes and chunk counts.
CSC_DEV void st_agent(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
CSC_DEV float ld_agent(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// First the subtrees that lie inside one workgroup's 256 consecutive leaves, with LDS counters and
// LDS hand-off (k_bvh_emit_fast numbering only: a leaf finds its node from the Karras arrays, and
// a node's Karras id is (left child - 1) / 2): a thread climbs with its box in registers; at a
// parent whose range of leaves lies within the workgroup's it leaves its box in the parent's slot
// for its side and bumps the parent's counter (workgroup-scope acq_rel: plain s_waitcnt on
// gfx950); the first arrival stops, the second one reads the sibling's slot, forms the union,
// stores it as the parent's box and goes on. A thread that reaches a parent outside the workgroup
// flags its node (ndepth, free in this numbering) and stops: k_bvh_refit below continues from the
// flagged nodes across workgroups, after the kernel boundary has made every box visible. At 1M
// objects 1 node in 40 is left to the device-scope atomics. min / max are exact and
// order-independent: same boxes bit for bit.
CSC_GLOBAL __launch_bounds__(256) void k_bvh_refit_local(bvh_bufs b)
{
    __shared__ float4 s_lb[512], s_ub[512];
    __shared__ std::uint32_t s_cnt[256];
    const std::uint32_t c = blockIdx.y;
    const std::uint32_t nn = b.n_nodes[c], M = (nn + 1u) / 2u;
    const std::uint32_t R0 = blockIdx.x * 256u;
    if (R0 >= M || M == 1u) {
        return; // (the whole workgroup; a root that is a leaf has its box from the emit kernel)
    }
    s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const std::uint32_t r = R0 + threadIdx.x;
    if (r >= M) {
        return; // (no barrier below)
    }
    const std::uint32_t R1 = R0 + 256u < M ? R0 + 256u : M;
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max, po = static_cast<std::size_t>(c) * b.n;
    const std::int32_t kid_leaf = static_cast<std::int32_t>((M - 1u) + r);
    const std::int32_t kp = b.k_parent[no + kid_leaf];
    std::uint32_t cur = 2u * static_cast<std::uint32_t>(kp) + 1u + (b.k_right[po + kp] == kid_leaf ? 1u : 0u);
    float4 lb = b.nlb[no + cur], ub = b.nub[no + cur];
    while (true) {
        const std::int32_t par = b.npar[no + cur];
        if (par < 0) {
            return; // cur is the root and its box is stored
        }
        const std::uint32_t left = static_cast<std::uint32_t>(b.nleft[no + par]);
        const std::uint32_t kid = (left - 1u) >> 1;
        if (b.k_first[po + kid] < R0 || b.k_last[po + kid] >= R1) {
            b.ndepth[no + cur] = 1u; // continued by k_bvh_refit
            return;
        }
        const std::uint32_t slot = kid - R0, side = cur == left ? 0u : 1u;
        s_lb[2u * slot + side] = lb;
        s_ub[2u * slot + side] = ub;
        if (__hip_atomic_fetch_add(&s_cnt[slot], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
            return;
        }
        // (left, right) operand order as in k_bvh_refit: the same bits even for zeros of opposite sign
        const float4 l0 = s_lb[2u * slot], l1 = s_lb[2u * slot + 1u], u0 = s_ub[2u * slot], u1 = s_ub[2u * slot + 1u];
        lb = make_float4(fminf_std(l0.x, l1.x), fminf_std(l0.y, l1.y), fminf_std(l0.z, l1.z), fminf_std(l0.w, l1.w));
        ub = make_float4(fmaxf_std(u0.x, u1.x), fmaxf_std(u0.y, u1.y), fmaxf_std(u0.z, u1.z), fmaxf_std(u0.w, u1.w));
        cur = static_cast<std::uint32_t>(par);
        b.nlb[no + cur] = lb;
        b.nub[no + cur] = ub;
    }
}

// flagged != 0: the climbs start at the nodes k_bvh_refit_local flagged (ndepth) instead of at the leaves.
CSC_GLOBAL __launch_bounds__(256) void k_bvh_refit(bvh_bufs b, int flagged)
{
    const std::uint32_t c = blockIdx.y, p0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (p0 >= b.n_nodes[c]) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    if (flagged ? b.ndepth[no + p0] == 0u : b.nleft[no + p0] >= 0) {
        return; // climbs start at the leaves (at the flagged roots of the workgroup-local subtrees)
    }
    std::int32_t cur = b.npar[no + p0];
    while (cur >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const std::uint32_t arrived = atomicAdd(&b.visit[no + cur], 1u);
        if (arrived == 0u) {
            return;
        }
        asm volatile("" ::: "memory");
        const std::int32_t l = b.nleft[no + cur];
        const float *la = reinterpret_cast<const float *>(b.nlb + no + l), *ua = reinterpret_cast<const float *>(b.nub + no + l);
        float *lo = reinterpret_cast<float *>(b.nlb + no + cur), *uo = reinterpret_cast<float *>(b.nub + no + cur);
        // All sixteen loads are issued before the first value is used (one memory round trip per
        // level instead of eight: each store depends on its loads and the compiler waits for
        // everything outstanding in front of it).
        float cl[8], cu[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cl[k] = ld_agent(la + k);
            cu[k] = ld_agent(ua + k);
        }
        const std::int32_t up = b.npar[no + cur];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st_agent(lo + k, fminf_std(cl[k], cl[4 + k]));
            st_agent(uo + k, fmaxf_std(cu[k], cu[4 + k]));
        }
        cur = up;
    }
}

// Reference node layout (include/cascade/detail/sim_data.hpp:193-207), 60 bytes.
struct bvh_node_ref {
    std::uint32_t begin, end;
    std::int32_t parent, left, right;
    float lb[4], ub[4];
    std::uint32_t nn_level;
    int split_idx;
};
static_assert(sizeof(bvh_node_ref) == 60, "bvh_node layout");

CSC_GLOBAL __launch_bounds__(256) void k_bvh_export(bvh_bufs b, std::uint32_t chunk, bvh_node_ref *out)
{
    const std::uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const std::uint32_t nn = b.n_nodes[chunk];
    if (p >= nn) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(chunk) * b.nn_max;
    bvh_node_ref o;
    o.begin = b.nb[no + p];
    o.end = b.ne[no + p];
    o.parent = b.npar[no + p];
    o.left = b.nleft[no + p];
    o.right = o.left < 0 ? -1 : o.left + 1;
    const float4 lb = b.nlb[no + p], ub = b.nub[no + p];
    o.lb[0] = lb.x;
    o.lb[1] = lb.y;
    o.lb[2] = lb.z;
    o.lb[3] = lb.w;
    o.ub[0] = ub.x;
    o.ub[1] = ub.y;
    o.ub[2] = ub.z;
    o.ub[3] = ub.w;
    o.nn_level = b.level_cnt[chunk * BVH_MAX_LEVELS + b.ndepth[no + p]];
    o.split_idx = b.nsplit[no + p];
    out[p] = o;
}

// ---------------------------------------------------------------------------
// K6: broad phase. Replaces src/sim_broad_phase.cpp:92-202.
//
// What the reference computes. For every particle p it walks the tree from the root and
// descends into every node whose AABB overlaps p's AABB (closed intervals, 4 dimensions,
// :149-151); at a leaf it emits (orig_p, orig_i) for every member i with orig_p < orig_i and
// at least one activity flag (:175-185). A parent box is the exact float union of its
// children's boxes, so "p overlaps the leaf's box" implies "p overlaps every ancestor": the
// pair set is {(p, i) : box(p) overlaps box(leaf(i)), orig_p < orig_i, active}. Any traversal
// that reaches a superset of those leaves and then applies that final test yields the
// IDENTICAL set; only the order of the list differs (the reference's own order is
// unspecified: tbb::concurrent_vector).
//
// Wavefront-cooperative traversal. One wave owns 64 Morton-consecutive query particles and
// walks the tree ONCE for all of them with the union of their boxes (Morton neighbours have
// nearly coincident boxes, so the union visits about the nodes a single query visits, at
// 1/64 of the fetches). The walk itself is breadth-parallel: every round up to 64 pending
// nodes are popped from the wave's LDS stack, one per lane, each lane tests both children
// of its node against the group box (one 96-byte "wide" record per node: both child boxes
// and descriptors), overlapping internal children are pushed back (ballot-compacted),
// overlapping leaves go to an LDS leaf queue. Then the queue is drained: per leaf, all 64
// lanes test their OWN box against the leaf box and the hits are ballot-compacted into an
// LDS pair buffer that is flushed with one global atomic per ~200 pairs and fully coalesced
// 16-byte stores. There is no per-lane divergence anywhere and no dependent-load chain
// longer than the tree depth.
//
// A child descriptor is {tag, begin, end, orig}: tag = the child's own canonical index when it
// is internal; for a leaf tag = -1 - (index << 2 | act) where act is the activity byte of its
// (first) particle; [begin, end) the particle range; orig the original index of the first particle.
// ---------------------------------------------------------------------------
// (Padding the record to a full 128-byte line was measured and is slower for the dual traversal,
// 1.81 against 1.65 ms: the table outgrows the caches faster than the straddling lines cost.)
struct tnode {
    float4 lb0, ub0, lb1, ub1;
    int4 c0, c1;
};
static_assert(sizeof(tnode) == 96, "wide node layout");

// One thread per canonical node; internal nodes fill their record.
// sinfo / sleaf (for the tile kernel of k_bp_dual.hpp; null = not wanted): per sorted position,
// orig | act << 24 | (member of a multi-particle leaf) << 26, and for such members the canonical
// index of their leaf (whose box, not the particle's, is what a query is tested against).
constexpr std::uint32_t SINFO_MULTI = 1u << 26;
CSC_GLOBAL __launch_bounds__(256) void k_bvh_pack(bvh_bufs b, const std::uint32_t *vidx, const std::uint8_t *active,
                                                  tnode *tn, std::uint8_t *sact, std::uint32_t *sinfo, std::uint32_t *sleaf,
                                                  std::uint32_t *multi_flag)
{
    const std::uint32_t c = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.n_nodes[c]) {
        return;
    }
    const std::size_t no = static_cast<std::size_t>(c) * b.nn_max;
    const std::int32_t l = b.nleft[no + p];
    if (l < 0) {
        return;
    }
    const std::uint32_t *vx = vidx + static_cast<std::size_t>(c) * b.n;
    tnode t;
    t.lb0 = b.nlb[no + l];
    t.ub0 = b.nub[no + l];
    t.lb1 = b.nlb[no + l + 1];
    t.ub1 = b.nub[no + l + 1];
    const std::uint32_t b0 = b.nb[no + l], b1 = b.nb[no + l + 1];
    const std::uint32_t o0 = vx[b0], o1 = vx[b1];
    const int tag0 = b.nleft[no + l] < 0 ? -1 - ((l << 2) | static_cast<int>(active[o0] & 3u)) : l;
    const int tag1 = b.nleft[no + l + 1] < 0 ? -1 - (((l + 1) << 2) | static_cast<int>(active[o1] & 3u)) : l + 1;
    // Activity flags by sorted position, for the single-particle leaves (k_bp_dual.hpp).
    if (tag0 < 0) {
        sact[static_cast<std::size_t>(c) * b.n + b0] = active[o0];
    }
    if (tag1 < 0) {
        sact[static_cast<std::size_t>(c) * b.n + b1] = active[o1];
    }
    t.c0 = make_int4(tag0, static_cast<int>(b0), static_cast<int>(b.ne[no + l]), static_cast<int>(o0));
    t.c1 = make_int4(tag1, static_cast<int>(b1), static_cast<int>(b.ne[no + l + 1]), static_cast<int>(o1));
    tn[no + p] = t;
    if (sinfo != nullptr) {
        const std::size_t po = static_cast<std::size_t>(c) * b.n;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if ((k ? tag1 : tag0) < 0) {
                const std::uint32_t lb_ = k ? b1 : b0, le_ = b.ne[no + l + k];
                const std::uint32_t multi = le_ - lb_ > 1u ? SINFO_MULTI : 0u;
                if (multi) {
                    atomicOr(multi_flag, 1u); // (rare: equal Morton codes; k_bp_tiles picks its loop by it)
                }
                for (std::uint32_t m = lb_; m != le_; ++m) {
                    const std::uint32_t o = vx[m];
                    sinfo[po + m] = o | (static_cast<std::uint32_t>(active[o] & 3u) << 24) | multi;
                    if (multi) {
                        sleaf[po + m] = static_cast<std::uint32_t>(l + k);
                    }
                }
            }
        }
    }
}

// Output of the broad phase: pair blocks of up to PB_SIZE pairs {i, j}. A wave flushes its LDS
// staging buffer into a freshly allocated block, so every block holds pairs of ONE group of 64
// Morton-consecutive queries; the blocks of a group are chained (newest first) and k_pl_order
// then lists all blocks group by group. The narrow phase follows that list, which is what gives
// its Taylor-row gathers L1/L2 locality (the 64 query rows of a group are contiguous, the
// partner rows of neighbouring groups largely coincide).
// Blocks are allocated from PL_SHARDS independent pools (one counter each, on its own cache
// line): a single device-wide counter serialises ~10^5 atomics per launch at the memory side.
constexpr int PB_SIZE = 256;
constexpr int PL_SHARDS = 32;
constexpr int PL_CTR_STRIDE = 16; // counters are 128 bytes apart

struct pair_lists {
    uint2 *pairs;            // [cap_blocks][PB_SIZE] {pos_i, pos_j}: sorted positions, vidx[pos_i] < vidx[pos_j]
    std::uint32_t *pb_count; // [cap_blocks] pairs in the block
    std::int32_t *pb_next;   // [cap_blocks] next block of the same list, -1 = end
    std::int32_t *head;      // [nlists] first block of the list (valid when nblk > 0)
    std::uint32_t *nblk;     // [nlists] blocks in the list (zeroed before the launch)
    std::uint32_t *tchunk;   // [nlists] time chunk of the list
    unsigned long long *shard_ctr; // [PL_SHARDS][PL_CTR_STRIDE]: blocks allocated << 40 | pairs
    std::uint32_t cap_per_shard;   // pool s owns blocks [s * cap_per_shard, (s + 1) * cap_per_shard)
    std::uint32_t nlists;
};

// Group-major list of the pair blocks: order[off[g] + k] = {k-th block of list g, time chunk}.
CSC_GLOBAL __launch_bounds__(256) void k_pl_order(pair_lists pl, const std::uint32_t *off, uint2 *order)
{
    const std::uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.nlists) {
        return;
    }
    const std::uint32_t nb = pl.nblk[g];
    if (nb == 0u) {
        return;
    }
    const std::uint32_t c = pl.tchunk[g];
    std::int32_t b = pl.head[g];
    for (std::uint32_t k = 0; k < nb && b >= 0; ++k, b = pl.pb_next[b]) {
        order[off[g] + k] = make_uint2(static_cast<std::uint32_t>(b), c);
    }
}

struct bp_params {
    std::uint32_t n;
    std::uint32_t nn_max;
    const float4 *slb, *sub;     // [C][n]
    const std::uint32_t *vidx;   // [C][n]
    const tnode *tn;             // [C][nn_max]
    const float4 *nlb, *nub;     // [C][nn_max] (root box)
    const std::int32_t *nleft;   // root leaf test
    const std::uint8_t *active;  // [n] bit0 coll_active, bit1 conj_active
    pair_lists pl;
    unsigned long long *counter; // [1] statistics (diagnostic kernel)
    int *err_flag;               // traversal stack overflow (cannot happen for depth <= 64)
    std::uint32_t *gstats;       // debug builds: per group {nodes visited, cycles}
    // Query sharding across GPUs: this process walks the tree for the query groups dealt to
    // shard_rank (runs of BP_SHARD_RUN consecutive groups, round-robin over shard_n processes);
    // every process holds the whole tree. ngroups = all groups of a chunk.
    std::uint32_t shard_rank, shard_n, ngroups;
};

// Closed-interval overlap on the 4 dimensions (src/sim_broad_phase.cpp:149-151). Written with
// wave-wide ballots: every comparison is one v_cmp into an SGPR pair and the conjunction is
// scalar s_and_b64 work (per-lane boolean algebra costs the compiler ~3 VALU ops per term).
CSC_DEV unsigned long long bp_overlap_mask(const float4 &plb, const float4 &pub, const float4 &nl, const float4 &nu)
{
    unsigned long long m = __ballot(pub.x >= nl.x);
    m &= __ballot(plb.x <= nu.x);
    m &= __ballot(pub.y >= nl.y);
    m &= __ballot(plb.y <= nu.y);
    m &= __ballot(pub.z >= nl.z);
    m &= __ballot(plb.z <= nu.z);
    m &= __ballot(pub.w >= nl.w);
    m &= __ballot(plb.w <= nu.w);
    return m;
}
CSC_DEV bool bp_overlap(const float4 &plb, const float4 &pub, const float4 &nl, const float4 &nu)
{
    return (pub.x >= nl.x && plb.x <= nu.x) && (pub.y >= nl.y && plb.y <= nu.y) && (pub.z >= nl.z && plb.z <= nu.z)
           && (pub.w >= nl.w && plb.w <= nu.w);
}

constexpr int BPG_STACK = 768; // pending internal nodes of one wave
constexpr unsigned BP_SHARD_RUN = 16; // consecutive query groups that stay on one process
constexpr int BPG_PAIRQ = PB_SIZE + 64; // staged pairs of one wave: one pair block + one emission

// tn is passed on its own (not inside bp_params) so that it can carry __restrict__: with the
// node index made wave-uniform (v_readlane) the compiler then fetches the 96-byte record with
// scalar loads (s_load_dwordx8) into SGPRs, and the per-lane box tests take their node
// operands straight from SGPRs: no LDS traffic and no VGPRs for the broadcast.
CSC_GLOBAL __launch_bounds__(64) void k_broad_phase(const tnode *__restrict__ tn, bp_params q)
{
    __shared__ std::int32_t stack[BPG_STACK];
    __shared__ uint2 pairq[BPG_PAIRQ];
    // Chunk-major linear id, remapped so that each XCD walks one contiguous range of queries.
    const std::uint32_t lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const std::uint32_t c = lin / gridDim.x;
    const int lane = threadIdx.x;
    // Local group -> global group of 64 Morton-consecutive queries (identity for one process).
    const std::uint32_t kloc = lin % gridDim.x;
    const std::uint32_t grp = ((kloc / BP_SHARD_RUN) * q.shard_n + q.shard_rank) * BP_SHARD_RUN + kloc % BP_SHARD_RUN;
    if (grp >= q.ngroups) {
        return;
    }
    const std::uint32_t pidx = grp * 64u + lane;
    const bool valid = pidx < q.n;
    const std::size_t po = static_cast<std::size_t>(c) * q.n;
    const std::size_t no = static_cast<std::size_t>(c) * q.nn_max;
    tn += no;
    const float finf = __builtin_huge_valf();
    // Lanes past the end carry an empty box: they never overlap anything.
    float4 plb = make_float4(finf, finf, finf, finf), pub = make_float4(-finf, -finf, -finf, -finf);
    std::uint32_t orig_p = 0xffffffffu;
    bool act_p = false;
    if (valid) {
        plb = q.slb[po + pidx];
        pub = q.sub[po + pidx];
        orig_p = q.vidx[po + pidx];
        act_p = q.active[orig_p] != 0;
    }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int sp = 0, npq = 0;
#ifdef CSC_BP_GROUP_STATS
    std::uint32_t dbg_nodes = 0;
    const long long dbg_t0 = clock64();
#endif

    // Staging buffer -> a freshly allocated pair block, chained in front of this group's list.
    std::int32_t head_local = -1;
    std::uint32_t nblk_local = 0;
    const std::uint32_t shard = (blockIdx.y * gridDim.x + blockIdx.x) % PL_SHARDS;
    // Moves the first min(npq, PB_SIZE) staged pairs into a new block (mid-walk flushes are
    // always full blocks, so that the narrow phase's waves are full) and keeps the rest.
    auto flush = [&]() {
        __syncthreads();
        const int nout = npq < PB_SIZE ? npq : PB_SIZE;
        unsigned long long old = 0;
        if (lane == 0) {
            old = atomicAdd(q.pl.shard_ctr + shard * PL_CTR_STRIDE, (1ull << 40) | static_cast<unsigned long long>(nout));
        }
        const unsigned long long local = __shfl(old, 0) >> 40;
        if (local < q.pl.cap_per_shard) {
            const std::uint32_t blk = shard * q.pl.cap_per_shard + static_cast<std::uint32_t>(local);
            if (lane == 0) {
                q.pl.pb_count[blk] = static_cast<std::uint32_t>(nout);
                q.pl.pb_next[blk] = head_local;
            }
            head_local = static_cast<std::int32_t>(blk);
            ++nblk_local;
            uint2 *dst = q.pl.pairs + static_cast<std::size_t>(blk) * PB_SIZE;
            for (int k = lane; k < nout; k += 64) {
                dst[k] = pairq[k];
            }
        }
        const int rem = npq - nout; // < 64
        uint2 carry = make_uint2(0u, 0u);
        if (lane < rem) {
            carry = pairq[nout + lane];
        }
        __syncthreads();
        if (lane < rem) {
            pairq[lane] = carry;
        }
        npq = rem;
        __syncthreads();
    };
    const unsigned long long my_bit = 1ull << lane;
    const unsigned long long act_mask = __ballot(act_p);
    // mask: the lanes that emit the pair (their query, particle at sorted position pos). Pairs
    // are recorded as SORTED POSITIONS within the chunk (vidx maps them to original indices).
    auto emit = [&](unsigned long long mask, std::uint32_t pos) {
        if (mask != 0ull) {
            if (mask & my_bit) {
                pairq[npq + __popcll(mask & lt_mask)] = make_uint2(pidx, pos);
            }
            npq += __popcll(mask);
            if (npq >= PB_SIZE) {
                flush();
            }
        }
    };
    // A leaf whose box the lanes of `hit` overlap.
    auto leaf = [&](unsigned long long hit, int tag, int begin, int end, int orig) {
        if (end - begin == 1) {
            const std::uint32_t oi = static_cast<std::uint32_t>(orig);
            emit(hit & __ballot(orig_p < oi) & (((-1 - tag) & 3) != 0 ? ~0ull : act_mask), static_cast<std::uint32_t>(begin));
        } else {
            // Multi-particle leaf (equal Morton codes): the test is against the LEAF box for
            // every member (src/sim_broad_phase.cpp:161-168).
            for (std::uint32_t m = static_cast<std::uint32_t>(begin); m != static_cast<std::uint32_t>(end); ++m) {
                const std::uint32_t oi = q.vidx[po + m];
                emit(hit & __ballot(orig_p < oi) & (q.active[oi] != 0 ? ~0ull : act_mask), m);
            }
        }
    };

    {
        const unsigned long long hroot = bp_overlap_mask(plb, pub, q.nlb[no], q.nub[no]);
        if (hroot == 0ull) {
            return;
        }
        if (q.nleft[no] < 0) {
            // The whole tree is one leaf.
            const std::uint32_t o0 = q.vidx[po];
            leaf(hroot, -1 - static_cast<int>(q.active[o0]), 0, static_cast<int>(q.n), static_cast<int>(o0));
        } else {
            if (lane == 0) {
                stack[0] = 0;
            }
            sp = 1;
        }
    }
    __syncthreads();

    while (sp > 0) {
        // Pop up to 64 pending nodes, lane k holding the k-th. Near the stack capacity fall back
        // to one node per round (plain depth-first: the stack then grows by at most the
        // remaining tree depth).
        const int np = (sp > BPG_STACK - 192) ? 1 : (sp < 64 ? sp : 64);
        const std::int32_t my_node = (lane < np) ? stack[sp - 1 - lane] : 0;
        sp -= np;
#ifdef CSC_BP_GROUP_STATS
        dbg_nodes += np;
#endif
        __syncthreads();
        for (int k = 0; k < np; ++k) {
            const std::int32_t node = __builtin_amdgcn_readlane(my_node, k);
            const tnode *t = tn + node; // wave-uniform address
            const float4 lb0 = t->lb0, ub0 = t->ub0, lb1 = t->lb1, ub1 = t->ub1;
            const int4 c0 = t->c0, c1 = t->c1;
            const unsigned long long h0 = bp_overlap_mask(plb, pub, lb0, ub0), h1 = bp_overlap_mask(plb, pub, lb1, ub1);
            if (h0 != 0ull) {
                if (c0.x >= 0) {
                    if (lane == 0) {
                        stack[sp] = c0.x;
                    }
                    ++sp;
                } else {
                    leaf(h0, c0.x, c0.y, c0.z, c0.w);
                }
            }
            if (h1 != 0ull) {
                if (c1.x >= 0) {
                    if (lane == 0) {
                        stack[sp] = c1.x;
                    }
                    ++sp;
                } else {
                    leaf(h1, c1.x, c1.y, c1.z, c1.w);
                }
            }
        }
        if (sp > BPG_STACK - 130) {
            // Cannot happen for trees of depth <= 64 (see np above); never write out of bounds.
            if (lane == 0) {
                atomicOr(q.err_flag, 1);
            }
            return;
        }
        __syncthreads();
    }
    if (npq > 0) {
        flush();
    }
    if (lane == 0) {
        q.pl.head[lin] = head_local;
        q.pl.nblk[lin] = nblk_local;
        q.pl.tchunk[lin] = c;
#ifdef CSC_BP_GROUP_STATS
        q.gstats[2 * lin] = dbg_nodes;
        q.gstats[2 * lin + 1] = static_cast<std::uint32_t>(clock64() - dbg_t0);
#endif
    }
}

// Diagnostic: the reference's per-particle traversal (one lane per query, DFS stack in LDS),
// writing nothing and adding the number of node AABB tests (the reference's "nodes visited":
// every node popped from the stack of src/sim_broad_phase.cpp:121-196) to counter[1]. This is
// the unit count behind the broad phase's algorithmic bytes.
constexpr int BP_BLOCK = 64;
constexpr int BP_STACK = 72; // depth <= 64 -> at most 65 pending entries

CSC_GLOBAL __launch_bounds__(BP_BLOCK) void k_broad_phase_stats(bp_params q)
{
    __shared__ std::int32_t stack[BP_STACK][BP_BLOCK];
    const std::uint32_t c = blockIdx.y;
    const std::uint32_t pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= q.n) {
        return;
    }
    const int tid = threadIdx.x;
    const std::size_t po = static_cast<std::size_t>(c) * q.n;
    const std::size_t no = static_cast<std::size_t>(c) * q.nn_max;
    const float4 plb = q.slb[po + pidx], pub = q.sub[po + pidx];
    unsigned long long visits = 1;
    if (bp_overlap(plb, pub, q.nlb[no], q.nub[no]) && q.nleft[no] >= 0) {
        int sp = 0;
        stack[sp++][tid] = 0;
        do {
            const std::int32_t node = stack[--sp][tid];
            const tnode *t = q.tn + no + node;
            visits += 2;
            if (bp_overlap(plb, pub, t->lb0, t->ub0) && t->c0.x >= 0) {
                stack[sp++][tid] = t->c0.x;
            }
            if (bp_overlap(plb, pub, t->lb1, t->ub1) && t->c1.x >= 0) {
                stack[sp++][tid] = t->c1.x;
            }
        } while (sp > 0);
    }
    atomicAdd(q.counter + 1, visits);
}

} // namespace csc
