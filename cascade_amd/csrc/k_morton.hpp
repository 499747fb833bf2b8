// cascade_amd/csrc/k_morton.hpp
//
// Per-chunk global AABB (exact f32 min/max, two-stage deterministic reduction), 64-bit
// Morton codes of the AABB centres, and the gather of the sorted AABBs.
//
// Replaces:
//   src/sim_propagate.cpp:883-920 + src/detail/atomic_utils.cpp:20-55  local/global AABB fold
//   src/sim_propagate.cpp:94-124   disc_single_coord (all in f32)
//   src/sim_propagate.cpp:440-479  Morton loop, MortonNDLutEncoder<4,16,8> (mortonND_LUT.h:166-323)
//   src/sim_propagate.cpp:493-504  gather of srt_lbs / srt_ubs / srt_mcodes
// All stages are HBM bound: 32 B read per (chunk, particle) for the reduction, 32 B read +
// 8 B written for the codes, 4 + 32 B read (gathered) + 32 B written for the gather.
#pragma once

#include "dev_math.hpp"

namespace csc
{

constexpr int GA_BLOCK = 256;
constexpr int GA_MAX_PARTIAL = 1024;

CSC_DEV float4 f4min(float4 a, float4 b)
{
    return make_float4(fminf_std(a.x, b.x), fminf_std(a.y, b.y), fminf_std(a.z, b.z), fminf_std(a.w, b.w));
}
CSC_DEV float4 f4max(float4 a, float4 b)
{
    return make_float4(fmaxf_std(a.x, b.x), fmaxf_std(a.y, b.y), fmaxf_std(a.z, b.z), fmaxf_std(a.w, b.w));
}

CSC_DEV float4 wave_min4(float4 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        float4 o;
        o.x = __shfl_xor(v.x, d);
        o.y = __shfl_xor(v.y, d);
        o.z = __shfl_xor(v.z, d);
        o.w = __shfl_xor(v.w, d);
        v = f4min(v, o);
    }
    return v;
}
CSC_DEV float4 wave_max4(float4 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        float4 o;
        o.x = __shfl_xor(v.x, d);
        o.y = __shfl_xor(v.y, d);
        o.z = __shfl_xor(v.z, d);
        o.w = __shfl_xor(v.w, d);
        v = f4max(v, o);
    }
    return v;
}

// Stage 1: grid (nb, C). partial: [C][nb][2] float4 (min, max).
__global__ __launch_bounds__(GA_BLOCK) void k_gaabb_partial(const float4 *lbs, const float4 *ubs, std::uint32_t n,
                                                            float4 *partial)
{
    __shared__ float4 smin[GA_BLOCK / 64], smax[GA_BLOCK / 64];
    const std::uint32_t chunk = blockIdx.y, nb = gridDim.x;
    const float finf = __builtin_huge_valf();
    float4 mn = make_float4(finf, finf, finf, finf), mx = make_float4(-finf, -finf, -finf, -finf);
    const std::size_t off = static_cast<std::size_t>(chunk) * n;
    for (std::uint32_t i = blockIdx.x * GA_BLOCK + threadIdx.x; i < n; i += nb * GA_BLOCK) {
        mn = f4min(mn, lbs[off + i]);
        mx = f4max(mx, ubs[off + i]);
    }
    mn = wave_min4(mn);
    mx = wave_max4(mx);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        smin[wid] = mn;
        smax[wid] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < GA_BLOCK / 64; ++w) {
            mn = f4min(mn, smin[w]);
            mx = f4max(mx, smax[w]);
        }
        partial[(static_cast<std::size_t>(chunk) * nb + blockIdx.x) * 2u] = mn;
        partial[(static_cast<std::size_t>(chunk) * nb + blockIdx.x) * 2u + 1u] = mx;
    }
}

// Stage 2: one block (64 threads) per chunk. glb/gub: [C] float4 (NOT bumped).
__global__ __launch_bounds__(64) void k_gaabb_final(const float4 *partial, std::uint32_t nb, float4 *glb, float4 *gub)
{
    const std::uint32_t chunk = blockIdx.x;
    const float finf = __builtin_huge_valf();
    float4 mn = make_float4(finf, finf, finf, finf), mx = make_float4(-finf, -finf, -finf, -finf);
    for (std::uint32_t b = threadIdx.x; b < nb; b += 64) {
        mn = f4min(mn, partial[(static_cast<std::size_t>(chunk) * nb + b) * 2u]);
        mx = f4max(mx, partial[(static_cast<std::size_t>(chunk) * nb + b) * 2u + 1u]);
    }
    mn = wave_min4(mn);
    mx = wave_max4(mx);
    if (threadIdx.x == 0) {
        glb[chunk] = mn;
        gub[chunk] = mx;
    }
}

// src/sim_propagate.cpp:94-124, all in f32.
CSC_DEV std::uint64_t disc_single_coord(float x, float mn, float mx)
{
    const float isize = mx - mn;
    float rx = (x - mn) / isize;
    rx = rx >= 0.f ? rx : 0.f;
    rx *= 65536.f;
    std::uint64_t retval;
    if (rx >= 18446744073709551616.f) {
        retval = ~0ull;
    } else {
        retval = static_cast<std::uint64_t>(rx);
    }
    return retval < 65535ull ? retval : 65535ull;
}

// Bit b of the 16-bit field goes to bit 4 b of the result.
CSC_DEV std::uint64_t spread4(std::uint64_t v)
{
    v &= 0xffffull;
    v = (v | (v << 24)) & 0x000000ff000000ffull;
    v = (v | (v << 12)) & 0x000f000f000f000full;
    v = (v | (v << 6)) & 0x0303030303030303ull;
    v = (v | (v << 3)) & 0x1111111111111111ull;
    return v;
}

// grid (ceil(n / 256), C). glb/gub are the raw per-chunk global bounds; the upper bound is
// bumped by one ulp here (:448-449). err_flag is raised if the bumped size is not finite (:453-456).
__global__ __launch_bounds__(256) void k_morton(const float4 *lbs, const float4 *ubs, std::uint32_t n,
                                                const float4 *glb, const float4 *gub, std::uint64_t *codes,
                                                int *err_flag)
{
    const std::uint32_t chunk = blockIdx.y;
    const std::uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float finf = __builtin_huge_valf();
    const float4 gl = glb[chunk];
    float4 gu = gub[chunk];
    gu.x = nextafterf(gu.x, finf);
    gu.y = nextafterf(gu.y, finf);
    gu.z = nextafterf(gu.z, finf);
    gu.w = nextafterf(gu.w, finf);
    if (i == 0) {
        const bool ok = (fabsf(gu.x - gl.x) <= FLT_MAX) && (fabsf(gu.y - gl.y) <= FLT_MAX)
                        && (fabsf(gu.z - gl.z) <= FLT_MAX) && (fabsf(gu.w - gl.w) <= FLT_MAX);
        if (!ok) {
            atomicOr(err_flag, 1);
        }
    }
    if (i >= n) {
        return;
    }
    const std::size_t off = static_cast<std::size_t>(chunk) * n + i;
    const float4 lb = lbs[off], ub = ubs[off];
    const float cx = lb.x / 2 + ub.x / 2, cy = lb.y / 2 + ub.y / 2, cz = lb.z / 2 + ub.z / 2, cr = lb.w / 2 + ub.w / 2;
    const std::uint64_t n0 = disc_single_coord(cx, gl.x, gu.x);
    const std::uint64_t n1 = disc_single_coord(cy, gl.y, gu.y);
    const std::uint64_t n2 = disc_single_coord(cz, gl.z, gu.z);
    const std::uint64_t n3 = disc_single_coord(cr, gl.w, gu.w);
    codes[off] = spread4(n0) | (spread4(n1) << 1) | (spread4(n2) << 2) | (spread4(n3) << 3);
}

// grid (ceil(n / 256), C).
// recs / srt_recs: the 64-byte sampled-trajectory records (may be null).
__global__ __launch_bounds__(256) void k_gather_aabb(const float4 *lbs, const float4 *ubs, const std::uint32_t *vidx,
                                                     std::uint32_t n, float4 *srt_lbs, float4 *srt_ubs,
                                                     const float4 *recs, float4 *srt_recs)
{
    const std::uint32_t chunk = blockIdx.y;
    const std::uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const std::size_t off = static_cast<std::size_t>(chunk) * n;
    const std::uint32_t s = i < n ? vidx[off + i] : 0u;
    if (i < n) {
        srt_lbs[off + i] = lbs[off + s];
        srt_ubs[off + i] = ubs[off + s];
    }
    if (recs != nullptr) {
        // Four lanes per record: one 64-byte request per gathered record, and every store of the
        // wave covers 1024 contiguous bytes (one lane per record would write 16 bytes of 64 lines).
        const std::uint32_t lane = threadIdx.x & 63u, piece = lane & 3u;
        const std::uint32_t wave_base = i - lane;
#pragma unroll
        for (std::uint32_t q = 0; q < 4; ++q) {
            const std::uint32_t src_lane = 16u * q + (lane >> 2);
            const std::uint32_t sq = static_cast<std::uint32_t>(__shfl(static_cast<int>(s), static_cast<int>(src_lane)));
            const std::uint32_t obj = wave_base + src_lane;
            if (obj < n) {
                srt_recs[(off + obj) * 4u + piece] = recs[(off + sq) * 4u + piece];
            }
        }
    }
}

} // namespace csc
