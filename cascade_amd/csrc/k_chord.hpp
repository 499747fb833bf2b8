// cascade_amd/csrc/k_chord.hpp
//
// The geometric exclusion test in front of the polynomial narrow phase (no counterpart in the
// reference; see k_narrow.hpp for what it guarantees): used by k_np_prefilter* over a stored pair
// list and by k_bp_tiles<.., FUSE> on the hits of a tile while they are still on chip.
#pragma once

#include "dev_math.hpp"

namespace csc
{

// The test proper: the sample differences d_s = a_s - b_s, s = 0..4, and the two radii.
CSC_DEV bool np_prefilter_core(const float (&dx)[5], const float (&dy)[5], const float (&dz)[5], float thr_extra, float deva,
                               float devb)
{
    const float thr = (thr_extra + deva + devb) * 1.000002f;
    const float thr2 = thr * thr * 1.000002f;
    bool keep = !(thr2 < __builtin_huge_valf()); // invalid record (dev = +inf) or overflow: never excluded
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        // min over u in [0, 1] of |d_s + u (d_{s+1} - d_s)|^2
        const float ex = dx[sidx + 1] - dx[sidx], ey = dy[sidx + 1] - dy[sidx], ez = dz[sidx + 1] - dz[sidx];
        // (explicit fused multiply-adds: the library is built with -ffp-contract=off for the FP64 stages)
        const float ee = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
        const float de = fmaf(dx[sidx], ex, fmaf(dy[sidx], ey, dz[sidx] * ez));
        // (v_rcp_f32, 1 ulp, instead of the 10-instruction IEEE division: at the minimiser the
        // distance is stationary in u, so an error of 3e-7 in u changes m2 by ~1e-13 |e|^2)
        float u = ee > 0.f ? -de * __builtin_amdgcn_rcpf(ee) : 0.f;
        u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u);
        const float mx = fmaf(u, ex, dx[sidx]), my = fmaf(u, ey, dy[sidx]), mz = fmaf(u, ez, dz[sidx]);
        const float m2 = fmaf(mx, mx, fmaf(my, my, mz * mz));
        // Any u in [0, 1] gives an upper bound of the true minimum only if it IS the minimiser;
        // rounding of u moves the point along the segment by ~1e-7 |e|, which the margin in dev
        // and the factor on thr absorb. NaNs compare false: the pair is kept.
        keep = keep || !(m2 > thr2);
    }
    return keep;
}

// The particles at sorted positions pa, pb of the chunk whose arrays start at offset co; srt_recs:
// [C][n][4] float4 sampled-trajectory records in sorted order. Returns false when the pair can be dropped.
// (the two 64-byte records already loaded)
CSC_DEV bool chord_keep_recs(const float4 (&ra)[4], const float4 (&rb)[4], float thr_extra)
{
    const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3];
    const float4 b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
    // Sample differences d_s = a_s - b_s, s = 0..4 (x, y, z).
    const float dx[5] = {a0.x - b0.x, a0.w - b0.w, a1.z - b1.z, a2.y - b2.y, a3.x - b3.x};
    const float dy[5] = {a0.y - b0.y, a1.x - b1.x, a1.w - b1.w, a2.z - b2.z, a3.y - b3.y};
    const float dz[5] = {a0.z - b0.z, a1.y - b1.y, a2.x - b2.x, a2.w - b2.w, a3.z - b3.z};
    // First a much cheaper test that settles 99.5 % of the box-overlap pairs of a LEO population
    // (profiles/exp_filter_selectivity.py): if, along one axis, the five sample differences all lie
    // beyond the threshold on the same side, then so does every chord difference (a chord runs
    // between consecutive samples), and the distance never drops to the threshold. Its margin
    // (1.000002 on the length against 1.000002 on the square below) makes it exclude only what the
    // test below excludes as well; NaNs and an infinite radius compare false: the pair goes on.
    const float thr1 = (thr_extra + a3.w + b3.w) * 1.000002f * 1.000002f;
    const float nx = fminf(fminf(fminf(dx[0], dx[1]), fminf(dx[2], dx[3])), dx[4]), mx = fmaxf(fmaxf(fmaxf(dx[0], dx[1]), fmaxf(dx[2], dx[3])), dx[4]);
    const float ny = fminf(fminf(fminf(dy[0], dy[1]), fminf(dy[2], dy[3])), dy[4]), my = fmaxf(fmaxf(fmaxf(dy[0], dy[1]), fmaxf(dy[2], dy[3])), dy[4]);
    const float nz = fminf(fminf(fminf(dz[0], dz[1]), fminf(dz[2], dz[3])), dz[4]), mz = fmaxf(fmaxf(fmaxf(dz[0], dz[1]), fmaxf(dz[2], dz[3])), dz[4]);
    // (fminf / fmaxf drop a NaN operand, and a NaN sample must not exclude: the sum of the
    // differences is NaN as soon as one of them is)
    const float sm = (((dx[0] + dx[1]) + (dx[2] + dx[3])) + ((dy[0] + dy[1]) + (dy[2] + dy[3]))) + (((dz[0] + dz[1]) + (dz[2] + dz[3])) + ((dx[4] + dy[4]) + dz[4]));
    const bool finite = sm == sm;
    if (finite && (nx > thr1 || mx < -thr1 || ny > thr1 || my < -thr1 || nz > thr1 || mz < -thr1)) {
        return false;
    }
    return np_prefilter_core(dx, dy, dz, thr_extra, a3.w, b3.w);
}

CSC_DEV bool chord_keep(const float4 *srt_recs, std::size_t co, std::uint32_t pa, std::uint32_t pb, float thr_extra)
{
    const float4 *pa4 = srt_recs + (co + pa) * 4u, *pb4 = srt_recs + (co + pb) * 4u;
    const float4 ra[4] = {pa4[0], pa4[1], pa4[2], pa4[3]}, rb[4] = {pb4[0], pb4[1], pb4[2], pb4[3]};
    return chord_keep_recs(ra, rb, thr_extra);
}

} // namespace csc
