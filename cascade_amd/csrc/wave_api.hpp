// cascade_amd/csrc/wave_api.hpp
//
// Host interface of the several-lanes-per-object propagation kernel (k_propagate_wave.hpp; its
// translation unit is cascade_wave.hip): the tape fused into sum-of-products operations and
// scheduled into rows of mutually independent ones, and the launch.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "k_propagate_wave.hpp"

namespace csc
{

constexpr std::size_t WAVE_LDS_MAX = 160u * 1024u;

struct wave_plan {
    std::vector<std::uint32_t> blk; // [nrows][row_units][ng] 16-byte units, 4 words each
    std::vector<wave_op0> ops0;     // [nrows][ng]
    std::vector<std::uint32_t> place_level; // [nrows][ng]: dependency level of the operation there (0: empty place)
    std::uint32_t nrows = 0, ng = 0, nslots = 0, obj_doubles = 0, row_units = 0, one = 0;
    std::uint32_t lpg = 8;          // lanes that share one operation of a row (4 or 8): an object owns lpg x ng lanes
    std::uint32_t wpb = 1;          // waves of a workgroup (they share the schedule in LDS)
    // What it was built from, and what the fusion made of it.
    std::uint32_t tape_nslots = 0, fused_ops = 0, depth = 0;
    std::int32_t rhs[7] = {}, tape_rhs[7] = {};
    [[nodiscard]] std::size_t blk_bytes() const
    {
        return blk.size() * sizeof(std::uint32_t);
    }
    [[nodiscard]] std::size_t wave_bytes() const
    {
        return static_cast<std::size_t>(64u / (lpg * ng)) * obj_doubles * sizeof(double);
    }
    // (of a workgroup: the objects' series of every wave, then the schedule)
    [[nodiscard]] std::size_t lds_bytes() const
    {
        return wpb * wave_bytes() + blk_bytes();
    }
    [[nodiscard]] unsigned waves_per_cu() const
    {
        // (LDS is handed out in blocks: three workgroups of 53 824 bytes did not share a compute
        // unit, three of 52 928 did; counted with 2 KB blocks)
        const std::size_t granted = (lds_bytes() + 2047u) / 2048u * 2048u;
        return static_cast<unsigned>(WAVE_LDS_MAX / granted) * wpb;
    }
};

// Fuses and schedules the tape (validated by check_cfg) for `ng` groups per object (1, 2, 4, 8; 0:
// the layout with the best estimated throughput). False: the series of one object do not fit
// the LDS of a compute unit.
bool wave_build_plan(const tape_op *tape, std::uint32_t nops, std::uint32_t nslots, const std::int32_t rhs[7], int ng,
                     std::uint32_t order, wave_plan &out);
// Launches k_propagate_wave<20, plan.ng, plan.lpg, events> over prm.n objects (the plan's blk / ops0 on the device).
hipError_t wave_launch(const wave_plan &plan, const std::uint32_t *d_blk, const wave_op0 *d_ops0, bool events, const prop_params &prm,
                       hipStream_t stream);
// What identifies a schedule (the compiled tapes of csrc/gen/compiled_tapes.hpp are found by it).
std::uint64_t wave_plan_hash(const wave_plan &plan);
// Straight-line source of a plan with one operation per row, as wave_compiled_rows<id> (build tool).
std::string wave_emit_source(const wave_plan &plan, std::uint32_t order, int id);
// Compiled tapes (cascade_tapes.hip): the id of the schedule with this hash, 0 when there is none;
// the launch of k_propagate_wave<20, 1, 8, events, id> for a plan with one operation per row.
int wave_compiled_lookup(std::uint64_t hash);
hipError_t wave_launch_compiled(int id, const wave_plan &plan, bool events, const prop_params &prm, hipStream_t stream);
// Run-time compilation (cascade_tapes.hip; CASCADE_WAVE_JIT=0 switches it off): the same kernel for a schedule that has
// no ahead-of-time tape, compiled by hipRTC at first use, kept for the life of the process.
// hipErrorNotSupported with `err` set when it cannot be built (the caller falls back on the interpreted rows).
hipError_t wave_launch_jit(const wave_plan &plan, std::uint32_t order, bool events, const prop_params &prm, hipStream_t stream,
                           std::string &err);
// The plan executed on the host, operation by operation as the kernel does (tests of the fusion
// and the schedule without a GPU): Taylor coefficients [7][order + 1] of x, y, z, vx, vy, vz, r.
void wave_plan_eval(const wave_plan &plan, std::uint32_t order, const double state[7], const double *pars, double time, double *out);

} // namespace csc
