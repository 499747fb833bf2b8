// cascade_amd/csrc/wave_api.hpp
//
// Host interface of the several-lanes-per-object propagation kernel (k_propagate_wave.hpp; its
// translation unit is cascade_wave.hip): the tape scheduled into rows of mutually independent
// operations, and the launch.
#pragma once

#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "k_propagate_wave.hpp"

namespace csc
{

struct wave_plan {
    std::vector<wave_op> ops;   // [nrows][ng]
    std::vector<wave_op0> ops0; // [nrows][ng]
    std::uint32_t nrows = 0, ng = 0, nslots = 0, obj_doubles = 0;
    std::int32_t rhs[7] = {};
    [[nodiscard]] std::size_t lds_bytes() const
    {
        // (the objects' series, then the schedule of the orders >= 1)
        return static_cast<std::size_t>(64u / (8u * ng)) * obj_doubles * sizeof(double) + static_cast<std::size_t>(nrows) * ng * 32u;
    }
};

constexpr std::size_t WAVE_LDS_MAX = 160u * 1024u;

// Groups per object (1, 2, 4, 8) for a tape with `nslots` series at order P (0: the series of one
// object do not fit the LDS of a compute unit).
int wave_pick_ng(std::uint32_t nslots, std::uint32_t nops, std::uint32_t order);
// Schedules the tape (validated by check_cfg) for `ng` groups per object.
void wave_build_plan(const tape_op *tape, std::uint32_t nops, std::uint32_t nslots, const std::int32_t rhs[7], int ng,
                     std::uint32_t order, wave_plan &out);
// Launches k_propagate_wave<20, ng, events> over prm.n objects (sc.ops / sc.ops0 on the device).
hipError_t wave_launch(int ng, bool events, const prop_params &prm, const wave_sched &sc, std::size_t lds_bytes, hipStream_t stream);

} // namespace csc
