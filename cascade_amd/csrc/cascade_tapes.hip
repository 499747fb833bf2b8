// cascade_amd/csrc/cascade_tapes.hip
//
// Translation unit of the COMPILED tapes: k_propagate_wave with the Taylor recurrences of a fused
// schedule as straight-line code (csrc/gen/compiled_tapes.hpp, generated ahead of time by
// profiles/gen_compiled_tapes.py from cascade_wave.hip::wave_emit_source) instead of rows
// interpreted from LDS -- for the dynamics known when the library is built: what the reference
// gets from heyoka's LLVM JIT for any dynamics (src/sim_basic.cpp:553-577, EXTERNAL). A schedule is
// recognised by the hash of its one-operation-per-row plan; anything else runs interpreted.
#include "wave_api.hpp"

#include "gen/compiled_tapes.hpp"

namespace csc
{

int wave_compiled_lookup(std::uint64_t hash)
{
    for (const wave_compiled_entry &e : wave_compiled_table) {
        if (e.hash == hash) {
            return e.id;
        }
    }
    return 0;
}

hipError_t wave_launch_compiled(int id, const wave_plan &plan, bool events, const prop_params &prm, hipStream_t stream)
{
    if (plan.ng != 1u || plan.lpg != 8u) {
        return hipErrorInvalidValue;
    }
    // 8 objects per wave, one wave per workgroup: no schedule in LDS, only the objects' series.
    const std::size_t lds_bytes = plan.wave_bytes();
    const dim3 grid(static_cast<unsigned>((static_cast<std::uint64_t>(prm.n) + 7u) / 8u)), block(64);
    wave_sched sc{};
    sc.nrows = plan.nrows;
    sc.nslots = plan.nslots;
    sc.obj_doubles = plan.obj_doubles;
    sc.row_units = plan.row_units;
    sc.one = plan.one;
    for (int c = 0; c < 7; ++c) {
        sc.rhs[c] = plan.rhs[c];
    }
#define CSC_TAPE_CASE(ID)                                                                                               \
    case ID: {                                                                                                          \
        auto ke = k_propagate_wave<20, 1, 8, true, ID>;                                                                 \
        auto kn = k_propagate_wave<20, 1, 8, false, ID>;                                                                \
        const void *kern = events ? reinterpret_cast<const void *>(ke) : reinterpret_cast<const void *>(kn);            \
        if (lds_bytes > 48u * 1024u) {                                                                                  \
            const hipError_t rc = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)); \
            if (rc != hipSuccess) {                                                                                     \
                return rc;                                                                                              \
            }                                                                                                           \
        }                                                                                                               \
        if (events) {                                                                                                   \
            hipLaunchKernelGGL(ke, grid, block, lds_bytes, stream, prm, sc);                                            \
        } else {                                                                                                        \
            hipLaunchKernelGGL(kn, grid, block, lds_bytes, stream, prm, sc);                                            \
        }                                                                                                               \
        break;                                                                                                          \
    }
    switch (id) {
        CSC_TAPE_CASE(1)
        CSC_TAPE_CASE(2)
        CSC_TAPE_CASE(3)
        default:
            return hipErrorInvalidValue;
    }
#undef CSC_TAPE_CASE
    return hipGetLastError();
}

} // namespace csc
