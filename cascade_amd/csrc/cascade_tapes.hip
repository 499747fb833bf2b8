// cascade_amd/csrc/cascade_tapes.hip
//
// Translation unit of the COMPILED tapes: k_propagate_wave with the Taylor recurrences of a fused
// schedule as straight-line code (csrc/gen/compiled_tapes.hpp, generated ahead of time by
// profiles/gen_compiled_tapes.py from cascade_wave.hip::wave_emit_source) instead of rows
// interpreted from LDS -- for the dynamics known when the library is built: what the reference
// gets from heyoka's LLVM JIT for any dynamics (src/sim_basic.cpp:553-577, EXTERNAL). A schedule is
// recognised by the hash of its one-operation-per-row plan; anything else runs interpreted.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hiprtc.h>

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

// ---- run-time compilation -----------------------------------------------------------------
//
// What heyoka's LLVM JIT is to the reference (src/sim_basic.cpp:553-577, EXTERNAL): dynamics the
// library did not know when it was built get their straight-line code at first use. The same text
// wave_emit_source writes for the ahead-of-time tapes is compiled by hipRTC together with the kernel
// headers, which are read from csrc/ next to the shared library (11 s per kernel variant for the
// default simple_earth() on the MI355X box's host, 31 s on a slow one). The code object lives as
// long as the process: nothing is written to disk. CASCADE_WAVE_JIT=0 switches it off (the
// interpreted rows then), 2 compiles the dynamics that have an ahead-of-time tape too.
namespace
{

constexpr int JIT_ID = 1000;

struct jit_kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    std::string err;
};

std::mutex jit_mutex;
std::map<std::pair<std::uint64_t, int>, jit_kernel> jit_cache; // (schedule hash, events + 2 x device): a module belongs to the device it was loaded on

std::string lib_dir()
{
    Dl_info info{};
    if (dladdr(reinterpret_cast<const void *>(&wave_compiled_lookup), &info) == 0 || info.dli_fname == nullptr) {
        return ".";
    }
    std::string p = info.dli_fname;
    const std::size_t k = p.find_last_of('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

void jit_build(const wave_plan &plan, std::uint32_t order, bool events, jit_kernel &out)
{
    // (the kernel headers: csrc/ beside the shared library, or where CASCADE_JIT_INCLUDE_DIR says)
    const char *inc_env = std::getenv("CASCADE_JIT_INCLUDE_DIR");
    const std::string inc = inc_env ? std::string(inc_env) : lib_dir() + "/csrc";
    for (const char *h : {"/dev_math.hpp", "/k_propagate.hpp", "/k_propagate_wave.hpp"}) {
        if (!std::ifstream(inc + h)) {
            out.err = "run-time compilation needs the kernel headers in " + inc + " (" + (h + 1) + " is missing)";
            return;
        }
    }
    const std::string src = "#include \"k_propagate_wave.hpp\"\nnamespace csc\n{\n" + wave_emit_source(plan, order, JIT_ID) + "}\n";
    char name[160];
    std::snprintf(name, sizeof(name), "csc::k_propagate_wave<20, 1, 8, %s, %d>", events ? "true" : "false", JIT_ID);
    int dev = 0;
    hipDeviceProp_t prop{};
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        out.err = "no device to compile for";
        return;
    }
    std::string arch = prop.gcnArchName; // ("gfx950:sramecc+:xnack-": the processor is what hipRTC wants)
    arch = arch.substr(0, arch.find(':'));
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "cascade_jit_tape.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        out.err = "hiprtcCreateProgram failed";
        return;
    }
    hiprtcAddNameExpression(prog, name);
    const std::string oarch = "--offload-arch=" + arch, oinc = "-I" + inc;
    const char *opts[] = {oarch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", oinc.c_str()};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    if (rc != HIPRTC_SUCCESS) {
        std::size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, 0);
        if (ls != 0u) {
            hiprtcGetProgramLog(prog, log.data());
        }
        out.err = std::string("hipRTC: ") + hiprtcGetErrorString(rc) + "\n" + log.substr(0, 2000);
        hiprtcDestroyProgram(&prog);
        return;
    }
    const char *low = nullptr;
    std::size_t cs = 0;
    if (hiprtcGetLoweredName(prog, name, &low) != HIPRTC_SUCCESS || low == nullptr || hiprtcGetCodeSize(prog, &cs) != HIPRTC_SUCCESS) {
        out.err = "hipRTC: no code for the kernel";
        hiprtcDestroyProgram(&prog);
        return;
    }
    const std::string lowered = low;
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (hipModuleLoadData(&out.mod, code.data()) != hipSuccess) {
        out.err = "hipModuleLoadData failed for the compiled tape";
        out.mod = nullptr;
        return;
    }
    if (hipModuleGetFunction(&out.fn, out.mod, lowered.c_str()) != hipSuccess) {
        out.err = "the compiled tape has no kernel " + lowered;
        out.fn = nullptr;
    }
}

} // namespace

hipError_t wave_launch_jit(const wave_plan &plan, std::uint32_t order, bool events, const prop_params &prm, hipStream_t stream, std::string &err)
{
    if (plan.ng != 1u || plan.lpg != 8u) {
        return hipErrorInvalidValue;
    }
    hipFunction_t fn = nullptr;
    {
        std::lock_guard<std::mutex> lock(jit_mutex);
        int dev = 0;
        if (const hipError_t rc = hipGetDevice(&dev); rc != hipSuccess) {
            return rc;
        }
        const auto key = std::make_pair(wave_plan_hash(plan), (events ? 1 : 0) + 2 * dev);
        auto it = jit_cache.find(key);
        if (it == jit_cache.end()) {
            it = jit_cache.emplace(key, jit_kernel{}).first;
            const auto t0 = std::chrono::steady_clock::now();
            jit_build(plan, order, events, it->second);
            if (std::getenv("CASCADE_WAVE_DEBUG")) {
                std::fprintf(stderr, "[wave] run-time compilation: %s (%.1f s)\n", it->second.fn ? "ok" : it->second.err.c_str(),
                             std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
        }
        if (it->second.fn == nullptr) {
            err = it->second.err;
            return hipErrorNotSupported;
        }
        fn = it->second.fn;
    }
    const std::size_t lds_bytes = plan.wave_bytes();
    wave_sched sc{};
    sc.nrows = plan.nrows;
    sc.nslots = plan.nslots;
    sc.obj_doubles = plan.obj_doubles;
    sc.row_units = plan.row_units;
    sc.one = plan.one;
    for (int c = 0; c < 7; ++c) {
        sc.rhs[c] = plan.rhs[c];
    }
    prop_params p = prm;
    void *args[] = {&p, &sc};
    if (lds_bytes > 48u * 1024u) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
        (void)hipGetLastError();
    }
    const unsigned grid = static_cast<unsigned>((static_cast<std::uint64_t>(prm.n) + 7u) / 8u);
    return hipModuleLaunchKernel(fn, grid, 1, 1, 64, 1, 1, static_cast<unsigned>(lds_bytes), stream, args, nullptr);
}

} // namespace csc
