// oracle/ref_pin.cpp
//
// TEST INFRASTRUCTURE ONLY. Thin C entry points over the two pieces of the
// reference that compile standalone (STL only): src/detail/mortonND_LUT.h and
// src/detail/ival.hpp. They are #included from where they lie under
// /root/reference (never copied); the resulting library goes to oracle/_ref/
// (git-ignored) and is used to pin the oracle's restatements of the Morton
// encoder and of interval arithmetic, and to generate tests/golden/ vectors.
//
// Build: make -C oracle ref   (needs /root/reference; skipped when absent)

#include <cstdint>

#include <detail/ival.hpp>
#include <detail/mortonND_LUT.h>

extern "C" {

// sim_propagate.cpp:411,477: MortonNDLutEncoder<4, 16, 8>().Encode(n0, n1, n2, n3)
std::uint64_t ref_morton_encode(std::uint64_t n0, std::uint64_t n1, std::uint64_t n2, std::uint64_t n3)
{
    constexpr auto enc = mortonnd::MortonNDLutEncoder<4, 16, 8>();
    return enc.Encode(n0, n1, n2, n3);
}

// sim_propagate.cpp:342-349: interval Horner with cascade::detail::ival
void ref_ival_horner(const double *ptr, std::uint32_t order, double lo, double hi, double *out2)
{
    using cascade::detail::ival;
    const ival h_int(lo, hi);
    auto acc = ival(ptr[order]);
    for (auto o = 1u; o <= order; ++o) {
        acc = ival(ptr[order - o]) + acc * h_int;
    }
    out2[0] = acc.lower;
    out2[1] = acc.upper;
}

void ref_ival_mul(double al, double au, double bl, double bu, double *out2)
{
    using cascade::detail::ival;
    const auto r = ival(al, au) * ival(bl, bu);
    out2[0] = r.lower;
    out2[1] = r.upper;
}
}
