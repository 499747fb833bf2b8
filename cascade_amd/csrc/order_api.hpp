// cascade_amd/csrc/order_api.hpp
//
// Taylor orders other than 20. The reference builds its integrators for any kw::tol
// (src/sim_basic.cpp:553-577; heyoka's order rule p = ceil(-ln(tol) / 2 + 1));
// here the Taylor order is a template parameter of the propagation, narrow-phase and dense-output
// kernels. Order 20 (tol = DBL_EPSILON, the default and every BASELINE configuration) is compiled
// into the main translation unit with the hand-unrolled dynamics; every other order
// CSC_ORDER_MIN..CSC_ORDER_MAX is one small translation unit (cascade_order.hip compiled with
// -DCSC_ORDER=P) holding the tape-interpreter propagation (any right-hand side, the built-in ones
// included) and the order-P narrow-phase / dense-output kernels behind this table of launchers.
#pragma once

#include <hip/hip_runtime.h>

#include "k_narrow.hpp"
#include "k_propagate.hpp"

namespace csc
{

constexpr int ORDER_MIN = 6, ORDER_MAX = 24;

struct order_api {
    int order;
    // k_propagate<P, events, store, DYN_TAPE>
    void (*propagate)(bool store, bool events, dim3 grid, dim3 block, hipStream_t stream, const prop_params &p);
    void (*np_fex)(dim3 grid, hipStream_t stream, const np_params &q);
    void (*np_fex_tr)(dim3 grid, hipStream_t stream, const np_params &q);
    void (*np_roots)(dim3 grid, hipStream_t stream, const np_params &q);
    void (*dense)(dim3 grid, hipStream_t stream, std::uint32_t n, double t, const std::uint64_t *tc_row, const std::uint32_t *nsub,
                  const double *tcs_xyz, const double *tcs_vr, const double *tcoords, double *out, int *err_flag);
};

} // namespace csc

// One per built order (defined by cascade_order.hip); csc_order_api(p) finds it (nullptr: not built).
extern "C" const csc::order_api *csc_order_api(int order);
