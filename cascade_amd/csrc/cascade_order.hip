// cascade_amd/csrc/cascade_order.hip -- the order-P kernels of libcascade_hip.so for ONE Taylor
// order P != 20 (see order_api.hpp). Compiled once per order:
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -c -DCSC_ORDER=15 \
//         -Iinclude -Icascade_amd/csrc cascade_amd/csrc/cascade_order.hip -o build/order_15.o
#include "order_api.hpp"

#ifndef CSC_ORDER
#error "compile with -DCSC_ORDER=<Taylor order>"
#endif

namespace
{

constexpr int P = CSC_ORDER;
static_assert(P >= csc::ORDER_MIN && P <= csc::ORDER_MAX && P != 20, "order outside the built range");

void propagate(bool store, bool events, dim3 grid, dim3 block, hipStream_t stream, const csc::prop_params &p)
{
    if (store) {
        if (events) {
            hipLaunchKernelGGL((csc::k_propagate<P, true, true, csc::DYN_TAPE>), grid, block, 0, stream, p);
        } else {
            hipLaunchKernelGGL((csc::k_propagate<P, false, true, csc::DYN_TAPE>), grid, block, 0, stream, p);
        }
    } else {
        if (events) {
            hipLaunchKernelGGL((csc::k_propagate<P, true, false, csc::DYN_TAPE>), grid, block, 0, stream, p);
        } else {
            hipLaunchKernelGGL((csc::k_propagate<P, false, false, csc::DYN_TAPE>), grid, block, 0, stream, p);
        }
    }
}

void np_fex(dim3 grid, hipStream_t stream, const csc::np_params &q)
{
    hipLaunchKernelGGL((csc::k_np_fex<P>), grid, dim3(64), 0, stream, q);
}

void np_fex_tr(dim3 grid, hipStream_t stream, const csc::np_params &q)
{
    hipLaunchKernelGGL((csc::k_np_fex_tr<P>), grid, dim3(64), 0, stream, q);
}

void np_roots(dim3 grid, hipStream_t stream, const csc::np_params &q)
{
    hipLaunchKernelGGL((csc::k_np_roots<P>), grid, dim3(64), 0, stream, q);
}

void dense(dim3 grid, hipStream_t stream, std::uint32_t n, double t, const std::uint64_t *tc_row, const std::uint32_t *nsub,
           const double *tcs_xyz, const double *tcs_vr, const double *tcoords, double *out, int *err_flag)
{
    hipLaunchKernelGGL((csc::k_dense<P>), grid, dim3(256), 0, stream, n, t, tc_row, nsub, tcs_xyz, tcs_vr, tcoords, out, err_flag);
}

const csc::order_api api{P, propagate, np_fex, np_fex_tr, np_roots, dense};

} // namespace

#define CSC_CAT_(a, b) a##b
#define CSC_CAT(a, b) CSC_CAT_(a, b)
extern "C" __attribute__((visibility("hidden"))) const csc::order_api *CSC_CAT(csc_order_api_, CSC_ORDER)()
{
    return &api;
}
