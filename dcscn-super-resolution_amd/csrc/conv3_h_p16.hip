// conv3_h variants that read a P16 tensor (conv3_h.hpp: IN16; p16.hpp), one translation unit to parallelise the build.
#include "conv3_h.hpp"

namespace dcscn {

template <int NT>
static hipError_t c3h16_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_h<NT, 2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C3HGeom<NT>::LDS_BYTES);
}

hipError_t c3h16_init_kernels() {
    hipError_t e = c3h16_set_attr<1>();
    if (e == hipSuccess) e = c3h16_set_attr<2>();
    if (e == hipSuccess) e = c3h16_set_attr<3>();
    if (e == hipSuccess) e = c3h16_set_attr<4>();
    if (e == hipSuccess) e = c3h16_set_attr<5>();
    return e != hipSuccess ? e : c3h16_set_attr<6>();
}

template <int NT>
static hipError_t c3h16_launch_one(ConvArgs a, int n_groups, hipStream_t stream) {
    a.n_groups = n_groups;
    a.group_span = n_groups < 3 ? n_groups : 3;
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    const int phases = (n_groups + a.group_span - 1) / a.group_span;
    const long long ids = ((tiles + 7) / 8) * 8 * a.group_span * phases;
    if (ids > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv3_h<NT, 2, 0, true>), dim3((unsigned)ids), dim3(256), C3HGeom<NT>::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// (arguments checked by c3h_launch)
hipError_t c3h16_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream) {
    if ((long long)a.N * a.H * a.W > kP16MaxPixels) return hipErrorInvalidValue;
    switch (nt) {
        case 1: return c3h16_launch_one<1>(a, n_groups, stream);
        case 2: return c3h16_launch_one<2>(a, n_groups, stream);
        case 3: return c3h16_launch_one<3>(a, n_groups, stream);
        case 4: return c3h16_launch_one<4>(a, n_groups, stream);
        case 5: return c3h16_launch_one<5>(a, n_groups, stream);
        case 6: return c3h16_launch_one<6>(a, n_groups, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
