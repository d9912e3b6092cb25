// Winograd F(2x2,3x3) variants (conv_wino.hpp), one translation unit to parallelise the build.
#include "conv_wino.hpp"

namespace dcscn {

template <int NT>
static hipError_t wino_launch_one(const ConvArgs& a, int n_groups, hipStream_t stream) {
    using G = WinoGeom<NT, kWinoKC>;
    const size_t lds = (size_t)G::BUF * sizeof(float);
    const dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)n_groups);
    // single LDS buffer, 4 waves, filter reads software-pipelined three frequencies ahead (tools/wino_tune.hip)
    hipLaunchKernelGGL((conv_wino<NT, kWinoKC, 2, false, 0, 4, false, 0, 3>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t wino_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream) {
    if (a.nt_last < 1 || a.nt_last > nt) return hipErrorInvalidValue;
    switch (nt) {
        case 1: return wino_launch_one<1>(a, n_groups, stream);
        case 2: return wino_launch_one<2>(a, n_groups, stream);
        case 3: return wino_launch_one<3>(a, n_groups, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
