// Winograd F(2x2,3x3) variants (conv_wino.hpp), one translation unit to parallelise the build.
#include "conv_wino.hpp"

namespace dcscn {

template <int NT, int KC, int WPS>
static hipError_t wino_launch_one(const ConvArgs& a, int n_groups, hipStream_t stream) {
    using G = WinoGeom<NT, KC>;
    const size_t lds = (size_t)G::BUF * sizeof(float);
    const dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)n_groups);
    // single LDS buffer, 4 waves, filter reads software-pipelined three frequencies ahead (tools/wino_tune.hip)
    hipLaunchKernelGGL((conv_wino<NT, KC, WPS>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t wino_launch(int nt, int kc, const ConvArgs& a, int n_groups, hipStream_t stream) {
    if (a.nt_last < 1 || a.nt_last > nt) return hipErrorInvalidValue;
    // the 1-tile tail group of a layer: 8 input channels per chunk at 4 waves per SIMD -- there the input-tile
    // loads (one 16-byte piece per pixel and chunk) are the bottleneck, and 32-byte pieces halve them
    if (nt == 1 && kc == kWinoTailKC) return wino_launch_one<1, kWinoTailKC, 4>(a, n_groups, stream);
    if (kc != kWinoKC) return hipErrorInvalidValue;
    switch (nt) {
        case 1: return wino_launch_one<1, kWinoKC, 2>(a, n_groups, stream);
        case 2: return wino_launch_one<2, kWinoKC, 2>(a, n_groups, stream);
        case 3: return wino_launch_one<3, kWinoKC, 2>(a, n_groups, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
