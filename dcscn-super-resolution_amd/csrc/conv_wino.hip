// Winograd F(2x2,3x3) variants (conv_wino.hpp), one translation unit to parallelise the build.
#include <cstdlib>

#include "conv_wino.hpp"

namespace dcscn {

// groups whose workgroups are interleaved on the XCDs (see conv_wino): up to three at a time
static int wino_group_span(int n_groups) { return n_groups < 3 ? n_groups : 3; }   // measured: 1 / 2 / 3 / 4 / 8 -> 49.19 / 48.63 / 47.95 / 48.39 / 48.30 ms per step

template <int NT, int KC, int WPS>
static hipError_t wino_launch_one(const ConvArgs& a, int n_groups, hipStream_t stream) {
    using G = WinoGeom<NT, KC>;
    const size_t lds = (size_t)G::BUF * sizeof(float);
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    ConvArgs b = a;
    b.n_groups = n_groups;
    static const int env_span = getenv("DCSCN_WINO_SPAN") ? atoi(getenv("DCSCN_WINO_SPAN")) : 0;   // tuning aid
    b.group_span = env_span > 0 ? (env_span < n_groups ? env_span : n_groups) : wino_group_span(n_groups);
    const int phases = (n_groups + b.group_span - 1) / b.group_span;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8 * b.group_span * phases));      // 1-D, decoded XCD-aware in the kernel
    // single LDS buffer, 4 waves, filter reads software-pipelined three frequencies ahead (tools/wino_tune.hip)
    hipLaunchKernelGGL((conv_wino<NT, KC, WPS>), grid, dim3(256), lds, stream, b);
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
