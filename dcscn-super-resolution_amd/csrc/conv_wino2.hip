// Winograd F(2x2,3x3) variants (conv_wino2.hpp), one translation unit to parallelise the build.
// Built with -fno-slp-vectorize: hipcc would otherwise pair the input transform's adds into v_pk_add_f32, which costs
// v_mov shuffles and issues worse beside MFMAs (tools/wino2_tune: 36.1 vs 35.6 ms over the 3x3 layers of the bench model).
#include <cstdlib>

#include "conv_wino2.hpp"

namespace dcscn {

// groups whose workgroups are interleaved on the XCDs (see conv_wino2): up to three at a time
static int wino_group_span(int n_groups) { return n_groups < 3 ? n_groups : 3; }   // r01, measured: 1 / 2 / 3 / 4 / 8 -> 49.19 / 48.63 / 47.95 / 48.39 / 48.30 ms per step

template <int NT>
static hipError_t wino_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino2<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, Wino2Geom<NT>::LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino2_redo<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, Wino2Geom<NT>::LDS_BYTES);
}

hipError_t wino_init_kernels() {
    hipError_t e = wino_set_attr<1>();
    if (e == hipSuccess) e = wino_set_attr<2>();
    return e != hipSuccess ? e : wino_set_attr<3>();
}

template <int NT>
static hipError_t wino_launch_one(const ConvArgs& a, int n_groups, hipStream_t stream) {
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    ConvArgs b = a;
    b.n_groups = n_groups;
    static const int env_span = getenv("DCSCN_WINO_SPAN") ? atoi(getenv("DCSCN_WINO_SPAN")) : 0;   // tuning aid
    b.group_span = env_span > 0 ? (env_span < n_groups ? env_span : n_groups) : wino_group_span(n_groups);
    const int phases = (n_groups + b.group_span - 1) / b.group_span;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8 * b.group_span * phases));      // 1-D, decoded XCD-aware in the kernel
    if (b.redo_check)                                          // behind conv3_h: 64 tile flags per workgroup (conv_wino2_redo)
        hipLaunchKernelGGL((conv_wino2_redo<NT>), dim3((unsigned)((tiles + 63) / 64)), dim3(256), Wino2Geom<NT>::LDS_BYTES, stream, b);
    else
        hipLaunchKernelGGL((conv_wino2<NT>), grid, dim3(256), Wino2Geom<NT>::LDS_BYTES, stream, b);
    return hipGetLastError();
}

hipError_t wino_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream) {
    if (a.n_full < 1 || a.n_full > n_groups || (nt == 1 && a.n_full != n_groups)) return hipErrorInvalidValue;
    switch (nt) {
        case 1: return wino_launch_one<1>(a, n_groups, stream);
        case 2: return wino_launch_one<2>(a, n_groups, stream);
        case 3: return wino_launch_one<3>(a, n_groups, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
