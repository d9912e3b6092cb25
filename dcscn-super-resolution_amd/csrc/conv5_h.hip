// conv5_h variants (conv5_h.hpp): the folded tail of x2 / x3 / x4 models.
#include "conv5_h.hpp"

namespace dcscn {

template <int NT>
static hipError_t c5h_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_h<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C5HGeom<NT>::LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_h<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C5HGeom<NT>::LDS_BYTES);
}

hipError_t c5h_init_kernels() {
    hipError_t e = c5h_set_attr<1>();
    if (e == hipSuccess) e = c5h_set_attr<3>();
    return e != hipSuccess ? e : c5h_set_attr<4>();
}

template <int NT>
static hipError_t c5h_launch_one(const ConvArgs& a, hipStream_t stream) {
    const long long tiles = (long long)a.N * a.tiles_y * a.tiles_x;
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.in16.base) {
        if ((long long)a.N * a.H * a.W > kP16MaxPixels) return hipErrorInvalidValue;
        hipLaunchKernelGGL((conv5_h<NT, true>), dim3((unsigned)tiles), dim3(256), C5HGeom<NT>::LDS_BYTES, stream, a);
    } else
        hipLaunchKernelGGL((conv5_h<NT, false>), dim3((unsigned)tiles), dim3(256), C5HGeom<NT>::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t c5h_launch(int nt, const ConvArgs& a, hipStream_t stream) {
    if (!a.fold || (a.fold == 2 && nt != 1) || !a.wpack16 || a.tiles_x != (a.W + 15) / 16 || a.tiles_y != (a.H + 15) / 16 || a.n_chunks < 1) return hipErrorInvalidValue;
    switch (nt) {
        case 1: return c5h_launch_one<1>(a, stream);
        case 3: return c5h_launch_one<3>(a, stream);
        case 4: return c5h_launch_one<4>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t c5h_border_launch(const ConvArgs& a, hipStream_t stream) {
    if (a.fold != 2 || !a.wpack16 || !a.bias || a.n_chunks < 1 || a.ps < 2 || a.ps * a.ps > 16 || a.N < 1) return hipErrorInvalidValue;
    const long long grid = (fold_border_jobs(a.N, a.H, a.W).total + 3) / 4;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.in16.base) {
        if ((long long)a.N * a.H * a.W > kP16MaxPixels) return hipErrorInvalidValue;
        hipLaunchKernelGGL((fold_border<true>), dim3((unsigned)grid), dim3(256), 4 * kFbWinBytes, stream, a);
    } else
        hipLaunchKernelGGL((fold_border<false>), dim3((unsigned)grid), dim3(256), 4 * kFbWinBytes, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
