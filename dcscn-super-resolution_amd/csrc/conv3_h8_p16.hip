// conv3_h8 variants with P16 tensors in and out (conv3_h8.hpp: P16; p16.hpp), one translation unit to parallelise the build.
#include "conv3_h8.hpp"

namespace dcscn {

template <int NT, int C1>
static hipError_t c3e16_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_h8<NT, C1, 0, NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C3EGeom<NT>::LDS_BYTES);
}

hipError_t c3e16_init_kernels() {
    hipError_t e = c3e16_set_attr<6, 6>();
    if (e == hipSuccess) e = c3e16_set_attr<6, 5>();
    if (e == hipSuccess) e = c3e16_set_attr<5, 5>();
    if (e == hipSuccess) e = c3e16_set_attr<5, 4>();
    if (e == hipSuccess) e = c3e16_set_attr<4, 4>();
    return e != hipSuccess ? e : c3e16_set_attr<4, 3>();
}

template <int NT, int C1>
static hipError_t c3e16_launch_one(const ConvArgs& a, int wgs, hipStream_t stream) {
    hipLaunchKernelGGL((conv3_h8<NT, C1, 0, NT, true>), dim3((unsigned)wgs), dim3(512), C3EGeom<NT>::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// (a: eligible and completed by c3e_launch)
hipError_t c3e16_launch(int nt, const ConvArgs& a, int wgs, hipStream_t stream) {
    if ((long long)a.N * a.H * a.W > kP16MaxPixels) return hipErrorInvalidValue;
    const bool eq = a.n_full == 2;
    switch (nt) {
        case 6: return eq ? c3e16_launch_one<6, 6>(a, wgs, stream) : c3e16_launch_one<6, 5>(a, wgs, stream);
        case 5: return eq ? c3e16_launch_one<5, 5>(a, wgs, stream) : c3e16_launch_one<5, 4>(a, wgs, stream);
        case 4: return eq ? c3e16_launch_one<4, 4>(a, wgs, stream) : c3e16_launch_one<4, 3>(a, wgs, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
