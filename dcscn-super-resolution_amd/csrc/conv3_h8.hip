// conv3_h8 variants (conv3_h8.hpp): layers whose output channels form two channel groups (7 .. 12 tiles of 16), one translation unit.
#include "conv3_h8.hpp"

namespace dcscn {

template <int NT, int C1>
static hipError_t c3e_set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_h8<NT, C1>), hipFuncAttributeMaxDynamicSharedMemorySize, C3EGeom<NT>::LDS_BYTES);
}

hipError_t c3e16_init_kernels();                                // conv3_h8_p16.hip: P16 tensors in and out
hipError_t c3e16_launch(int nt, const ConvArgs& a, int wgs, hipStream_t stream);

hipError_t c3e_init_kernels() {
    hipError_t e = c3e16_init_kernels();
    if (e != hipSuccess) return e;
    e = c3e_set_attr<6, 6>();
    if (e == hipSuccess) e = c3e_set_attr<6, 5>();
    if (e == hipSuccess) e = c3e_set_attr<5, 5>();
    if (e == hipSuccess) e = c3e_set_attr<5, 4>();
    if (e == hipSuccess) e = c3e_set_attr<4, 4>();
    return e != hipSuccess ? e : c3e_set_attr<4, 3>();
}

// conv3_h8 takes the launches conv3_h would run with exactly two channel groups of nt >= 4 tiles (the pair (nt, nt) or (nt, nt - 1)),
// no depth_to_space and no residual (its lean epilogue); everything else stays on conv3_h
bool c3e_eligible(int nt, const ConvArgs& a, int n_groups) {
    return n_groups == 2 && nt >= 4 && nt <= 6 && a.n_full >= 1 && a.n_full <= 2 && a.ps == 1 && a.res == nullptr && (a.act == ACT_ALPHA || a.act == ACT_NONE) &&
           a.n_chunks >= 3;
}

template <int NT, int C1>
static hipError_t c3e_launch_one(const ConvArgs& a, int wgs, hipStream_t stream) {
    hipLaunchKernelGGL((conv3_h8<NT, C1>), dim3((unsigned)wgs), dim3(512), C3EGeom<NT>::LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t c3e_launch(int nt, const ConvArgs& args, int n_groups, int n_cus, hipStream_t stream) {
    if (!c3e_eligible(nt, args, n_groups) || !args.wpack16 || args.tiles_x != (args.W + 15) / 16 || args.tiles_y != (args.H + 15) / 16) return hipErrorInvalidValue;
    ConvArgs a = args;
    a.n_groups = n_groups;
    a.nt_pack = nt;
    const long long units = (long long)a.N * a.tiles_y * a.tiles_x;     // one (pixel tile, group pair) per unit
    if (units > 0x7fffffffLL) return hipErrorInvalidValue;
    const int wgs = (int)(units < n_cus ? units : n_cus);                 // one persistent workgroup per CU
    // P16 in and out (p16.hpp): the variant that stages its image by LDS-DMA; float32 in and out: the r04 kernel; anything mixed is not ours
    const bool out16 = a.out0.p16.base != nullptr && (a.split >= (1 << 29) || a.out1.p16.base != nullptr);
    if (a.in16.base && out16) return c3e16_launch(nt, a, wgs, stream);
    if (a.in16.base || a.out0.p16.base || a.out1.p16.base) return hipErrorInvalidValue;
    const bool eq = a.n_full == 2;
    switch (nt) {
        case 6: return eq ? c3e_launch_one<6, 6>(a, wgs, stream) : c3e_launch_one<6, 5>(a, wgs, stream);
        case 5: return eq ? c3e_launch_one<5, 5>(a, wgs, stream) : c3e_launch_one<5, 4>(a, wgs, stream);
        case 4: return eq ? c3e_launch_one<4, 4>(a, wgs, stream) : c3e_launch_one<4, 3>(a, wgs, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
