// conv_nin_h with 8 waves x two 16-pixel tiles = 256 pixels per workgroup, ONE workgroup per CU (conv_nin_h.hpp built with other constants under
// other names): half the filter traffic per pixel.  Same products in the same order per pixel: bit-identical to the 128-pixel workgroups.
// Measured same-box in r05 (profiles/r05_ninh_ablation.txt): -3 % on the 1301-channel GEMM of the L12 nets, +7 ... +9 % on 540 / 131 channels --
// so only the wide K axes take it (conv_nin_h.hip: nin_h_launch), P16 sources, six output tiles.
#define NINH_WAVES 8
#define NINH_MT 2
#define NinHGeom NinHGeomW8
#define conv_nin_h_body conv_nin_h_body_w8
#define conv_nin_h conv_nin_h_w8
#include "conv_nin_h.hpp"

namespace dcscn {

constexpr int kNinH8Stages = 3;
constexpr int kNinH8MaxTable = 16 * 1024;

hipError_t nin_h8_init_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_nin_h_w8<6, 2, kNinH8Stages>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               NinHGeomW8<6, kNinH8Stages>::LDS_BYTES + kNinH8MaxTable);
}

hipError_t nin_h8_launch(const ConvArgs& a, int n_groups, hipStream_t stream) {
    using G = NinHGeomW8<6, kNinH8Stages>;
    const long long npix = (long long)a.N * a.H * a.W;
    const size_t table = (size_t)a.n_chunks * 64;
    if (!a.in16.base || !a.srctab || table > (size_t)kNinH8MaxTable || npix > kP16MaxPixels) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((npix + G::PIX - 1) / G::PIX), (unsigned)n_groups);
    hipLaunchKernelGGL((conv_nin_h_w8<6, 2, kNinH8Stages>), grid, dim3(G::THREADS), G::LDS_BYTES + table, stream, a);
    return hipGetLastError();
}

}  // namespace dcscn
