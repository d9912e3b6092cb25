// conv_nin_h variants (conv_nin_h.hpp), one translation unit to parallelise the build.
#include <cstdlib>

#include "conv_nin_h.hpp"

namespace dcscn {

constexpr int kNinHStages = 3;                   // input stages: chunk c + 3 is fetched while chunk c computes (3.02 vs 3.08 ms with 2)
constexpr int kNinHMaxTable = 16 * 1024;         // LDS bytes for the multi-source quad table: 1024 quads = 4096 input channels

template <int NT>
static hipError_t nin_h_set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_nin_h<NT, 0, kNinHStages>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       NinHGeom<NT, kNinHStages>::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_nin_h<NT, 2, kNinHStages>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            NinHGeom<NT, kNinHStages>::LDS_BYTES + kNinHMaxTable);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_nin_h<NT, 1, kNinHStages>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               NinHGeom<NT, kNinHStages>::LDS_BYTES + kNinHMaxTable);
}

hipError_t nin_h8_init_kernels();                                // conv_nin_h_w8.hip: 256 pixels per workgroup, for the wide K axes
hipError_t nin_h8_launch(const ConvArgs& a, int n_groups, hipStream_t stream);
constexpr int kNinH8MinChunks = 32;                             // K >= 1024 channels (-3 % at 1301; slower at 540: profiles/r05_ninh_ablation.txt)

hipError_t nin_h_init_kernels() {
    hipError_t e = nin_h8_init_kernels();
    if (e != hipSuccess) return e;
    e = nin_h_set_attr<1>();
    if (e == hipSuccess) e = nin_h_set_attr<2>();
    if (e == hipSuccess) e = nin_h_set_attr<3>();
    if (e == hipSuccess) e = nin_h_set_attr<4>();
    if (e == hipSuccess) e = nin_h_set_attr<5>();
    return e != hipSuccess ? e : nin_h_set_attr<6>();
}

template <int NT>
static hipError_t nin_h_launch_one(const ConvArgs& a, int n_groups, hipStream_t stream) {
    using G = NinHGeom<NT, kNinHStages>;
    const long long npix = (long long)a.N * a.H * a.W;
    const dim3 grid((unsigned)((npix + G::PIX - 1) / G::PIX), (unsigned)n_groups);
    if (a.in16.base) {                                           // P16 sources: a.srctab holds one entry per channel OCTET (4 per chunk)
        const size_t table = (size_t)a.n_chunks * 64;
        if (!a.srctab || table > (size_t)kNinHMaxTable || npix > kP16MaxPixels) return hipErrorInvalidValue;
        hipLaunchKernelGGL((conv_nin_h<NT, 2, kNinHStages>), grid, dim3(NinHGeom<NT, kNinHStages>::THREADS), G::LDS_BYTES + table, stream, a);
    } else if (a.srctab) {
        const size_t table = (size_t)a.n_chunks * 128;           // 8 quads of 16 bytes per 32-channel chunk
        if (table > (size_t)kNinHMaxTable) return hipErrorInvalidValue;
        hipLaunchKernelGGL((conv_nin_h<NT, 1, kNinHStages>), grid, dim3(NinHGeom<NT, kNinHStages>::THREADS), G::LDS_BYTES + table, stream, a);
    } else {
        hipLaunchKernelGGL((conv_nin_h<NT, 0, kNinHStages>), grid, dim3(NinHGeom<NT, kNinHStages>::THREADS), G::LDS_BYTES, stream, a);
    }
    return hipGetLastError();
}

hipError_t nin_h_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream) {
    if (a.n_full < 1 || a.n_full > n_groups || (nt == 1 && a.n_full != n_groups) || !a.wpack16) return hipErrorInvalidValue;
    static const bool w8 = !(getenv("DCSCN_NINH8") && getenv("DCSCN_NINH8")[0] == '0');     // (A/B aid: DCSCN_NINH8=0 keeps the 128-pixel workgroups)
    if (w8 && nt == 6 && a.in16.base && a.n_chunks >= kNinH8MinChunks && a.n_full == n_groups) return nin_h8_launch(a, n_groups, stream);
    switch (nt) {
        case 1: return nin_h_launch_one<1>(a, n_groups, stream);
        case 2: return nin_h_launch_one<2>(a, n_groups, stream);
        case 3: return nin_h_launch_one<3>(a, n_groups, stream);
        case 4: return nin_h_launch_one<4>(a, n_groups, stream);
        case 5: return nin_h_launch_one<5>(a, n_groups, stream);
        case 6: return nin_h_launch_one<6>(a, n_groups, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dcscn
