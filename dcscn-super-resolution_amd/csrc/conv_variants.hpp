// Variant table of conv_igemm shared by the per-family translation units (conv_k1.hip, conv_k3.hip) and
// the dispatcher (kernels.hip).  The families are compiled separately only to parallelise the build.
#pragma once
#include "conv_igemm.hpp"

namespace dcscn {

// (mt, kc, wps) as a function of (ks, nt), from tools/conv_tune.hip sweeps on MI355X (1024 48x48
// patches, profiles/r01_conv_tune_sweep*.txt): the single-LDS-buffer form with KC = 4 (3x3) / 16 (1x1) and
// as many waves per SIMD as the accumulators allow beat the double-buffered form everywhere (CNN2 124 ->
// 134 TFLOP/s, CNN7 104 -> 121, Up-PS 124 -> 138): extra resident workgroups hide the staging phases
// better than intra-workgroup double buffering does, and larger KC only cost occupancy.
__host__ __device__ constexpr int pick_mt(int ks, int nt) {
    if (ks == 1) return 2;
    return nt >= 8 ? 2 : (nt >= 5 ? 3 : 4);
}
__host__ __device__ constexpr int pick_kc(int ks, int nt) { return ks == 1 ? 16 : 4; }
__host__ __device__ constexpr int pick_wps(int ks, int nt) {
    if (nt >= 13) return 2;
    if (ks == 1) return nt <= 8 ? 4 : 3;
    return nt >= 7 ? 3 : 4;
}
constexpr bool kDoubleBuffer = false;

inline size_t lds_bytes_for(int ks, int mt, int nt, int kc, int dwk = 0) {
    const int halo = ks / 2;
    const int hp = (4 * mt + 2 * halo) * (16 + 2 * halo);
    const int ps = conv_plane_stride(hp);
    const int ns = conv_ns(nt);
    size_t floats = (kDoubleBuffer ? 2 : 1) * (size_t)(kc * ps + ks * ks * kc * ns);
    if (dwk > 0) floats += (size_t)kc * conv_plane_stride((4 * mt + dwk - 1) * (16 + dwk - 1)) + (size_t)dwk * dwk * kc;   // DwGeom
    return floats * sizeof(float);
}

#define DCSCN_FOR_NT(X, KS) \
    X(KS, 1) X(KS, 2) X(KS, 3) X(KS, 4) X(KS, 5) X(KS, 6) X(KS, 7) X(KS, 8) X(KS, 9) X(KS, 10) X(KS, 11) X(KS, 12) X(KS, 13)
// the fused-depthwise pointwise kernels are instantiated only for the channel-tile widths separable
// models use (<= 8 tiles of 16)
#define DCSCN_FOR_NT_DW(X, DWK) X(1, 1, DWK) X(1, 2, DWK) X(1, 3, DWK) X(1, 4, DWK) X(1, 5, DWK) X(1, 6, DWK) X(1, 7, DWK) X(1, 8, DWK)
constexpr int kMaxDwNt = 8;

template <int KS, int NT, int DWK = 0>
struct Variant {
    static constexpr int MT = pick_mt(KS, NT), KC = pick_kc(KS, NT), WPS = pick_wps(KS, NT);
    static constexpr auto kernel = &conv_igemm<KS, MT, NT, KC, kDoubleBuffer, WPS, DWK>;
    static size_t lds() { return lds_bytes_for(KS, MT, NT, KC, DWK); }
    static hipError_t set_attr() {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds());
    }
    static hipError_t launch(const ConvArgs& a, int n_tiles, hipStream_t stream) {
        const dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)n_tiles);
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds(), stream, a);
        return hipGetLastError();
    }
};

// per-family entry points (one translation unit each)
hipError_t conv_init_k1();
hipError_t conv_init_k3();
hipError_t conv_launch_k1(int nt, int dwk, const ConvArgs& a, int n_tiles, hipStream_t stream);
hipError_t conv_launch_k3(int nt, const ConvArgs& a, int n_tiles, hipStream_t stream);
// 5x5 (--cnn_size=5 and the folded linear tail) and 7x7 (--cnn_size=7); the 7x7 filter block of 49 taps
// limits the channel tile to 8 x 16 (113 KB of LDS)
#define DCSCN_FOR_NT_K7(X) X(7, 1) X(7, 2) X(7, 3) X(7, 4) X(7, 5) X(7, 6) X(7, 7) X(7, 8)
constexpr int kMaxK7Nt = 8;
hipError_t conv_init_k5();
hipError_t conv_init_k7();
hipError_t conv_launch_k5(int nt, const ConvArgs& a, int n_tiles, hipStream_t stream);
hipError_t conv_launch_k7(int nt, const ConvArgs& a, int n_tiles, hipStream_t stream);

}  // namespace dcscn
