// P16: activations stored ONCE in the form the split16 kernels consume (kernels.h: P16Desc) -- the f16 (hi, lo) pair of split16.hpp
// instead of the float32 value, same 4 bytes per value.  The producer's epilogue does the split (3 VALU per 2 values, once per value
// instead of once per consumer and channel group), the consumers stage their input by LDS-DMA: no global load into registers, no
// conversion, no ds_write in any K loop.
//
// Layout.  Per 32-channel chunk one plane; a plane starts with a 128-byte ZERO record (what out-of-image halo pixels and channel
// octets past the tensor's last are fetched from: SAME padding without a branch or an EXEC mask in the DMA), then one record per pixel:
//
//     record of a full chunk (128 bytes, 128-byte aligned):  [hi c0-7][lo c0-7][hi c8-15][lo c8-15][hi c16-23][lo c16-23][hi c24-31][lo c24-31]
//     record of the last chunk: the same with 1 / 2 / 3 / 4 octets (32 / 64 / 96 / 128 bytes)
//
// A consumer's lane fetches one 16-byte unit (global_load_lds_dwordx4); the lane -> unit permutation on the SOURCE side gives the
// swizzled LDS image (c3h_unit) although the DMA destination is lane-linear.  Channels past the tensor's last inside its last octet
// are written as exact zeros by the producer (zero filter columns, zero bias, PReLU(0) = 0), like the float32 layout's padding.
//
// Range: hi = f16(x) is +-inf for |x| >= 65520.  The producer's epilogue checks the hi pieces it stores (v_dot2_f32_f16 against
// zero: NaN for inf / NaN) and raises the image's redo flag; a flagged image is recomputed by the float32 plan (exec.hip).
#pragma once
#include "split16.hpp"

namespace dcscn {

__device__ __forceinline__ h2 p16_opaque_zero2() {
    unsigned z = 0u;
    asm volatile("" : "+v"(z));                 // keeps the compiler from folding the dot product with zero away
    return __builtin_bit_cast(h2, z);
}

// chk stays 0 while both halves of `pk` are finite f16 values, NaN otherwise (inf * 0, NaN * 0)
__device__ __forceinline__ float p16_check(float chk, unsigned pk, h2 zero2) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pk), zero2, chk, false);
}

// Epilogue side.  A lane of an MFMA accumulator tile holds 4 consecutive output channels (4 lk .. 4 lk + 3 of the 16-channel tile,
// lk = lane >> 4) of pixel lane & 15.  After the split, lane groups 0 / 1 (and 2 / 3) exchange halves (v_permlane16_swap_b32) so that
// every lane ends up with one whole 16-byte unit of its pixel's record:
//     lk 0: hi of channels 0-7,  lk 1: lo of channels 0-7,  lk 2: hi of channels 8-15,  lk 3: lo of channels 8-15
// i.e. unit index lk of the tile's two octets.  All 64 lanes must be active.
__device__ __forceinline__ u32x4 p16_unit(const f32x4 v, float m1, float& chk, h2 zero2) {
    h4 hi, lo;
    split4(v, m1, hi, lo);
    const u32x2 hu = __builtin_bit_cast(u32x2, hi), lu = __builtin_bit_cast(u32x2, lo);
    chk = p16_check(chk, hu.x, zero2);
    chk = p16_check(chk, hu.y, zero2);
    // vdst = hi, src = lo: odd rows of vdst <-> even rows of src.  Row 0 then holds (own hi, row 1's hi), row 1 (row 0's lo, own lo).
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(hu.x, lu.x, false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(hu.y, lu.y, false, false);
    return u32x4{s0.x, s1.x, s0.y, s1.y};
}

// The same for a lane whose value may lie outside the tensor (a pixel row / column past the image edge, a channel octet past the last):
// such a lane takes part in the swap but must not raise the image's redo flag -- the float32-tensor epilogues only check what they store
// (ADVICE r05), and "a flagged image equals its split16 = 0 run, every other image keeps its bits" needs the same rule here.
__device__ __forceinline__ u32x4 p16_unit(const f32x4 v, float m1, float& chk, h2 zero2, bool live) {
    float c = chk;
    const u32x4 unit = p16_unit(v, m1, c, zero2);
    chk = live ? c : chk;
    return unit;
}

// float32 NHWC -> P16 (tests, the harness, and tensors a float32 kernel produced for a split16 consumer); one thread per (pixel, octet)
static __global__ void p16_pack_kernel(const float* in, int in_stride, int channels, long long npix, P16Desc d) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * d.octs) return;
    const long long p = idx / d.octs;
    const int oct = (int)(idx - p * d.octs);
    const int chunk = oct >> 2, rec = p16_rec_bytes(d.octs, chunk);
    const float m1 = opaque_minus_one();
    f32x4 x0 = {0.0f, 0.0f, 0.0f, 0.0f}, x1 = x0;
    const float* src = in + (size_t)p * in_stride + oct * 8;
    float t[8];
    for (int i = 0; i < 8; ++i) t[i] = oct * 8 + i < channels ? src[i] : 0.0f;
    x0 = f32x4{t[0], t[1], t[2], t[3]};
    x1 = f32x4{t[4], t[5], t[6], t[7]};
    h8 hi, lo;
    split8(x0, x1, m1, hi, lo);
    char* dst = d.base + (long long)chunk * d.plane + 128 + p * rec + (oct & 3) * 32;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 16) = lo;
}

// P16 -> two float32 NHWC tensors of the stored pieces (hi, lo as floats): bit-level comparison of a P16 tensor with split(x)
static __global__ void p16_unpack_kernel(P16Desc d, long long npix, float* hi_out, float* lo_out, int out_stride) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * d.octs) return;
    const long long p = idx / d.octs;
    const int oct = (int)(idx - p * d.octs);
    const int chunk = oct >> 2, rec = p16_rec_bytes(d.octs, chunk);
    const char* src = d.base + (long long)chunk * d.plane + 128 + p * rec + (oct & 3) * 32;
    const h8 hi = *reinterpret_cast<const h8*>(src), lo = *reinterpret_cast<const h8*>(src + 16);
    for (int i = 0; i < 8; ++i)
        if (oct * 8 + i < out_stride) {
            hi_out[(size_t)p * out_stride + oct * 8 + i] = (float)hi[i];
            lo_out[(size_t)p * out_stride + oct * 8 + i] = (float)lo[i];
        }
}

}  // namespace dcscn
