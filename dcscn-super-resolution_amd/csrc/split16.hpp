// f32-accurate contractions on the 16-bit matrix pipe (numerics: tools/f16x3_numerics.py,
// profiles/r03_f16x3_numerics.txt).
//
// An f32 operand x is carried as two f16 pieces, x = hi + lo with hi = f16(x) and lo = f16(x - hi) (x - hi is exact in f32,
// so the pair holds x to 2^-22..2^-24 relative while lo is a normal f16, i.e. for |x| >= ~2^-3; below that lo is subnormal and the
// pair has an absolute error floor of 2^-25 ~ 3e-8 -- weights are scaled per layer to stay clear of it, activations are not: see
// include/dcscn.h "split16" and test_small_magnitude_inputs_on_split16), and a product a*b is taken as  ah*bh + ah*bl + al*bh  -- three
// v_mfma_f32_16x16x{16,32}_f16 instead of four (eight) v_mfma_f32_16x16x4_f32, at 1/16 of their cycles per MAC.  Every
// f16 x f16 product is exact in f32, the instruction sums its 16 / 32 products and adds them to the f32 accumulator, so the
// result is at least as accurate as the f32 fma chain the f32-input MFMA computes (measured: 0.3-0.7x its error).  The
// dropped al*bl term is <= 2^-22 relative.
//
//   * filters are split ONCE on the host (split16_pack.hpp) after scaling the layer by a power of two 2^e that puts its
//     largest weight in [2^13, 2^14): the lo pieces of all but vanishing weights stay normal f16 numbers.  The epilogue
//     multiplies the accumulators by 2^-e (exact).
//   * activations are split unscaled -- in the consumer's registers, or (r05, p16.hpp) once, in the producer's epilogue: |x| < 65520 is
//     required for hi to be finite.  A launch that meets a value beyond that (its accumulators turn inf / NaN) or produces one (a P16
//     output whose hi piece is not finite) raises redo[0] and redo[1 + image]; behind the pass the float32 kernels of ALL launches run once
//     more, gated by those flags (ConvArgs::redo_check): a flagged image is recomputed from the first layer on -- bit-identical to a
//     split16 = 0 run of it -- and every other image keeps its bits (exec.hip: run_forward; r04 recomputed tiles per layer, which tensors
//     holding (hi, lo) pairs no longer allow).  So the path is safe for ANY finite f32 input and deterministic per image whatever else is
//     in the batch.  Finite-but-huge inputs (6.5e4 > |x| >> 255) are covered by the 2^-22 relative bound of the split, not by the fallback.
#pragma once
#include "conv_igemm.hpp"

namespace dcscn {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) -> packed hi pair, packed lo pair: v_cvt_pk_f16_f32, v_fma_mixlo_f16, v_fma_mixhi_f16 (3 VALU per two values).
// m1 is -1.0f in a register the compiler cannot see through (it would turn fma(h, -1, x) into a subtract of converted
// values: cvt + cvt + sub + cvt per value); the packed hi goes through an empty asm so that the mix instructions read ITS
// halves instead of converting x a second time.
__device__ __forceinline__ void split2(float x0, float x1, float m1, unsigned& hi, unsigned& lo) {
    h2 p = {(_Float16)x0, (_Float16)x1};
    unsigned pu = __builtin_bit_cast(unsigned, p);
    asm volatile("" : "+v"(pu));
    p = __builtin_bit_cast(h2, pu);
    const h2 l = {(_Float16)__builtin_fmaf((float)p[0], m1, x0), (_Float16)__builtin_fmaf((float)p[1], m1, x1)};
    hi = pu;
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split4(const f32x4 x, float m1, h4& hi, h4& lo) {
    u32x2 h, l;
    unsigned a, b;
    split2(x.x, x.y, m1, a, b); h.x = a; l.x = b;
    split2(x.z, x.w, m1, a, b); h.y = a; l.y = b;
    hi = __builtin_bit_cast(h4, h);
    lo = __builtin_bit_cast(h4, l);
}

__device__ __forceinline__ void split8(const f32x4 x, const f32x4 y, float m1, h8& hi, h8& lo) {
    u32x4 h, l;
    unsigned a, b;
    split2(x.x, x.y, m1, a, b); h.x = a; l.x = b;
    split2(x.z, x.w, m1, a, b); h.y = a; l.y = b;
    split2(y.x, y.y, m1, a, b); h.z = a; l.z = b;
    split2(y.z, y.w, m1, a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(h8, h);
    lo = __builtin_bit_cast(h8, l);
}

__device__ __forceinline__ float opaque_minus_one() {
    float m1 = -1.0f;
    asm volatile("" : "+s"(m1));                // a scalar register: the fma takes it as its one SGPR operand
    return m1;
}

__device__ __forceinline__ float opaque_zero() {
    float z = 0.0f;
    asm volatile("" : "+v"(z));                 // keeps the compiler from folding v * 0 to 0
    return z;
}

// chk stays 0 while every v is finite and turns NaN otherwise (inf * 0, NaN * 0); z = opaque_zero()
__device__ __forceinline__ float nonfinite_acc(float chk, const f32x4 v, float z) {
    chk = __builtin_fmaf(v.x, z, chk);
    chk = __builtin_fmaf(v.y, z, chk);
    chk = __builtin_fmaf(v.z, z, chk);
    chk = __builtin_fmaf(v.w, z, chk);
    return chk;
}

}  // namespace dcscn
