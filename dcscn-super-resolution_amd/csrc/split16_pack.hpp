// Host side of split16.hpp: filters as f16 (hi, lo) pairs in the exact LDS image of the kernel that consumes them.
// Shared by the plan (pack.hip) and the stand-alone harness (tools/h16_tune.hip).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dcscn {

// IEEE binary16 round-to-nearest-even of a float (what v_cvt_f16_f32 does); subnormals kept, overflow to infinity
inline uint16_t f16_bits_rn(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // NaN / inf
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                  // >= 65520 rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                                               // <= 2^-25 rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }                  // subnormal result: value = m * 2^(e-23), unit 2^-24
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    const uint32_t half = 1u << (shift - 1);
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    return (uint16_t)(sign | (base + r));                               // a carry out of the mantissa bumps the exponent
}

inline float f16_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    uint32_t b;
    std::memcpy(&b, &v, 4);
    b |= sign;
    std::memcpy(&v, &b, 4);
    return v;
}

inline void split16_host(float x, uint16_t* hi, uint16_t* lo) {
    *hi = f16_bits_rn(x);
    *lo = f16_bits_rn(x - f16_bits_to_float(*hi));
}

// exponent e with max|w| * 2^e in [2^13, 2^14) (0 for an all-zero filter)
inline int split16_scale_exp(const float* w, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
    if (!(m > 0.0f) || !std::isfinite(m)) return 0;
    int ex;
    std::frexp(m, &ex);                                                 // m = f * 2^ex, f in [0.5, 1)
    return 14 - ex;
}

// conv_nin_h image: dense [k_rows][cols] (k_rows = n_chunks * 16 physical input channels, cols = n_groups * nt * 16 padded
// output channels) -> [group][chunk][n][lane = kq * 16 + i][hi0..3, lo0..3] halfs: the A fragments of
// v_mfma_f32_16x16x16_f16 (row i = output channel, k = 4 kq + t), one 16-byte LDS read per lane and tile.
inline std::vector<uint16_t> pack_nin16(const std::vector<float>& dense, int k_rows, int cols, int n_groups, int nt, int n_chunks, int scale_exp) {
    std::vector<uint16_t> out((size_t)n_groups * n_chunks * nt * 64 * 8, 0);
    for (int g = 0; g < n_groups; ++g)
        for (int c = 0; c < n_chunks; ++c)
            for (int n = 0; n < nt; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, kq = lane >> 4;
                    uint16_t* dst = &out[((((size_t)g * n_chunks + c) * nt + n) * 64 + lane) * 8];
                    for (int t = 0; t < 4; ++t) {
                        const int k = c * 16 + 4 * kq + t, col = (g * nt + n) * 16 + i;
                        const float w = k < k_rows ? dense[(size_t)k * cols + col] : 0.0f;
                        split16_host(std::ldexp(w, scale_exp), &dst[t], &dst[4 + t]);
                    }
                }
    return out;
}

// conv3_h image: dense [tap][k_rows][cols] -> [group][chunk of 32 channels][tap][n][part: 0 hi, 1 lo][lane = kq * 16 + i][8 halfs]:
// the A fragments of v_mfma_f32_16x16x32_f16 (row i = output channel, k = 8 kq + t).
// tail_octs = 1 / 2 / 3 (conv3_h's packed last chunk, taps == 9): the last chunk holds at most 8 / 16 / 24 real channels, so its
// (tap, octet) pairs -- pair p = tap * octs + octet -- are packed four to an instruction: tap slot s < ceil(9 octs / 4) holds, for
// lane group kq, pair 4 s + kq (zeros past the last pair); the other slots of that chunk stay zero and are never fetched.
inline std::vector<uint16_t> pack_conv16(const std::vector<float>& dense, int taps, int k_rows, int cols, int n_groups, int nt, int n_chunks, int scale_exp,
                                         int tail_octs = 0) {
    std::vector<uint16_t> out((size_t)n_groups * n_chunks * taps * nt * 2 * 64 * 8, 0);
    for (int g = 0; g < n_groups; ++g)
        for (int c = 0; c < n_chunks; ++c) {
            const bool packed = tail_octs > 0 && c == n_chunks - 1;
            const int slots = packed ? (taps * tail_octs + 3) / 4 : taps;
            for (int slot = 0; slot < slots; ++slot)
                for (int n = 0; n < nt; ++n)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 15, kq = lane >> 4;
                        uint16_t* hi = &out[((((((size_t)g * n_chunks + c) * taps + slot) * nt + n) * 2 + 0) * 64 + lane) * 8];
                        uint16_t* lo = hi + 64 * 8;
                        const int pair = 4 * slot + kq;
                        const int tap = packed ? pair / tail_octs : slot;
                        const int oct = packed ? pair % tail_octs : kq;
                        for (int t = 0; t < 8; ++t) {
                            const int k = c * 32 + 8 * oct + t, col = (g * nt + n) * 16 + i;
                            const float w = (k < k_rows && tap < taps) ? dense[((size_t)tap * k_rows + k) * cols + col] : 0.0f;
                            split16_host(std::ldexp(w, scale_exp), &hi[t], &lo[t]);
                        }
                    }
        }
    return out;
}

}  // namespace dcscn
