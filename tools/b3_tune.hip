// PROTOTYPE (tools only, not part of the library): 3x3 SAME conv with f32 accuracy on the bf16 matrix pipe.
//
// An f32 value is split into three bf16 pieces x = x1 + x2 + x3; a product w * x is taken as the six cross terms
// w3x1 + w1x3 + w2x2 + w2x1 + w1x2 + w1x1 (dropped terms <= 2^-24 relative, tools/bf16x3_numerics.py); every
// bf16 x bf16 product is exact in f32 and v_mfma_f32_16x16x32_bf16 accumulates in f32.  Six MFMAs of 8192 MACs / 16
// cycles replace eight v_mfma_f32_16x16x4_f32 of 1024 MACs / 32 cycles: 2.7x the f32 MAC rate of the chip.
//
// Direct implicit GEMM (no Winograd): D[cout][pixel] += W[cout][k] * X[k][pixel] per filter tap, k = 32 input
// channels.  Activations arrive pre-split as [pixel][32-channel block][piece][32] bf16 (in a real pipeline the
// producing layer's epilogue writes that form), filters are split on the host.  Workgroup = 4 waves, 8 x 16 output
// pixels x NT*16 output channels; per channel block the halo tile sits in LDS, the filter slices of the 9 taps
// stream through a double-buffered LDS region (one barrier per tap).
//
// Checked against conv_igemm (exact f32) on the same data; prints time and "f32-equivalent" TFLOP/s.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv_igemm.hpp"

using namespace dcscn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static int N = 1024, H = 48, W = 48;

// ---- bf16 helpers (round to nearest even), host and device ----
__host__ __device__ inline uint16_t f2bf(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline void split3(float x, uint16_t* p) {
    p[0] = f2bf(x);
    const float r1 = x - bf2f(p[0]);
    p[1] = f2bf(r1);
    p[2] = f2bf(r1 - bf2f(p[1]));
}

// x3[img][y][x][cb][piece][32] bf16 from NHWC f32 channels [in_off, in_off + cin)
__global__ void split_input(const float* in, uint16_t* x3, long long pixels, int in_stride, int in_off, int cin, int cb) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // one thread per (pixel, channel)
    const int cpad = cb * 32;
    if (idx >= pixels * cpad) return;
    const long long px = idx / cpad;
    const int c = (int)(idx - px * cpad);
    const float v = c < cin ? in[px * in_stride + in_off + c] : 0.0f;
    uint16_t p[3];
    split3(v, p);
    uint16_t* d = x3 + (px * cb + c / 32) * 96 + (c % 32);
    d[0] = p[0];
    d[32] = p[1];
    d[64] = p[2];
}

struct B3Args {
    const uint16_t* x3;      // [N][H][W][cb][3][32]
    const uint16_t* w3;      // [group][cb][tap][piece][NT][16][40]  (rows padded to 80 bytes)
    const float* bias;
    const float* alpha;
    float* out;
    int N, H, W, cb, out_stride, out_off, cout, tiles_x, tiles_y;
};

constexpr int kTH = 8, kTW = 16, kHTH = kTH + 2, kHTW = kTW + 2, kHP = kHTH * kHTW;   // 180 halo pixels
constexpr int kPixB = 208;                       // bytes per halo pixel in LDS: 3 pieces x 64 B + 16 B pad
constexpr int kRowB = 80;                        // bytes per filter row (32 bf16 + pad)
constexpr int kXBytes = kHP * kPixB;             // 37440

template <int NT>
constexpr int w_tap_bytes() { return 3 * NT * 16 * kRowB; }

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_b3(const B3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Xs = lds;
    unsigned char* Ws = lds + kXBytes;           // two buffers of w_tap_bytes<NT>()
    constexpr int WB = w_tap_bytes<NT>();
    constexpr int W_VEC = WB / 16;               // 16-byte pieces per tap
    constexpr int W_LOADS = (W_VEC + 255) / 256;
    constexpr int X_VEC = kHP * 12;              // 12 pieces of 16 B per halo pixel (192 B)
    constexpr int X_LOADS = (X_VEC + 255) / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int group = blockIdx.y;
    const int y0 = ty * kTH, x0 = tx * kTW;
    const int H = a.H, W = a.W;

    // staging descriptors of the halo tile
    const uint16_t* x_src[X_LOADS];
    int x_dst[X_LOADS];
    bool x_ok[X_LOADS], x_item[X_LOADS];
    static_for<0, X_LOADS>([&](auto i_) DCSCN_INL {
        constexpr int i = decltype(i_)::value;
        const int item = tid + 256 * i;
        const int hp = item / 12, pc = item - hp * 12;
        const int hy = hp / kHTW, hx = hp - hy * kHTW;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        x_item[i] = item < X_VEC;
        x_ok[i] = x_item[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
        x_dst[i] = hp * kPixB + pc * 16;
        x_src[i] = a.x3 + (((size_t)img * H + (x_ok[i] ? gy : 0)) * W + (x_ok[i] ? gx : 0)) * a.cb * 96 + pc * 8;
    });
    const uint16_t* w_base = a.w3 + (size_t)group * a.cb * 9 * (WB / 2);

    f32x4 acc[2][NT];
    static_for<0, 2>([&](auto m_) DCSCN_INL {
        static_for<0, NT>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0, 0, 0, 0}; });
    });

    u32x4 wreg[W_LOADS];
    auto load_w = [&](int blk, int tap) DCSCN_INL {
        const uint16_t* src = w_base + ((size_t)blk * 9 + tap) * (WB / 2);
        static_for<0, W_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (tid + 256 * i < W_VEC) wreg[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(tid + 256 * i) * 8);
        });
    };
    auto store_w = [&](int buf) DCSCN_INL {
        static_for<0, W_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (tid + 256 * i < W_VEC) *reinterpret_cast<u32x4*>(Ws + buf * WB + (tid + 256 * i) * 16) = wreg[i];
        });
    };

    const int row0 = 2 * wave;                   // this wave's two tile rows
    for (int blk = 0; blk < a.cb; ++blk) {
        // halo tile of this channel block -> LDS (zero outside the image = SAME padding)
        u32x4 xreg[X_LOADS];
        static_for<0, X_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            xreg[i] = u32x4{0, 0, 0, 0};
            if (x_ok[i]) xreg[i] = *reinterpret_cast<const u32x4*>(x_src[i] + (size_t)blk * 96);
        });
        load_w(blk, 0);
        __syncthreads();                         // every wave is done with the previous block's tile and filters
        static_for<0, X_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (x_item[i]) *reinterpret_cast<u32x4*>(Xs + x_dst[i]) = xreg[i];
        });
        store_w(0);
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_w(blk, tap + 1);
            const int dy = tap / 3, dx = tap - 3 * dy;
            const unsigned char* wb = Ws + (tap & 1) * WB + col * kRowB + kg * 16;
            // X operands: pieces of the two rows
            bf16x8 xo[2][3];
            static_for<0, 2>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                const unsigned char* xp = Xs + ((row0 + m + dy) * kHTW + col + dx) * kPixB + kg * 16;
                static_for<0, 3>([&](auto p_) DCSCN_INL {
                    constexpr int p = decltype(p_)::value;
                    xo[m][p] = *reinterpret_cast<const bf16x8*>(xp + p * 64);
                });
            });
            static_for<0, NT>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                bf16x8 wo[3];
                static_for<0, 3>([&](auto p_) DCSCN_INL {
                    constexpr int p = decltype(p_)::value;
                    wo[p] = *reinterpret_cast<const bf16x8*>(wb + (p * NT + n) * 16 * kRowB);
                });
                static_for<0, 2>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    f32x4 c = acc[m][n];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[2], xo[m][0], c, 0, 0, 0);   // small terms first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[0], xo[m][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[1], xo[m][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[1], xo[m][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[0], xo[m][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[0], xo[m][0], c, 0, 0, 0);
                    acc[m][n] = c;
                });
            });
            if (tap + 1 < 9) store_w((tap + 1) & 1);
            __syncthreads();
        }
    }

    // epilogue: bias, PReLU, NHWC f32 store (lane: 4 consecutive output channels of one pixel)
    const int gx = x0 + col;
    if (gx >= W) return;
    static_for<0, NT>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        const int c = group * NT * 16 + n * 16 + 4 * kg;
        if (c < a.cout) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c);
            const f32x4 av = *reinterpret_cast<const f32x4*>(a.alpha + c);
            static_for<0, 2>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                const int gy = y0 + row0 + m;
                if (gy < H) {
                    f32x4 v = acc[m][n] + bv;
                    v.x = v.x > 0 ? v.x : av.x * v.x;
                    v.y = v.y > 0 ? v.y : av.y * v.y;
                    v.z = v.z > 0 ? v.z : av.z * v.z;
                    v.w = v.w > 0 ? v.w : av.w * v.w;
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)img * H + gy) * W + gx) * a.out_stride + a.out_off + c) = v;
                }
            });
        }
    });
}

// ---- second form: v_mfma_f32_32x32x16_bf16, 2 x 2 register tiles (64 pixels x 64 output channels per wave) ----
// The 16x16x32 form above needs 427 B of LDS operand reads per 16-cycle MFMA; a 32x32x16 MFMA does twice the MACs
// per operand byte, and a 2x2 tile block brings it to 512 B per 32-cycle MFMA (64 B/clk per CU).  Workgroup = 4 waves
// = 16 x 16 output pixels x 64 output channels; wave w owns tile rows 4w .. 4w+3.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTH2 = 16, kHP2 = (kTH2 + 2) * kHTW;          // 324 halo pixels
constexpr int kXBytes2 = kHP2 * kPixB;                      // 67392
constexpr int kWTap2 = 3 * 2 * 32 * kRowB;                  // 15360: [piece][32-channel tile][32 couts][80 B]

__global__ __launch_bounds__(256, 1) void conv_b3w(const B3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* Xs = lds;
    unsigned char* Ws = lds + kXBytes2;
    constexpr int W_VEC = kWTap2 / 16, W_LOADS = (W_VEC + 255) / 256;
    constexpr int X_VEC = kHP2 * 12, X_LOADS = (X_VEC + 255) / 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kgl = lane >> 5;
    const int pr = li >> 4, pc = li & 15;        // pixel of this lane inside a 2 x 16 M-tile

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int group = blockIdx.y;
    const int y0 = ty * kTH2, x0 = tx * kTW;
    const int H = a.H, W = a.W;

    const uint16_t* w_base = a.w3 + (size_t)group * a.cb * 9 * (kWTap2 / 2);
    f32x16 acc[2][2];
    static_for<0, 2>([&](auto m_) DCSCN_INL {
        static_for<0, 2>([&](auto n_) DCSCN_INL {
            static_for<0, 16>([&](auto r_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value][decltype(r_)::value] = 0.0f; });
        });
    });
    u32x4 wreg[W_LOADS];
    auto load_w = [&](int blk, int tap) DCSCN_INL {
        const uint16_t* src = w_base + ((size_t)blk * 9 + tap) * (kWTap2 / 2);
        static_for<0, W_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (tid + 256 * i < W_VEC) wreg[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(tid + 256 * i) * 8);
        });
    };
    auto store_w = [&](int buf) DCSCN_INL {
        static_for<0, W_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (tid + 256 * i < W_VEC) *reinterpret_cast<u32x4*>(Ws + buf * kWTap2 + (tid + 256 * i) * 16) = wreg[i];
        });
    };

    for (int blk = 0; blk < a.cb; ++blk) {
        load_w(blk, 0);
        __syncthreads();
        // halo tile: 324 pixels x 192 B, zero outside the image (loaded in two halves to bound the staging registers)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            u32x4 xreg[(X_LOADS + 1) / 2];
            static_for<0, (X_LOADS + 1) / 2>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                const int item = tid + 256 * (i + half * ((X_LOADS + 1) / 2));
                xreg[i] = u32x4{0, 0, 0, 0};
                if (item < X_VEC) {
                    const int hp = item / 12, pcs = item - hp * 12;
                    const int hy = hp / kHTW, hx = hp - hy * kHTW;
                    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                    if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                        xreg[i] = *reinterpret_cast<const u32x4*>(a.x3 + ((((size_t)img * H + gy) * W + gx) * a.cb + blk) * 96 + pcs * 8);
                }
            });
            static_for<0, (X_LOADS + 1) / 2>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                const int item = tid + 256 * (i + half * ((X_LOADS + 1) / 2));
                if (item < X_VEC) {
                    const int hp = item / 12, pcs = item - hp * 12;
                    *reinterpret_cast<u32x4*>(Xs + hp * kPixB + pcs * 16) = xreg[i];
                }
            });
        }
        store_w(0);
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_w(blk, tap + 1);
            const int dy = tap / 3, dx = tap - 3 * dy;
            const unsigned char* wb = Ws + (tap & 1) * kWTap2 + li * kRowB + kgl * 16;
            static_for<0, 2>([&](auto h_) DCSCN_INL {                       // two 16-channel halves of the block
                constexpr int h = decltype(h_)::value;
                bf16x8 xo[2][3], wo[2][3];
                static_for<0, 2>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const unsigned char* xp = Xs + ((4 * wave + 2 * m + pr + dy) * kHTW + pc + dx) * kPixB + h * 32 + kgl * 16;
                    static_for<0, 3>([&](auto p_) DCSCN_INL { xo[m][decltype(p_)::value] = *reinterpret_cast<const bf16x8*>(xp + decltype(p_)::value * 64); });
                });
                static_for<0, 2>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    static_for<0, 3>([&](auto p_) DCSCN_INL {
                        constexpr int p = decltype(p_)::value;
                        wo[n][p] = *reinterpret_cast<const bf16x8*>(wb + (p * 2 + n) * 32 * kRowB + h * 32);
                    });
                });
                static_for<0, 2>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    static_for<0, 2>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        f32x16 c = acc[m][n];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][2], xo[m][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][0], xo[m][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][1], xo[m][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][1], xo[m][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][0], xo[m][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[n][0], xo[m][0], c, 0, 0, 0);
                        acc[m][n] = c;
                    });
                });
            });
            if (tap + 1 < 9) store_w((tap + 1) & 1);
            __syncthreads();
        }
    }

    // epilogue: C/D layout of 32x32: column = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    const int gx = x0 + pc;
    if (gx >= W) return;
    static_for<0, 2>([&](auto m_) DCSCN_INL {
        constexpr int m = decltype(m_)::value;
        const int gy = y0 + 4 * wave + 2 * m + pr;
        if (gy < H) {
            float* o = a.out + (((size_t)img * H + gy) * W + gx) * a.out_stride + a.out_off;
            static_for<0, 2>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                static_for<0, 4>([&](auto q_) DCSCN_INL {
                    constexpr int q = decltype(q_)::value;
                    const int c = group * 64 + n * 32 + 8 * q + 4 * kgl;
                    if (c < a.cout) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c);
                        const f32x4 av = *reinterpret_cast<const f32x4*>(a.alpha + c);
                        f32x4 v = {acc[m][n][4 * q] + bv.x, acc[m][n][4 * q + 1] + bv.y, acc[m][n][4 * q + 2] + bv.z, acc[m][n][4 * q + 3] + bv.w};
                        v.x = v.x > 0 ? v.x : av.x * v.x;
                        v.y = v.y > 0 ? v.y : av.y * v.y;
                        v.z = v.z > 0 ? v.z : av.z * v.z;
                        v.w = v.w > 0 ? v.w : av.w * v.w;
                        *reinterpret_cast<f32x4*>(o + c) = v;
                    }
                });
            });
        }
    });
}

struct Layer { const char* name; int cin, cout, in_stride, in_off, out_stride, out_off; };

static std::vector<float> rand_vec(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
    return h;
}

static std::vector<float> pack_direct(const std::vector<float>& w, int cin, int cout, int cin_phys, int kc, int nt, int* n_chunks) {
    const int ns = conv_ns(nt);
    *n_chunks = (cin_phys + kc - 1) / kc;
    std::vector<float> p((size_t)*n_chunks * 9 * kc * ns, 0.0f);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < cin; ++c)
            for (int o = 0; o < cout; ++o)
                p[((size_t)(c / kc) * 9 + t) * kc * ns + (size_t)(c % kc) * ns + o] = w[((size_t)t * cin + c) * cout + o];
    return p;
}

static float *g_in, *g_ref, *g_out, *g_w, *g_bias;
static uint16_t *g_x3, *g_w3;

template <int NTD, int NT>
void run(const Layer& L) {
    const int cin_phys = (L.cin + 3) & ~3;
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777, 0.2f);
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;
    // ---- exact f32 reference: conv_igemm ----
    {
        constexpr int MT = 2;
        ConvArgs a{};
        a.in = g_in; a.in_stride = L.in_stride; a.in_off = L.in_off; a.cin_phys = cin_phys;
        a.bias = g_bias; a.alpha = g_bias; a.act = ACT_ALPHA;
        a.N = N; a.H = H; a.W = W;
        a.split = 1 << 30; a.ps = 1; a.ps_c = 1; a.vec4 = 1; a.res = nullptr; a.res_stride = 1;
        using Gd = ConvGeom<3, MT, NTD, 4>;
        int nch;
        std::vector<float> p = pack_direct(w, L.cin, L.cout, cin_phys, 4, NTD, &nch);
        CK(hipMemcpy(g_w, p.data(), p.size() * sizeof(float), hipMemcpyHostToDevice));
        a.wpack = g_w; a.n_chunks = nch;
        a.tiles_x = (W + 15) / 16; a.tiles_y = (H + Gd::TH - 1) / Gd::TH;
        a.out0 = OutDesc{g_ref, L.out_stride, L.out_off, (L.cout + 3) & ~3};
        a.out1 = a.out0;
        auto kern = conv_igemm<3, MT, NTD, 4, false, 3>;
        const size_t lds = (size_t)Gd::BUF * sizeof(float);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(N * a.tiles_y * a.tiles_x, 1), dim3(256), lds, 0, a);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(N * a.tiles_y * a.tiles_x, 1), dim3(256), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-8s %4d->%-4d f32 direct conv_igemm NT%-2d        %8.3f ms  %7.2f TFLOP/s\n", L.name, L.cin, L.cout, NTD, ms, flop / (ms * 1e-3) / 1e12);
    }
    // ---- bf16x3 ----
    const int cb = (L.cin + 31) / 32;
    const long long pixels = (long long)N * H * W;
    {
        const long long total = pixels * cb * 32;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(split_input, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, g_in, g_x3, pixels, L.in_stride, L.in_off, L.cin, cb);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("         split_input (stand-in for a producer epilogue writing bf16 triplets): %.3f ms\n", ms);
    }
    const int tiles16 = (L.cout + 15) / 16;
    const int groups = (tiles16 + NT - 1) / NT;
    constexpr int WB = w_tap_bytes<NT>();
    std::vector<uint16_t> w3((size_t)groups * cb * 9 * (WB / 2), 0);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < L.cin; ++c)
            for (int o = 0; o < L.cout; ++o) {
                uint16_t p[3];
                split3(w[((size_t)t * L.cin + c) * L.cout + o], p);
                const int g = o / (NT * 16), n = (o % (NT * 16)) / 16, r = o % 16;
                for (int q = 0; q < 3; ++q)
                    w3[(((size_t)g * cb + c / 32) * 9 + t) * (WB / 2) + ((size_t)(q * NT + n) * 16 + r) * (kRowB / 2) + c % 32] = p[q];
            }
    CK(hipMemcpy(g_w3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
    B3Args a{};
    a.x3 = g_x3; a.w3 = g_w3; a.bias = g_bias; a.alpha = g_bias; a.out = g_out;
    a.N = N; a.H = H; a.W = W; a.cb = cb; a.out_stride = L.out_stride; a.out_off = L.out_off; a.cout = (L.cout + 3) & ~3;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH - 1) / kTH;
    auto kern = conv_b3<NT>;
    const size_t lds = kXBytes + 2 * WB;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
    CK(hipMemset(g_out, 0, (size_t)N * H * W * L.out_stride * sizeof(float)));
    const dim3 grid(N * a.tiles_y * a.tiles_x, groups);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 5; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const size_t cnt = (size_t)8 * H * W * L.out_stride;
    std::vector<float> r(cnt), o(cnt);
    CK(hipMemcpy(r.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t px = 0; px < (size_t)8 * H * W; ++px)
        for (int c = 0; c < L.cout; ++c) {
            const size_t i = px * L.out_stride + L.out_off + c;
            maxd = std::fmax(maxd, std::fabs((double)r[i] - o[i]));
            maxv = std::fmax(maxv, std::fabs((double)r[i]));
        }
    printf("%-8s %4d->%-4d bf16x3 direct NT%d groups%d cb%d lds %5.1f KB occ %d  %8.3f ms  %7.2f TFLOP/s (f32-equivalent)  max|diff vs f32| %.3g (max|ref| %.3g)\n",
           L.name, L.cin, L.cout, NT, groups, cb, lds / 1024.0, occ, best, flop / (best * 1e-3) / 1e12, maxd, maxv);
    fflush(stdout);
}

void run_wide(const Layer& L) {
    std::vector<float> w = rand_vec((size_t)9 * L.cin * L.cout, 777, 0.2f);
    const double flop = 2.0 * 9 * L.cin * (double)L.cout * N * H * W;
    const int cb = (L.cin + 31) / 32;
    const int groups = (L.cout + 63) / 64;
    std::vector<uint16_t> w3((size_t)groups * cb * 9 * (kWTap2 / 2), 0);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < L.cin; ++c)
            for (int o = 0; o < L.cout; ++o) {
                uint16_t p[3];
                split3(w[((size_t)t * L.cin + c) * L.cout + o], p);
                const int g = o / 64, n = (o % 64) / 32, r = o % 32;
                for (int q = 0; q < 3; ++q)
                    w3[(((size_t)g * cb + c / 32) * 9 + t) * (kWTap2 / 2) + ((size_t)(q * 2 + n) * 32 + r) * (kRowB / 2) + c % 32] = p[q];
            }
    CK(hipMemcpy(g_w3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
    B3Args a{};
    a.x3 = g_x3; a.w3 = g_w3; a.bias = g_bias; a.alpha = g_bias; a.out = g_out;
    a.N = N; a.H = H; a.W = W; a.cb = cb; a.out_stride = L.out_stride; a.out_off = L.out_off; a.cout = (L.cout + 3) & ~3;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH2 - 1) / kTH2;
    auto kern = conv_b3w;
    const size_t lds = kXBytes2 + 2 * kWTap2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
    CK(hipMemset(g_out, 0, (size_t)N * H * W * L.out_stride * sizeof(float)));
    const dim3 grid(N * a.tiles_y * a.tiles_x, groups);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 5; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    const size_t cnt = (size_t)8 * H * W * L.out_stride;
    std::vector<float> r(cnt), o(cnt);
    CK(hipMemcpy(r.data(), g_ref, cnt * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(o.data(), g_out, cnt * sizeof(float), hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t px = 0; px < (size_t)8 * H * W; ++px)
        for (int c = 0; c < L.cout; ++c) {
            const size_t i = px * L.out_stride + L.out_off + c;
            maxd = std::fmax(maxd, std::fabs((double)r[i] - o[i]));
            maxv = std::fmax(maxv, std::fabs((double)r[i]));
        }
    printf("%-8s %4d->%-4d bf16x3 32x32x16 2x2 tiles, groups%d cb%d lds %5.1f KB occ %d  %8.3f ms  %7.2f TFLOP/s (f32-equivalent)  max|diff vs f32| %.3g (max|ref| %.3g)\n",
           L.name, L.cin, L.cout, groups, cb, lds / 1024.0, occ, best, flop / (best * 1e-3) / 1e12, maxd, maxv);
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) N = atoi(argv[1]);
    const size_t act = (size_t)N * H * W * 1316;
    CK(hipMalloc(&g_in, act * sizeof(float)));
    CK(hipMalloc(&g_ref, act * sizeof(float)));
    CK(hipMalloc(&g_out, act * sizeof(float)));
    CK(hipMalloc(&g_w, (size_t)(32u << 20) * sizeof(float)));
    CK(hipMalloc(&g_bias, 4096 * sizeof(float)));
    CK(hipMalloc(&g_x3, (size_t)N * H * W * 7 * 96 * 2));
    CK(hipMalloc(&g_w3, (size_t)64 << 20));
    {
        std::vector<float> h = rand_vec(16u << 20, 4242, 100.0f);
        for (size_t off = 0; off < act; off += h.size())
            CK(hipMemcpy(g_in + off, h.data(), std::min(h.size(), act - off) * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float> b = rand_vec(4096, 99, 0.5f);
        for (auto& v : b) v = std::fabs(v) * 0.5f;
        CK(hipMemcpy(g_bias, b.data(), 4096 * sizeof(float), hipMemcpyHostToDevice));
    }
    CK(hipMemset(g_ref, 0, act * sizeof(float)));
    const Layer cnn2{"CNN2", 196, 166, 1316, 0, 1316, 196};
    const Layer cnn5{"CNN5", 133, 120, 1316, 512, 1316, 648};
    run<11, 3>(cnn2);
    run_wide(cnn2);
    run<8, 4>(cnn5);
    run_wide(cnn5);
    return 0;
}
