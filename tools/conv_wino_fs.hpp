// EXPERIMENT (tuner only, not part of the library; measured equal to conv_wino: CNN2 7.11 vs 7.08 ms, CNN5 3.86 vs 3.85)
// conv_wino_fs: Winograd F(2x2,3x3) with the 16 frequencies SPLIT OVER THE FOUR WAVES of a workgroup.
//
// conv_wino (conv_wino.hpp) gives every wave all 16 frequencies of 16 tiles; each of its MFMAs then needs its
// own filter operand from LDS: one ds_read_b32 per MFMA.  Measured on MI355X the matrix pipe pays for every
// instruction that sits between two MFMAs (~6 cycles each for a single wave, tuner log), so the instruction
// count per MFMA -- not LDS bandwidth, not memory -- is what holds that kernel at 55-58 % of the pipe.
//
// Here wave w owns frequency row xi = w (frequencies 4w .. 4w+3) for ALL 64 tiles of the 16x16-pixel workgroup
// tile (4 blocks of 16 tiles) and NT*16 output channels:
//   * a filter operand (frequency, channel tile) is read once and feeds 4 MFMAs (the 4 tile blocks):
//     12 filter reads per 48 MFMAs instead of 48;
//   * the input transform of a tile for row xi needs only 2 of the 4 raw patch rows (B^T has two non-zeros per
//     row): r = d[ra] +- d[rb] (4 values), v[nu] = B-column combinations of r (4 values): 8 VALU and 4
//     ds_read_b64 per (tile, channel) -- 32 VALU / 16 reads per chunk, the same VALU count as before;
//   * accumulators: 4 frequencies x 4 blocks x NT tiles = 16 NT, as before (192 VGPRs at NT = 3);
//   * the output transform needs all four xi of a tile: every wave first applies the column half
//     z[b'] = sum_nu A^T[b'][nu] m[xi][nu] (2 values instead of 4), the waves exchange z through LDS (32 KB, one
//     channel tile at a time, reusing the staging buffers) and wave w finishes the tiles of block w:
//     y[a][b'] = sum_xi A^T[a][xi] z_xi[b'].  Same products, same filters; only the association of the final
//     adds differs from conv_wino.
// Staging (input halo tile + filter block of a 4-channel chunk in one LDS buffer, register prefetch of the next
// chunk, two barriers per chunk) is the scheme of conv_wino.
#pragma once
#include "conv_wino_lab.hpp"

namespace dcscn_lab {
using namespace dcscn;

constexpr int kWinoFsZFloats = 4 * 4 * 2 * 64 * 4;   // exchange buffer of the epilogue: [wave][block][2][lane] float4

template <int NT>
constexpr size_t wino_fs_lds_bytes() {
    using G = WinoGeom<NT, kWinoKC, 4>;
    return sizeof(float) * (size_t)(G::BUF > kWinoFsZFloats ? G::BUF : kWinoFsZFloats);
}

template <int NT, int NTV>
__device__ __forceinline__ void conv_wino_fs_body(const ConvArgs& a, float* smem) {
    constexpr int KC = kWinoKC;
    using G = WinoGeom<NT, KC, 4>;
    static_assert(G::KQ == 1 && !G::SCATTER, "one MFMA k-step per chunk, linear filter image");
    constexpr int THREADS = 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int ntile = blockIdx.y;
    const int y0 = ty * G::TH;
    const int x0 = tx * G::TW;
    const int H = a.H, W = a.W;
    const float* in_img = a.in + (size_t)img * H * W * a.in_stride + a.in_off;

    // ---- staging descriptors (as conv_wino) ----
    const float* a_src[G::A_LOADS];
    int a_dst[G::A_LOADS];
    bool a_item[G::A_LOADS];
    unsigned a_mask[G::A_LOADS];
    static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
        constexpr int i = decltype(i_)::value;
        const int hp = tid + THREADS * i;
        const int hy = hp / G::HTW;
        const int hx = hp - hy * G::HTW;
        const int gy = y0 + hy - 1;
        const int gx = x0 + hx - 1;
        a_item[i] = hp < G::A_ITEMS;
        const bool inb = a_item[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
        a_mask[i] = inb ? 0xffffffffu : 0u;            // SAME zero padding, applied when the tile is written to LDS
        a_dst[i] = hp;
        a_src[i] = in_img + ((size_t)(inb ? gy : 0) * W + (inb ? gx : 0)) * a.in_stride;
    });
    const float* b_src = a.wpack + (size_t)ntile * a.n_chunks * G::GB_FLOATS + 4 * tid;
    const int c_last = a.cin_phys - 4;

    f32x4 areg[G::A_LOADS];
    f32x4 breg[G::B_LOADS];
    auto load_chunk = [&](int chunk) DCSCN_INL {
        const int c = chunk * KC < c_last ? chunk * KC : c_last;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            areg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + c);
        });
        const float* bs = b_src + (size_t)chunk * G::GB_FLOATS;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            breg[i] = *reinterpret_cast<const f32x4*>(bs + 4 * THREADS * i);   // filter buffer has one sweep of slack
        });
    };
    auto store_chunk = [&](float* buf) DCSCN_INL {
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (a_item[i]) {
                float* d = buf + a_dst[i];
                d[0] = __uint_as_float(__float_as_uint(areg[i].x) & a_mask[i]);
                d[G::PS] = __uint_as_float(__float_as_uint(areg[i].y) & a_mask[i]);
                d[2 * G::PS] = __uint_as_float(__float_as_uint(areg[i].z) & a_mask[i]);
                d[3 * G::PS] = __uint_as_float(__float_as_uint(areg[i].w) & a_mask[i]);
            }
        });
        float* bd = buf + G::A_FLOATS + 4 * tid;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (i < G::B_LOADS - 1 || tid + THREADS * i < G::B_VEC)
                *reinterpret_cast<f32x4*>(bd + 4 * THREADS * i) = breg[i];
        });
    };

    // ---- this wave's frequency row xi = wave:  (B^T d)[xi] = d[ra] + sgn * d[rb] ----
    //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    const int xi = wave;
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int rb = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
    const float sgn = xi == 1 ? 1.0f : -1.0f;
    // lane (lj, lk): tile (row 2b + (lj >> 3), column lj & 7) of block b, channel lk of the chunk
    const int t_lane = lk * G::PS + (2 * (lj >> 3)) * G::HTW + 2 * (lj & 7);
    const int a_ra = t_lane + ra * G::HTW;
    const int a_rb = t_lane + rb * G::HTW;
    const int b_lane = G::A_FLOATS + (4 * xi * KC + lk) * G::NS + lj;

    f32x4 acc[4][4][NTV];                                  // [nu][block][channel tile]
    static_for<0, 4>([&](auto u_) DCSCN_INL {
        static_for<0, 4>([&](auto b_) DCSCN_INL {
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                acc[decltype(u_)::value][decltype(b_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            });
        });
    });

    auto read_w = [&](const float* Bs, auto nu_, float (&w)[NTV]) DCSCN_INL {
        constexpr int nu = decltype(nu_)::value;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            w[n] = Bs[(nu * KC) * G::NS + n * 16];
        });
    };
    auto compute = [&](const float* buf) DCSCN_INL {
        const float* Bs = buf + b_lane;
        float wq[2][NTV];
        read_w(Bs, std::integral_constant<int, 0>{}, wq[0]);
        float v[4][4];                                     // [block][nu]
        static_for<0, 4>([&](auto b_) DCSCN_INL {
            constexpr int b = decltype(b_)::value;
            const float* pa = buf + a_ra + b * 4 * G::HTW;
            const float* pb = buf + a_rb + b * 4 * G::HTW;
            const float2 a01 = *reinterpret_cast<const float2*>(pa), a23 = *reinterpret_cast<const float2*>(pa + 2);
            const float2 b01 = *reinterpret_cast<const float2*>(pb), b23 = *reinterpret_cast<const float2*>(pb + 2);
            const float r0 = fmaf(sgn, b01.x, a01.x), r1 = fmaf(sgn, b01.y, a01.y);   // sgn = +-1: exact product, one rounding
            const float r2 = fmaf(sgn, b23.x, a23.x), r3 = fmaf(sgn, b23.y, a23.y);
            v[b][0] = r0 - r2;
            v[b][1] = r1 + r2;
            v[b][2] = r2 - r1;
            v[b][3] = r1 - r3;
        });
        static_for<0, 4>([&](auto u_) DCSCN_INL {
            constexpr int nu = decltype(u_)::value;
            if constexpr (nu + 1 < 4) read_w(Bs, std::integral_constant<int, nu + 1>{}, wq[(nu + 1) & 1]);
            static_for<0, 4>([&](auto b_) DCSCN_INL {
                constexpr int b = decltype(b_)::value;
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[nu][b][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[nu & 1][n], v[b][nu], acc[nu][b][n], 0, 0, 0);
                });
            });
        });
    };

    // ---- K loop ----
    load_chunk(0);
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        store_chunk(smem);
        if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
        __syncthreads();
        compute(smem);
        __syncthreads();
    }

    // ---- output transform across the waves + epilogue ----
    const int tr = 2 * wave + (lj >> 3);               // after the exchange wave w finishes block w
    const int tc = lj & 7;
    const int gy0 = y0 + 2 * tr;
    const int gx0 = x0 + 2 * tc;
    const int cbase = ntile * NT * 16 + 4 * lk;
    const int act = a.act;
    const int ps = a.ps;
    const int orow = W * ps;
    const bool ok_y1 = gy0 + 1 < H, ok_x1 = gx0 + 1 < W;
    const bool ok_00 = gy0 < H && gx0 < W;
    f32x4 bv[NTV], av[NTV];
    static_for<0, NTV>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        bv[n] = *reinterpret_cast<const f32x4*>(a.bias + cbase + n * 16);
        av[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (act == ACT_ALPHA) av[n] = *reinterpret_cast<const f32x4*>(a.alpha + cbase + n * 16);
    });
    f32x4* zbuf = reinterpret_cast<f32x4*>(smem);
    auto finish = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            // column half of A^T m A for this wave's xi: z[b'] = sum_nu A^T[b'][nu] m[xi][nu], A^T = [1 1 1 0; 0 1 -1 -1]
            static_for<0, 4>([&](auto b_) DCSCN_INL {
                constexpr int b = decltype(b_)::value;
                const f32x4 z0 = acc[0][b][n] + acc[1][b][n] + acc[2][b][n];
                const f32x4 z1 = acc[1][b][n] - acc[2][b][n] - acc[3][b][n];
                zbuf[((wave * 4 + b) * 2 + 0) * 64 + lane] = z0;
                zbuf[((wave * 4 + b) * 2 + 1) * 64 + lane] = z1;
            });
            __syncthreads();
            // row half for the tiles of block `wave`: y[a][b'] = sum_xi A^T[a][xi] z_xi[b']
            f32x4 yv[2][2];
            static_for<0, 2>([&](auto pb_) DCSCN_INL {
                constexpr int pb = decltype(pb_)::value;
                const f32x4 q0 = zbuf[((0 * 4 + wave) * 2 + pb) * 64 + lane];
                const f32x4 q1 = zbuf[((1 * 4 + wave) * 2 + pb) * 64 + lane];
                const f32x4 q2 = zbuf[((2 * 4 + wave) * 2 + pb) * 64 + lane];
                const f32x4 q3 = zbuf[((3 * 4 + wave) * 2 + pb) * 64 + lane];
                yv[0][pb] = q0 + q1 + q2;
                yv[1][pb] = q1 - q2 - q3;
            });
            if constexpr (n + 1 < NTV) __syncthreads();     // the next channel tile overwrites the exchange buffer

            const int c = cbase + n * 16;
            const bool first = c < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc = first ? c : c - a.split;
            int ch = cc, ay = 0, bx = 0;
            if (ps != 1) {
                const int sub = cc / a.ps_c;
                ch = cc - sub * a.ps_c;
                ay = sub / ps;
                bx = sub - ay * ps;
            }
            const size_t pix00 = (size_t)((img * H + gy0) * ps + ay) * orow + (size_t)(gx0 * ps + bx);
            float* o00 = optr + pix00 * ostride + ooff + ch;
            const size_t dx = (size_t)ps * ostride;
            const size_t dy = (size_t)ps * orow * ostride;
            const bool live = ok_00 && cc < owidth;
            static_for<0, 2>([&](auto pa_) DCSCN_INL {
                static_for<0, 2>([&](auto pb_) DCSCN_INL {
                    constexpr int pa = decltype(pa_)::value, pb = decltype(pb_)::value;
                    f32x4 o = yv[pa][pb] + bv[n];
                    o.x = activate1(o.x, av[n].x, act_e);
                    o.y = activate1(o.y, av[n].y, act_e);
                    o.z = activate1(o.z, av[n].z, act_e);
                    o.w = activate1(o.w, av[n].w, act_e);
                    if (live && (pa == 0 || ok_y1) && (pb == 0 || ok_x1)) {
                        if (a.res) {
                            const size_t pix = pix00 + (size_t)(pa * ps) * orow + (size_t)(pb * ps);
                            o += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch);
                        }
                        *reinterpret_cast<f32x4*>(o00 + pa * dy + pb * dx) = o;
                    }
                });
            });
        });
    };
    if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
}

template <int NT, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_wino_fs(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nt_valid = (blockIdx.y == gridDim.y - 1) ? a.n_full : NT;   // block uniform
    if (nt_valid == NT) conv_wino_fs_body<NT, NT>(a, smem);
    else if constexpr (NT >= 2) {
        if (nt_valid == NT - 1) conv_wino_fs_body<NT, NT - 1>(a, smem);
        else if constexpr (NT >= 3) {
            if (nt_valid == NT - 2) conv_wino_fs_body<NT, NT - 2>(a, smem);
        }
    }
}

}  // namespace dcscn_lab
