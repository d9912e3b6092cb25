// conv_nin_h: conv_nin (the wide 1x1 convs A1 || B1 over the skip-concat, DCSCN.py:273-277, and the non-NIN "C" layer) with
// the contraction on the 16-bit matrix pipe at f32 accuracy (split16.hpp): v_mfma_f32_16x16x16_f16 x 3 products instead of
// four v_mfma_f32_16x16x4_f32 per 16-channel chunk.  Everything around the MFMAs is conv_nin's pipeline, unchanged:
//
// * 256 consecutive pixels of the flat pixel list per workgroup, wave w owns 64 of them (four MFMA column tiles) for all
//   NT*16 <= 96 output channels of the group; 16-channel chunks; input [256 pixels][64 bytes] by LDS-DMA (four adjacent
//   lanes fetch the 64 contiguous bytes of a pixel's chunk, or of one source quad each with MULTI), two input stages with the
//   fragments of chunk c+1 read while chunk c computes, one barrier per chunk.
// * the B operand of the 16x16x16 instruction is exactly the ds_read_b128 fragment of before -- lane (j, kq) holds channels
//   4kq..4kq+3 of pixel j -- split into (hi, lo) in registers: 6 VALU per fragment, 24 per chunk and wave, once per value.
// * filters: [chunk][tile n][lane][hi0..3 | lo0..3] halfs (split16_pack.hpp: pack_nin16) = ONE ds_read_b128 per lane and tile
//   for both pieces; NT KB per chunk instead of 7 KB of f32.
// * epilogue: accumulators * 2^-e (the filter scale), bias, activator, two destinations, float4 stores -- plus the non-finite
//   check that raises redo[pixel block] for the f32 kernel behind this one (split16.hpp).
//
// With the matrix time down 5x the launch is bound by reading the 5.3 KB of concat per pixel from HBM.
#pragma once
#include "conv_nin.hpp"
#include "split16.hpp"

namespace dcscn {

template <int NT>
struct NinHGeom {
    static constexpr int THREADS = 256;
    static constexpr int KC = 16;
    static constexpr int PIX = 256;
    static constexpr int MT = 4;
    static constexpr int PSTRIDE = 64;
    static constexpr int A_SLOTS = PIX * 4;
    static constexpr int A_DMA = A_SLOTS / 64;
    static constexpr int A_BYTES = A_SLOTS * 16;              // 16384
    static constexpr int A_ROUNDS = A_DMA / 4;
    static constexpr int B_BYTES = NT * 1024;                 // [n][64 lanes][16 bytes]
    static constexpr int B_PIECES = NT;
    static constexpr int B_ROUNDS = (B_PIECES + 3) / 4;
    static constexpr int B_STAGE = B_BYTES;
    static constexpr int B_BASE = 2 * A_BYTES;
    static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_STAGE;
};

template <int NT, int NTV, bool MULTI>
__device__ __forceinline__ void conv_nin_h_body(const ConvArgs& a, float* smem, long long pix0, int ntile) {
    using G = NinHGeom<NT>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    const long long npix = (long long)a.N * a.H * a.W;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA sources (conv_nin.hpp): piece p = wave + 4r covers slots [64p, 64p + 64); slot = 4 * pixel + quad ----
    unsigned a_off[G::A_ROUNDS];
    bool a_on[G::A_ROUNDS];
    int a_q[G::A_ROUNDS];
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int slot = (wave + 4 * r) * 64 + lane;
        const int p = slot >> 2;
        const int q = slot & 3;
        a_q[r] = 4 * q;
        a_on[r] = pix0 + p < npix;
        a_off[r] = (unsigned)((p * a.in_stride + 4 * q) * 4);
    });
    const float* a_base = a.in + (size_t)pix0 * a.in_stride + a.in_off;                           // wave-uniform
    const char* b_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)ntile * a.n_chunks * G::B_BYTES;
    const unsigned b_off = (unsigned)(lane * 16);

    auto dma_b = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int piece = wave + 4 * r;
        if (piece < G::B_PIECES)                              // wave-uniform
            glds16(b_base + (size_t)chunk * G::B_BYTES + 1024 * piece, b_off, lds0 + G::B_BASE + stage * G::B_STAGE + (unsigned)piece * 1024u);
    };
    typedef const volatile __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
    u32x4 ent = {0u, 0u, 0u, 0u};
    auto load_ent = [&](int chunk) DCSCN_INL {
        if constexpr (MULTI) ent = *(lds_u32x4_ptr)(uintptr_t)(lds0 + G::LDS_BYTES + (unsigned)(chunk * 4 + (lane & 3)) * 16u);
    };
    auto dma_a = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if constexpr (MULTI) {
            const unsigned long long pix = (unsigned long long)(pix0 + (wave + 4 * r) * 16 + (lane >> 2));
            const char* src = reinterpret_cast<const char*>(((unsigned long long)ent.y << 32) | ent.x) + pix * ent.z;
            if (a_on[r] && ent.w) glds16v(src, lds0 + stage * G::A_BYTES + (unsigned)(wave + 4 * r) * 1024u);
        } else {
            if (a_on[r] && chunk * G::KC + a_q[r] < a.cin_phys)
                glds16(a_base + chunk * G::KC, a_off[r], lds0 + stage * G::A_BYTES + (unsigned)(wave + 4 * r) * 1024u);
        }
    };

    // clear both input stages once: channel-tail slots are never written, and 0 * stale must not be 0 * NaN
    {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = tid; i < 2 * G::A_BYTES / 16; i += G::THREADS) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + 16 * i) = z;
        if constexpr (MULTI)
            for (int i = tid; i < 4 * a.n_chunks; i += G::THREADS)
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::LDS_BYTES + 16 * i) = reinterpret_cast<const f32x4*>(a.srctab)[i];
        __syncthreads();
    }

    f32x4 acc[G::MT][NTV];
    static_for<0, G::MT>([&](auto m_) DCSCN_INL {
        static_for<0, NTV>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
    });

    const unsigned a_lane = lds0 + (unsigned)((64 * wave + lj) * G::PSTRIDE + lk * 16);
    const unsigned b_lane = lds0 + (unsigned)(G::B_BASE + lane * 16);
    typedef const volatile __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    typedef const volatile __attribute__((address_space(3))) h8* lds_h8_ptr;
    const float m1 = opaque_minus_one();

    const int last = a.n_chunks - 1;
    static_for<0, G::B_ROUNDS>([&](auto r_) DCSCN_INL { dma_b(r_, 0, 0); });
    load_ent(0);
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, 0, 0); });
    load_ent(last < 1 ? last : 1);
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, last < 1 ? last : 1, 1); });
    load_ent(last < 2 ? last : 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 xv[G::MT];
    static_for<0, G::MT>([&](auto m_) DCSCN_INL {
        constexpr int m = decltype(m_)::value;
        xv[m] = *(lds_f32x4_ptr)(uintptr_t)(a_lane + m * 16 * G::PSTRIDE);
    });
    __syncthreads();                                          // every wave holds its fragments of chunk 0: input stage 0 may be refilled
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        const unsigned sb = chunk & 1;
        const int cb = chunk + 1 < last ? chunk + 1 : last;   // filters to fetch (clamped: redundant copies land in a dead stage)
        const int ca = chunk + 2 < last ? chunk + 2 : last;   // input to fetch
        const unsigned Bs = b_lane + sb * G::B_STAGE;
        const unsigned An = a_lane + (sb ^ 1) * G::A_BYTES;
        h4 wh[NTV], wl[NTV];
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const h8 w = *(lds_h8_ptr)(uintptr_t)(Bs + n * 1024);
            wh[n] = __builtin_shufflevector(w, w, 0, 1, 2, 3);
            wl[n] = __builtin_shufflevector(w, w, 4, 5, 6, 7);
        });
        f32x4 xn[G::MT];
        static_for<0, G::MT>([&](auto m_) DCSCN_INL {
            constexpr int m = decltype(m_)::value;
            h4 xh, xl;
            split4(xv[m], m1, xh, xl);
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl[n], xh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[n], xl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[n], xh, acc[m][n], 0, 0, 0);
            });
            // the chunk's DMA pieces behind the MFMA groups; the next chunk's fragments right after
            static_for<0, G::B_ROUNDS + G::A_ROUNDS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                if constexpr (i % G::MT == m) {
                    if constexpr (i < G::B_ROUNDS) dma_b(std::integral_constant<int, i>{}, cb, sb ^ 1);
                    else dma_a(std::integral_constant<int, i - G::B_ROUNDS>{}, ca, sb);
                }
            });
            xn[m] = *(lds_f32x4_ptr)(uintptr_t)(An + m * 16 * G::PSTRIDE);
        });
        load_ent(chunk + 3 < last ? chunk + 3 : last);      // the next iteration's ca
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        static_for<0, G::MT>([&](auto m_) DCSCN_INL { xv[decltype(m_)::value] = xn[decltype(m_)::value]; });
    }

    // ---- epilogue: un-scale, bias, activator, store (float4 per lane: channels cbase..cbase+3 of one pixel) ----
    const int cbase = ntile * NT * 16 + 4 * lk;
    const int obase = cbase - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);
    const int act = a.act;
    const float inv = a.inv_scale;
    float chk = 0.0f;
    const float zero = opaque_zero();
    auto finish = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c = obase + n * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + cbase + n * 16);
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            if (act_e == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(a.alpha + cbase + n * 16);
            const bool first = c < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc = first ? c : c - a.split;
            if (cc < owidth) {
                static_for<0, G::MT>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const long long p = pix0 + 64 * wave + 16 * m + lj;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act_e);
                    v.y = activate1(v.y, av.y, act_e);
                    v.z = activate1(v.z, av.z, act_e);
                    v.w = activate1(v.w, av.w, act_e);
                    if (p < npix) {
                        chk = nonfinite_acc(chk, acc[m][n], zero);
                        *reinterpret_cast<f32x4*>(optr + (size_t)p * ostride + ooff + cc) = v;
                    }
                });
            }
        });
    };
    if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
    if (chk != chk && a.redo) a.redo[blockIdx.x] = 1;          // any lane, any group of the block: same value, benign race
}

// grid = (pixel blocks of 256, channel groups)
template <int NT, bool MULTI = false, int WPS = 3>
__global__ __launch_bounds__(256, WPS) void conv_nin_h(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long pix0 = (long long)blockIdx.x * NinHGeom<NT>::PIX;
    const int ntile = blockIdx.y;
    if (ntile < a.n_full) conv_nin_h_body<NT, NT, MULTI>(a, smem, pix0, ntile);          // block uniform
    else if constexpr (NT >= 2) conv_nin_h_body<NT, NT - 1, MULTI>(a, smem, pix0, ntile);
}

}  // namespace dcscn
