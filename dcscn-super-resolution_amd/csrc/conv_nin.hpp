// conv_nin: the wide 1x1 convs (A1 || B1 over the 1301-channel skip-concat, DCSCN.py:273-277; the non-NIN "C" layer) as a
// plain GEMM on v_mfma_f32_16x16x4_f32 with LDS-DMA staging -- the conv_wino2 pipeline without the Winograd transforms.
//
//   D[cout][pixel] += W[cout][cin] * X[cin][pixel]      per pixel, no spatial coupling: the pixels of all images are ONE
//                                                        flat list and a workgroup takes 256 consecutive ones.
//
// * 4 waves; wave w owns pixels [64w, 64w + 64) = four MFMA column tiles, for all NT*16 <= 96 output channels of the
//   group: 4*NT accumulator tiles (96 VGPRs at NT = 6), 4 + ... operand reads per 4*NT MFMAs.
// * K walks in chunks of 16 input channels = four MFMA k-steps per barrier.  Both operands arrive by
//   `global_load_lds_dwordx4` (conv_wino2.hpp: glds16):
//     input    [256 pixels][4 slots of 16 bytes] = the chunk's 16 channels of each pixel, 64 contiguous bytes of the NHWC row
//              fetched by four adjacent lanes.  The operand read is one ds_read_b128 per lane (channels 4k..4k+3 of pixel j);
//              with the dense 64-byte pixel stride it is 2-way bank conflicted, which does not matter at 4 reads per 96
//              MFMAs -- an 80-byte padded stride (conflict free, tried first) cost a fifth of the DMA lanes and, at 55 KB
//              of LDS, the third resident workgroup per CU: 4.77 ms vs the dense layout's time in profiles/.
//     filters  [s*4 + k][NS] with row (s, k) = channel 4k + s of the chunk: k-step s multiplies channel 4k+s of every lane
//              group k, so one 16-byte input read feeds four k-steps.  Packed on the host, copied linearly.
// * two filter stages and two input stages, the input running one chunk further ahead: the operand fragments of chunk
//   c+1 are read during the last k-step of chunk c, the MFMA stream continues straight across the single barrier.
// * pixels past the end of the list and channels past cin never issue their DMA lanes; the stale LDS they leave is
//   multiplied into accumulator columns nobody stores (pixels) or by zero filter rows (channels; the stages are cleared
//   once so that "stale" is finite).
//
// MULTI (multi-source input): the K axis is the concatenation of several dense NHWC tensors (the per-layer buffers of the
// feature stack, api.hip: densify_features) instead of one wide concat tensor.  A device table with one entry per 16-byte
// channel quad of the virtual concat -- {pointer to that quad of pixel 0, pixel stride in bytes, valid} -- is copied to
// LDS once; a DMA lane always serves the same quad position (lane & 3) of its pixels, reads its entry per chunk and
// forms a 64-bit source address.  Nothing else changes: the LDS image, the filter image and the MFMA loop are the same.
//
// Epilogue = conv_igemm's: bias, activator, two destinations (B1 -> T1, A1 -> its slice of Concat2), float4 NHWC stores.
#pragma once
#include "conv_wino2.hpp"

namespace dcscn {

// LDS-DMA with a per-lane 64-bit source address (conv_wino2.hpp: glds16 takes an SGPR base + 32-bit lane offset)
__device__ __forceinline__ void glds16v(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// the same with the non-temporal policy: data that ONE workgroup reads ONCE (MI355X_MICROARCH.md "nt-weights": issued -> landed -18 %)
__device__ __forceinline__ void glds16v_nt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int NT>
struct NinGeom {
    static constexpr int THREADS = 256;
    static constexpr int KC = 16;
    static constexpr int PIX = 256;                           // pixels per workgroup
    static constexpr int MT = 4;                              // 16-pixel tiles per wave
    static constexpr int PSTRIDE = 64;                        // bytes per pixel record: the 16 channels of the chunk
    static constexpr int A_SLOTS = PIX * 4;
    static constexpr int A_DMA = A_SLOTS / 64;                // 16 wave instructions, 4 per wave
    static constexpr int A_BYTES = A_SLOTS * 16;              // 16384
    static constexpr int NS = conv_ns(NT);
    static constexpr int B_FLOATS = KC * NS;
    static constexpr int B_BYTES = B_FLOATS * 4;
    static constexpr int B_PIECES = (B_BYTES + 1023) / 1024;  // 1 KB DMA pieces (the last may be partial)
    static constexpr int B_ROUNDS = (B_PIECES + 3) / 4;
    static constexpr int B_STAGE = B_PIECES * 1024;
    static constexpr int B_BASE = 2 * A_BYTES;
    static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_STAGE;
    static constexpr int A_ROUNDS = A_DMA / 4;
};

template <int NT, int NTV, bool MULTI>
__device__ __forceinline__ void conv_nin_body(const ConvArgs& a, float* smem, long long pix0, int ntile) {
    using G = NinGeom<NT>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    const long long npix = (long long)a.N * a.H * a.W;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA sources: piece p = wave + 4r covers slots [64p, 64p + 64); slot = 4 * pixel + quad ----
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
        a_off[r] = (unsigned)((p * a.in_stride + 4 * q) * 4);                  // < 256 * stride * 4 bytes
    });
    const float* a_base = a.in + (size_t)pix0 * a.in_stride + a.in_off;        // wave-uniform
    const float* b_base = a.wpack + (size_t)ntile * a.n_chunks * G::B_FLOATS;  // wave-uniform
    const unsigned b_off = (unsigned)(lane * 16);

    auto dma_b = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int piece = wave + 4 * r;
        if (piece < G::B_PIECES) {                            // wave-uniform
            const float* src = b_base + (size_t)chunk * G::B_FLOATS + 256 * piece;
            const unsigned dst = lds0 + G::B_BASE + stage * G::B_STAGE + (unsigned)piece * 1024u;
            if constexpr (G::B_BYTES % 1024 == 0) glds16(src, b_off, dst);
            else if (piece * 1024 + lane * 16 < G::B_BYTES) glds16(src, b_off, dst);     // partial last piece
        }
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
    u32x4 ent = {0u, 0u, 0u, 0u};
    auto load_ent = [&](int chunk) DCSCN_INL {
        if constexpr (MULTI) ent = *(lds_u32x4_ptr)(uintptr_t)(lds0 + G::LDS_BYTES + (unsigned)(chunk * 4 + (lane & 3)) * 16u);
    };
    auto dma_a = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if constexpr (MULTI) {
            // ent = this lane's quad of the chunk (entry chunk * 4 + (lane & 3) of the source table), fetched one chunk ahead
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

    // operand addresses: pixel 64*wave + 16*m + lj, channels 4*lk .. 4*lk+3 of the chunk
    const unsigned a_lane = lds0 + (unsigned)((64 * wave + lj) * G::PSTRIDE + lk * 16);
    const int b_lane = G::B_BASE + (lk * G::NS + lj) * 4;
    typedef const volatile __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;

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
        const float* Bs = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + b_lane + sb * G::B_STAGE);
        const unsigned An = a_lane + (sb ^ 1) * G::A_BYTES;
        f32x4 xn[G::MT];
        static_for<0, 4>([&](auto s_) DCSCN_INL {
            constexpr int s = decltype(s_)::value;
            float wv[NTV];
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                wv[n] = Bs[(s * 4) * G::NS + n * 16];
            });
            static_for<0, G::MT>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n], xv[m][s], acc[m][n], 0, 0, 0);
                });
                // the chunk's DMA pieces behind the first MFMA groups; the next chunk's fragments behind the last ones
                if constexpr (s * G::MT + m < G::B_ROUNDS) dma_b(std::integral_constant<int, s * G::MT + m>{}, cb, sb ^ 1);
                else if constexpr (s * G::MT + m < G::B_ROUNDS + G::A_ROUNDS)
                    dma_a(std::integral_constant<int, s * G::MT + m - G::B_ROUNDS>{}, ca, sb);
                if constexpr (s == 3) xn[m] = *(lds_f32x4_ptr)(uintptr_t)(An + m * 16 * G::PSTRIDE);
            });
        });
        load_ent(chunk + 3 < last ? chunk + 3 : last);      // the next iteration's ca
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        static_for<0, G::MT>([&](auto m_) DCSCN_INL { xv[decltype(m_)::value] = xn[decltype(m_)::value]; });
    }

    // ---- epilogue: bias, activator, store (float4 per lane: channels cbase..cbase+3 of one pixel) ----
    const int cbase = ntile * NT * 16 + 4 * lk;                                  // bias / slope index: padded group layout
    const int obase = cbase - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);    // conv channel: groups past n_full are one tile narrower
    const int act = a.act;
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
                    f32x4 v = acc[m][n] + bv;
                    v.x = activate1(v.x, av.x, act_e);
                    v.y = activate1(v.y, av.y, act_e);
                    v.z = activate1(v.z, av.z, act_e);
                    v.w = activate1(v.w, av.w, act_e);
                    if (p < npix) *reinterpret_cast<f32x4*>(optr + (size_t)p * ostride + ooff + cc) = v;
                });
            }
        });
    };
    if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
}

// grid = (pixel blocks of 256, channel groups)
template <int NT, bool MULTI = false, int WPS = 3>
__global__ __launch_bounds__(256, WPS) void conv_nin(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long pix0 = (long long)blockIdx.x * NinGeom<NT>::PIX;
    const int ntile = blockIdx.y;
    if (a.redo_check) {                                        // float32 plan behind a split16 pass: blocks that touch a flagged image
        if (a.redo[0] == 0) return;
        const long long hw = (long long)a.H * a.W, npix = hw * a.N;
        const long long last = pix0 + NinGeom<NT>::PIX - 1 < npix ? pix0 + NinGeom<NT>::PIX - 1 : npix - 1;
        bool any = false;
        for (int i = (int)(pix0 / hw); i <= (int)(last / hw); ++i) any = any || a.redo[1 + i] != 0;
        if (!any) return;
    }
    if (ntile < a.n_full) conv_nin_body<NT, NT, MULTI>(a, smem, pix0, ntile);          // block uniform
    else if constexpr (NT >= 2) conv_nin_body<NT, NT - 1, MULTI>(a, smem, pix0, ntile);
}

}  // namespace dcscn
