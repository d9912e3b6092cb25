// conv_nin_h: the wide 1x1 convs (A1 || B1 over the skip-concat, DCSCN.py:273-277; the non-NIN "C" layer) on
// v_mfma_f32_16x16x32_f16 at f32 accuracy (split16.hpp) -- the conv_nin pipeline with 32-channel chunks, so that the matrix instruction
// is the full-rate K = 32 form (v_mfma_f32_16x16x16_f16 takes the same 16 cycles for half the work: profiles/r03_pipe_probe.txt).
//
// * 128 consecutive pixels of the flat pixel list per workgroup, wave w owns 32 of them (two MFMA column tiles) for all
//   NT*16 <= 96 output channels of the group.  Two workgroups per CU.
// * input stage = [128 pixels][8 slots of 16 bytes]: the chunk's 32 channels of each pixel, 128 contiguous bytes fetched by eight
//   adjacent lanes of an LDS-DMA (or one source quad each with MULTI).  Slot order inside a pixel record is swizzled on the SOURCE
//   side (the DMA destination is lane-linear): quad q = 2 kq + half sits at slot 2 * ((kq + (p >> 1)) & 3) + (half ^ (kq & 1)), which
//   makes both ds_read_b128 of a B fragment (lane (j, kq): channels 8kq..8kq+3, then 8kq+4..8kq+7 of pixel j) conflict free.
// * S input stages (2 or 3): chunk c + S is fetched while chunk c computes; fragments of chunk c + 1 are read during chunk c;
//   a chunk's filter pieces are issued before its input pieces so that the counted vmcnt wait covers them.
// * filters: pack_conv16 image with one tap: [chunk][n][hi | lo][lane][8 halfs], 2 NT KB per chunk, two stages.
// * epilogue as conv_nin plus the split16 parts (scale 2^-e, bias, activator, two destinations -- float32 or P16, each on its own --, the
//   redo flag of the image of every pixel whose value left the f16 range).
#pragma once
#include "conv_nin.hpp"
#include "split16.hpp"
#include "p16.hpp"

#ifndef NINH_A_NT
#define NINH_A_NT 0        // 1: input pieces with the non-temporal policy -- measured r05: 2.91 -> 4.18 ms (the producers' lines are still on their way through L2 / MALL)
#endif

#ifndef NINH_ABL
#define NINH_ABL 0         // tuner only (tools/ninh_abl.sh; results wrong by design): 1 no filter DMA in the K loop, 2 no MFMAs, 4 no epilogue stores, 8 no input DMA in the K loop
#endif

namespace dcscn {

#ifndef NINH_WAVES
#define NINH_WAVES 4       // waves per workgroup (tuner: 8)
#endif
#ifndef NINH_MT
#define NINH_MT 2          // 16-pixel tiles per wave; pixels per workgroup = 16 * NINH_MT * NINH_WAVES (tuner: 8 waves x 1 = 128 pixels with half the serial work
#endif                     // per wave; 8 x 2 = 256 pixels, ONE workgroup per CU, half the filter traffic per pixel)

template <int NT, int S = 2>
struct NinHGeom {
    static constexpr int W = NINH_WAVES;
    static constexpr int THREADS = 64 * W;
    static constexpr int KC = 32;
    static constexpr int PIX = 16 * NINH_MT * W;
    static constexpr int MT = NINH_MT;                        // 16-pixel tiles per wave
    static constexpr int PSTRIDE = 128;
    static constexpr int A_SLOTS = PIX * 8;
    static constexpr int A_DMA = A_SLOTS / 64;                // 16 wave instructions, 4 per wave
    static constexpr int A_BYTES = A_SLOTS * 16;              // 16384
    static constexpr int A_ROUNDS = A_DMA / W;
    static constexpr int B_BYTES = NT * 2048;
    static constexpr int B_PIECES = 2 * NT;
    static constexpr int B_ROUNDS = (B_PIECES + W - 1) / W;
    static constexpr int B_STAGE = B_BYTES;
    static constexpr int B_BASE = S * A_BYTES;
    static constexpr int BA_BASE = S * A_BYTES + 2 * B_STAGE;  // bias | slopes of the channel group (NT * 16 floats each), staged at workgroup start
    static constexpr int LDS_BYTES = BA_BASE + NT * 128;
    static_assert(S == 2 || S == 3, "2 or 3 input stages");
};

// SRC: 0 = one float32 tensor, 1 = MULTI (float32 sources through a per-quad table), 2 = P16 sources (p16.hpp) through a per-OCTET table:
// entry = {address of the octet's hi unit in the record of pixel 0, record bytes}; the staged slot is then a ready (hi | lo) unit -- the B
// fragments are read as they are, no split in registers.  Entries past the last octet point at a plane's zero record with stride 0.
template <int NT, int NTV, int SRC, int S>
__device__ __forceinline__ void conv_nin_h_body(const ConvArgs& a, float* smem, long long pix0, int ntile) {
    using G = NinHGeom<NT, S>;
    constexpr bool MULTI = SRC != 0, IN16 = SRC == 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    const long long npix = (long long)a.N * a.H * a.W;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA sources: piece p = wave + 4r covers slots [64p, 64p + 64) = pixels [8p, 8p + 8); this lane's slot position
    // (lane & 7) of pixel 8p + (lane >> 3) holds channel quad dq (the inverse of the slot swizzle; (pixel >> 1) & 3 == (lane >> 4) & 3) ----
    const int d_kq = (((lane & 7) >> 1) - ((lane >> 4) & 3)) & 3;
    const int dq = 2 * d_kq + ((lane & 1) ^ (d_kq & 1));
    // Every lane ALWAYS issues its piece (the counted vmcnt wait of the K loop needs a fixed number of operations per wave):
    // pixels past the end of the list re-read the last pixel (their columns are never stored), channel quads past cin re-read
    // quad 0 (their filter rows are zero, the data is finite), invalid table entries point at valid memory with stride 0.
    unsigned a_off[G::A_ROUNDS];
    long long a_pix[G::A_ROUNDS];
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        long long p = pix0 + (wave + G::W * r) * 8 + (lane >> 3);
        p = p < npix ? p : npix - 1;
        a_pix[r] = p;
        a_off[r] = (unsigned)((p - pix0) * a.in_stride * 4);
    });
    const float* a_base = a.in + (size_t)pix0 * a.in_stride + a.in_off;                           // wave-uniform
    const char* b_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)ntile * a.n_chunks * G::B_BYTES;
    const unsigned b_off = (unsigned)(lane * 16);

    auto dma_b = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int piece = (wave + G::W * r) % G::B_PIECES;       // waves without a piece of their own repeat one: same count for every wave
        glds16(b_base + (size_t)chunk * G::B_BYTES + 1024 * piece, b_off, lds0 + G::B_BASE + stage * G::B_STAGE + (unsigned)piece * 1024u);
    };
    typedef const volatile __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
    u32x4 ent = {0u, 0u, 0u, 0u};
    auto load_ent = [&](int chunk) DCSCN_INL {
        if constexpr (IN16) ent = *(lds_u32x4_ptr)(uintptr_t)(lds0 + G::LDS_BYTES + (unsigned)(chunk * 4 + d_kq) * 16u);
        else if constexpr (MULTI) ent = *(lds_u32x4_ptr)(uintptr_t)(lds0 + G::LDS_BYTES + (unsigned)(chunk * 8 + dq) * 16u);
    };
    auto dma_a = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if constexpr (MULTI) {
            const char* src = reinterpret_cast<const char*>(((unsigned long long)ent.y << 32) | ent.x) + (unsigned long long)a_pix[r] * ent.z + (IN16 ? (dq & 1) * 16 : 0);
            if constexpr (NINH_A_NT) glds16v_nt(src, lds0 + stage * G::A_BYTES + (unsigned)(wave + G::W * r) * 1024u);
            else glds16v(src, lds0 + stage * G::A_BYTES + (unsigned)(wave + G::W * r) * 1024u);
        } else {
            const int c0 = chunk * G::KC + 4 * dq;
            glds16(a_base, a_off[r] + (unsigned)((c0 < a.cin_phys ? c0 : 0) * 4), lds0 + stage * G::A_BYTES + (unsigned)(wave + G::W * r) * 1024u);
        }
    };

    {
        if constexpr (MULTI)
            for (int i = tid; i < (IN16 ? 4 : 8) * a.n_chunks; i += G::THREADS)
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::LDS_BYTES + 16 * i) = reinterpret_cast<const f32x4*>(a.srctab)[i];
        __syncthreads();
    }

    if (tid < NTV * 4) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::BA_BASE + tid * 16) = reinterpret_cast<const f32x4*>(a.bias + ntile * NT * 16)[tid];
    else if (tid >= 64 && tid < 64 + NTV * 4 && a.act == ACT_ALPHA)
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::BA_BASE + NT * 64 + (tid - 64) * 16) = reinterpret_cast<const f32x4*>(a.alpha + ntile * NT * 16)[tid - 64];

    f32x4 acc[G::MT][NTV];
    static_for<0, G::MT>([&](auto m_) DCSCN_INL {
        static_for<0, NTV>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
    });

    // B fragment of pixel tile m: pixel 16 * G::MT * wave + 16 * m + lj, channel group lk: halves at slot 2 * ((lk + (pixel >> 1)) & 3) + (lk & 1), ^ 1
    const unsigned a_rel = (unsigned)((16 * G::MT * wave + lj) * G::PSTRIDE + ((((lk + (lj >> 1)) & 3) << 1) | (lk & 1)) * 16);
    const unsigned a_lane = lds0 + a_rel, a_lane2 = lds0 + (a_rel ^ 16u);
    const unsigned b_lane = lds0 + (unsigned)(G::B_BASE + lane * 16);
    typedef const volatile __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    typedef const volatile __attribute__((address_space(3))) h8* lds_h8_ptr;
    const float m1 = opaque_minus_one();

    const int last = a.n_chunks - 1;
    static_for<0, G::B_ROUNDS>([&](auto r_) DCSCN_INL { dma_b(r_, 0, 0); });
    load_ent(0);
    static_for<0, S>([&](auto st_) DCSCN_INL {                // chunks 0 .. S-1 into stages 0 .. S-1
        constexpr int st = decltype(st_)::value;
        static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, last < st ? last : st, st); });
        load_ent(last < st + 1 ? last : st + 1);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 xa[G::MT], xb[G::MT];
    static_for<0, G::MT>([&](auto m_) DCSCN_INL {
        constexpr int m = decltype(m_)::value;
        xa[m] = *(lds_f32x4_ptr)(uintptr_t)(a_lane + m * 16 * G::PSTRIDE);
        xb[m] = *(lds_f32x4_ptr)(uintptr_t)(a_lane2 + m * 16 * G::PSTRIDE);
    });
    __syncthreads();                                          // every wave holds its fragments of chunk 0: input stage 0 may be refilled
    unsigned sa = 0;                                          // chunk % S
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        const unsigned sb = chunk & 1;
        const int cb = chunk + 1 < last ? chunk + 1 : last;   // filters to fetch (clamped: redundant copies land in a dead stage)
        const int ca = chunk + S < last ? chunk + S : last;   // input to fetch
        const unsigned sn = sa + 1 == S ? 0 : sa + 1;
        const unsigned Bs = b_lane + sb * G::B_STAGE;
        const unsigned An = a_lane + sn * G::A_BYTES, An2 = a_lane2 + sn * G::A_BYTES;
        // filters first, then input: the counted wait below relies on this order
        if constexpr (!(NINH_ABL & 1)) static_for<0, G::B_ROUNDS>([&](auto r_) DCSCN_INL { dma_b(r_, cb, sb ^ 1); });
        f32x4 na[G::MT], nb[G::MT];
        static_for<0, G::MT>([&](auto m_) DCSCN_INL {
            constexpr int m = decltype(m_)::value;
            h8 xh, xl;
            if constexpr (IN16) { xh = __builtin_bit_cast(h8, xa[m]); xl = __builtin_bit_cast(h8, xb[m]); }
            else split8(xa[m], xb[m], m1, xh, xl);
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const h8 wh = *(lds_h8_ptr)(uintptr_t)(Bs + (2 * n) * 1024);
                const h8 wl = *(lds_h8_ptr)(uintptr_t)(Bs + (2 * n + 1) * 1024);
                if constexpr (NINH_ABL & 2) asm volatile("" :: "v"(wh), "v"(wl), "v"(xh), "v"(xl));
                else {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[m][n], 0, 0, 0);
                }
                // the chunk's input pieces behind the first MFMA groups
                if constexpr (!(NINH_ABL & 8) && m * NTV + n < G::A_ROUNDS) dma_a(std::integral_constant<int, m * NTV + n>{}, ca, sa);
            });
            if constexpr (!(NINH_ABL & 8) && m == G::MT - 1 && G::MT * NTV < G::A_ROUNDS)
                static_for<G::MT * NTV, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, ca, sa); });
            na[m] = *(lds_f32x4_ptr)(uintptr_t)(An + m * 16 * G::PSTRIDE);
            nb[m] = *(lds_f32x4_ptr)(uintptr_t)(An2 + m * 16 * G::PSTRIDE);
        });
        load_ent(chunk + S + 1 < last ? chunk + S + 1 : last);      // the next iteration's ca
        // the next chunk reads the fragments of chunk c + 2 and the filters of chunk c + 1: with S = 3 only this iteration's
        // input pieces (the youngest A_ROUNDS operations) may stay in flight
        if constexpr (S == 2 || NINH_ABL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::A_ROUNDS) : "memory");
        __syncthreads();
        static_for<0, G::MT>([&](auto m_) DCSCN_INL { xa[decltype(m_)::value] = na[decltype(m_)::value]; xb[decltype(m_)::value] = nb[decltype(m_)::value]; });
        sa = sn;
    }
    if constexpr (S > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: un-scale, bias, activator, store (float4 per lane: channels cbase..cbase+3 of one pixel) ----
    const int cbase = ntile * NT * 16 + 4 * lk;
    const int obase = cbase - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);
    const int act = a.act;
    const float inv = a.inv_scale;
    float chk[G::MT];                                          // per pixel tile: the two pixels of a lane may belong to different images
    static_for<0, G::MT>([&](auto m_) DCSCN_INL { chk[decltype(m_)::value] = 0.0f; });
    const float zero = opaque_zero();
    const h2 zero2 = p16_opaque_zero2();
    auto finish = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c = obase + n * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + G::BA_BASE + (n * 4 + lk) * 16);
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            if (act_e == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(smem) + G::BA_BASE + NT * 64 + (n * 4 + lk) * 16);
            const bool first = c < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc = first ? c : c - a.split;
            const OutDesc& od = first ? a.out0 : a.out1;
            if (od.p16.base != nullptr) {                         // block uniform: a P16 destination takes one (hi | lo) unit per lane
                const int oct0 = (ooff + cc - 4 * lk) >> 3;       // first octet of the 16-channel tile
                const int chunk = oct0 >> 2, rem = od.p16.octs - 4 * chunk;
                const int rec = rem >= 4 ? 128 : 32 * rem;
                char* base = od.p16.base + (long long)chunk * od.p16.plane + 128 + (oct0 & 3) * 32 + lk * 16;
                const bool chan_ok = oct0 + (lk >> 1) < od.p16.octs;
                static_for<0, G::MT>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const long long p = pix0 + 16 * G::MT * wave + 16 * m + lj;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act_e);
                    v.y = activate1(v.y, av.y, act_e);
                    v.z = activate1(v.z, av.z, act_e);
                    v.w = activate1(v.w, av.w, act_e);
                    if (ACT_C < 0 && p < npix && chan_ok) chk[m] = nonfinite_acc(chk[m], acc[m][n], zero);
                    const u32x4 unit = p16_unit(v, m1, chk[m], zero2, p < npix && chan_ok);
                    if (p < npix && chan_ok && (!(NINH_ABL & 4) || unit.x == 0x12345u)) *reinterpret_cast<u32x4*>(base + (size_t)p * rec) = unit;
                });
            } else if (cc < owidth) {
                static_for<0, G::MT>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const long long p = pix0 + 16 * G::MT * wave + 16 * m + lj;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act_e);
                    v.y = activate1(v.y, av.y, act_e);
                    v.z = activate1(v.z, av.z, act_e);
                    v.w = activate1(v.w, av.w, act_e);
                    if (p < npix && (!(NINH_ABL & 4) || v.x == 12345.678f)) {
                        chk[m] = nonfinite_acc(chk[m], acc[m][n], zero);
                        *reinterpret_cast<f32x4*>(optr + (size_t)p * ostride + ooff + cc) = v;
                    }
                });
            }
        });
    };
    if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
    // the image of a pixel with a value beyond the f16 range goes to the float32 plan (exec.hip); columns past the end of the pixel list
    // hold copies of the last pixel
    static_for<0, G::MT>([&](auto m_) DCSCN_INL {
        constexpr int m = decltype(m_)::value;
        if (chk[m] != chk[m] && a.redo) {
            long long p = pix0 + 16 * G::MT * wave + 16 * m + lj;
            p = p < npix ? p : npix - 1;
            a.redo[0] = 1;
            a.redo[1 + (int)(p / ((long long)a.H * a.W))] = 1;
        }
    });
}

// grid = (pixel blocks of 128, channel groups)
template <int NT, int SRC = 0, int S = 2, int WPS = 2>
__global__ __launch_bounds__(64 * NINH_WAVES, (WPS * NINH_WAVES / 4) * 128 / (16 * NINH_MT * NINH_WAVES) > 0 ? (WPS * NINH_WAVES / 4) * 128 / (16 * NINH_MT * NINH_WAVES) : 1) void conv_nin_h(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long pix0 = (long long)blockIdx.x * NinHGeom<NT, S>::PIX;
    const int ntile = blockIdx.y;
    if (ntile < a.n_full) conv_nin_h_body<NT, NT, SRC, S>(a, smem, pix0, ntile);          // block uniform
    else if constexpr (NT >= 2) conv_nin_h_body<NT, NT - 1, SRC, S>(a, smem, pix0, ntile);
}

}  // namespace dcscn
