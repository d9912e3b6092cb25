// conv_wino2: 3x3 SAME convolution as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32, staged by LDS-DMA.
//
// Same arithmetic mapping as conv_wino (conv_wino.hpp: 16x16-pixel workgroup tile = 8x8 Winograd tiles, wave w owns tile
// rows 2w / 2w+1 for all 16 frequencies and NT*16 output channels, input transform in registers, wave-local output
// transform) -- what changed is everything AROUND the MFMA stream, which r01's counters showed to be the loss (matrix pipe
// 57 % busy: two barriers and a register-staged global -> LDS copy per 4-channel chunk):
//
// * Both operands reach LDS by `global_load_lds_dwordx4` (1 KB per wave instruction, no staging VGPRs, no ds_write pass,
//   no mask VALU).  The DMA destination is lane-linear, so the LDS image is chosen by each lane's SOURCE address:
//     filters  [f][s][k][NS]  -- the host packs exactly this image, the copy is linear;
//     input    2 planes [x parity] of [18 rows][9 pixels][2 channel quads] 16-byte slots: lanes 2p, 2p+1 of a DMA fetch the
//              32 contiguous bytes of one halo pixel (the 8 channels of the chunk).  Lanes whose halo pixel lies outside the image are masked off (EXEC): their slots are
//              cleared once at kernel start and never written again -- that IS the SAME zero padding.
// * 8 input channels per chunk = two MFMA k-steps per barrier, ONE barrier per chunk.  Two filter stages (filters of
//   chunk c+1 land while chunk c computes) and two input stages running one chunk further ahead: the raw patch of chunk
//   c+1 is read into registers during the last MFMAs of chunk c (the registers of chunk c's patch are dead by then), so
//   the MFMA stream continues straight across the barrier with the input transform's 32 adds as the only gap.
// * The DMA instructions of a chunk are spread over the MFMAs of its first k-step; they are waited for (vmcnt(0), the
//   wave's own loads only) just before the barrier that ends the chunk, ~2000 cycles after issue.
// * Raw-patch reads are `ds_read_b64`: lane (tile j, k) takes channels 2k, 2k+1 of its 4x4 patch with 16 reads per chunk;
//   k-step s uses channel 2k+s (the filter image is packed in the same order).  The 32 lanes of a read group cover 16
//   slots x 16 bytes; tiles 2 px apart sit in consecutive slots of a parity plane.
//
// The DMA is issued from inline asm: hipcc would otherwise make every later ds_read wait for it (it cannot tell the LDS
// stages apart), and __syncthreads() would drain it.  Nothing else in the K loop touches VMEM, so the only vmcnt wait
// is the explicit one.
#pragma once
#include "conv_igemm.hpp"

#ifndef DCSCN_GLDS_AUX
#define DCSCN_GLDS_AUX ""          // cache-policy suffix of the LDS-DMA loads (tuner: " nt", " sc1")
#endif

namespace dcscn {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT>
struct Wino2Geom {
    static constexpr int THREADS = 256;
    static constexpr int KC = 8;
    static constexpr int TH = 16, TW = 16;
    static constexpr int HTH = TH + 2, HTW = TW + 2;
    static constexpr int ROW_SLOTS = HTW / 2;                 // 9 pixels per plane row (one x parity)
    static constexpr int PL = HTH * ROW_SLOTS;                // 162 pixels per parity plane
    static constexpr int A_SLOTS = 4 * PL;                    // 648 slots: 2 planes x 162 pixels x 2 channel quads
    static constexpr int A_DMA = (A_SLOTS + 63) / 64;         // 11 wave instructions
    static constexpr int A_BYTES = A_DMA * 1024;
    static constexpr int NS = conv_ns(NT);
    static constexpr int B_FLOATS = 16 * KC * NS;
    static constexpr int B_BYTES = B_FLOATS * 4;
    static constexpr int B_DMA = B_BYTES / 1024;
    static constexpr int B_BASE = 2 * A_BYTES;                // LDS carve: input stage 0 | input stage 1 | filter stage 0 | filter stage 1
    static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;
    static constexpr int B_ROUNDS = B_DMA / 4;                // filter pieces per wave
    static constexpr int A_ROUNDS = (A_DMA + 3) / 4;          // input pieces per wave (the last round is partial)
    static_assert(B_BYTES % 4096 == 0, "filter block must be whole 1 KB pieces, the same number for every wave");
};

// One 16-byte-per-lane LDS-DMA: LDS[lds_dst + 16*lane] = *(sbase + voff) for the ACTIVE lanes (inactive lanes neither
// load nor write).  sbase is wave-uniform (SGPR pair), voff a 32-bit byte offset.  M0 (the DMA's LDS base) is
// compiler-reserved and is restored inside the same statement.  Completion is tracked by vmcnt; hipcc does not see the load.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" DCSCN_GLDS_AUX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// ABL (tuner only, tools/wino2_tune.hip): 0 shipped; 1 no DMA in the K loop; 2 no vmcnt wait; 3 no filter DMA; 4 no input DMA; 5 no barrier (+ no wait)
template <int NT, int NTV, int PF, int ABL = 0>
__device__ __forceinline__ void conv_wino2_body(const ConvArgs& a, float* smem, int tile_id, int ntile) {
    using G = Wino2Geom<NT>;
    static_assert(PF >= 1 && PF < 16, "filter operands are read 1..15 frequencies ahead");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;

    int bid = tile_id;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int y0 = ty * G::TH;
    const int x0 = tx * G::TW;
    const int H = a.H, W = a.W;
    const float* in_img = a.in + (size_t)img * H * W * a.in_stride + a.in_off;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;

    // ---- DMA sources of this lane's input slots (piece p = wave + 4r, slot = 64p + lane) ----
    // Byte offset from the halo tile's origin pixel (y0-1, x0-1); lanes whose slot is outside the image (SAME padding) or
    // past the last slot never issue: their LDS slots are cleared once, below, and stay zero.
    unsigned a_off[G::A_ROUNDS];
    bool a_inb[G::A_ROUNDS];
    bool a_hi[G::A_ROUNDS];                                  // slot holds channels 4..7 of a chunk
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int slot = (wave + 4 * r) * 64 + lane;
        const int pix = slot >> 1;
        const int par = pix / G::PL;
        const int rem = pix - par * G::PL;
        const int row = rem / G::ROW_SLOTS;
        const int xs = rem - row * G::ROW_SLOTS;
        const int hx = 2 * xs + par;
        // channel quad held by this slot: lanes 2p, 2p+1 fetch the 32 contiguous bytes of one pixel, in an order that
        // alternates with (xs + row / 2) -- see the raw-patch read below
        const int hq = (slot & 1) ^ ((xs + (row >> 1)) & 1);
        const int gy = y0 - 1 + row;
        const int gx = x0 - 1 + hx;
        a_inb[r] = slot < G::A_SLOTS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        a_hi[r] = hq != 0;
        a_off[r] = (unsigned)(((row * W + hx) * a.in_stride + 4 * hq) * 4);
    });
    const float* a_base = in_img + ((ptrdiff_t)(y0 - 1) * W + (x0 - 1)) * a.in_stride;    // wave-uniform
    const float* b_base = a.wpack + (size_t)ntile * a.n_chunks * G::B_FLOATS;             // wave-uniform
    const unsigned b_off = (unsigned)(wave * 1024 + lane * 16);
    const bool tail4 = (a.cin_phys & 7) != 0;                // the last chunk holds 4 channels only

    // filter piece r of this wave (chunk -> filter stage), input piece r (chunk -> input stage)
    auto dma_b = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        glds16(b_base + (size_t)chunk * G::B_FLOATS + 1024 * r, b_off, lds0 + G::B_BASE + stage * G::B_BYTES + (unsigned)(wave + 4 * r) * 1024u);
    };
    auto dma_a = [&](auto r_, int chunk, unsigned stage) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if (wave + 4 * r < G::A_DMA) {                       // wave-uniform
            // channels 4..7 of a 4-channel last chunk do not exist: their slots keep older (finite) data, the filter rows are zero
            const bool last4 = tail4 && chunk == a.n_chunks - 1;
            if (a_inb[r] && !(last4 && a_hi[r])) glds16(a_base + chunk * G::KC, a_off[r], lds0 + stage * G::A_BYTES + (unsigned)(wave + 4 * r) * 1024u);
        }
    };

    // SAME zero padding: the slots of halo pixels outside the image are never written by the DMA (their lanes are masked
    // off) -- the owning lane clears them once, in both stages.  Interior tiles clear nothing.  (Channel-tail slots of
    // the last chunk keep the data of chunk n-3: a Winograd layer has >= 4 chunks, so that is real, finite data.)
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int slot = (wave + 4 * r) * 64 + lane;
        if (!a_inb[r] && slot < G::A_SLOTS) {
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + 16 * slot) = z;
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + G::A_BYTES + 16 * slot) = z;
        }
    });

    f32x4 acc[16][NTV];

    // this lane's Winograd tile: rows 2w, 2w+1 of the 8x8 tile grid, 8 tiles per row
    const int tr = 2 * wave + (lj >> 3);
    const int tc = lj & 7;
    // byte offset of the lane's raw-patch origin inside an input stage: plane pair of channel quad lk>>1, halves by lk&1
    // byte offset of the lane's raw-patch origin inside an input stage.  A pixel record is 32 bytes = two channel quads whose
    // order alternates with (xs + row / 2): the 32 lanes of a ds_read_b64 group (16 tiles x the two halves of ONE quad) then
    // touch, in the two tile rows of the wave, complementary halves of the pixel records -- all 64 banks, no conflict.
    const int sw = (tc + tr) & 1;
    const int a_lane_e = ((2 * tr) * G::ROW_SLOTS + tc) * 32 + (((lk >> 1) ^ sw) * 16) + (lk & 1) * 8;   // patch elements with (i/2 + jj/2) even;
                                                                                                         // odd ones: a_lane_e ^ 16
    const int b_lane = G::B_BASE + (lk * G::NS + lj) * 4;

    // V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], on component C (channel 2k + C) of the raw patch
    auto transform = [&](auto c_, const f32x2 (&rr)[4][4], float (&v)[16]) DCSCN_INL {
        constexpr int C = decltype(c_)::value;
        float r[4][4];
        static_for<0, 4>([&](auto j_) DCSCN_INL {
            constexpr int j = decltype(j_)::value;
            r[0][j] = rr[0][j][C] - rr[2][j][C];
            r[1][j] = rr[1][j][C] + rr[2][j][C];
            r[2][j] = rr[2][j][C] - rr[1][j][C];
            r[3][j] = rr[1][j][C] - rr[3][j][C];
        });
        static_for<0, 4>([&](auto x_) DCSCN_INL {
            constexpr int x = decltype(x_)::value;
            v[4 * x + 0] = r[x][0] - r[x][2];
            v[4 * x + 1] = r[x][1] + r[x][2];
            v[4 * x + 2] = r[x][2] - r[x][1];
            v[4 * x + 3] = r[x][1] - r[x][3];
        });
    };
    // raw-patch element (i, jj) of this lane's tile from an input stage.  volatile: keeps the 8-byte reads single
    // (hipcc would pair them into ds_read2_b64, which has half the LDS rate and the narrow banking)
    typedef const volatile __attribute__((address_space(3))) f32x2* lds_f32x2_ptr;
    // (be, bo): the stage's lane bases for even / odd elements, formed per chunk (two persistent registers would not fit)
    auto read_raw1 = [&](auto i_, auto j_, unsigned be, unsigned bo, f32x2 (&rr)[4][4]) DCSCN_INL {
        constexpr int i = decltype(i_)::value, jj = decltype(j_)::value;
        const unsigned base = (((i >> 1) ^ (jj >> 1)) & 1) ? bo : be;
        rr[i][jj] = *(lds_f32x2_ptr)(uintptr_t)(base + ((jj & 1) * G::PL + i * G::ROW_SLOTS + (jj >> 1)) * 32);
    };
    auto lane_bases = [&](unsigned stage_base, unsigned& be, unsigned& bo) DCSCN_INL {
        be = stage_base + a_lane_e;
        bo = stage_base + (a_lane_e ^ 16);
        asm volatile("" : "+v"(be), "+v"(bo));               // keep them out of the loop-invariant (long-lived) set
    };
    // the 16*NTV MFMAs of one k-step, filter operands read PF frequencies ahead; hook(f) runs after the MFMAs of f
    // filter operands are read as single ds_read_b32 with 16-bit immediate offsets (volatile LDS pointer): paired into
    // ds_read2_b32 -- 8-bit offsets -- every pair needs a v_add for its base, 24 VALU instructions per chunk, and VALU time is
    // matrix time on this chip
    typedef const volatile __attribute__((address_space(3))) float* lds_f32_ptr;
    auto mfma_step = [&](unsigned Bs, const float (&v)[16], auto&& hook) DCSCN_INL {
        float wq[PF + 1][NTV];
        static_for<0, PF>([&](auto p_) DCSCN_INL {
            constexpr int pf = decltype(p_)::value;
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                wq[pf][n] = *(lds_f32_ptr)(uintptr_t)(Bs + ((pf * G::KC) * G::NS + n * 16) * 4);
            });
        });
        static_for<0, 16>([&](auto f_) DCSCN_INL {
            constexpr int f = decltype(f_)::value;
            if constexpr (f + PF < 16) {
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    wq[(f + PF) % (PF + 1)][n] = *(lds_f32_ptr)(uintptr_t)(Bs + (((f + PF) * G::KC) * G::NS + n * 16) * 4);
                });
            }
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[f % (PF + 1)][n], v[f], acc[f][n], 0, 0, 0);
            });
            hook(f_);
        });
    };

    // ---- K loop ----
    // chunk c reads filter stage c&1 (k-steps 0, 1) and, during k-step 1, the raw patch of chunk c+1 from input stage
    // (c+1)&1; it issues the DMA of filters c+1 -> filter stage (c+1)&1 (last read in chunk c-1) and of input c+2 -> input
    // stage c&1 (last read, as a raw patch, during chunk c-1).  The last iterations skip the copies that have no chunk
    // left to fetch (uniform branches; 1 filter block and 2 input tiles per workgroup -- a quarter of the input DMA of an
    // 8-chunk layer); the raw-patch read of the last iteration lands on stale data and is not used.
    const int last = a.n_chunks - 1;
    static_for<0, G::B_ROUNDS>([&](auto r_) DCSCN_INL { dma_b(r_, 0, 0); });
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, 0, 0); });
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL { dma_a(r_, last < 1 ? last : 1, 1); });
    // the accumulators are cleared while the first chunks are in flight
    static_for<0, 16>([&](auto f_) DCSCN_INL {
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            acc[decltype(f_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        });
    });

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x2 rr[4][4];
    {
        unsigned be, bo;
        lane_bases(lds0, be, bo);
        static_for<0, 4>([&](auto i_) DCSCN_INL {
            static_for<0, 4>([&](auto j_) DCSCN_INL { read_raw1(i_, j_, be, bo, rr); });
        });
    }
    __syncthreads();                                          // every wave holds its patch of chunk 0: input stage 0 may be refilled
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        const unsigned sb = chunk & 1;
        const bool more_b = chunk + 1 <= last;               // filters of chunk c+1 / input of chunk c+2 exist (wave uniform)
        const bool more_a = chunk + 2 <= last;
        const unsigned Bs = lds0 + (unsigned)b_lane + sb * G::B_BYTES;
        const unsigned An = lds0 + (sb ^ 1) * G::A_BYTES;
        unsigned be = 0, bo = 0;
        float v[16];
        transform(std::integral_constant<int, 0>{}, rr, v);
        mfma_step(Bs, v, [&](auto f_) DCSCN_INL {
            constexpr int f = decltype(f_)::value;
            // one DMA piece behind each of the first B_ROUNDS + A_ROUNDS frequencies
            if constexpr (f >= 1 && f <= G::B_ROUNDS) {
                if constexpr (ABL != 1 && ABL != 3) { if (more_b) dma_b(std::integral_constant<int, f - 1>{}, chunk + 1, sb ^ 1); }
            } else if constexpr (f > G::B_ROUNDS && f <= G::B_ROUNDS + G::A_ROUNDS) {
                if constexpr (ABL != 1 && ABL != 4) { if (more_a) dma_a(std::integral_constant<int, f - 1 - G::B_ROUNDS>{}, chunk + 2, sb); }
            }
        });
        transform(std::integral_constant<int, 1>{}, rr, v);
        mfma_step(Bs + 4 * G::NS * 4, v, [&](auto f_) DCSCN_INL {
            constexpr int f = decltype(f_)::value;
            // raw patch of the next chunk, two elements behind each of the last 8 frequencies
            if constexpr (f == 7) lane_bases(An, be, bo);
            if constexpr (f >= 8) {
                read_raw1(std::integral_constant<int, (2 * (f - 8)) / 4>{}, std::integral_constant<int, (2 * (f - 8)) % 4>{}, be, bo, rr);
                read_raw1(std::integral_constant<int, (2 * (f - 8) + 1) / 4>{}, std::integral_constant<int, (2 * (f - 8) + 1) % 4>{}, be, bo, rr);
            }
        });
        if constexpr (ABL != 2 && ABL != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (ABL != 5) __syncthreads();
    }
    if constexpr (ABL == 2 || ABL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- output transform (wave-local) + epilogue: identical to conv_wino ----
    const int gy0 = y0 + 2 * tr;
    const int gx0 = x0 + 2 * tc;
    const int cbase = ntile * NT * 16 + 4 * lk;                                  // bias / slope index: padded group layout
    const int obase = cbase - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);    // conv channel: groups past n_full are one tile narrower
    const int act = a.act;
    const int ps = a.ps;
    const int orow = W * ps;                                   // destination pixels per row
    const bool ok_y1 = gy0 + 1 < H, ok_x1 = gx0 + 1 < W;
    const bool ok_00 = gy0 < H && gx0 < W;
    f32x4 bv[NTV], av[NTV];
    static_for<0, NTV>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        bv[n] = *reinterpret_cast<const f32x4*>(a.bias + cbase + n * 16);
        av[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (act == ACT_ALPHA) av[n] = *reinterpret_cast<const f32x4*>(a.alpha + cbase + n * 16);
    });
    float m1 = -1.0f;
    asm volatile("" : "+v"(m1));
    auto finish = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c = obase + n * 16;
            const bool first = c < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc = first ? c : c - a.split;
            int ch = cc, ay = 0, bx = 0;
            if (ps != 1) {                                         // depth_to_space: channel (ay*ps + bx)*ps_c + ch
                const int sub = cc / a.ps_c;
                ch = cc - sub * a.ps_c;
                ay = sub / ps;
                bx = sub - ay * ps;
            }
            const size_t pix00 = (size_t)((img * H + gy0) * ps + ay) * orow + (size_t)(gx0 * ps + bx);
            float* o00 = optr + pix00 * ostride + ooff + ch;
            const size_t dx = (size_t)ps * ostride;                // one LR pixel to the right / down
            const size_t dy = (size_t)ps * orow * ostride;
            const bool live = ok_00 && cc < owidth;
            // t[a][nu] = sum_xi A^T[a][xi] m[xi][nu],  A^T = [1 1 1 0; 0 1 -1 -1]
            // a - b is written as fma(b, -1, a) with a -1 the compiler cannot see through: there is no packed f32 subtract, so
            // `a - b` on float4 becomes four v_sub_f32, while this is two v_pk_fma_f32 with the same (exact) result -- and on
            // this chip every VALU instruction of the epilogue is matrix-pipe time taken from the co-resident workgroup
            f32x4 t0[4], t1[4];
            static_for<0, 4>([&](auto nu_) DCSCN_INL {
                constexpr int nu = decltype(nu_)::value;
                t0[nu] = acc[0 + nu][n] + acc[4 + nu][n] + acc[8 + nu][n];
                t1[nu] = acc[12 + nu][n] * m1 + (acc[8 + nu][n] * m1 + acc[4 + nu][n]);
            });
            f32x4 yv[2][2];
            yv[0][0] = t0[0] + t0[1] + t0[2];
            yv[0][1] = t0[3] * m1 + (t0[2] * m1 + t0[1]);
            yv[1][0] = t1[0] + t1[1] + t1[2];
            yv[1][1] = t1[3] * m1 + (t1[2] * m1 + t1[1]);
            static_for<0, 2>([&](auto pa_) DCSCN_INL {
                static_for<0, 2>([&](auto pb_) DCSCN_INL {
                    constexpr int pa = decltype(pa_)::value, pb = decltype(pb_)::value;
                    f32x4 v = yv[pa][pb] + bv[n];
                    v.x = activate1(v.x, av[n].x, act_e);
                    v.y = activate1(v.y, av[n].y, act_e);
                    v.z = activate1(v.z, av[n].z, act_e);
                    v.w = activate1(v.w, av[n].w, act_e);
                    if (live && (pa == 0 || ok_y1) && (pb == 0 || ok_x1)) {
                        if (a.res) {
                            const size_t pix = pix00 + (size_t)(pa * ps) * orow + (size_t)(pb * ps);
                            v += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch);
                        }
                        *reinterpret_cast<f32x4*>(o00 + pa * dy + pb * dx) = v;
                    }
                });
            });
        });
    };
    if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
}

// launch_bounds' second argument is waves per SIMD = resident 4-wave workgroups per CU
template <int NT, int WPS = 2, int PF = 3, int ABL = 0>
__global__ __launch_bounds__(256, WPS) void conv_wino2(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware decode of the 1-D grid (see conv_wino): the channel groups of one pixel tile get ids that are congruent
    // mod 8 and close together, so they run on one XCD at about the same time and share the input tile in its L2.
    const int G = a.n_groups, S = a.group_span;
    const int tiles8 = (a.N * a.tiles_y * a.tiles_x + 7) >> 3;            // blocks of 8 pixel tiles
    int id = blockIdx.x;
    const int phase_ids = tiles8 * 8 * S;
    const int phase = id / phase_ids;
    id -= phase * phase_ids;
    const int gs = (G - phase * S) < S ? (G - phase * S) : S;               // groups in this phase (the last may be short)
    const int q = id / (8 * gs), r = id - q * 8 * gs;
    if (q >= tiles8) return;                                                // padding ids of a short last phase
    const int ntile = phase * S + (r >> 3);
    const int tile_id = q * 8 + (r & 7);
    if (tile_id >= a.N * a.tiles_y * a.tiles_x) return;
    if (a.redo_check && (a.redo[0] == 0 || a.redo[1 + tile_id / (a.tiles_y * a.tiles_x)] == 0)) return;   // float32 plan: flagged images only
    if (ntile < a.n_full) conv_wino2_body<NT, NT, PF, ABL>(a, smem, tile_id, ntile);                 // block uniform
    else if constexpr (NT >= 2) conv_wino2_body<NT, NT - 1, PF, ABL>(a, smem, tile_id, ntile);
}

// The launch behind conv3_h (split16.hpp): recompute in f32 the pixel tiles whose redo flag is set -- normally none.  A full grid
// of early exits costs the latency of one flag load per ROUND of workgroups (12-59 us per layer, 0.33 ms per pass of the bench
// model); here a workgroup reads the flags of 64 tiles at once (one per lane), leaves if none is set, and otherwise walks the
// flagged tiles and their channel groups.  One workgroup per CU (512 VGPRs allowed): the loop around the body needs more
// scalar registers than the 2-per-CU kernel has room to spill into.
template <int NT, int PF = 3>
__global__ __launch_bounds__(256, 1) void conv_wino2_redo(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;
    const int t0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63;
    if (a.redo[0] == 0) return;                               // no image of the pass was flagged (the normal case)
    const int flag = t0 + lane < n_tiles ? a.redo[1 + (t0 + lane) / (a.tiles_y * a.tiles_x)] : 0;
    unsigned long long mask = __ballot(flag != 0);            // the same in all four waves
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mask), hi = __builtin_amdgcn_readfirstlane((unsigned)(mask >> 32));
    mask = ((unsigned long long)hi << 32) | lo;
    while (mask) {
        const int tile_id = t0 + __builtin_ctzll(mask);
        mask &= mask - 1;
        for (int ntile = 0; ntile < a.n_groups; ++ntile) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // clamped DMAs of the previous body
            __syncthreads();
            if (ntile < a.n_full) conv_wino2_body<NT, NT, PF, 0>(a, smem, tile_id, ntile);
            else if constexpr (NT >= 2) conv_wino2_body<NT, NT - 1, PF, 0>(a, smem, tile_id, ntile);
        }
    }
}

}  // namespace dcscn
