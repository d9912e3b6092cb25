// conv3_h: 3x3 SAME convolution + bias + activator (tf.nn.conv2d, helper/tf_graph.py:104-153) as a DIRECT implicit GEMM on
// v_mfma_f32_16x16x32_f16 at f32 accuracy (split16.hpp: f16 (hi, lo) pieces, 3 products).
//
// Why direct form and not the Winograd domain of conv_wino2: at 1/16 of the f32 matrix cycles per MAC the 2.25x fewer
// multiplies of F(2x2,3x3) no longer pay for what they cost around the MFMAs -- 16 frequencies x (hi, lo) filter fragments are
// 4x the filter bytes per MAC (the kernel would be bound by L2 -> LDS traffic: ~50 B/clk/CU at full matrix rate against
// ~58 available), and the input transform + split is 3.5 VALU per MFMA where 2 are free (profiles/r03_pipe_probe.txt).  The
// direct form reads 0.28-0.33 LDS fragments per MFMA and splits every input value once per workgroup.
//
// GEMM view per tap: D[cout][pixel] += W_tap[cout][cin] * X[cin][pixel + tap offset]; A operand (rows) = filter fragment,
// B operand (columns) = 16 consecutive pixels of one image row, so a lane ends up with 4 consecutive output channels of one
// pixel = one float4 NHWC store (the C/D layout of the f16 instruction is the f32 one's).
//
// * workgroup = 4 waves = 16 x 16 pixels x NT*16 output channels; wave w owns rows 4w..4w+3 (four column tiles).
// * K walks in chunks of 32 input channels.  The chunk's halo tile (18 x 18 pixels) is fetched into REGISTERS while the
//   previous chunk computes (11 dwordx4 loads per thread, 8 adjacent lanes = the 128 contiguous bytes of a pixel), split into
//   f16 (hi, lo) once, and written to LDS as the B-operand image: per pixel eight 16-byte units [kq][hi | lo] of 8 channels,
//   position of unit (kq, part) in the pixel record = 2 * ((kq + (x >> 1)) & 3) + (part ^ (kq & 1)): the 16 lanes of every
//   ds_read_b128 group then hit 16 different bank quads for all three tap columns (no padding: 41,472 bytes).  Halo pixels
//   outside the image and channels past cin are written as zeros: that is the SAME padding.
// * filters: f16 (hi, lo) A fragments packed by the host (split16_pack.hpp: pack_conv16) in read order, one tap = NT * 2 KB,
//   a ring of three tap slots in LDS filled by LDS-DMA (conv_wino2.hpp: glds16) two taps ahead; the wait before a tap's barrier
//   is a COUNTED vmcnt that leaves the next tap's pieces in flight.
//   One barrier per tap, two more per chunk around the write of the input image.  LDS = 41.5 + NT * 6.1 KB: two workgroups per CU.
// * the chunk's input values are converted to (hi, lo) in registers a few per tap while the chunk's later taps compute (the
//   f16 MFMA leaves two VALU issue slots per instruction free, profiles/r03_pipe_probe.txt); only the LDS writes sit between
//   the two barriers of the chunk boundary.
// * taps go column by column (kx outer, ky inner): down a column the wave's four pixel rows move by one row per tap, so a tap reads
//   ONE new B row: per tap and wave 2 (+ 6 at the head of a column) B-fragment reads + 2 NT A-fragment reads for 12 NT MFMAs.
// * a last chunk with at most 8 / 16 / 24 physical channels is PACKED: its (tap, octet) pairs go four to an instruction, 3 / 5 / 7
//   MFMA steps instead of 9 (kernels.h: c3h_tail_octs; the host packs the filters to match).
// * bias and slopes of the channel group are copied to LDS at workgroup start (no global round trip in the epilogue).
// * epilogue: accumulators * 2^-e, bias, activator, optional depth_to_space addressing, float4 stores -- or, for a P16 destination
//   (p16.hpp), one (hi | lo) unit per lane; a non-finite accumulator / hi piece raises the IMAGE's redo flag (split16.hpp).
//
#pragma once
#include "conv_wino2.hpp"
#include "split16.hpp"
#include "p16.hpp"
#ifndef C3H_EXP
#define C3H_EXP 0
#endif

namespace dcscn {

template <int NT>
struct C3HGeom {
    static constexpr int THREADS = 256;
    static constexpr int KC = 32;
    static constexpr int TH = 16, TW = 16;
    static constexpr int HT = 18;                             // halo tile edge
    static constexpr int HP = HT * HT;                        // 324 halo pixels
    static constexpr int PIX_BYTES = 128;                     // 32 channels x (hi, lo) f16
    static constexpr int ROW_BYTES = HT * PIX_BYTES;          // 2304
    static constexpr int IN_BYTES = HP * PIX_BYTES;           // 41472
    static constexpr int IN_ITEMS = HP * 8;                   // (pixel, channel quad) pieces of 16 bytes of f32
    static constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;   // 11
    static constexpr int F_TAP_BYTES = NT * 2048;             // [n][hi | lo][64 lanes][16 bytes]
    static constexpr int F_PIECES = 2 * NT;                   // 1 KB DMA pieces of a tap
    static constexpr int F_ROUNDS = (F_PIECES + 3) / 4;       // DMA instructions per wave and tap (waves without a piece of their own repeat one)
    static constexpr int F_SLOTS = 3;
    static constexpr int F_BASE = IN_BYTES;
    static constexpr int BA_BASE = IN_BYTES + F_SLOTS * F_TAP_BYTES;   // bias | slopes of the channel group (NT * 16 floats each)
    static constexpr int LDS_BYTES = BA_BASE + NT * 128;
};

// position (16-byte unit) of channel group kq, piece `part` (0 hi, 1 lo) inside the record of halo column hx.  Reads (ds_read_b128, 64
// banks = two records): the 16 lanes of a lane group see 16 distinct (column parity, unit) pairs for every tap column.  Writes
// (ds_write_b64, 32 banks = ONE record, 16 lanes = two neighbouring columns x 8 channel quads): the column parity in the low bit puts the
// hi pieces of an odd column where the even column has its lo pieces, so the two columns of a store fill one record's worth of banks
// exactly once (r03 had them on the same banks: 2-way conflicts on every store of the image).
__host__ __device__ constexpr int c3h_unit(int hx, int kq, int part) { return (((kq + ((hx >> 1) & 3)) & 3) << 1) | ((part ^ kq ^ hx) & 1); }

// compile-time proof of the two claims above, for the lane groups of MI355X_MICROARCH.md's LDS table
constexpr bool c3h_reads_conflict_free() {
    // ds_read_b128: four groups of 16 lanes; a lane reads the 16-byte unit of column (lane & 15) + kx, channel group lane >> 4
    constexpr int group[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                  {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int part = 0; part < 2; ++part)
        for (int kx = 0; kx < 5; ++kx)                                // 3x3 taps use 0..2, conv5_h's 5x5 taps 0..4
            for (int g = 0; g < 4; ++g) {
                unsigned seen = 0;                                    // (column parity, unit): 64 banks = two records of 8 units
                for (int i = 0; i < 16; ++i) {
                    const int l = group[g][i], hx = (l & 15) + kx;
                    const unsigned bit = 1u << ((hx & 1) * 8 + c3h_unit(hx, l >> 4, part));
                    if (seen & bit) return false;
                    seen |= bit;
                }
            }
    return true;
}
constexpr bool c3h_writes_conflict_free() {
    // ds_write_b64: groups of 16 contiguous lanes = two neighbouring halo columns x 8 channel quads, 8 bytes each at unit * 16 + (quad & 1) * 8
    for (int part = 0; part < 2; ++part)
        for (int hcol = 0; hcol < 20; hcol += 2) {
            unsigned seen = 0;                                        // 32 banks = one record: 16 slots of 8 bytes
            for (int i = 0; i < 16; ++i) {
                const int col = hcol + (i >> 3), cq = i & 7;
                const unsigned bit = 1u << (c3h_unit(col, cq >> 1, part) * 2 + (cq & 1));
                if (seen & bit) return false;
                seen |= bit;
            }
        }
    return true;
}
static_assert(c3h_reads_conflict_free(), "c3h_unit: a ds_read_b128 lane group must see 16 distinct (column parity, unit) pairs");
static_assert(c3h_writes_conflict_free(), "c3h_unit: the two columns of an image store must fill a record's banks exactly once");

// ABL (tuner only, tools/h16_tune.hip; results are wrong by design): 0 shipped; 1 no convert + write of the input image after the
// first chunk; 2 nor its global loads; 3 no filter staging after the first tap; 4 no per-tap barrier; 5 one MFMA product of three;
// 6 = 2 + 3 + 4 (LDS reads and MFMAs only); 7 / 8 = shipped + shader-clock probes (per wave, through a.srctab: [0] entry, [1] K loop
// start, [2] K loop end, [3] exit, [4] sum over taps of (wait + barrier) [7] or of the DMA / load issue behind it [8], [5] sum of the
// chunk-boundary barrier + image write, [6] HW_ID)
// IN16: the input is a P16 tensor (a.in16, p16.hpp): the staged item is one 16-byte (hi | lo) unit -- fetched with the c3h_unit
// permutation on the source side, written to LDS as it is (ds_write_b128 at the lane-linear position), out-of-image pixels and octets past
// the tensor's last read the plane's zero record: no conversion, no select, half the LDS store instructions.
// Destinations may be P16 tensors (OutDesc::p16), each on its own: the epilogue then stores (hi | lo) units.
template <int NT, int NTV, int ABL = 0, bool IN16 = false>
__device__ __forceinline__ void conv3_h_body(const ConvArgs& a, char* smem, int tile_id, int ntile) {
    using G = C3HGeom<NT>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    constexpr bool PROBE = ABL == 7 || ABL == 8;
    long long pr_t0 = 0, pr_t1 = 0, pr_t2 = 0, pr_sum = 0, pr_cb = 0, pr_a = 0;
    if constexpr (PROBE) pr_t0 = __builtin_readcyclecounter();

    int bid = tile_id;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int y0 = ty * G::TH;
    const int x0 = tx * G::TW;
    const int H = a.H, W = a.W;
    // origin of the halo tile; only in-image addresses are ever dereferenced (out-of-image items read the tile's own first pixel)
    const float* a_base = IN16 ? nullptr : a.in + (size_t)img * H * W * a.in_stride + a.in_off + ((ptrdiff_t)(y0 - 1) * W + (x0 - 1)) * a.in_stride;
    const int pix0 = (img * H + y0 - 1) * W + x0 - 1;         // IN16: flat pixel index of the halo tile's origin (may be negative)

    // ---- staging plan of the input image: item = r * 256 + tid = (halo pixel, channel quad) ----
    const int cq = tid & 7;                                   // channel quad of every item of this thread
    // The halo pixel of item r is hp = r * 32 + (tid >> 3): one row and 14 columns further per item.  Its position is re-derived
    // from tid wherever it is needed (a dozen VALU operations per chunk) instead of living in a register per item.
    unsigned ok_mask = 0;                                     // per item: inside the image
    {
        int hrow = (tid >> 3) >= G::HT ? 1 : 0, hcol = (tid >> 3) - G::HT * hrow;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int gy = y0 - 1 + hrow, gx = x0 - 1 + hcol;
            const bool ok = r * 32 + (tid >> 3) < G::HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            ok_mask |= ok ? (1u << r) : 0u;
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    }
    const bool all_in = __builtin_amdgcn_readfirstlane((int)(y0 >= 1 && x0 >= 1 && y0 + G::TH + 1 <= H && x0 + G::TW + 1 <= W)) != 0;   // no padding in this tile
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* f_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)ntile * a.n_chunks * 9 * G::F_TAP_BYTES;   // wave-uniform
    const unsigned f_off = (unsigned)(lane * 16);

    // bias and slopes of this channel group go to LDS now: read from global memory in the epilogue they are a dependent round trip
    // (1-2 us) at the end of every workgroup -- 10 % of a narrow layer's workgroup, 1-2 % of a wide one's
    if (tid < NTV * 4) *reinterpret_cast<f32x4*>(smem + G::BA_BASE + tid * 16) = reinterpret_cast<const f32x4*>(a.bias + ntile * NT * 16)[tid];
    else if (tid >= 64 && tid < 64 + NTV * 4 && a.act == ACT_ALPHA)
        *reinterpret_cast<f32x4*>(smem + G::BA_BASE + NT * 64 + (tid - 64) * 16) = reinterpret_cast<const f32x4*>(a.alpha + ntile * NT * 16)[tid - 64];

    f32x4 gin[G::IN_ROUNDS];
    auto load_in = [&](int chunk) DCSCN_INL {
        if constexpr (IN16) {
            const int rem = a.in16.octs - 4 * chunk;           // octets of this chunk (block uniform)
            const int rec = rem >= 4 ? 128 : 32 * rem;
            const char* base = a.in16.base + (long long)chunk * a.in16.plane;
            int hp0 = tid >> 3;
            asm volatile("" : "+v"(hp0));
            int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
            static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
                constexpr int r = decltype(r_)::value;
                const int kq = ((cq >> 1) - (hcol >> 1)) & 3;  // the unit c3h_unit puts at slot cq of this halo column
                const int part = (cq ^ kq ^ hcol) & 1;
                const bool ok = ((ok_mask >> r) & 1u) && kq < rem;
                const unsigned off = ok ? 128u + (unsigned)(pix0 + hrow * W + hcol) * (unsigned)rec + (unsigned)((2 * kq + part) * 16) : (unsigned)(cq * 16);
                gin[r] = *reinterpret_cast<const f32x4*>(base + (size_t)off);
                hcol += 32 - G::HT; hrow += 1;
                if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
            });
            return;
        }
        const int c0 = chunk * G::KC + cq * 4;
        const unsigned coff = (unsigned)((c0 < a.cin_phys ? c0 : 0) * 4);   // channels past cin: read something valid, written as zeros
        // wave-uniform 64-bit base + 32-bit lane offset: the loads take the SGPR-base form, no 64-bit pointer per item is kept live
        const char* base = reinterpret_cast<const char*>(a_base);
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));                         // keeps the offsets below out of the registers between chunks
        int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
        const int stride4 = a.in_stride * 4;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int pix = ((ok_mask >> r) & 1u) ? hrow * W + hcol : W + 1;
            gin[r] = *reinterpret_cast<const f32x4*>(base + (size_t)((unsigned)(pix * stride4) + coff));
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    };
    const float m1 = opaque_minus_one();
    // item r of the chunk in flight: f32 values -> (hi, lo) pairs, in place (gin[r] = {hi01, hi23, lo01, lo23})
    auto convert_in = [&](auto r_, int chunk) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if constexpr (IN16) return;
        f32x4 x = gin[r];
        const bool whole = all_in && (chunk + 1) * G::KC <= a.cin_phys;       // block uniform: nothing to zero
        if (!whole) {
            const bool ok = chunk * G::KC + cq * 4 < a.cin_phys && ((ok_mask >> r) & 1u);
            x.x = ok ? x.x : 0.0f; x.y = ok ? x.y : 0.0f; x.z = ok ? x.z : 0.0f; x.w = ok ? x.w : 0.0f;
        }
        h4 hi, lo;
        split4(x, m1, hi, lo);
        const u32x2 hu = __builtin_bit_cast(u32x2, hi), lu = __builtin_bit_cast(u32x2, lo);
        gin[r] = __builtin_bit_cast(f32x4, u32x4{hu.x, hu.y, lu.x, lu.y});
    };
    auto store_in = [&]() DCSCN_INL {
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hcol = hp0 >= G::HT ? hp0 - G::HT : hp0;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int hp = r * 32 + hp0;
            const int kq = cq >> 1;
            const int off = hp * G::PIX_BYTES + c3h_unit(hcol, kq, 0) * 16 + (cq & 1) * 8;
            const u32x4 v = __builtin_bit_cast(u32x4, gin[r]);
            if constexpr (IN16) {
                if (r < G::IN_ROUNDS - 1 || hp < G::HP) *reinterpret_cast<u32x4*>(smem + hp * G::PIX_BYTES + cq * 16) = v;
            } else
            if (r < G::IN_ROUNDS - 1 || hp < G::HP) {
                *reinterpret_cast<u32x2*>(smem + off) = u32x2{v.x, v.y};
                *reinterpret_cast<u32x2*>(smem + (off ^ 16)) = u32x2{v.z, v.w};
            }
            hcol += 32 - G::HT;
            if (hcol >= G::HT) hcol -= G::HT;
        });
    };
    // filter pieces of packed tap `src` (chunk * 9 + ky * 3 + kx in the image) -> ring slot
    auto dma_f = [&](int src, int slot) DCSCN_INL {
        static_for<0, G::F_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int piece = (wave + 4 * r) % G::F_PIECES;
            glds16(f_base + (size_t)src * G::F_TAP_BYTES + piece * 1024, f_off, lds0 + G::F_BASE + slot * G::F_TAP_BYTES + (unsigned)piece * 1024u);
        });
    };

    f32x4 acc[4][NTV];
    static_for<0, 4>([&](auto m_) DCSCN_INL {
        static_for<0, NTV>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
    });

    // B fragment address of tap column kx: pixel (row 4w, column lj + kx) of the halo tile, channel group lk; derived from the
    // lane at the head of each column (a few VALU operations per three taps, no register per column)
    auto b_col = [&](auto kx_) DCSCN_INL {
        constexpr int kx = decltype(kx_)::value;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int hx = (l & 15) + kx;
        return (4 * wave * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, l >> 4, 0) * 16;
    };
    int b_hi = 0;
    const int a_lane = G::F_BASE + lane * 16;

    const int n_chunks = a.n_chunks;
    dma_f(0, 0);                                              // step 0 = tap (ky 0, kx 0), step 1 = tap (ky 1, kx 0)
    dma_f(3, 1);
    load_in(0);
    static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL { convert_in(r_, 0); });
    store_in();
    // K loop.  Per tap: wait for this tap's filter pieces, barrier, refill the slot the previous tap was read from with the tap
    // after next, compute.  The wait is vmcnt(F_ROUNDS): everything but the youngest F_ROUNDS operations has retired.  The
    // LDS-DMA pieces of a wave retire in issue order, so if a piece of this tap were still in flight all F_ROUNDS pieces of the
    // next tap would be too -- more than the count allows; the argument needs no ordering between the DMA pieces and the plain
    // loads of the input values (issued at tap 0 behind that tap's DMA, so the wait of tap 1 retires them: one tap of latency
    // is hidden, the rest is covered by the other workgroup of the CU).
    // The loads, the conversion and the DMA are unconditional (the last chunk re-reads itself, the last taps re-fetch the last
    // tap): no value in the loop depends on a branch, so the compiler keeps ONE register set for the staged values.
    const int octs = a.tail_octs;                              // 0, or 1 / 2 / 3: the last chunk is a packed tail (below)
    const int n_main = octs ? n_chunks - 1 : n_chunks;
    if constexpr (PROBE) pr_t1 = __builtin_readcyclecounter();
    for (int chunk = 0; chunk < n_main; ++chunk) {
        const bool more = chunk + 1 < n_chunks;                // block uniform
        const int nchunk = more ? chunk + 1 : chunk;
        const bool to_tail = octs != 0 && chunk + 1 == n_main; // the steps two ahead of steps 7, 8 are steps 0, 1 of the packed tail
        // The nine taps go column by column (step s: kx = s / 3, ky = s % 3): down a column the four pixel rows of the wave move
        // by one row per tap, so only ONE new row of B fragments is read per tap (rows ky .. ky + 3 live in xh / xl[(ky + m) & 3])
        // -- 12 row reads per column instead of 24; with the 2 NT filter fragments per tap that is 14 LDS reads per 12 NT MFMAs.
        h8 xh[4], xl[4];
        static_for<0, 9>([&](auto s_) DCSCN_INL {
            constexpr int step = decltype(s_)::value;
            constexpr int kx = step / 3, ky = step % 3;
            constexpr int slot = step % 3;                     // (chunk * 9 + step) % 3
            constexpr int step2 = (step + 2) % 9;              // the step two ahead: packed tap (ky2 * 3 + kx2) of this or the next chunk
            constexpr int ptap2 = (step2 % 3) * 3 + step2 / 3;
            if constexpr (ABL == 7) pr_a = __builtin_readcyclecounter();
            if constexpr (ABL != 3 && ABL != 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::F_ROUNDS) : "memory");
            if constexpr ((ABL != 4 && ABL != 6) || step == 0) __syncthreads();
            if constexpr (ABL == 7) pr_sum += __builtin_readcyclecounter() - pr_a;
            if constexpr (ABL == 8) pr_a = __builtin_readcyclecounter();
            if constexpr (ABL != 3 && ABL != 6)             // past the end: a re-fetch nobody reads; the tail's slots are in step order
                dma_f(step + 2 < 9 ? chunk * 9 + ptap2 : nchunk * 9 + (to_tail ? step2 : ptap2), (step + 2) % 3);
            if constexpr (step == 0 && ABL != 2 && ABL != 6) load_in(nchunk);
            if constexpr (ABL == 8) pr_sum += __builtin_readcyclecounter() - pr_a;
            if constexpr (ky == 0) b_hi = b_col(std::integral_constant<int, kx>{});
            static_for<(ky == 0 ? 0 : 3), 4>([&](auto m_) DCSCN_INL {
                constexpr int row = ky + decltype(m_)::value;
                xh[row & 3] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                xl[row & 3] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
            });
            const char* fs = smem + a_lane + slot * G::F_TAP_BYTES;
#if C3H_EXP == 0
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const h8 wh = *reinterpret_cast<const h8*>(fs + (2 * n) * 1024);
                const h8 wl = *reinterpret_cast<const h8*>(fs + (2 * n + 1) * 1024);
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    constexpr int q = (ky + m) & 3;
                    if constexpr (ABL != 5) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[q], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[q], acc[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[q], acc[m][n], 0, 0, 0);
                });
            });
#else
            // experiment (tools/h16_tune, -DC3H_EXP=1/3): products outermost per tile (an accumulator is touched every 4th MFMA);
            // bit 1: the A fragments of tile n + 1 are read into a second register pair before tile n's MFMAs, order pinned
            {
                h8 whb[2], wlb[2];
                whb[0] = *reinterpret_cast<const h8*>(fs);
                wlb[0] = *reinterpret_cast<const h8*>(fs + 1024);
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    constexpr int cb = (C3H_EXP & 2) ? (n & 1) : 0;
                    if constexpr ((C3H_EXP & 2) != 0) {
                        if constexpr (n + 1 < NTV) {
                            whb[(n + 1) & 1] = *reinterpret_cast<const h8*>(fs + (2 * n + 2) * 1024);
                            wlb[(n + 1) & 1] = *reinterpret_cast<const h8*>(fs + (2 * n + 3) * 1024);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        }
                    } else if constexpr (n > 0) {
                        whb[0] = *reinterpret_cast<const h8*>(fs + (2 * n) * 1024);
                        wlb[0] = *reinterpret_cast<const h8*>(fs + (2 * n + 1) * 1024);
                    }
                    static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlb[cb], xh[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                    static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whb[cb], xl[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                    static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whb[cb], xh[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                    if constexpr ((C3H_EXP & 2) != 0) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                });
            }
#endif
            // the next chunk's input values become (hi, lo) pairs two items per tap from step 3 on
            if constexpr (step >= 3 && ABL != 1 && ABL != 2 && ABL != 6)
                static_for<2 * (step - 3), (2 * (step - 3) + 2 < G::IN_ROUNDS ? 2 * (step - 3) + 2 : G::IN_ROUNDS)>([&](auto r_) DCSCN_INL { convert_in(r_, nchunk); });
        });
        if constexpr (ABL != 1 && ABL != 2 && ABL != 6) {
            if (more) {
                if constexpr (PROBE) pr_a = __builtin_readcyclecounter();
                __syncthreads();                              // every wave is past its last read of this chunk's image
                store_in();                                   // made visible by the barrier in front of the next tap
                if constexpr (PROBE) pr_cb += __builtin_readcyclecounter() - pr_a;
            }
        }
    }
    // Packed tail (kernels.h: c3h_tail_octs): the last chunk holds at most 8 / 16 / 24 channels = 1 / 2 / 3 octets, so its
    // (tap, octet) pairs go four to a K = 32 instruction -- lane group lk multiplies pair 4 step + lk = (tap, octet); the host
    // packed the filters to match (split16_pack.hpp) and pairs past the last are zero filters on any valid pixel.
    // 3 / 5 / 7 MFMA steps instead of 9 for the chunk that is mostly padding.
    if (octs) {
        const int n_steps = (9 * octs + 3) >> 2;
        const int tail0 = n_main * 9;
        int l = lane;
        asm volatile("" : "+v"(l));
        for (int step = 0; step < n_steps; ++step) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::F_ROUNDS) : "memory");
            __syncthreads();
            const int slot = step % 3;                         // (tail0 + step) % 3
            dma_f(tail0 + (step + 2 < n_steps ? step + 2 : n_steps - 1), (step + 2) % 3);
            const int pair = 4 * step + (l >> 4);
            int tap = octs == 1 ? pair : octs == 2 ? pair >> 1 : (pair * 11) >> 5;      // pair / octs for pair < 36
            const int oct = pair - tap * octs;
            tap = tap < 8 ? tap : 8;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int hx = (l & 15) + kx;
            const int b = ((4 * wave + ky) * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, oct, 0) * 16;
            h8 xh[4], xl[4];
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                xh[m] = *reinterpret_cast<const h8*>(smem + b + m * G::ROW_BYTES);
                xl[m] = *reinterpret_cast<const h8*>(smem + (b ^ 16) + m * G::ROW_BYTES);
            });
            const char* fs = smem + a_lane + slot * G::F_TAP_BYTES;
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const h8 wh = *reinterpret_cast<const h8*>(fs + (2 * n) * 1024);
                const h8 wl = *reinterpret_cast<const h8*>(fs + (2 * n + 1) * 1024);
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[m], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[m], acc[m][n], 0, 0, 0);
                });
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clamped re-fetches of the last two taps
    if constexpr (PROBE) pr_t2 = __builtin_readcyclecounter();

    // ---- epilogue ----
    const int cbase = ntile * NT * 16 + 4 * lk;                                  // bias / slope index: padded group layout
    const int obase = cbase - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);    // conv channel: groups past n_full are one tile narrower
    const int act = a.act;
    const int ps = a.ps;
    const int orow = W * ps;
    const float inv = a.inv_scale;
    const float zero = opaque_zero();
    float chk = 0.0f;
    const int gx = x0 + lj;
    auto finish = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c = obase + n * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + (n * 4 + lk) * 16);
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            if (act_e == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + NT * 64 + (n * 4 + lk) * 16);
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
            const bool live = gx < W && cc < owidth;
            if (!a.vec4) {
                // block uniform: destinations without the 16-byte store form (a pixel shuffler to fewer than 4 channels per sub-pixel -- the
                // x3 stage of the c-DCSCN nets, 32 -> 9 -- or a width that is no multiple of 4): one store per channel
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const int gy = y0 + 4 * wave + m;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act_e);
                    v.y = activate1(v.y, av.y, act_e);
                    v.z = activate1(v.z, av.z, act_e);
                    v.w = activate1(v.w, av.w, act_e);
                    if (gx < W && gy < H) {
                        chk = nonfinite_acc(chk, acc[m][n], zero);
                        static_for<0, 4>([&](auto i_) DCSCN_INL {
                            constexpr int i = decltype(i_)::value;
                            const int cci = cc + i;
                            if (cci < owidth) {
                                int chi = cci, ayi = 0, bxi = 0;
                                if (ps != 1) {
                                    const int sub = cci / a.ps_c;
                                    chi = cci - sub * a.ps_c;
                                    ayi = sub / ps;
                                    bxi = sub - ayi * ps;
                                }
                                const size_t pix = (size_t)((img * H + gy) * ps + ayi) * orow + (size_t)(gx * ps + bxi);
                                float r = i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
                                if (a.res) r += a.res[pix * a.res_stride + chi];
                                optr[pix * ostride + ooff + chi] = r;
                            }
                        });
                    }
                });
                return;
            }
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                const int gy = y0 + 4 * wave + m;
                f32x4 v = acc[m][n] * inv + bv;
                v.x = activate1(v.x, av.x, act_e);
                v.y = activate1(v.y, av.y, act_e);
                v.z = activate1(v.z, av.z, act_e);
                v.w = activate1(v.w, av.w, act_e);
                if (live && gy < H) {
                    chk = nonfinite_acc(chk, acc[m][n], zero);
                    const size_t pix = (size_t)((img * H + gy) * ps + ay) * orow + (size_t)(gx * ps + bx);
                    if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch);
                    *reinterpret_cast<f32x4*>(optr + pix * ostride + ooff + ch) = v;
                }
            });
        });
    };
    // the common case -- whole tile inside the image, plain NHWC destination(s) split on a tile boundary, PReLU or no activator -- with
    // ONE 64-bit base per destination tile and 32-bit lane offsets (conv3_h8's epilogue: ~450 VALU instead of ~3000)
    auto finish_fast = [&](auto act_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        const int cb16 = ntile * NT * 16 - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);
        int le = lane;
        asm volatile("" : "+v"(le));
        const int lje = le & 15, lke = le >> 4;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c0 = cb16 + n * 16;
            const bool first = c0 < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc0 = first ? c0 : c0 - a.split;
            char* base = reinterpret_cast<char*>(optr + ((size_t)(img * H + y0) * W + x0) * ostride + ooff + cc0);
            const unsigned voff = (unsigned)(((4 * wave * W + lje) * ostride + 4 * lke) * 4);
            const unsigned rowb = (unsigned)(W * ostride * 4);
            const bool chan_ok = cc0 + 4 * lke < owidth;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + (n * 4 + lke) * 16);
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (ACT_C == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + NT * 64 + (n * 4 + lke) * 16);
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                f32x4 v = acc[m][n] * inv + bv;
                if constexpr (ACT_C == ACT_ALPHA) {
                    v.x = v.x > 0.0f ? v.x : av.x * v.x;
                    v.y = v.y > 0.0f ? v.y : av.y * v.y;
                    v.z = v.z > 0.0f ? v.z : av.z * v.z;
                    v.w = v.w > 0.0f ? v.w : av.w * v.w;
                }
                if (chan_ok) chk = nonfinite_acc(chk, acc[m][n], zero);
                if (chan_ok) *reinterpret_cast<f32x4*>(base + (size_t)(voff + m * rowb)) = v;
            });
        });
    };
    // P16 destination(s) (no residual; depth_to_space when a sub-pixel takes whole tiles): one (hi | lo) unit per lane and tile, one 64-bit base per
    // tile and 32-bit lane offsets; a float32 destination beside a P16 one takes the plain store
    auto finish16 = [&](auto act_c, auto mask_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        constexpr bool MASK = decltype(mask_c)::value;
        const h2 zero2 = p16_opaque_zero2();
        const int cb16 = ntile * NT * 16 - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);
        int le = lane;
        asm volatile("" : "+v"(le));
        const int lje = le & 15, lke = le >> 4;
        const bool col_ok = !MASK || x0 + lje < W;
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c0 = cb16 + n * 16;
            const bool first = c0 < a.split;
            const OutDesc& od = first ? a.out0 : a.out1;
            const int cc0 = first ? c0 : c0 - a.split;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + (n * 4 + lke) * 16);
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ACT_C == ACT_ALPHA || (ACT_C < 0 && act == ACT_ALPHA)) av = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + NT * 64 + (n * 4 + lke) * 16);
            char* base;
            unsigned voff, rowb;
            bool chan_ok;
            const bool o16 = od.p16.base != nullptr;              // block uniform
            if (o16) {
                // depth_to_space (ps_c a multiple of 16: the tile lies in ONE sub-pixel (ay, bx)): pixel (y, x) -> (y ps + ay, x ps + bx) of the
                // ps-times larger map; the lane offsets stay 32-bit
                int ch0 = cc0, ay = 0, bx = 0;
                if (ps != 1) {
                    const int sub = cc0 / a.ps_c;
                    ch0 = cc0 - sub * a.ps_c;
                    ay = sub / ps;
                    bx = sub - ay * ps;
                }
                const int oct0 = (od.off + ch0) >> 3;
                const int chunk = oct0 >> 2, rem = od.p16.octs - 4 * chunk;
                const int rec = rem >= 4 ? 128 : 32 * rem;
                base = od.p16.base + (long long)chunk * od.p16.plane + 128 + ((long long)((img * H + y0) * ps + ay) * orow + x0 * ps + bx) * rec + (oct0 & 3) * 32;
                voff = (unsigned)((4 * wave * ps * orow + lje * ps) * rec + lke * 16);
                rowb = (unsigned)(ps * orow * rec);
                chan_ok = col_ok && oct0 + (lke >> 1) < od.p16.octs && ay < ps;
            } else {
                base = reinterpret_cast<char*>(od.ptr + ((size_t)(img * H + y0) * W + x0) * od.stride + od.off + cc0);
                voff = (unsigned)(((4 * wave * W + lje) * od.stride + 4 * lke) * 4);
                rowb = (unsigned)(W * od.stride * 4);
                chan_ok = col_ok && cc0 + 4 * lke < od.width;
            }
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                f32x4 v = acc[m][n] * inv + bv;
                if constexpr (ACT_C == ACT_ALPHA) {
                    v.x = v.x > 0.0f ? v.x : av.x * v.x;
                    v.y = v.y > 0.0f ? v.y : av.y * v.y;
                    v.z = v.z > 0.0f ? v.z : av.z * v.z;
                    v.w = v.w > 0.0f ? v.w : av.w * v.w;
                } else if constexpr (ACT_C < 0) {
                    v.x = activate1(v.x, av.x, act);
                    v.y = activate1(v.y, av.y, act);
                    v.z = activate1(v.z, av.z, act);
                    v.w = activate1(v.w, av.w, act);
                }
                const bool row_ok = !MASK || y0 + 4 * wave + m < H;
                if (o16) {
                    if constexpr (ACT_C < 0) { if (chan_ok && row_ok) chk = nonfinite_acc(chk, acc[m][n], zero); }   // (a saturating activator hides a non-finite accumulator)
                    const u32x4 unit = p16_unit(v, m1, chk, zero2, chan_ok && row_ok);
                    if (chan_ok && row_ok) *reinterpret_cast<u32x4*>(base + (size_t)(voff + m * rowb)) = unit;
                } else {
                    if (chan_ok && row_ok) chk = nonfinite_acc(chk, acc[m][n], zero);
                    if (chan_ok && row_ok) *reinterpret_cast<f32x4*>(base + (size_t)(voff + m * rowb)) = v;
                }
            });
        });
    };
    const bool any16 = a.out0.p16.base != nullptr || a.out1.p16.base != nullptr;                                    // block uniform
    const bool whole_tile = y0 + G::TH <= H && x0 + G::TW <= W;
    if (any16) {
        if (whole_tile && act == ACT_ALPHA) finish16(std::integral_constant<int, ACT_ALPHA>{}, std::false_type{});
        else if (whole_tile && act == ACT_NONE) finish16(std::integral_constant<int, ACT_NONE>{}, std::false_type{});
        else if (act == ACT_ALPHA) finish16(std::integral_constant<int, ACT_ALPHA>{}, std::true_type{});
        else finish16(std::integral_constant<int, -1>{}, std::true_type{});
        if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }
        return;
    }
    const bool fast = ps == 1 && a.res == nullptr && (a.split & 15) == 0 && whole_tile && a.vec4;   // block uniform
    if (fast && act == ACT_ALPHA) finish_fast(std::integral_constant<int, ACT_ALPHA>{});
    else if (fast && act == ACT_NONE) finish_fast(std::integral_constant<int, ACT_NONE>{});
    else if (act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
    else if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
    if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }     // the image goes to the float32 plan (exec.hip)
    if constexpr (PROBE) {
        if (lane == 0 && a.srctab) {
            long long* pr = reinterpret_cast<long long*>(const_cast<void*>(a.srctab)) + ((size_t)blockIdx.x * 4 + wave) * 8;
            unsigned hw;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            pr[7] = xcc;
            pr[0] = pr_t0; pr[1] = pr_t1; pr[2] = pr_t2; pr[3] = __builtin_readcyclecounter(); pr[4] = pr_sum; pr[5] = pr_cb; pr[6] = hw;
        }
    }
}

// 1-D grid decoded as conv_wino2's: the channel groups of one pixel tile get ids that are congruent mod 8 and close together
// (same XCD, about the same time: the input tile is shared through that XCD's L2)
template <int NT, int WPS = 2, int ABL = 0, bool IN16 = false>
__global__ __launch_bounds__(256, WPS) void conv3_h(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_c3h[];
    const int Gn = a.n_groups, S = a.group_span;
    const int tiles8 = (a.N * a.tiles_y * a.tiles_x + 7) >> 3;
    int id = blockIdx.x;
    const int phase_ids = tiles8 * 8 * S;
    const int phase = id / phase_ids;
    id -= phase * phase_ids;
    const int gs = (Gn - phase * S) < S ? (Gn - phase * S) : S;
    const int q = id / (8 * gs), r = id - q * 8 * gs;
    if (q >= tiles8) return;
    const int ntile = phase * S + (r >> 3);
    const int tile_id = q * 8 + (r & 7);
    if (tile_id >= a.N * a.tiles_y * a.tiles_x) return;
    if (ntile < a.n_full) conv3_h_body<NT, NT, ABL, IN16>(a, smem_c3h, tile_id, ntile);                 // block uniform
    else if constexpr (NT >= 2) conv3_h_body<NT, NT - 1, ABL, IN16>(a, smem_c3h, tile_id, ntile);
}

}  // namespace dcscn
