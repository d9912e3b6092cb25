// conv3_hc: conv3_h8's workgroup (one persistent 8-wave workgroup per CU, P16 image staged by LDS-DMA and double buffered, the halves
// running their LOAD and COMPUTE parts in opposite order between ONE barrier per step, epilogue inside the next item's first LOAD part)
// for 3x3 layers with ONE channel group (conv3_h's arithmetic, helper/tf_graph.py:104-153): the group's channel tiles are divided
// between the halves (C0 + C1 tiles) and a step is a whole tap COLUMN (kx; ky = 0 .. 2), not a tap.
//
// Why.  The one-group layers (L12: CNN8 .. CNN12, B2; every 3x3 layer of L8; the narrow nets' shuffler convs) ran on conv3_h: two
// free-running 4-wave workgroups per CU whose prologue (first image: global load -> LDS, nothing to overlap), epilogue and
// chunk-boundary barriers are 25-40 % of a workgroup's life at these depths (profiles/r04_conv3_h_probe.txt: CNN12 11.3 k + 7.2 k of
// 48 k cycles) -- matrix pipe 0.28 .. 0.67 busy.  conv3_h8 hides all of that, but with the tiles of ONE group divided between its
// halves a wave has 6 NT MFMAs per tap against the same barrier, filter DMA and B-row reads, and half 1 has one SEGMENT of latency
// budget for its filter pieces: the segment cannot be shorter than an L2 -> LDS DMA round trip (~2 k cycles measured, whatever NT).
// A column step carries three taps' MFMAs (36 per tile) between the same barrier pair: the fixed costs per MFMA fall by three and a
// segment is longer than the DMA latency by itself.  (Two PIXEL tiles per workgroup on one filter ring -- VERDICT r05 item 1 as
// written -- needs two double-buffered images: 4 x 41.5 KB, more than the CU's 160 KB of LDS; single-buffered images give the
// LDS-DMA staging and the overlap across chunk boundaries back.  DESIGN.md 3.2.)
//
// Same products in the same order as conv3_h (taps column by column, packed last chunk, wl.xh, wh.xl, wh.xh per accumulator):
// BIT-IDENTICAL outputs (tools/hc_tune.hip; tests/test_hip_parity.py::test_conv3_hc_is_bit_identical_to_conv3_h).
//
// * filters: per half a ring of three GROUP slots (group = the three taps of a column, or up to three steps of the packed tail:
//   3 x C0 x 2 KB), filled two groups ahead by LDS-DMA: half 0 in its LOAD part, half 1 right behind the barrier that ends its LOAD part
//   (conv3_h8.hpp: DMA_LATE); counted vmcnt waits (below).
// * image: conv3_h8's P16 staging -- 41 DMA pieces of 1 KB per 32-channel chunk, lane -> unit permutation on the source side -- all six
//   rounds of a wave in the SECOND segment of the phase before (a phase = the three groups of a main chunk, or the packed tail's
//   groups): the other buffer is read until the barrier that ends the phase's first segment (half 1's last COMPUTE of the phase
//   before runs there).  A phase of ONE group (a packed tail of three steps: cin = 32 k + 1 .. 8) has no second segment: the next
//   item's first image is then fetched at the head of that item, under the previous item's epilogue, behind one extra barrier.
// * B rows: a ring of five (rows ky .. ky + 3 of a tap; row ky + 4 is read during tap ky); A fragments two tiles ahead.
//
// vmcnt bookkeeping (in-order retirement; F = filter DMA instructions per wave and group, 6 = image rounds):
//   half 0, end of COMPUTE(g): group g + 1's pieces (issued in LOAD(g - 1)) must have landed.  Younger: LOAD(g)'s F pieces and, when g is
//     the phase's second group, the six image rounds issued behind them  ->  vmcnt(F) / vmcnt(F + 6); at the phase's LAST group the
//     image rounds must have landed as well: vmcnt(F) when they are older than LOAD(g)'s pieces (three-group phase), vmcnt(0) when
//     they were issued in this very LOAD part (two-group tail).
//   half 1, end of LOAD(g): the pieces issued behind the previous barrier (head of COMPUTE(g - 1)).  Younger: the six image rounds when
//     g - 1 was the phase's first group  ->  vmcnt(6) then, unless g is the phase's last group (two-group tail): vmcnt(0).
#pragma once
#include "../dcscn-super-resolution_amd/csrc/conv3_h8.hpp"

#ifndef C3C_ABL
#define C3C_ABL 0          // tuner only (results wrong): 1 no filter DMA, 2 no image staging, 4 no epilogue stores, 8 no MFMAs, 16 no LDS operand reads
#endif

namespace dcscn {

template <int C0, int R>
struct C3CGeom {
    static constexpr int THREADS = 512;
    static constexpr int KC = 32, TH = 16, TW = 16, HT = 18, HP = HT * HT;
    static constexpr int PIX_BYTES = 128, ROW_BYTES = HT * PIX_BYTES;
    static constexpr int IN_BUF = 41 * 1024;                      // an image buffer: 41 DMA pieces of 1 KB
    static constexpr int L = 6;                                   // image rounds per wave and chunk
    static constexpr int GRP_BYTES = 3 * C0 * 2048;               // one group of one half: [step][tile][hi | lo][64 lanes][16 bytes]
    static constexpr int F_ROUNDS = (6 * C0 + 3) / 4;             // DMA instructions per wave and group
    static constexpr int F_BASE = 2 * IN_BUF;                     // [half][slot]
    static constexpr int BA_BASE = F_BASE + 2 * R * GRP_BYTES;    // [parity][half][bias | slopes]: C0 * 128 bytes each
    static constexpr int LDS_BYTES = BA_BASE + 4 * C0 * 128;
};

template <int F>
__device__ __forceinline__ void c3c_wait_sel(int n) {             // n in {0, 6, F, F + 6} (wave uniform)
    if (n == 0) c3p_wait_vm<0>();
    else if (n == 6) c3p_wait_vm<6>();
    else if (n == F) c3p_wait_vm<F>();
    else c3p_wait_vm<F + 6>();
}

// C0 / C1 = channel tiles of half 0 / half 1 (C1 = C0 or C0 - 1, >= 1); every group of the launch holds C0 + C1 tiles (a.nt_pack).
// Input and destinations are P16 tensors (p16.hpp), no depth_to_space, no residual.
template <int C0, int C1, int R = 3>
__global__ __launch_bounds__(512, 2) void conv3_hc(const ConvArgs a) {
    static_assert(C1 >= 1 && (C1 == C0 || C1 == C0 - 1), "half 1 takes as many tiles as half 0 or one fewer");
    static_assert(R == 3, "the ring protocol below is the three-slot one");
    using G = C3CGeom<C0, R>;
    extern __shared__ __attribute__((aligned(16))) char smem_c3c[];
    char* const smem = smem_c3c;
    constexpr int F = G::F_ROUNDS, L = G::L;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;
    const int H = a.H, W = a.W;
    const int n_groups = a.n_groups, ntp = a.nt_pack;
    const int n_units = a.N * a.tiles_y * a.tiles_x * n_groups;
    const int n_chunks = a.n_chunks;
    const int octs = a.tail_octs;
    const int n_main = octs ? n_chunks - 1 : n_chunks;           // >= 1 (launcher)
    const int n_tail = (9 * octs + 3) >> 2;                       // packed steps: 0, 3, 5, 7
    const int g_tail = (n_tail + 2) / 3;                          // their groups: 0, 1, 2, 3
    const int g_total = n_main * 3 + g_tail;                      // groups of an item (>= 3)
    const bool late_mode = g_tail == 1;                           // the last phase has no second segment (header)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned f_off = (unsigned)(lane * 16);
    const int tap_stride = ntp * 2048;                            // bytes between taps of a group's filter image

    struct Unit {
        int valid, img, y0, x0, g, pix0;
        bool full;
        const char* f_base;
        unsigned ok_mask;
    };
    auto decode = [&](int id, Unit& u) DCSCN_INL {
        u.valid = id < n_units;
        const int idc = u.valid ? id : 0;
        const int tile_id = idc / n_groups;
        u.g = idc - tile_id * n_groups;
        int bid = tile_id;
        const int tx = bid % a.tiles_x;
        bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        u.img = bid / a.tiles_y;
        u.y0 = ty * G::TH; u.x0 = tx * G::TW;
        u.full = u.y0 + G::TH <= H && u.x0 + G::TW <= W;
        u.pix0 = (u.img * H + u.y0 - 1) * W + u.x0 - 1;
        u.f_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)u.g * n_chunks * 9 * tap_stride + (half ? C0 * 2048 : 0);
        unsigned m = 0;
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hrow = hp0 / G::HT, hcol = hp0 - G::HT * hrow;
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int gy = u.y0 - 1 + hrow, gx = u.x0 - 1 + hcol;
            bool ok = r * 64 + hp0 < G::HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            if constexpr (r == L - 1) {                            // every wave fetches DMA piece 40 (halo pixels 320 + (lane >> 3)) of the image
                const int hp = 320 + (hp0 & 7), hr = hp / G::HT, hc = hp - G::HT * hr;
                const int qy = u.y0 - 1 + hr, qx = u.x0 - 1 + hc;
                ok = hp < G::HP && qy >= 0 && qy < H && qx >= 0 && qx < W;
            }
            m |= ok ? (1u << r) : 0u;
            hcol += 64 - 3 * G::HT; hrow += 3;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
        u.ok_mask = m;
    };
    // conv3_h8's P16 staging: DMA piece r of this wave = 8 halo pixels x 8 units; out-of-image pixels and octets past the tensor's last
    // come from the plane's zero record: every lane always issues
    auto img_piece = [&](auto r_, int pix0, unsigned mask, int chunk, int buf) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int rem = a.in16.octs - 4 * chunk;
        const int rec = rem >= 4 ? 128 : 32 * rem;
        const char* base = a.in16.base + (long long)chunk * a.in16.plane;
        int t8 = tid >> 3;
        asm volatile("" : "+v"(t8));
        const int hp = r < L - 1 ? r * 64 + t8 : 320 + (t8 & 7);
        const int hrow = hp / G::HT, hcol = hp - G::HT * hrow;
        const int s = lane & 7;
        const int kq = ((s >> 1) - (hcol >> 1)) & 3;
        const int part = (s ^ kq ^ hcol) & 1;
        const bool ok = ((mask >> r) & 1u) && kq < rem;
        const unsigned voff = ok ? 128u + (unsigned)(pix0 + hrow * W + hcol) * (unsigned)rec + (unsigned)((2 * kq + part) * 16) : (unsigned)(s * 16);
        const int piece = r < L - 1 ? wave + 8 * r : 40;
        if constexpr (C3C_ABL & 2) return;
        glds16c(base, voff, lds0 + (unsigned)(buf * G::IN_BUF + piece * 1024));
    };
    auto img_rounds = [&](int pix0, unsigned mask, int chunk, int buf) DCSCN_INL {
        static_for<0, L>([&](auto r_) DCSCN_INL { img_piece(r_, pix0, mask, chunk, buf); });
    };

    auto run = [&](auto cnt_c, auto half_c) DCSCN_INL {
    constexpr int CNT = decltype(cnt_c)::value;                   // channel tiles of THIS half
    constexpr int HALF = decltype(half_c)::value;
    constexpr int PFD = 2, NB = 3;                                // A fragments two (tap, tile) pairs ahead, ring of three
    constexpr int TAPB = CNT * 2048;                              // one step of this half inside a group slot

    // group gi of unit u (gi may run past the item: then it is a group of the next unit): source of its first step, bytes between steps, steps
    auto group_src = [&](const Unit& u, const Unit& un, int gi, int& tstep, int& nsteps) DCSCN_INL -> const char* {
        const Unit& w = gi < g_total ? u : un;
        const int g = gi < g_total ? gi : gi - g_total;
        if (g < n_main * 3) {
            const int c = g / 3, kx = g - 3 * c;
            tstep = 3 * tap_stride; nsteps = 3;
            return w.f_base + (size_t)(c * 9 + kx) * tap_stride;
        }
        const int j = g - n_main * 3;
        tstep = tap_stride;
        nsteps = n_tail - 3 * j < 3 ? n_tail - 3 * j : 3;
        return w.f_base + (size_t)(n_main * 9 + 3 * j) * tap_stride;
    };
    auto dma_group = [&](const Unit& u, const Unit& un, int gi, int slot) DCSCN_INL {
        int tstep, nsteps;
        const char* src = group_src(u, un, gi, tstep, nsteps);
        const int pieces = nsteps * 2 * CNT;
        if constexpr (C3C_ABL & 1) return;
        static_for<0, F>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            int p = w4 + 4 * r;
            p = p < pieces ? p : pieces - 1;                       // (a wave without a piece of its own repeats the last one)
            const int i = p / (2 * CNT), q = p - i * 2 * CNT;
            glds16c(src + (size_t)i * tstep + q * 1024, f_off, lds0 + (unsigned)(slot + i * TAPB + q * 1024));
        });
    };

    Unit cur, nxt;
    int id = blockIdx.x;
    decode(id, cur);
    if (!cur.valid) return;
    decode(id + (int)gridDim.x, nxt);
    int ibuf = 0;
    int sl0 = G::F_BASE + (half * R + 0) * G::GRP_BYTES, sl1 = G::F_BASE + (half * R + 1) * G::GRP_BYTES, sl2 = G::F_BASE + (half * R + 2) * G::GRP_BYTES;
    dma_group(cur, nxt, 0, sl0);
    dma_group(cur, nxt, 1, sl1);
    img_rounds(cur.pix0, cur.ok_mask, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c3p_barrier();
    if (HALF == 1) asm volatile("s_setprio 1");

    f32x4 acc[4][CNT];
    h8 xh[5], xl[5], wa[NB], wb[NB];
    bool pending = false, first_item = true;
    int parity = 0;
    int e_img = 0, e_y0 = 0, e_x0 = 0, e_g = 0;
    bool e_full = false;
    const float m1 = opaque_minus_one();

    auto epilogue = [&]() DCSCN_INL {
        const int ba = G::BA_BASE + ((parity ^ 1) * 2 + half) * C0 * 128;
        const int cb16 = e_g * ntp * 16 + (half ? C0 * 16 : 0);    // conv channel of the half's first tile (every group holds ntp tiles)
        const float inv = a.inv_scale;
        const float zero = opaque_zero();
        float chk = 0.0f;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int lje = le & 15, lke = le >> 4;
        const h2 zero2 = p16_opaque_zero2();
        // P16 destinations (conv3_h8.hpp): every lane stores ONE 16-byte unit per accumulator tile
        auto finish = [&](auto act_c, auto mask_c) DCSCN_INL {
            constexpr int ACT_C = decltype(act_c)::value;
            constexpr bool MASK = decltype(mask_c)::value;
            const bool col_ok = !MASK || e_x0 + lje < W;
            static_for<0, CNT>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const int c0 = cb16 + n * 16;
                const bool first = c0 < a.split;
                const P16Desc& od = first ? a.out0.p16 : a.out1.p16;
                const int oct0 = ((first ? a.out0.off : a.out1.off) + (first ? c0 : c0 - a.split)) >> 3;
                const int chunk = oct0 >> 2, rem = od.octs - 4 * chunk;
                const int rec = rem >= 4 ? 128 : 32 * rem;
                char* base = od.base + (long long)chunk * od.plane + 128 + (long long)((e_img * H + e_y0) * W + e_x0) * rec + (oct0 & 3) * 32;
                const unsigned voff = (unsigned)((4 * w4 * W + lje) * rec + lke * 16);
                const unsigned rowb = (unsigned)(W * rec);
                const bool chan_ok = col_ok && oct0 + (lke >> 1) < od.octs;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + ba + (n * 4 + lke) * 16);
                f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
                if (ACT_C == ACT_ALPHA || (ACT_C < 0 && a.act == ACT_ALPHA)) av = *reinterpret_cast<const f32x4*>(smem + ba + C0 * 64 + (n * 4 + lke) * 16);
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const bool live = chan_ok && (!MASK || e_y0 + 4 * w4 + m < H);
                    f32x4 v = acc[m][n] * inv + bv;
                    if constexpr (ACT_C == ACT_ALPHA) {
                        v.x = v.x > 0.0f ? v.x : av.x * v.x;
                        v.y = v.y > 0.0f ? v.y : av.y * v.y;
                        v.z = v.z > 0.0f ? v.z : av.z * v.z;
                        v.w = v.w > 0.0f ? v.w : av.w * v.w;
                    } else if constexpr (ACT_C < 0) {
                        if (live) chk = nonfinite_acc(chk, acc[m][n], zero);     // (a saturating activator hides a non-finite accumulator)
                        v.x = activate1(v.x, av.x, a.act);
                        v.y = activate1(v.y, av.y, a.act);
                        v.z = activate1(v.z, av.z, a.act);
                        v.w = activate1(v.w, av.w, a.act);
                    }
                    const u32x4 unit = p16_unit(v, m1, chk, zero2, live);
                    if (live && (!(C3C_ABL & 4) || unit.x == 0x12345u)) *reinterpret_cast<u32x4*>(base + (size_t)(voff + m * rowb)) = unit;
                });
            });
        };
        if (e_full) {
            if (a.act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{}, std::false_type{});
            else if (a.act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{}, std::false_type{});
            else finish(std::integral_constant<int, -1>{}, std::false_type{});
        } else {
            if (a.act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{}, std::true_type{});
            else finish(std::integral_constant<int, -1>{}, std::true_type{});
        }
        if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + e_img] = 1; }     // the image goes to the float32 plan (exec.hip)
        pending = false;
    };
    auto barrier_here = [&]() DCSCN_INL {
        __builtin_amdgcn_sched_barrier(0);                        // MFMAs have no memory effect: without this the scheduler moves some across the barrier
        c3p_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // 12 MFMAs of one (step, tile): the three products of the four pixel rows, in conv3_h's order
    auto tile_mfmas = [&](auto n_, auto j_, const h8& x0h, const h8& x0l, const h8& x1h, const h8& x1l, const h8& x2h, const h8& x2l, const h8& x3h, const h8& x3l) DCSCN_INL {
        constexpr int n = decltype(n_)::value, jb = decltype(j_)::value % NB;
        if constexpr (C3C_ABL & 8) {
            asm volatile("" :: "v"(wa[jb]), "v"(wb[jb]));
            asm volatile("" :: "v"(x0h), "v"(x0l), "v"(x1h), "v"(x1l));
            asm volatile("" :: "v"(x2h), "v"(x2l), "v"(x3h), "v"(x3l));
            return;
        }
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jb], x0h, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jb], x1h, acc[1][n], 0, 0, 0);
        acc[2][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jb], x2h, acc[2][n], 0, 0, 0);
        acc[3][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jb], x3h, acc[3][n], 0, 0, 0);
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x0l, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x1l, acc[1][n], 0, 0, 0);
        acc[2][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x2l, acc[2][n], 0, 0, 0);
        acc[3][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x3l, acc[3][n], 0, 0, 0);
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x0h, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x1h, acc[1][n], 0, 0, 0);
        acc[2][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x2h, acc[2][n], 0, 0, 0);
        acc[3][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jb], x3h, acc[3][n], 0, 0, 0);
    };
    // A fragments of (step i, tile n) of the group in slot `fs` (lane address): wb = hi piece, wa = lo piece (conv3_h8.hpp)
    auto read_a = [&](const char* fs, int i, int n, int jb) DCSCN_INL {
        wb[jb] = *reinterpret_cast<const h8*>(fs + i * TAPB + (2 * n) * 1024);
        wa[jb] = *reinterpret_cast<const h8*>(fs + i * TAPB + (2 * n + 1) * 1024);
    };

    // One group = one segment of this half.  MAIN: column POS (= kx) of a main chunk; else group POS of the packed tail with `gs` steps from
    // packed step `s0`.  plen = groups of the phase; li_* = the image to stage for the NEXT phase.
    auto group = [&](auto main_c, auto pos_c, int pos_rt, int plen, int gs, int s0, int gi, int li_pix0, unsigned li_ok, int lchunk) DCSCN_INL {
        constexpr bool MAIN = decltype(main_c)::value;
        constexpr int KX = decltype(pos_c)::value;
        const int pos = MAIN ? KX : pos_rt;
        const bool first_g = gi == 0, last_g = gi + 1 == g_total;
        const int img_off = ibuf * G::IN_BUF;
        const int a_lane = lane * 16;
        int l = lane;
        asm volatile("" : "+v"(l));
        // ================= LOAD part =================
        if (HALF == 0 && first_g && late_mode && !first_item) {
            // the item's first image could not be staged during the previous item (its last phase is one segment long): fetch it now, under
            // the previous item's epilogue; one extra barrier publishes it (half 1 meets it behind its last COMPUTE part of that item)
            img_rounds(cur.pix0, cur.ok_mask, 0, ibuf);
            if (pending) epilogue();
            dma_group(cur, nxt, gi + 2, sl2);
            c3p_wait_vm<F>();
            barrier_here();
        }
        int b_hi = 0;
        if constexpr (MAIN) {
            const int hx = (l & 15) + KX;
            b_hi = img_off + (4 * w4 * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, l >> 4, 0) * 16;
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int row = decltype(m_)::value;
                xh[row] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                xl[row] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
            });
        }
        auto tail_rows = [&](int step) DCSCN_INL {                // B rows of packed step `step`: (tap, octet) pair 4 step + lane group
            const int pair = 4 * step + (l >> 4);
            int tap = octs == 1 ? pair : octs == 2 ? pair >> 1 : (pair * 11) >> 5;
            const int oct = pair - tap * octs;
            tap = tap < 8 ? tap : 8;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int hx = (l & 15) + kx;
            const int b = img_off + ((4 * w4 + ky) * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, oct, 0) * 16;
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                xh[m] = *reinterpret_cast<const h8*>(smem + b + m * G::ROW_BYTES);
                xl[m] = *reinterpret_cast<const h8*>(smem + (b ^ 16) + m * G::ROW_BYTES);
            });
        };
        if constexpr (!MAIN) tail_rows(s0);
        const char* fs = smem + sl0 + a_lane;
        static_for<0, PFD>([&](auto p_) DCSCN_INL {
            constexpr int p = decltype(p_)::value;
            if constexpr (MAIN) read_a(fs, p / CNT, p % CNT, p % NB);
            else if constexpr (p < CNT) read_a(fs, 0, p, p % NB);
        });
        if constexpr (HALF == 0) {
            if (!(first_g && late_mode && !first_item)) {
                if (first_g && pending) epilogue();             // (its stores are older than the pieces below: the counted waits stay exact)
                dma_group(cur, nxt, gi + 2, sl2);
                if (pos == 1 && plen >= 2) img_rounds(li_pix0, li_ok, lchunk, ibuf ^ 1);
            }
        } else {
            // the wait first: behind the epilogue it would also wait for the epilogue's stores to be acknowledged
            c3c_wait_sel<F>(pos == 1 && plen == 3 ? 6 : 0);
            if (first_g && pending) epilogue();
            barrier_here();
        }
        // ================= COMPUTE part =================
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HALF == 1) {
            dma_group(cur, nxt, gi + 2, sl2);
            if (pos == 0 && plen >= 2) img_rounds(li_pix0, li_ok, lchunk, ibuf ^ 1);
            if (last_g && late_mode && nxt.valid) img_rounds(nxt.pix0, nxt.ok_mask, 0, ibuf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (first_g)
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                static_for<0, CNT>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
            });
        if constexpr (MAIN) {
            static_for<0, 3>([&](auto ky_) DCSCN_INL {
                constexpr int ky = decltype(ky_)::value;
                if constexpr (ky < 2) {                             // the row the NEXT tap adds, into the ring slot of the row the previous tap dropped
                    constexpr int row = ky + 4;
                    xh[row % 5] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                    xl[row % 5] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
                }
                static_for<0, CNT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value, j = ky * CNT + n;
                    if constexpr (j + PFD < 3 * CNT) read_a(fs, (j + PFD) / CNT, (j + PFD) % CNT, (j + PFD) % NB);
                    tile_mfmas(n_, std::integral_constant<int, j>{}, xh[ky % 5], xl[ky % 5], xh[(ky + 1) % 5], xl[(ky + 1) % 5],
                               xh[(ky + 2) % 5], xl[(ky + 2) % 5], xh[(ky + 3) % 5], xl[(ky + 3) % 5]);
                    __builtin_amdgcn_sched_barrier(0);             // a tile's reads and MFMAs stay where they are
                });
            });
        } else {
            for (int i = 0; i < gs; ++i) {
                if (i > 0) {
                    tail_rows(s0 + i);
                    static_for<0, (PFD < CNT ? PFD : CNT)>([&](auto p_) DCSCN_INL { constexpr int p = decltype(p_)::value; read_a(fs, i, p, p % NB); });
                }
                static_for<0, CNT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    if constexpr (n + PFD < CNT) read_a(fs, i, n + PFD, (n + PFD) % NB);
                    tile_mfmas(n_, n_, xh[0], xl[0], xh[1], xl[1], xh[2], xl[2], xh[3], xl[3]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        if constexpr (HALF == 1) {
            if (last_g && late_mode && nxt.valid) {                // (header: the next item's first image, fetched late)
                c3p_wait_vm<0>();
                barrier_here();
            }
        } else {
            c3c_wait_sel<F>(pos == 1 ? (plen == 3 ? F + 6 : 0) : F);
            barrier_here();
        }
        { const int t = sl0; sl0 = sl1; sl1 = sl2; sl2 = t; }      // the ring moves on by one slot per group
    };

    while (true) {
        {   // bias / slopes of this item's tiles (the previous item's epilogue reads the other parity)
            const int ba = G::BA_BASE + (parity * 2 + half) * C0 * 128;
            const int t4 = tid & 255;
            const int boff = cur.g * ntp * 16 + (half ? C0 * 16 : 0);
            if (t4 < CNT * 4) *reinterpret_cast<f32x4*>(smem + ba + t4 * 16) = reinterpret_cast<const f32x4*>(a.bias + boff)[t4];
            else if (t4 >= 64 && t4 < 64 + CNT * 4 && a.act == ACT_ALPHA)
                *reinterpret_cast<f32x4*>(smem + ba + C0 * 64 + (t4 - 64) * 16) = reinterpret_cast<const f32x4*>(a.alpha + boff)[t4 - 64];
        }
        int gi = 0;
        for (int chunk = 0; chunk < n_main; ++chunk) {
            const bool ends = chunk + 1 == n_main && octs == 0;    // the next phase is the next item's first chunk
            const int li_pix0 = ends ? nxt.pix0 : cur.pix0;
            const unsigned li_ok = ends ? nxt.ok_mask : cur.ok_mask;
            const int lchunk = ends ? 0 : chunk + 1;
            group(std::true_type{}, std::integral_constant<int, 0>{}, 0, 3, 3, 0, gi, li_pix0, li_ok, lchunk); ++gi;
            group(std::true_type{}, std::integral_constant<int, 1>{}, 1, 3, 3, 0, gi, li_pix0, li_ok, lchunk); ++gi;
            group(std::true_type{}, std::integral_constant<int, 2>{}, 2, 3, 3, 0, gi, li_pix0, li_ok, lchunk); ++gi;
            ibuf ^= 1;
        }
        if (octs) {
            for (int j = 0; j < g_tail; ++j) {
                const int gs = n_tail - 3 * j < 3 ? n_tail - 3 * j : 3;
                group(std::false_type{}, std::integral_constant<int, 0>{}, j, g_tail, gs, 3 * j, gi, nxt.pix0, nxt.ok_mask, 0); ++gi;
            }
            ibuf ^= 1;
        }
        // ---- item done: its epilogue runs inside the first LOAD part of the next item (or below, for the last one) ----
        pending = true;
        first_item = false;
        e_img = cur.img; e_y0 = cur.y0; e_x0 = cur.x0; e_g = cur.g; e_full = cur.full;
        parity ^= 1;
        if (!nxt.valid) break;
        id += (int)gridDim.x;
        decode(id, cur);
        decode(id + (int)gridDim.x, nxt);
    }
    epilogue();
    };
    if (half == 0) run(std::integral_constant<int, C0>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, C1>{}, std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may land after the workgroup has given its LDS back
}

}  // namespace dcscn
