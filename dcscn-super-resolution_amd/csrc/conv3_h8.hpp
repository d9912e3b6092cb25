// conv3_h8: conv3_h's arithmetic (3x3 SAME conv + bias + activator, tf.nn.conv2d of helper/tf_graph.py:104-153, direct implicit GEMM on
// v_mfma_f32_16x16x32_f16 with f16 (hi, lo) operands, three products per MAC: split16.hpp) with the workgroup rebuilt around what the
// r04 probes measured (profiles/r04_conv3_h_probe.txt, r04_h16_conv3_harness.txt):
//
//   * a conv3_h wave needs ~2300 cycles per tap when it has its SIMD to itself -- 1152 of MFMAs and as much again of everything
//     around them (barrier, filter DMA issue, LDS reads it has to wait for, staging of the next image) -- and two independent
//     workgroups per CU hide only a quarter of that from each other: their non-MFMA phases are not forced apart;
//   * every pixel tile's input is fetched, split into (hi, lo) and written to LDS once per channel GROUP (2.5x the algorithmic reads).
//
// Here ONE persistent workgroup of 8 waves per CU owns a pixel tile for ALL its output channels: waves 0-3 (half 0) take one channel
// group, waves 4-7 (half 1) the other (or the other half of the tiles of a single group), wave w and w + 4 share a SIMD.  Per tap a
// half has a LOAD part (everything but MFMAs: the first fragments of the tap read into registers, filter DMA two taps ahead, a slice of
// the next image staged, the epilogue of the previous item) and a COMPUTE part (12 MFMAs per channel tile and the LDS reads feeding
// them, straight-line).  The halves run them in OPPOSITE order between ONE workgroup barrier per tap:
//
//   half 0:  | LOAD(t)      COMPUTE(t) | LOAD(t+1)    COMPUTE(t+1) | ...
//   half 1:  | COMPUTE(t-1) LOAD(t)    | COMPUTE(t)   LOAD(t+1)    | ...                   ( | = s_barrier )
//
// so that a SIMD's matrix pipe goes from one wave to the other inside a segment without anybody waiting at a barrier for it (the
// first r04 build separated the parts by a second barrier -- C3E_SB = 0, kept for the tuner: a segment then lasted as long as its
// slowest wave, twice per tap).  Half 1 runs at s_setprio 1: its MFMAs and, behind them, its load part win the issue arbitration
// against the partner wave of the SIMD (-4 % on the two-group layers; raising the priority per phase instead, or for half 0, gains
// nothing: profiles/r04_h16_conv3_harness.txt).
//
// * input image: ONE (hi, lo) image per 32-channel chunk for both halves, double buffered (2 x 41.5 KB): the 2592 (pixel, channel quad)
//   items of chunk c + 1 are loaded at step 0 of chunk c by all 512 threads (6 loads each, hand-written so that the compiler's wait
//   bookkeeping -- which cannot see the LDS-DMA instructions issued in between -- does not wait for the youngest DMA in front of every
//   use), split one round per step at steps 3..8 and written straight into the other buffer: no chunk-boundary barrier, no second
//   fetch for the second channel group.
// * filters: a ring of three tap slots per half, filled by LDS-DMA two taps ahead.  Half 0 issues tap t + 2 in LOAD(t) (its slot was
//   last read in COMPUTE(t - 1), before the barrier) and waits for tap t + 1 at the END of COMPUTE(t), right in front of the barrier
//   that publishes it: two segments of latency budget.  Half 1 may not issue tap t + 2 before the barrier behind its LOAD(t) (sibling
//   waves still read the slot in COMPUTE(t - 1)), so it issues right behind that barrier and waits at the end of LOAD(t + 1).  Either
//   way a tap's slot is complete one segment before its compute part, so the tap's first fragments are read in the load part.
// * persistent, items (pixel tiles) dealt statically -- equal work, no competition inside a CU: the next item's first image and first
//   taps are in flight during the last taps of the current one, the epilogue of a half (lean fast path) runs in its first load part
//   of the next item.
// * LDS: 2 x 41.5 + 2 x 3 x NT x 2 KB + bias = 156 KB at NT = 6.  VGPRs: 96 accumulators + 32 B + 16 A (ring of 3 tiles) + 24 staged.
//
// vmcnt bookkeeping (a wave's vector-memory operations retire in order; anything issued that the counts below do not name -- epilogue
// stores -- only makes a wait longer than needed, never shorter).  F = DMA instructions per wave and tap, 6 = image loads of a step 0.
//   half 0, end of COMPUTE(t): the pieces of LOAD(t - 1) must have landed; younger: LOAD(t)'s F pieces, plus the six image loads when
//     step t or t - 1 is a step 0  ->  vmcnt(F + 6) at steps 0 and 1, vmcnt(F) otherwise; the wait at step 2 thereby completes the
//     image loads before their first use at step 3.  Packed tail: the same with its steps 0 / 1, and vmcnt(F) in front of step 2's
//     conversion.
//   half 1, end of LOAD(t): the pieces issued behind the previous barrier; younger: only this part's image loads  ->  vmcnt(6) at a
//     step 0, vmcnt(0) otherwise.
#pragma once
#include "conv3_h.hpp"
#include "p16.hpp"

namespace dcscn {

template <int N>
__device__ __forceinline__ void c3p_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS operations of this wave done, then the workgroup barrier; no fence semantics wanted (vector memory stays in flight across it)
__device__ __forceinline__ void c3p_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// conv_wino2.hpp's glds16 with M0 declared clobbered instead of saved and restored around every piece (two s_mov fewer per DMA)
__device__ __forceinline__ void glds16c(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" DCSCN_GLDS_AUX : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <int NT>
struct C3EGeom {
    static constexpr int THREADS = 512;
    static constexpr int KC = 32, TH = 16, TW = 16, HT = 18, HP = HT * HT;
    static constexpr int PIX_BYTES = 128, ROW_BYTES = HT * PIX_BYTES, IN_BYTES = HP * PIX_BYTES;   // 41472
    static constexpr int IN_BUF = 41 * 1024;                      // an image buffer: 41 DMA pieces of 1 KB (P16 staging; the last piece is half padding)
    static constexpr int IN_ITEMS = HP * 8;
    static constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;                            // 6 (the last one: 32 items)
    static constexpr int F_TAP_BYTES = NT * 2048;                 // one tap of one half: [n][hi | lo][64 lanes][16 bytes]
    static constexpr int F_ROUNDS = (2 * NT + 3) / 4;             // DMA instructions per wave and tap
    static constexpr int IMG = 0;                                 // two image buffers
    static constexpr int F_BASE = 2 * IN_BUF;                     // [half][slot]
    static constexpr int BA_BASE = F_BASE + 2 * 3 * F_TAP_BYTES;  // [parity][half][bias | slopes]: NT * 128 bytes each
    static constexpr int LDS_BYTES = BA_BASE + 4 * NT * 128;
};

// DBG (tuner only) 1: per wave through a.srctab: [0] entry, [1] exit, [2] items, [3] sum of load phases, [4] sum of compute phases,
// [5] sum of the waits at the barrier ending a load phase, [6] same for compute phases, [7] sum of epilogues
#ifndef C3E_SB
#define C3E_SB 1           // 1: one workgroup barrier per tap (see the header); 0: the two-barrier ping-pong of the first r04 build
#endif
#ifndef C3E_ABL
#define C3E_ABL 0          // tuner only (results wrong): 1 no filter DMA, 2 no image staging, 4 no A-fragment reads, 8 no MFMAs, 16 no B-row reads
#endif
#ifndef C3E_PRIO
#define C3E_PRIO 2         // waves of half (C3E_PRIO - 1) run at s_setprio 1 for the whole kernel; 0 = nobody
#endif
#ifndef C3E_PRIO_LEVEL
#define C3E_PRIO_LEVEL 1
#endif
#ifndef C3E_PFD
#define C3E_PFD 2          // A fragments are read this many channel tiles ahead of their MFMAs (the first PFD tiles in the load phase)
#endif
// NT = channel tiles of half 0, C1 = of half 1 (NT or NT - 1; 0 for a one-tile layer): launch constants -- the host instantiates the
// pair a layer needs, every loop over tiles is straight-line code (a branch around a tile makes the compiler wait for ALL outstanding
// LDS reads in front of every tile, and merging code variants cost 60 VGPRs in copies of the accumulators).
// NTP = channel tiles per group in the filter image when that is a launch constant too (two-group layers: NTP = NT, every tap address an
// immediate offset), 0 = args.nt_pack (tuner: one group split between the halves)
// P16 = the input is a pre-split tensor (a.in16, p16.hpp) staged by LDS-DMA, and the destinations are P16 tensors too (the layers this
// kernel runs feed split16 consumers only); false = float32 NHWC in and out, split in registers (r04)
template <int NT, int C1, int DBG = 0, int NTP = NT, bool P16 = false>
__global__ __launch_bounds__(512, 2) void conv3_h8(const ConvArgs a) {
    static_assert(C1 == NT || C1 == NT - 1, "half 1 takes as many tiles as half 0 or one fewer");
    static_assert(!P16 || C3E_SB, "P16 staging is written for the one-barrier schedule");
    constexpr int PFD = C3E_PFD < NT ? C3E_PFD : (NT > 1 ? NT - 1 : 1), NB = PFD + 1;
    using G = C3EGeom<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem_c3e[];
    char* const smem = smem_c3e;
    constexpr int F = G::F_ROUNDS, L = G::IN_ROUNDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;
    const int H = a.H, W = a.W;
    const int n_groups = a.n_groups, ntp = NTP ? NTP : a.nt_pack;
    const int n_pairs = (n_groups + 1) >> 1;
    const int n_units = a.N * a.tiles_y * a.tiles_x * n_pairs;
    const int n_chunks = a.n_chunks;
    const int octs = a.tail_octs;
    const int n_main = octs ? n_chunks - 1 : n_chunks;
    const int n_tail = (9 * octs + 3) >> 2;
    const int t_total = n_main * 9 + n_tail;                   // taps (MFMA steps) of an item
    const bool fastable = a.ps == 1 && a.res == nullptr && (a.act == ACT_ALPHA || a.act == ACT_NONE) && (a.split & 15) == 0;   // (two destinations: split on a tile boundary)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned f_off = (unsigned)(lane * 16);
    const int cq = tid & 7;
    const int tap_stride = ntp * 2048;                         // bytes between taps of a group's filter image
    long long pr_t0 = 0, pr_load = 0, pr_comp = 0, pr_bl = 0, pr_bc = 0, pr_epi = 0, pr_a = 0, pr_b = 0;
    int pr_items = 0;
    if constexpr (DBG == 1) pr_t0 = __builtin_readcyclecounter();

    // ---- what this half does of an item: channel group g, its tiles [o, o + cnt) ----
    struct Unit {
        int valid, tile_id, img, y0, x0, g, o;
        int pix0;                                              // P16: flat pixel index of the halo tile's origin (may be negative)
        bool all_in, full;
        const char* a_base;
        const char* f_base;
        unsigned ok_mask;
    };
    auto decode = [&](int id, Unit& u) DCSCN_INL {
        u.valid = id < n_units;
        const int idc = u.valid ? id : 0;
        const int tile_id = idc / n_pairs, p = idc - tile_id * n_pairs;
        const int g0 = 2 * p;
        if (g0 + 1 < n_groups) {
            u.g = g0 + half; u.o = 0;
        } else {                                               // one group left: its tiles are split between the halves
            u.g = g0; u.o = half ? NT : 0;                      // half 0 takes the group's first NT tiles
        }
        int bid = tile_id;
        const int tx = bid % a.tiles_x;
        bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        u.tile_id = tile_id;
        u.img = bid / a.tiles_y;
        u.y0 = ty * G::TH; u.x0 = tx * G::TW;
        u.all_in = u.y0 >= 1 && u.x0 >= 1 && u.y0 + G::TH + 1 <= H && u.x0 + G::TW + 1 <= W;
        u.full = u.y0 + G::TH <= H && u.x0 + G::TW <= W;
        if constexpr (P16) { u.a_base = nullptr; u.pix0 = (u.img * H + u.y0 - 1) * W + u.x0 - 1; }
        else { u.pix0 = 0; u.a_base = reinterpret_cast<const char*>(a.in + (size_t)u.img * H * W * a.in_stride + a.in_off + ((ptrdiff_t)(u.y0 - 1) * W + (u.x0 - 1)) * a.in_stride); }
        u.f_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)u.g * n_chunks * 9 * tap_stride + (size_t)u.o * 2048;
        unsigned m = 0;
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hrow = hp0 / G::HT, hcol = hp0 - G::HT * hrow;     // item r: halo pixel hp0 + 64 r = 3 rows and 10 columns further per round
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int gy = u.y0 - 1 + hrow, gx = u.x0 - 1 + hcol;
            bool ok = r * 64 + hp0 < G::HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            if constexpr (P16 && r == L - 1) {                 // every wave fetches DMA piece 40 (halo pixels 320 + (lane >> 3)) of the image
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

    // ---- image staging: item = r * 512 + tid = (halo pixel r * 64 + (tid >> 3), channel quad tid & 7) ----
    f32x4 gin[L];
    auto load_in = [&](const char* base, unsigned mask, int chunk) DCSCN_INL {
        const int c0 = chunk * G::KC + cq * 4;
        const unsigned coff = (unsigned)((c0 < a.cin_phys ? c0 : 0) * 4);
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hrow = hp0 / G::HT, hcol = hp0 - G::HT * hrow;
        const int stride4 = a.in_stride * 4;
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int pix = ((mask >> r) & 1u) ? hrow * W + hcol : W + 1;
            // by hand: the compiler's own vmcnt bookkeeping does not see the LDS-DMA instructions between these loads and their use and
            // would wait for the youngest of THEM in front of every round of convert_store; the waits below cover these loads
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gin[r]) : "v"((unsigned)(pix * stride4) + coff), "s"(base) : "memory");
            hcol += 64 - 3 * G::HT; hrow += 3;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    };
    const float m1 = opaque_minus_one();
    // round r of the image in flight: zero what is padding, split, write to image buffer `buf`
    auto convert_store = [&](auto r_, bool all_in, unsigned mask, int chunk, int buf) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        asm volatile("" : "+v"(gin[r]));                       // (stays behind the wait that made it valid)
        f32x4 x = gin[r];
        const bool whole = all_in && (chunk + 1) * G::KC <= a.cin_phys;
        if (!whole) {
            const bool ok = chunk * G::KC + cq * 4 < a.cin_phys && ((mask >> r) & 1u);
            x.x = ok ? x.x : 0.0f; x.y = ok ? x.y : 0.0f; x.z = ok ? x.z : 0.0f; x.w = ok ? x.w : 0.0f;
        }
        h4 hi, lo;
        split4(x, m1, hi, lo);
        const u32x2 hu = __builtin_bit_cast(u32x2, hi), lu = __builtin_bit_cast(u32x2, lo);
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        const int hp = r * 64 + hp0;
        const int hcol = hp % G::HT;
        const int kq = cq >> 1;
        const int off = buf * G::IN_BUF + hp * G::PIX_BYTES + c3h_unit(hcol, kq, 0) * 16 + (cq & 1) * 8;
        if (r < L - 1 || hp < G::HP) {
            *reinterpret_cast<u32x2*>(smem + off) = hu;
            *reinterpret_cast<u32x2*>(smem + (off ^ 16)) = lu;
        }
    };
    // P16 staging: DMA piece r of this wave = 8 halo pixels x 8 units; lane (pixel, slot) fetches the unit c3h_unit puts at that slot.
    // Out-of-image pixels and octets past the tensor's last come from the plane's zero record: every lane always issues.
    auto img_piece = [&](auto r_, int pix0, unsigned mask, int chunk, int buf) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int rem = a.in16.octs - 4 * chunk;               // octets of this chunk (wave-uniform)
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
        if constexpr (C3E_ABL & 2) return;
        glds16c(base, voff, lds0 + (unsigned)(buf * G::IN_BUF + piece * 1024));
    };
    // one tap of this half's filters -> ring slot
    auto dma_f = [&](const char* src, int slot /* byte offset of the ring slot */, int cnt) DCSCN_INL {
        const int pieces = cnt > 0 ? 2 * cnt : 1;
        if constexpr (C3E_ABL & 1) return;
        static_for<0, F>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int piece = (w4 + 4 * r) % pieces;
            glds16c(src + piece * 1024, f_off, lds0 + (unsigned)slot + (unsigned)piece * 1024u);
        });
    };
    // address of tap `tt` of unit u's filter image (tt may run past the item: then it is a tap of the next unit)
    auto tap_src = [&](const Unit& u, const Unit& un, int tt) DCSCN_INL -> const char* {
        const Unit& w = tt < t_total ? u : un;
        const int t = tt < t_total ? tt : tt - t_total;
        int idx;
        if (t < n_main * 9) { const int c = t / 9, s = t - 9 * c; idx = c * 9 + (s % 3) * 3 + s / 3; }
        else idx = t;                                          // the packed tail's slots are in step order
        return w.f_base + (size_t)idx * tap_stride;
    };

    auto run = [&](auto cnt_c, auto half_c) DCSCN_INL {
    constexpr int CNT = decltype(cnt_c)::value;                // channel tiles of THIS half
    constexpr int HALF = decltype(half_c)::value;
    // one barrier per tap: half 0 keeps the one behind its compute phase, half 1 the one behind its load phase
    constexpr bool BAR_L = !C3E_SB || HALF == 1, BAR_C = !C3E_SB || HALF == 0;
    constexpr bool DMA_LATE = C3E_SB && HALF == 1;             // half 1 issues a tap's DMA behind the barrier (its target slot is read until then)
    // ---- first item ----
    Unit cur, nxt;
    int id = blockIdx.x;
    decode(id, cur);
    if (!cur.valid) return;
    decode(id + (int)gridDim.x, nxt);
    int ibuf = 0;                                              // image buffer of the chunk being computed
    // byte offsets of this half's three ring slots, in the order (this tap, next tap, tap after next) at step % 3 == 0 of a chunk:
    // nine taps per chunk leave the order alone, the packed tail rotates it once per step
    int sl0 = G::F_BASE + (half * 3 + 0) * G::F_TAP_BYTES, sl1 = G::F_BASE + (half * 3 + 1) * G::F_TAP_BYTES, sl2 = G::F_BASE + (half * 3 + 2) * G::F_TAP_BYTES;
    dma_f(tap_src(cur, nxt, 0), sl0, CNT);
    dma_f(tap_src(cur, nxt, 1), sl1, CNT);
    if constexpr (P16) {
        static_for<0, L>([&](auto r_) DCSCN_INL { img_piece(r_, cur.pix0, cur.ok_mask, 0, 0); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_in(cur.a_base, cur.ok_mask, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<0, L>([&](auto r_) DCSCN_INL { convert_store(r_, cur.all_in, cur.ok_mask, 0, 0); });
    }
    c3p_barrier();
    if (!C3E_SB && half == 1) c3p_barrier();                   // half 1 runs one phase behind half 0
    if (C3E_PRIO && half == C3E_PRIO - 1) asm volatile("s_setprio %0" :: "n"(C3E_PRIO_LEVEL));

    f32x4 acc[4][NT];
    h8 xh[4], xl[4], wa[NB], wb[NB];                           // B rows (hi, lo), A fragments (wl, wh) in a ring of PFD + 1 tiles
    if constexpr (C3E_ABL != 0) {                              // finite operands for the ablation builds (non-finite ones slow the MFMAs down)
        u32x4 z = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        asm volatile("" : "+v"(z));
        static_for<0, 4>([&](auto m_) DCSCN_INL { xh[decltype(m_)::value] = xl[decltype(m_)::value] = __builtin_bit_cast(h8, z); });
        static_for<0, NB>([&](auto m_) DCSCN_INL { wa[decltype(m_)::value] = wb[decltype(m_)::value] = __builtin_bit_cast(h8, z); });
    }
    bool pending = false;                                      // an epilogue is owed (previous item)
    int parity = 0;
    // epilogue state of the previous item (its accumulators are still in acc until the first compute phase of the next item)
    int e_tile = 0, e_img = 0, e_y0 = 0, e_x0 = 0, e_g = 0, e_o = 0;
    bool e_full = false;

    auto epilogue = [&]() DCSCN_INL {
        if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
        const int ba = G::BA_BASE + ((parity ^ 1) * 2 + half) * NT * 128;
        const int cb16 = e_g * ntp * 16 - 16 * (e_g > a.n_full ? e_g - a.n_full : 0) + e_o * 16;   // conv channel of the half's first tile
        const float inv = a.inv_scale;
        const float zero = opaque_zero();
        float chk = 0.0f;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int lje = le & 15, lke = le >> 4;
        if constexpr (P16) {
            // P16 destinations (p16.hpp): every lane stores ONE 16-byte unit per accumulator tile -- unit lke of the tile's two octets of
            // its pixel's record; one 64-bit base per tile, 32-bit lane offsets.  MASK: tiles that stick out of the image.
            const h2 zero2 = p16_opaque_zero2();
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
                    if (ACT_C == ACT_ALPHA || (ACT_C < 0 && a.act == ACT_ALPHA)) av = *reinterpret_cast<const f32x4*>(smem + ba + NT * 64 + (n * 4 + lke) * 16);
                    static_for<0, 4>([&](auto m_) DCSCN_INL {
                        constexpr int m = decltype(m_)::value;
                        f32x4 v = acc[m][n] * inv + bv;
                        if constexpr (ACT_C == ACT_ALPHA) {
                            v.x = v.x > 0.0f ? v.x : av.x * v.x;
                            v.y = v.y > 0.0f ? v.y : av.y * v.y;
                            v.z = v.z > 0.0f ? v.z : av.z * v.z;
                            v.w = v.w > 0.0f ? v.w : av.w * v.w;
                        } else if constexpr (ACT_C < 0) {
                            if (chan_ok && (!MASK || e_y0 + 4 * w4 + m < H)) chk = nonfinite_acc(chk, acc[m][n], zero);     // (a saturating activator hides a non-finite accumulator)
                            v.x = activate1(v.x, av.x, a.act);
                            v.y = activate1(v.y, av.y, a.act);
                            v.z = activate1(v.z, av.z, a.act);
                            v.w = activate1(v.w, av.w, a.act);
                        }
                        const u32x4 unit = p16_unit(v, m1, chk, zero2, chan_ok && (!MASK || e_y0 + 4 * w4 + m < H));
                        if (chan_ok && (!MASK || e_y0 + 4 * w4 + m < H)) *reinterpret_cast<u32x4*>(base + (size_t)(voff + m * rowb)) = unit;
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
            if constexpr (DBG == 1) { pr_epi += __builtin_readcyclecounter() - pr_a; ++pr_items; }
            return;
        }
        if (fastable && e_full) {
            auto finish = [&](auto act_c) DCSCN_INL {
                constexpr int ACT_C = decltype(act_c)::value;
                static_for<0, CNT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    const int c0 = cb16 + n * 16;
                    const bool first = c0 < a.split;
                    float* optr = first ? a.out0.ptr : a.out1.ptr;
                    const int ostride = first ? a.out0.stride : a.out1.stride;
                    const int ooff = first ? a.out0.off : a.out1.off;
                    const int owidth = first ? a.out0.width : a.out1.width;
                    const int cc0 = first ? c0 : c0 - a.split;
                    char* base = reinterpret_cast<char*>(optr + ((size_t)(e_img * H + e_y0) * W + e_x0) * ostride + ooff + cc0);
                    const unsigned voff = (unsigned)(((4 * w4 * W + lje) * ostride + 4 * lke) * 4);
                    const unsigned rowb = (unsigned)(W * ostride * 4);
                    const bool chan_ok = cc0 + 4 * lke < owidth;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + ba + (n * 4 + lke) * 16);
                    f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (ACT_C == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(smem + ba + NT * 64 + (n * 4 + lke) * 16);
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
            if (a.act == ACT_ALPHA) finish(std::integral_constant<int, ACT_ALPHA>{});
            else finish(std::integral_constant<int, ACT_NONE>{});
        } else {
            const int obase = cb16 + 4 * lke;
            const int act = a.act, ps = a.ps, orow = W * ps;
            const int gx = e_x0 + lje;
            static_for<0, CNT>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const int c = obase + n * 16;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + ba + (n * 4 + lke) * 16);
                f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
                if (act == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(smem + ba + NT * 64 + (n * 4 + lke) * 16);
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
                const bool live = gx < W && cc < owidth;
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const int gy = e_y0 + 4 * w4 + m;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act);
                    v.y = activate1(v.y, av.y, act);
                    v.z = activate1(v.z, av.z, act);
                    v.w = activate1(v.w, av.w, act);
                    if (live && gy < H) {
                        chk = nonfinite_acc(chk, acc[m][n], zero);
                        const size_t pix = (size_t)((e_img * H + gy) * ps + ay) * orow + (size_t)(gx * ps + bx);
                        if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch);
                        *reinterpret_cast<f32x4*>(optr + pix * ostride + ooff + ch) = v;
                    }
                });
            });
        }
        if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + e_img] = 1; }     // the image goes to the float32 plan (exec.hip)
        pending = false;
        if constexpr (DBG == 1) { pr_epi += __builtin_readcyclecounter() - pr_a; ++pr_items; }
    };
    auto phase_barrier = [&](auto on_c, long long& acc_wait) DCSCN_INL {
        if constexpr (DBG == 1) pr_b = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);                     // MFMAs have no memory effect: without this the scheduler moves some across the barrier
        if constexpr (decltype(on_c)::value) {
            c3p_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DBG == 1) acc_wait += __builtin_readcyclecounter() - pr_b;
    };
    constexpr std::integral_constant<bool, BAR_L> bar_l{};
    constexpr std::integral_constant<bool, BAR_C> bar_c{};

    while (true) {
        // bias / slopes of this item's tiles (the previous item's epilogue reads the other parity)
        {
            const int ba = G::BA_BASE + (parity * 2 + half) * NT * 128;
            const int t4 = tid & 255;
            const int boff = cur.g * ntp * 16 + cur.o * 16;
            if (t4 < NT * 4) {
                if (t4 < CNT * 4) *reinterpret_cast<f32x4*>(smem + ba + t4 * 16) = reinterpret_cast<const f32x4*>(a.bias + boff)[t4];
            } else if (t4 >= 64 && t4 < 64 + NT * 4 && a.act == ACT_ALPHA) {
                if (t4 - 64 < CNT * 4) *reinterpret_cast<f32x4*>(smem + ba + NT * 64 + (t4 - 64) * 16) = reinterpret_cast<const f32x4*>(a.alpha + boff)[t4 - 64];
            }
        }
        int b_hi = 0;
        const int a_lane = lane * 16;
        int tt = 0;                                            // tap index inside the item
        for (int chunk = 0; chunk < n_main; ++chunk) {
            const bool last_main = chunk + 1 == n_main;
            const bool ends = last_main && octs == 0;
            const char* li_base = ends ? nxt.a_base : cur.a_base;
            const unsigned li_ok = ends ? nxt.ok_mask : cur.ok_mask;
            const bool li_all_in = ends ? nxt.all_in : cur.all_in;
            const int li_pix0 = ends ? nxt.pix0 : cur.pix0;
            const int lchunk = ends ? 0 : chunk + 1;
            const int img_off = ibuf * G::IN_BUF;
            static_for<0, 9>([&](auto s_) DCSCN_INL {
                constexpr int step = decltype(s_)::value;
                constexpr int kx = step / 3, ky = step % 3;
                // ================= LOAD phase of this tap =================
                if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
                // first fragments of this tap, read FIRST (their latency runs under the rest of the phase): the new B row(s) and the
                // A fragments of the first PFD tiles -- the tap's filter slot has been complete since the previous load phase's wait
                if constexpr (ky == 0) {
                    int l = lane;
                    asm volatile("" : "+v"(l));
                    const int hx = (l & 15) + kx;
                    b_hi = img_off + (4 * w4 * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, l >> 4, 0) * 16;
                }
                if constexpr (!(C3E_ABL & 16))
                static_for<(ky == 0 ? 0 : 3), 4>([&](auto m_) DCSCN_INL {
                    constexpr int row = ky + decltype(m_)::value;
                    xh[row & 3] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                    xl[row & 3] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
                });
                const char* fs = smem + (step % 3 == 0 ? sl0 : step % 3 == 1 ? sl1 : sl2) + a_lane;
                if constexpr (!(C3E_ABL & 4))
                static_for<0, PFD>([&](auto p_) DCSCN_INL {
                    constexpr int p = decltype(p_)::value;
                    wb[p] = *reinterpret_cast<const h8*>(fs + (2 * p) * 1024);       // (tiles past the half's last: stale bytes of its own slot, never used)
                    wa[p] = *reinterpret_cast<const h8*>(fs + (2 * p + 1) * 1024);
                });
                if constexpr (!C3E_SB) { if constexpr (step == 1) c3p_wait_vm<L>(); else c3p_wait_vm<0>(); }   // the pieces issued in the previous load phase
                auto dma_ahead = [&]() DCSCN_INL {   // the tap two steps ahead: of this chunk, of the next one, of the packed tail, or of the next item
                    constexpr int step2 = (step + 2) % 9;
                    constexpr int ptap2 = (step2 % 3) * 3 + step2 / 3;
                    const char* src;
                    if constexpr (step + 2 < 9) src = cur.f_base + (size_t)(chunk * 9 + ptap2) * tap_stride;
                    else src = !last_main ? cur.f_base + (size_t)((chunk + 1) * 9 + ptap2) * tap_stride
                             : octs      ? cur.f_base + (size_t)(n_main * 9 + step2) * tap_stride
                                         : nxt.f_base + (size_t)ptap2 * tap_stride;
                    dma_f(src, (step + 2) % 3 == 0 ? sl0 : (step + 2) % 3 == 1 ? sl1 : sl2, CNT);
                };
                if constexpr (!DMA_LATE) dma_ahead();
                // P16: one image piece of the next chunk per step at steps 1 .. 6, BEHIND the filter pieces (buffer ibuf ^ 1 is read until the
                // barrier that ends step 0: half 1's COMPUTE(8) of the previous chunk runs in that segment)
                constexpr int IMG = P16 && step >= 1 && step <= L ? 1 : 0, IMG_PREV = P16 && step >= 2 && step <= L + 1 ? 1 : 0;
                if constexpr (P16) { if constexpr (IMG) img_piece(std::integral_constant<int, step - 1>{}, li_pix0, li_ok, lchunk, ibuf ^ 1); }
                else {
                    if constexpr (step == 0 && !(C3E_ABL & 2)) load_in(li_base, li_ok, lchunk);
                    if constexpr (step >= 3 && !(C3E_ABL & 2)) convert_store(std::integral_constant<int, step - 3>{}, li_all_in, li_ok, lchunk, ibuf ^ 1);
                }
                if constexpr (step == 0) { if (chunk == 0 && pending) epilogue(); }
                // half 1, one barrier per tap: the next tap's pieces were issued behind the previous barrier; only this step's image loads / piece are younger
                if constexpr (DMA_LATE) { if constexpr (P16) c3p_wait_vm<IMG>(); else if constexpr (step == 0) c3p_wait_vm<L>(); else c3p_wait_vm<0>(); }
                if constexpr (DBG == 1) pr_load += __builtin_readcyclecounter() - pr_a;
                phase_barrier(bar_l, pr_bl);
                // ================= COMPUTE phase =================
                if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
                if constexpr (DMA_LATE) { dma_ahead(); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (step == 0) {
                    if (chunk == 0)
                        static_for<0, 4>([&](auto m_) DCSCN_INL {
                            static_for<0, NT>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
                        });
                }
                auto mfmas = [&](auto cnt_c) DCSCN_INL {
                    constexpr int CNT = decltype(cnt_c)::value;
                    static_for<0, CNT>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        if constexpr (n + PFD < CNT && !(C3E_ABL & 4)) {
                            wb[(n + PFD) % NB] = *reinterpret_cast<const h8*>(fs + (2 * (n + PFD)) * 1024);
                            wa[(n + PFD) % NB] = *reinterpret_cast<const h8*>(fs + (2 * (n + PFD) + 1) * 1024);
                        }
                        if constexpr (C3E_ABL & 8) {
                            asm volatile("" :: "v"(wa[n % NB]), "v"(wb[n % NB]));
                            asm volatile("" :: "v"(xh[0]), "v"(xl[0]), "v"(xh[1]), "v"(xl[1]));
                            asm volatile("" :: "v"(xh[2]), "v"(xl[2]), "v"(xh[3]), "v"(xl[3]));
                        } else {
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[n % NB], xh[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[n % NB], xl[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[n % NB], xh[(ky + m) & 3], acc[m][n], 0, 0, 0); });
                        }
                        __builtin_amdgcn_sched_barrier(0);     // a tile's reads and MFMAs stay where they are (hoisted, the reads of all tiles cost 40 more registers)
                    });
                };
                mfmas(std::integral_constant<int, CNT>{});
                // half 0, one barrier per tap: the NEXT tap's pieces (issued in the previous load phase) must have landed before the barrier;
                // younger than them: this load phase's F pieces, and the six image loads of a step 0 during the two steps after it
                // P16: younger than LOAD(t - 1)'s filter pieces are its image piece (issued behind them), LOAD(t)'s F pieces and its image piece
                if constexpr (C3E_SB && HALF == 0) { if constexpr (P16) c3p_wait_vm<F + IMG + IMG_PREV>(); else if constexpr (step <= 1) c3p_wait_vm<F + L>(); else c3p_wait_vm<F>(); }
                if constexpr (DBG == 1) pr_comp += __builtin_readcyclecounter() - pr_a;
                phase_barrier(bar_c, pr_bc);
                ++tt;
            });
            ibuf ^= 1;
        }
        // ---- packed tail: (tap, octet) pairs four to an instruction; the next item's first image is staged here ----
        if (octs) {
            const int img_off = ibuf * G::IN_BUF;
            int l = lane;
            asm volatile("" : "+v"(l));
            auto tail_step = [&](int step, auto first_c) DCSCN_INL {
                constexpr bool FIRST = decltype(first_c)::value;
                if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
                if constexpr (!C3E_SB) { if (!FIRST && step == 1) c3p_wait_vm<L>(); else c3p_wait_vm<0>(); }
                if constexpr (!P16)
                if (!FIRST && step == 2 && !(C3E_ABL & 2)) {
                    if constexpr (C3E_SB && HALF == 0) c3p_wait_vm<F>();      // the image loads of tail step 0 (younger: step 1's pieces)
                    static_for<0, L>([&](auto r_) DCSCN_INL { convert_store(r_, nxt.all_in, nxt.ok_mask, 0, ibuf ^ 1); });
                }
                if constexpr (!DMA_LATE) dma_f(tap_src(cur, nxt, tt + 2), sl2, CNT);
                if constexpr (P16) {
                    // the next item's first image: all L pieces at tail step 1 (buffer ibuf ^ 1 is read until the barrier that ends tail step 0),
                    // behind the filter pieces; complete at the end of tail step 2 (n_tail >= 3)
                    if constexpr (!FIRST) { if (step == 1) static_for<0, L>([&](auto r_) DCSCN_INL { img_piece(r_, nxt.pix0, nxt.ok_mask, 0, ibuf ^ 1); }); }
                } else if constexpr (FIRST && !(C3E_ABL & 2)) load_in(nxt.a_base, nxt.ok_mask, 0);
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
                const char* fs = smem + sl0 + a_lane;
                static_for<0, PFD>([&](auto p_) DCSCN_INL {
                    constexpr int p = decltype(p_)::value;
                    wb[p] = *reinterpret_cast<const h8*>(fs + (2 * p) * 1024);       // (tiles past the half's last: stale bytes of its own slot, never used)
                    wa[p] = *reinterpret_cast<const h8*>(fs + (2 * p + 1) * 1024);
                });
                if constexpr (DMA_LATE) {
                    if constexpr (P16) { if (!FIRST && step == 1) c3p_wait_vm<L>(); else c3p_wait_vm<0>(); }
                    else if constexpr (FIRST) c3p_wait_vm<L>(); else c3p_wait_vm<0>();
                }
                if constexpr (DBG == 1) pr_load += __builtin_readcyclecounter() - pr_a;
                phase_barrier(bar_l, pr_bl);
                if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
                if constexpr (DMA_LATE) { dma_f(tap_src(cur, nxt, tt + 2), sl2, CNT); __builtin_amdgcn_sched_barrier(0); }
                auto mfmas = [&](auto cnt_c) DCSCN_INL {
                    constexpr int CNT = decltype(cnt_c)::value;
                    static_for<0, CNT>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        if constexpr (n + PFD < CNT) {
                            wb[(n + PFD) % NB] = *reinterpret_cast<const h8*>(fs + (2 * (n + PFD)) * 1024);
                            wa[(n + PFD) % NB] = *reinterpret_cast<const h8*>(fs + (2 * (n + PFD) + 1) * 1024);
                        }
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[n % NB], xh[m], acc[m][n], 0, 0, 0); });
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[n % NB], xl[m], acc[m][n], 0, 0, 0); });
                        static_for<0, 4>([&](auto m_) DCSCN_INL { constexpr int m = decltype(m_)::value; acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[n % NB], xh[m], acc[m][n], 0, 0, 0); });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
                mfmas(std::integral_constant<int, CNT>{});
                if constexpr (C3E_SB && HALF == 0) {
                    if constexpr (P16) { if (!FIRST && step == 1) c3p_wait_vm<F + L>(); else c3p_wait_vm<F>(); }
                    else if (FIRST || step == 1) c3p_wait_vm<F + L>(); else c3p_wait_vm<F>();
                }
                if constexpr (DBG == 1) pr_comp += __builtin_readcyclecounter() - pr_a;
                phase_barrier(bar_c, pr_bc);
                ++tt;
                { const int t = sl0; sl0 = sl1; sl1 = sl2; sl2 = t; }   // the ring moves on by one slot per tail step
            };
            tail_step(0, std::true_type{});
            for (int step = 1; step < n_tail; ++step) tail_step(step, std::false_type{});
            ibuf ^= 1;
        }
        // ---- item done: its epilogue is run inside the first load phase of the next item (or below, for the last one) ----
        pending = true;
        e_tile = cur.tile_id; e_img = cur.img; e_y0 = cur.y0; e_x0 = cur.x0; e_g = cur.g; e_o = cur.o; e_full = cur.full;
        parity ^= 1;
        if (!nxt.valid) break;
        id += (int)gridDim.x;
        decode(id, cur);
        decode(id + (int)gridDim.x, nxt);
    }
    epilogue();
    if (!C3E_SB && half == 0) c3p_barrier();                   // pairs with half 1's extra barrier at the start
    };
    if (half == 0) run(std::integral_constant<int, NT>{}, std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, C1>{}, std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may land after the workgroup has given its LDS back
    if constexpr (DBG == 1) {
        if (lane == 0 && a.srctab) {
            long long* pr = reinterpret_cast<long long*>(const_cast<void*>(a.srctab)) + ((size_t)blockIdx.x * 8 + wave) * 8;
            pr[0] = pr_t0; pr[1] = __builtin_readcyclecounter(); pr[2] = pr_items; pr[3] = pr_load; pr[4] = pr_comp; pr[5] = pr_bl; pr[6] = pr_bc; pr[7] = pr_epi;
        }
    }
}

}  // namespace dcscn
