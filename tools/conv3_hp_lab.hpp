// LAB KERNEL (tools/h16_tune -DH16_HP; not part of the library): measured neutral against conv3_h (profiles/r04_h16_conv3_harness.txt),
// its persistent item loop and lean epilogue live on in csrc/conv3_h8.hpp.
// conv3_hp: conv3_h's arithmetic and LDS pipeline (conv3_h.hpp: 3x3 SAME conv + bias + activator, tf.nn.conv2d of
// helper/tf_graph.py:104-153, as a direct implicit GEMM on v_mfma_f32_16x16x32_f16 with f16 (hi, lo) operands, 3 products per
// MAC) as a PERSISTENT kernel: two workgroups per CU for the whole launch, each pulling (pixel tile, channel group) items from
// one device counter and running them back to back with everything that crosses an item boundary kept in flight.
//
// Why (profiles/r04_conv3_h_probe.txt: shader-clock probes inside conv3_h on the bench layers): a conv3_h workgroup of CNN2 lives
// 163 k cycles of which 8.7 k are its prologue (first image: global load -> split -> LDS, nothing to overlap with), 13.4 k its
// epilogue (per-store address arithmetic, image-edge branches, three activator variants in one body), and while it is in either
// its SIMDs have one wave instead of two; on top the hardware kept only 1.77 of the 2 workgroup slots per CU filled (77 % of the
// time both: the two channel groups of a layer take different times and the dispatcher refills slots in order).  The narrow
// layers are worse off: CNN12 11.3 k + 7.2 k of 48 k.  The MFMAs a wave issues are 37 % of its life; two resident waves make the
// measured 0.60-0.64 matrix-pipe busy.  None of this is the K loop's fault, and all of it goes away when the workgroup does not end:
//
//   * item k+1's first image chunk is loaded, split and written to LDS while item k's last chunk computes, and its first two
//     filter taps are in the ring before item k's epilogue starts: the K loop of k+1 begins with its first MFMA;
//   * the epilogue has a fast path for what the stack runs (tile inside the image, no depth_to_space, no residual): one 64-bit
//     base per (item, destination), 32-bit lane offsets, PReLU or none -- ~450 VALU instructions instead of ~3000;
//   * residency is 2 workgroups per CU by construction, and the counter hands the next item to whichever workgroup is free, so
//     wide and narrow channel groups balance themselves (the older workgroup of a CU wins the issue arbitration and runs ahead).
//
// vmcnt bookkeeping (all waits are counted, s_waitcnt vmcnt(N) = "everything but the youngest N vector-memory operations has
// retired"; a wave's operations retire in issue order, stores and atomics included -- the compiler relies on the same rule):
// per tap F filter DMA instructions; at step 0 of a chunk, behind that tap's DMA, the IN_ROUNDS loads of the next image; in the
// epilogue one atomic (the item fetch) and, on the fast path, exactly 4 * NTV stores (the general path under-counts: safe, it only
// waits longer).  The counts per step are spelled out at the waits.  LDS-DMA must not be in flight when the workgroup ends
// (the LDS is handed to the next one): vmcnt(0) before s_endpgm.
#pragma once
#include "../dcscn-super-resolution_amd/csrc/conv3_h.hpp"

namespace dcscn {

template <int NT>
struct C3PGeom : C3HGeom<NT> {
    using B = C3HGeom<NT>;
    static constexpr int BA_BYTES = NT * 128;                     // bias | slopes of one channel group
    static constexpr int BA0 = B::BA_BASE;                        // two of them: item k+1's is written while item k's epilogue may still read
    static constexpr int MAIL = BA0 + 2 * BA_BYTES;               // item ids from the fetching wave to the others
    static constexpr int LDS_BYTES = MAIL + 64;
};

struct C3Item {
    int id, valid, tile_id, ntile, img, y0, x0;
    bool all_in, full;
    const char* a_base;      // origin of the halo tile (only in-image addresses are dereferenced)
    const char* f_base;      // the channel group's filter image
    unsigned ok_mask;        // per thread: staged item r lies inside the image
};
struct C3Next {              // what the item in progress needs to know about the one behind it: where its first image chunk and taps are
    int id, valid;
    bool all_in;
    const char* a_base;
    const char* f_base;
    unsigned ok_mask;
};

template <int N>
__device__ __forceinline__ void c3pl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS operations of this wave done, then the workgroup barrier; no fence semantics wanted (vector memory stays in flight across it)
__device__ __forceinline__ void c3pl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DBG (tuner only): 1 = shader-clock probes per wave through a.srctab: [0] entry, [1] exit, [2] items, [3] sum of epilogues,
// [4] sum of chunk boundaries (barrier + image write), [5] sum of (wait + barrier) per tap, [6] HW_ID, [7] XCC_ID
template <int NT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void conv3_hp(const ConvArgs a) {
    using G = C3PGeom<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem_c3p[];
    char* const smem = smem_c3p;
    constexpr int F = G::F_ROUNDS, L = G::IN_ROUNDS;
    constexpr int E_FULL = 1 + 4 * (NT >= 2 ? NT - 1 : 1);        // atomic + the stores of the narrower group: what a fast-path epilogue issues at least

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    const int H = a.H, W = a.W;
    const int n_groups = a.n_groups;
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;
    const int n_chunks = a.n_chunks;
    const int octs = a.tail_octs;                              // 0, or 1 / 2 / 3: the last chunk is a packed tail
    const int n_main = octs ? n_chunks - 1 : n_chunks;
    const int n_tail = (9 * octs + 3) >> 2;
    const bool fastable = a.ps == 1 && a.res == nullptr && (a.act == ACT_ALPHA || a.act == ACT_NONE);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned f_off = (unsigned)(lane * 16);
    const int cq = tid & 7;                                    // channel quad of every staged item of this thread
    long long pr_t0 = 0, pr_epi = 0, pr_cb = 0, pr_bar = 0, pr_a = 0;
    int pr_items = 0;
    if constexpr (DBG == 1) pr_t0 = __builtin_readcyclecounter();

    // ---- items ----
    auto decode = [&](int id, auto& it, auto full_c) DCSCN_INL {
        constexpr bool FULLREC = decltype(full_c)::value;
        // id = 8 k + x: the k-th item of XCD x's queue = channel group k % n_groups of pixel tile 8 (k / n_groups) + x -- the groups of one
        // tile are taken by workgroups of one XCD at about the same time and share the tile's input through that L2
        it.id = id;
        const int k = id >> 3, tl = k / n_groups;
        int tile_id = tl * 8 + (id & 7);
        int ntile = k - tl * n_groups;
        it.valid = tile_id < n_tiles;
        if (!it.valid) { tile_id = 0; ntile = 0; }            // past the end: a valid item nobody computes (its prefetches are harmless)
        int bid = tile_id;
        const int tx = bid % a.tiles_x;
        bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        const int img = bid / a.tiles_y;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        if constexpr (FULLREC) {
            it.tile_id = tile_id; it.ntile = ntile; it.img = img; it.y0 = y0; it.x0 = x0;
            it.full = y0 + G::TH <= H && x0 + G::TW <= W;
        }
        it.all_in = y0 >= 1 && x0 >= 1 && y0 + G::TH + 1 <= H && x0 + G::TW + 1 <= W;
        it.a_base = reinterpret_cast<const char*>(a.in + (size_t)img * H * W * a.in_stride + a.in_off + ((ptrdiff_t)(y0 - 1) * W + (x0 - 1)) * a.in_stride);
        it.f_base = reinterpret_cast<const char*>(a.wpack16) + (size_t)ntile * n_chunks * 9 * G::F_TAP_BYTES;
        unsigned m = 0;
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));                          // re-derived per item: hoisted out of the item loop the 22 positions stay in registers
        int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int gy = y0 - 1 + hrow, gx = x0 - 1 + hcol;
            const bool ok = r * 32 + hp0 < G::HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            m |= ok ? (1u << r) : 0u;
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
        it.ok_mask = m;
    };
    // Items are dealt per XCD: counter x (one cache line each) hands out k = 0, 1, ... and k stands for item 8 k + x -- 64 workgroups per
    // counter instead of 512 on one word (an agent-scope atomic is ~12 ns of a serialised L2 resource: one shared counter, fetched by
    // 2048 waves per round of items, cost ~50 k cycles per item).  Wave 0 does the atomic; the other waves issue a plain load of a
    // read-only word instead, so that every wave has the same count of vector-memory operations in flight.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    int* const my_counter = a.work + xcc * 16;
    auto fetch = [&](int n) DCSCN_INL -> int {
        int got = 0;
        if (lane == 0) {
            if (wave == 0) got = __hip_atomic_fetch_add(my_counter, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8 + (int)xcc;
            else got = *reinterpret_cast<const volatile int*>(a.bias);
        }
        return got;
    };
    auto post = [&](int v) DCSCN_INL { if (tid == 0) *reinterpret_cast<volatile int*>(smem + G::MAIL) = v; };
    auto collect = [&]() DCSCN_INL -> int { return __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem + G::MAIL)); };

    // ---- staging of an input image chunk: item = r * 256 + tid = (halo pixel, channel quad), conv3_h.hpp ----
    f32x4 gin[L];
    auto load_in = [&](const char* it_a_base, unsigned it_ok_mask, int chunk) DCSCN_INL {
        const int c0 = chunk * G::KC + cq * 4;
        const unsigned coff = (unsigned)((c0 < a.cin_phys ? c0 : 0) * 4);
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
        const int stride4 = a.in_stride * 4;
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int pix = ((it_ok_mask >> r) & 1u) ? hrow * W + hcol : W + 1;
            gin[r] = *reinterpret_cast<const f32x4*>(it_a_base + (size_t)((unsigned)(pix * stride4) + coff));
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    };
    const float m1 = opaque_minus_one();
    auto convert_in = [&](auto r_, bool it_all_in, unsigned it_ok_mask, int chunk) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        f32x4 x = gin[r];
        const bool whole = it_all_in && (chunk + 1) * G::KC <= a.cin_phys;       // block uniform: nothing to zero
        if (!whole) {
            const bool ok = chunk * G::KC + cq * 4 < a.cin_phys && ((it_ok_mask >> r) & 1u);
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
        static_for<0, L>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int hp = r * 32 + hp0;
            const int kq = cq >> 1;
            const int unit = c3h_unit(hcol, kq, 0);
            const int off = hp * G::PIX_BYTES + unit * 16 + (cq & 1) * 8;
            const u32x4 v = __builtin_bit_cast(u32x4, gin[r]);
            if (r < L - 1 || hp < G::HP) {
                *reinterpret_cast<u32x2*>(smem + off) = u32x2{v.x, v.y};
                *reinterpret_cast<u32x2*>(smem + (off ^ 16)) = u32x2{v.z, v.w};
            }
            hcol += 32 - G::HT;
            if (hcol >= G::HT) hcol -= G::HT;
        });
    };
    // one tap of filters (F_TAP_BYTES at src) -> ring slot at byte offset slot_off
    auto dma_f = [&](const char* src, int slot_off) DCSCN_INL {
        static_for<0, F>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int piece = (wave + 4 * r) % G::F_PIECES;
            glds16(src + piece * 1024, f_off, lds0 + (unsigned)slot_off + (unsigned)piece * 1024u);
        });
    };
    auto b_col = [&](auto kx_) DCSCN_INL {
        constexpr int kx = decltype(kx_)::value;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int hx = (l & 15) + kx;
        return (4 * wave * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, l >> 4, 0) * 16;
    };

    // ---- start: three item ids, the first item's image and first two taps ----
    C3Item cur;
    C3Next nxt;
    int nn_id;
    int fetched = 0;
    {
        const int got = fetch(3);
        post(got);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c3pl_barrier();
        const int id0 = collect();
        decode(id0, cur, std::true_type{});
        decode(id0 + 8, nxt, std::false_type{});
        nn_id = id0 + 16;
        fetched = nn_id;                                      // what the first item posts at step 3 (afterwards: the previous epilogue's fetch)
    }
    if (!cur.valid) return;
    int s0 = G::F_BASE, s1 = G::F_BASE + G::F_TAP_BYTES, s2 = G::F_BASE + 2 * G::F_TAP_BYTES;      // ring slots of steps 0, 1, 2 (mod 3)
    dma_f(cur.f_base, s0);                                    // step 0 = tap (ky 0, kx 0), step 1 = tap (ky 1, kx 0) = packed tap 3
    dma_f(cur.f_base + 3 * G::F_TAP_BYTES, s1);
    load_in(cur.a_base, cur.ok_mask, 0);
    static_for<0, L>([&](auto r_) DCSCN_INL { convert_in(r_, cur.all_in, cur.ok_mask, 0); });
    store_in();
    bool extra = false;                                       // the previous epilogue issued (at least) E_FULL operations behind the tap DMAs
    int parity = 0;

    // ---- one item: K loop + epilogue.  A narrow channel group (ntile >= n_full) has NT - 1 tiles: its last tile is skipped under a
    // block-uniform branch (one body for both kinds of group: two instantiations of this loop nest cost ~60 VGPRs in hoisted addresses)
    auto run_item = [&]() DCSCN_INL {
        constexpr int NTV = NT;
        const bool wide = NT == 1 || cur.ntile < a.n_full;
        const int ba = G::BA0 + parity * G::BA_BYTES;
        if (tid < NTV * 4) *reinterpret_cast<f32x4*>(smem + ba + tid * 16) = reinterpret_cast<const f32x4*>(a.bias + cur.ntile * NT * 16)[tid];
        else if (tid >= 64 && tid < 64 + NTV * 4 && a.act == ACT_ALPHA)
            *reinterpret_cast<f32x4*>(smem + ba + NT * 64 + (tid - 64) * 16) = reinterpret_cast<const f32x4*>(a.alpha + cur.ntile * NT * 16)[tid - 64];

        f32x4 acc[4][NTV];
        static_for<0, 4>([&](auto m_) DCSCN_INL {
            static_for<0, NTV>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
        });
        int b_hi = 0;
        const int a_lane = lane * 16;

        for (int chunk = 0; chunk < n_main; ++chunk) {
            const bool last_main = chunk + 1 == n_main;
            const bool ends = last_main && octs == 0;           // what follows this chunk is the next item
            const bool first = chunk == 0 && extra;
            const char* li_a_base = ends ? nxt.a_base : cur.a_base;      // whose image is loaded during this chunk
            const unsigned li_ok = ends ? nxt.ok_mask : cur.ok_mask;
            const bool li_all_in = ends ? nxt.all_in : cur.all_in;
            const int lchunk = ends ? 0 : chunk + 1;
            h8 xh[4], xl[4];
            static_for<0, 9>([&](auto s_) DCSCN_INL {
                constexpr int step = decltype(s_)::value;
                constexpr int kx = step / 3, ky = step % 3;
                constexpr int step2 = (step + 2) % 9;
                constexpr int ptap2 = (step2 % 3) * 3 + step2 / 3;
                const int slot = step % 3 == 0 ? s0 : step % 3 == 1 ? s1 : s2;
                const int slot2 = (step + 2) % 3 == 0 ? s0 : (step + 2) % 3 == 1 ? s1 : s2;
                if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
                // operations issued behind this tap's pieces: step 0: the DMA of step 8 (+ the previous epilogue); step 1: the DMA and the
                // image loads of step 0 (+ the previous epilogue); step 2: those loads and the DMA of step 1; later steps: one DMA
                if constexpr (step == 0) { if (first) c3pl_wait_vm<F + E_FULL>(); else c3pl_wait_vm<F>(); }
                else if constexpr (step == 1) { if (first) c3pl_wait_vm<F + L + E_FULL>(); else c3pl_wait_vm<F + L>(); }
                else if constexpr (step == 2) c3pl_wait_vm<F + L>();
                else c3pl_wait_vm<F>();
                c3pl_barrier();
                if constexpr (DBG == 1) pr_bar += __builtin_readcyclecounter() - pr_a;
                {   // the tap two steps ahead: of this chunk, of the next one, of the packed tail, or of the next item
                    const char* src;
                    if constexpr (step + 2 < 9) src = cur.f_base + (size_t)(chunk * 9 + ptap2) * G::F_TAP_BYTES;
                    else src = !last_main ? cur.f_base + (size_t)((chunk + 1) * 9 + ptap2) * G::F_TAP_BYTES
                             : octs      ? cur.f_base + (size_t)(n_main * 9 + step2) * G::F_TAP_BYTES
                                         : nxt.f_base + (size_t)ptap2 * G::F_TAP_BYTES;
                    dma_f(src, slot2);
                }
                if constexpr (step == 0) load_in(li_a_base, li_ok, lchunk);
                if constexpr (step == 3) { if (chunk == 0) post(fetched); }          // the item id fetched by the previous epilogue has arrived
                if constexpr (step == 5) { if (chunk == 0) nn_id = collect(); }
                if constexpr (ky == 0) b_hi = b_col(std::integral_constant<int, kx>{});
                static_for<(ky == 0 ? 0 : 3), 4>([&](auto m_) DCSCN_INL {
                    constexpr int row = ky + decltype(m_)::value;
                    xh[row & 3] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                    xl[row & 3] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
                });
                const char* fs = smem + a_lane + slot;
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    if (n < NT - 1 || wide) {
                        const h8 wh = *reinterpret_cast<const h8*>(fs + (2 * n) * 1024);
                        const h8 wl = *reinterpret_cast<const h8*>(fs + (2 * n + 1) * 1024);
                        static_for<0, 4>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            constexpr int q = (ky + m) & 3;
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[q], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[q], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[q], acc[m][n], 0, 0, 0);
                        });
                    }
                });
                // the image in flight becomes (hi, lo) pairs two items per tap from step 3 on
                if constexpr (step >= 3)
                    static_for<2 * (step - 3), (2 * (step - 3) + 2 < L ? 2 * (step - 3) + 2 : L)>([&](auto r_) DCSCN_INL { convert_in(r_, li_all_in, li_ok, lchunk); });
            });
            if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
            c3pl_barrier();                                    // every wave is past its last read of this chunk's image
            store_in();                                       // made visible by the barrier in front of the next tap
            if constexpr (DBG == 1) pr_cb += __builtin_readcyclecounter() - pr_a;
        }
        // Packed tail (kernels.h: c3h_tail_octs): (tap, octet) pairs four to an instruction; the next item's first image is loaded here
        if (octs) {
            int l = lane;
            asm volatile("" : "+v"(l));
            auto tail_step = [&](int step, auto first_c) DCSCN_INL {
                constexpr bool FIRST = decltype(first_c)::value;             // step 0, peeled: the image loads must not sit under a condition
                if (FIRST) c3pl_wait_vm<F>(); else if (step == 1 || step == 2) c3pl_wait_vm<F + L>(); else c3pl_wait_vm<F>();
                c3pl_barrier();
                {
                    const int k = step + 2 - n_tail;          // >= 0: step k of the next item (packed taps 0 and 3)
                    const char* src = k < 0 ? cur.f_base + (size_t)(n_main * 9 + step + 2) * G::F_TAP_BYTES : nxt.f_base + (size_t)(3 * k) * G::F_TAP_BYTES;
                    dma_f(src, s2);
                }
                if constexpr (FIRST) load_in(nxt.a_base, nxt.ok_mask, 0);
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
                const char* fs = smem + a_lane + s0;
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    if (n < NT - 1 || wide) {
                        const h8 wh = *reinterpret_cast<const h8*>(fs + (2 * n) * 1024);
                        const h8 wl = *reinterpret_cast<const h8*>(fs + (2 * n + 1) * 1024);
                        static_for<0, 4>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[m], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[m], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[m], acc[m][n], 0, 0, 0);
                        });
                    }
                });
                const int t = s0; s0 = s1; s1 = s2; s2 = t;   // the ring moves on by one slot per step
            };
            tail_step(0, std::true_type{});
            for (int step = 1; step < n_tail; ++step) tail_step(step, std::false_type{});
            static_for<0, L>([&](auto r_) DCSCN_INL { convert_in(r_, nxt.all_in, nxt.ok_mask, 0); });
            if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
            c3pl_barrier();
            store_in();
            if constexpr (DBG == 1) pr_cb += __builtin_readcyclecounter() - pr_a;
        }

        // ---- epilogue ----
        if constexpr (DBG == 1) pr_a = __builtin_readcyclecounter();
        fetched = fetch(1);
        const int ntile = cur.ntile;
        const int cb16 = ntile * NT * 16 - 16 * (ntile > a.n_full ? ntile - a.n_full : 0);    // conv channel of the group's first tile
        const float inv = a.inv_scale;
        const float zero = opaque_zero();
        float chk = 0.0f;
        const bool fast = fastable && cur.full;
        int le = lane;
        asm volatile("" : "+v"(le));                           // the epilogue's lane arithmetic stays in the epilogue
        const int lje = le & 15, lke = le >> 4;
        if (fast) {
            auto finish = [&](auto act_c) DCSCN_INL {
                constexpr int ACT_C = decltype(act_c)::value;
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    if (!(n < NT - 1 || wide)) return;
                    const int c0 = cb16 + n * 16;                                    // uniform
                    const bool first = c0 < a.split;
                    float* optr = first ? a.out0.ptr : a.out1.ptr;
                    const int ostride = first ? a.out0.stride : a.out1.stride;
                    const int ooff = first ? a.out0.off : a.out1.off;
                    const int owidth = first ? a.out0.width : a.out1.width;
                    const int cc0 = first ? c0 : c0 - a.split;
                    char* base = reinterpret_cast<char*>(optr + ((size_t)(cur.img * H + cur.y0) * W + cur.x0) * ostride + ooff + cc0);
                    const unsigned voff = (unsigned)(((4 * wave * W + lje) * ostride + 4 * lke) * 4);
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
            // general path (image edges, depth_to_space, residual, other activators): conv3_h's epilogue
            const int cbase = ntile * NT * 16 + 4 * lke;
            const int obase = cb16 + 4 * lke;
            const int act = a.act, ps = a.ps, orow = W * ps;
            const int gx = cur.x0 + lje;
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                (void)cbase;
                if (!(n < NT - 1 || wide)) return;
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
                if (ps != 1) {                                         // depth_to_space: channel (ay*ps + bx)*ps_c + ch
                    const int sub = cc / a.ps_c;
                    ch = cc - sub * a.ps_c;
                    ay = sub / ps;
                    bx = sub - ay * ps;
                }
                const bool live = gx < W && cc < owidth;
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    const int gy = cur.y0 + 4 * wave + m;
                    f32x4 v = acc[m][n] * inv + bv;
                    v.x = activate1(v.x, av.x, act);
                    v.y = activate1(v.y, av.y, act);
                    v.z = activate1(v.z, av.z, act);
                    v.w = activate1(v.w, av.w, act);
                    if (live && gy < H) {
                        chk = nonfinite_acc(chk, acc[m][n], zero);
                        const size_t pix = (size_t)((cur.img * H + gy) * ps + ay) * orow + (size_t)(gx * ps + bx);
                        if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch);
                        *reinterpret_cast<f32x4*>(optr + pix * ostride + ooff + ch) = v;
                    }
                });
            });
        }
        if (chk != chk && a.redo) a.redo[cur.tile_id] = 1;
        extra = fast;
        if constexpr (DBG == 1) { pr_epi += __builtin_readcyclecounter() - pr_a; ++pr_items; }
    };

    while (true) {
        run_item();
        if (!nxt.valid) break;
        decode(nxt.id, cur, std::true_type{});
        decode(nn_id, nxt, std::false_type{});
        parity ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no LDS-DMA may land after the workgroup has given its LDS back
    if constexpr (DBG == 1) {
        if (lane == 0 && a.srctab) {
            long long* pr = reinterpret_cast<long long*>(const_cast<void*>(a.srctab)) + ((size_t)blockIdx.x * 4 + wave) * 8;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            pr[0] = pr_t0; pr[1] = __builtin_readcyclecounter(); pr[2] = pr_items; pr[3] = pr_epi; pr[4] = pr_cb; pr[5] = pr_bar; pr[6] = hw; pr[7] = xcc;
        }
    }
}

}  // namespace dcscn
