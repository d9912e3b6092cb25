// conv_wino: 3x3 SAME convolution as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          (Lavin & Gray 2015; 16 multiplies per 2x2 outputs
//                                                         instead of 36 = 2.25x fewer MFMAs than conv_igemm)
//
// 16 "frequency" GEMMs D_f[cout][tile] += U_f[cout][cin] * V_f[cin][tile], one per entry of the 4x4
// transformed tile.  Mapping (differs from the usual "16 batched GEMMs" on purpose, to fit CDNA4):
//
// * workgroup = 4 waves = a 16x16 output-pixel tile = 8x8 Winograd tiles; wave w owns tile rows 2w, 2w+1
//   (16 tiles = the 16 columns of the MFMA B operand) for ALL 16 frequencies and NT*16 output channels
//   (NT <= 3: 16*NT accumulators of 4 VGPRs).  Lane (j = lane & 15, k = lane >> 4) is tile j, channel k of
//   the 4-deep MFMA step.
// * the input transform B^T d B is done ON THE FLY in registers: the raw halo tile is staged in LDS
//   exactly as conv_igemm does (channel-major planes), each lane reads the 4x4 raw patch of its (tile,
//   channel) and forms its 16 V_f values with 32 adds -- VALU work that co-issues with the MFMAs.  No
//   transformed-input buffer exists anywhere.
// * the output transform A^T m A is wave-local too: after the K loop a lane holds m_f for its tile and 4
//   output channels for every f, so the 2x2 output pixels are 24 adds away; then bias/activator/store.
// * filters are transformed once on the host in float64 (G g G^T), rounded to f32 and packed in the LDS
//   image [f][kk][NS] per channel chunk.
//
// Numerics: F(2x2,3x3) has transform entries 0, +-1, +-1/2 only; measured error of one 196->166 layer is
// 1.8x the direct form's (2e-4 vs 1.1e-4 on outputs of magnitude 275) and the end-to-end max-abs error of
// the L12 network is unchanged at 1.6e-5 (dominated by the final add) -- inside the 1e-4 parity bar.
#pragma once
// LAB VERSION (tools/ only): the Winograd kernel with every experiment of profiles/r01_wino_tune_log.txt as a
// template flag (double buffering, LDS-DMA filters, software-pipelined loops, 8-wave workgroups, side-by-side filter
// layout, ablations, s_memtime instrumentation).  The library ships the plain single-buffer loop of
// dcscn-super-resolution_amd/csrc/conv_wino.hpp; tools/wino_tune.hip checks both against the direct kernel.
#include "../dcscn-super-resolution_amd/csrc/conv_igemm.hpp"

#ifndef DCSCN_WINO_BVEC
#define DCSCN_WINO_BVEC 0
#endif
namespace dcscn_lab {
using namespace dcscn;
constexpr bool kWinoBVec = DCSCN_WINO_BVEC != 0;
__host__ __device__ constexpr int wino_npad(int nt) { return nt == 3 ? 4 : nt; }
__host__ __device__ constexpr int wino_lds_ns(int nt) { return kWinoBVec ? 16 * wino_npad(nt) : conv_ns(nt); }
__host__ __device__ constexpr int wino_glb_ns(int nt) { return kWinoBVec ? 16 * nt : conv_ns(nt); }
__host__ __device__ constexpr int wino_glb_col(int nt, int jn) { return kWinoBVec ? (jn % 16) * nt + jn / 16 : jn; }
}  // namespace dcscn_lab

namespace dcscn_lab {

template <int NT, int KC, int WAVES = 4>
struct WinoGeom {
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TH = 4 * WAVES, TW = 16;
    static constexpr int HTH = TH + 2, HTW = TW + 2;
    static constexpr int HP = HTH * HTW;
    static constexpr int PS = conv_plane_stride(HP);
    static constexpr int NS = wino_lds_ns(NT);        // floats per (f, kk) filter row in LDS
    static constexpr int GNS = wino_glb_ns(NT);       // ... in the global image
    static constexpr int NPAD = wino_npad(NT);
    static constexpr bool SCATTER = NS != GNS;        // NT = 3: 12-byte items padded to 16 on the way into LDS
    static constexpr int KQ = KC / 4;
    static constexpr int A_FLOATS = KC * PS;
    static constexpr int B_FLOATS = 16 * KC * NS;
    static constexpr int BUF = A_FLOATS + B_FLOATS;
    static constexpr int A_ITEMS = HP * KQ;
    static constexpr int A_LOADS = (A_ITEMS + THREADS - 1) / THREADS;
    static constexpr int GB_FLOATS = 16 * KC * GNS;   // one chunk of one group in global memory
    // staging items: float4 of the linear image, or (SCATTER) one (f, kk, j) triple
    static constexpr int B_VEC = SCATTER ? 16 * KC * 16 : GB_FLOATS / 4;
    static constexpr int B_LOADS = (B_VEC + THREADS - 1) / THREADS;
};

// NT: channel tiles per group as packed (LDS image, bias indexing); NTV <= NT: tiles that are real in
// this workgroup's group (the last group of a layer may be narrower) -- compile time, so the MFMA
// stream stays branch free.
template <int NT, int NTV, int KC, bool DB = false, int ABLATE = 0, int WAVES = 4, bool DMA = false, int PRIO = 0, int PF = 0, bool VPIPE = false>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a, float* smem) {
    using G = WinoGeom<NT, KC, WAVES>;
    constexpr int THREADS = G::THREADS;

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

    // ---- staging (same scheme as conv_igemm with a 16x16 pixel tile, single LDS buffer) ----
    const float* a_src[G::A_LOADS];
    int a_dst[G::A_LOADS];
    int a_c4[G::A_LOADS];
    bool a_item[G::A_LOADS], a_inb[G::A_LOADS];
    static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
        constexpr int i = decltype(i_)::value;
        const int item = tid + THREADS * i;
        const int hp = item / G::KQ;
        const int q = item - hp * G::KQ;
        const int hy = hp / G::HTW;
        const int hx = hp - hy * G::HTW;
        const int gy = y0 + hy - 1;
        const int gx = x0 + hx - 1;
        a_item[i] = item < G::A_ITEMS;
        a_inb[i] = a_item[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
        a_c4[i] = 4 * q;
        a_dst[i] = 4 * q * G::PS + hp;
        // always a valid address (pixel (0,0) for halo positions outside the image): loads are issued
        // unconditionally and masked when written to LDS, so no branch sits between a load and its use
        a_src[i] = in_img + ((size_t)(a_inb[i] ? gy : 0) * W + (a_inb[i] ? gx : 0)) * a.in_stride;
    });
    const float* b_src = a.wpack + (size_t)ntile * a.n_chunks * G::GB_FLOATS + (G::SCATTER ? 3 : 4) * tid;
    const int c_last = a.cin_phys - 4;

    f32x4 areg[G::A_LOADS];
    using bvec_t = std::conditional_t<G::SCATTER, f32x3, f32x4>;
    bvec_t breg[G::B_LOADS];

    // NOTE: the filter buffer is over-allocated by one staging sweep, so the last (partial) sweep of a
    // chunk may be loaded by every thread; only its LDS store is predicated.
    auto load_a = [&](int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const int c = c0 + a_c4[i];
            if constexpr (ABLATE == 9 || ABLATE == 11)   // tuner only: same bytes, fully coalesced (wrong data)
                areg[i] = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(blockIdx.x & 1023) * 49 + chunk) * 2048 + 4 * (tid + THREADS * i));
            else
                areg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (c < c_last ? c : c_last));
        });
    };
    auto load_b = [&](int chunk) DCSCN_INL {
        const float* bs = b_src + ((ABLATE == 10 || ABLATE == 11) ? 0 : (size_t)chunk * G::GB_FLOATS);   // 10/11: tuner only
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if constexpr (G::SCATTER) {
                const float* q = bs + 3 * THREADS * i;           // 12-byte items: 4-byte aligned only
                breg[i] = f32x3{q[0], q[1], q[2]};
            } else {
                breg[i] = *reinterpret_cast<const f32x4*>(bs + 4 * THREADS * i);
            }
        });
    };
    auto load_chunk = [&](int chunk) DCSCN_INL {
        load_a(chunk);
        load_b(chunk);
    };
    auto store_a = [&](float* buf, int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (a_item[i]) {
                // zero padding (halo outside the image, channels past cin) by bit mask: exact for any
                // loaded bit pattern and, unlike `keep ? v : 0`, never compiled into branches
                const bool keep = a_inb[i] && (G::KQ == 1 || c0 + a_c4[i] < a.cin_phys);
                const unsigned m = keep ? 0xffffffffu : 0u;
                float* d = buf + a_dst[i];
                d[0] = __uint_as_float(__float_as_uint(areg[i].x) & m);
                d[G::PS] = __uint_as_float(__float_as_uint(areg[i].y) & m);
                d[2 * G::PS] = __uint_as_float(__float_as_uint(areg[i].z) & m);
                d[3 * G::PS] = __uint_as_float(__float_as_uint(areg[i].w) & m);
            }
        });
    };
    auto store_b = [&](float* buf) DCSCN_INL {
        float* bd = buf + G::A_FLOATS + 4 * tid;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (i < G::B_LOADS - 1 || tid + THREADS * i < G::B_VEC)
                *reinterpret_cast<bvec_t*>(bd + 4 * THREADS * i) = breg[i];   // SCATTER: 12 of each 16 bytes
        });
    };
    auto store_chunk = [&](float* buf, int chunk) DCSCN_INL {
        store_a(buf, chunk);
        store_b(buf);
    };

    f32x4 acc[16][NTV];
    static_for<0, 16>([&](auto f_) DCSCN_INL {
        static_for<0, NTV>([&](auto n_) DCSCN_INL {
            acc[decltype(f_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        });
    });

    // this lane's Winograd tile: rows 2w, 2w+1 of the 8x8 tile grid, 8 tiles per row
    const int tr = 2 * wave + (lj >> 3);
    const int tc = lj & 7;
    const int a_lane = lk * G::PS + (2 * tr) * G::HTW + 2 * tc;   // raw 4x4 patch origin in the halo tile
    const int b_lane = G::A_FLOATS + lk * G::NS + (kWinoBVec ? lj * G::NPAD : lj);

    // raw 4x4 patch of this lane's (tile, channel) for k-step `ks` -> transformed operands v[16]
    auto read_raw = [&](const float* As, float (&d)[4][4]) DCSCN_INL {
        static_for<0, 4>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const float2 lo = *reinterpret_cast<const float2*>(As + i * G::HTW);
            const float2 hi = *reinterpret_cast<const float2*>(As + i * G::HTW + 2);
            d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
        });
    };
    auto transform = [&](const float (&d)[4][4], float (&v)[16]) DCSCN_INL {
        // V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float r[4][4];
        static_for<0, 4>([&](auto j_) DCSCN_INL {
            constexpr int j = decltype(j_)::value;
            r[0][j] = d[0][j] - d[2][j];
            r[1][j] = d[1][j] + d[2][j];
            r[2][j] = d[2][j] - d[1][j];
            r[3][j] = d[1][j] - d[3][j];
        });
        static_for<0, 4>([&](auto x_) DCSCN_INL {
            constexpr int x = decltype(x_)::value;
            v[4 * x + 0] = r[x][0] - r[x][2];
            v[4 * x + 1] = r[x][1] + r[x][2];
            v[4 * x + 2] = r[x][2] - r[x][1];
            v[4 * x + 3] = r[x][1] - r[x][3];
        });
    };
    // the 16*NTV MFMAs of one k-step; filter operands read PF frequencies ahead (PF = 0: hipcc's order,
    // which keeps a single operand pair in flight and stalls on every frequency)
    // filter operands of frequency f for this lane's NTV channel tiles
    auto read_w = [&](const float* Bs, int f, float (&w)[NTV]) DCSCN_INL {
        const float* q = Bs + (f * KC) * G::NS;
        if constexpr (!kWinoBVec) {
            static_for<0, NTV>([&](auto n_) DCSCN_INL { w[decltype(n_)::value] = q[decltype(n_)::value * 16]; });
        } else if constexpr (NTV == 1) {
            w[0] = q[0];
        } else if constexpr (NTV == 2) {
            const float2 t = *reinterpret_cast<const float2*>(q);
            w[0] = t.x; w[1] = t.y;
        } else {
            const f32x4 t = *reinterpret_cast<const f32x4*>(q);   // b128 (4 LDS cycles) rather than b96 (8); the pad lane is dead
            w[0] = t.x; w[1] = t.y; w[2] = t.z;
        }
    };
    constexpr int W_READS = kWinoBVec ? 1 : NTV;   // DS instructions per read_w
    constexpr int kHookF = 3;                      // frequency after which the pipelined loop stages the next chunk
    auto mfma_step = [&](const float* Bs, const float (&v)[16], auto&& hook) DCSCN_INL {
        if constexpr (PF < 0) {
            // tuner only: let LLVM's IGroupLP strategy (-PF - 1) interleave DS reads and MFMAs
            __builtin_amdgcn_iglp_opt(-PF - 1);
            static_for<0, 16>([&](auto f_) DCSCN_INL {
                constexpr int f = decltype(f_)::value;
                float w[NTV];
                read_w(Bs, f, w);
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n], v[f], acc[f][n], 0, 0, 0);
                });
            });
        } else if constexpr (PF == 0) {
            static_for<0, 16>([&](auto f_) DCSCN_INL {
                constexpr int f = decltype(f_)::value;
                float w[NTV];
                if constexpr (ABLATE == 6) {   // tuner only: no filter reads
                    static_for<0, NTV>([&](auto n_) DCSCN_INL { w[decltype(n_)::value] = breg[0].x; });
                } else {
                    read_w(Bs, f, w);
                }
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n], v[f], acc[f][n], 0, 0, 0);
                });
            });
        } else {
            float wq[(PF > 0 ? PF : 0) + 1][NTV];
            static_for<0, PF>([&](auto p_) DCSCN_INL {
                constexpr int pf = decltype(p_)::value;
                read_w(Bs, pf, wq[pf]);
            });
            static_for<0, 16>([&](auto f_) DCSCN_INL {
                constexpr int f = decltype(f_)::value;
                if constexpr (f + PF < 16) {
                    read_w(Bs, f + PF, wq[(f + PF) % (PF + 1)]);
                    __builtin_amdgcn_sched_group_barrier(0x100, W_READS, 0);   // DS reads of f + PF
                }
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[f % (PF + 1)][n], v[f], acc[f][n], 0, 0, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, NTV, 0);          // MFMAs of f
                hook(f_);                                                   // pipelined loops hang work on chosen frequencies
            });
        }
    };

    auto no_hook = [](auto) DCSCN_INL {};
    auto compute_h = [&](const float* buf, auto&& hook) DCSCN_INL {
        const float* As = buf + a_lane;
        const float* Bs = buf + b_lane;
        if constexpr (PRIO > 0 && PRIO < 10) __builtin_amdgcn_s_setprio(PRIO);
        if constexpr (VPIPE && G::KQ > 1) {
            // k-steps fully unrolled; the raw patch of step ks+1 is read and transformed while the MFMAs
            // of step ks run (needs a second operand set: 16 more VGPRs)
            float va[16], vb[16];
            {
                float d[4][4];
                read_raw(As, d);
                transform(d, va);
            }
            static_for<0, G::KQ>([&](auto ks_) DCSCN_INL {
                constexpr int ks = decltype(ks_)::value;
                float d[4][4];
                if constexpr (ks + 1 < G::KQ) read_raw(As + (ks + 1) * 4 * G::PS, d);
                if constexpr (ks % 2 == 0) mfma_step(Bs + ks * 4 * G::NS, va, no_hook);
                else mfma_step(Bs + ks * 4 * G::NS, vb, no_hook);
                if constexpr (ks + 1 < G::KQ) {
                    if constexpr (ks % 2 == 0) transform(d, vb);
                    else transform(d, va);
                }
            });
        } else {
            // one 4-channel MFMA step per iteration; NOT unrolled across steps so that only one step's
            // raw patch / transformed operands are live next to the 16*NT accumulators
#pragma unroll 1
            for (int ks = 0; ks < G::KQ; ++ks, As += 4 * G::PS, Bs += 4 * G::NS) {
                float d[4][4], v[16];
                if constexpr (ABLATE == 5) {
                    // tuner only: no raw read / transform.  Operands forged from the lane id, made opaque once per
                    // chunk (forging them from a staging register would put a vmcnt wait into the compute phase)
                    float seed = (float)lane;
                    asm volatile("" : "+v"(seed));
                    static_for<0, 16>([&](auto f_) DCSCN_INL { v[decltype(f_)::value] = seed; });
                } else if constexpr (ABLATE == 12) {
                    // tuner only: raw patch read kept, transform replaced by copies (no VALU adds)
                    read_raw(As, d);
                    static_for<0, 16>([&](auto f_) DCSCN_INL { v[decltype(f_)::value] = d[decltype(f_)::value / 4][decltype(f_)::value % 4]; });
                } else {
                    read_raw(As, d);
                    transform(d, v);
                }
                if constexpr (G::KQ == 1) mfma_step(Bs, v, hook);   // the hook exists for one-step chunks only
                else mfma_step(Bs, v, no_hook);
            }
        }
        if constexpr (PRIO > 0 && PRIO < 10) __builtin_amdgcn_s_setprio(0);
    };
    auto compute = [&](const float* buf) DCSCN_INL { compute_h(buf, no_hook); };

    // ---- filters by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no registers) ----
    auto dma_filters = [&](int chunk, float* buf) DCSCN_INL {
        constexpr int PIECES = G::B_FLOATS / 256;
        static_assert(!DMA || G::B_FLOATS % 256 == 0, "filter block must be whole 1 KB DMA pieces");
        static_assert(!DMA || !G::SCATTER, "the DMA experiment copies the filter image linearly");
        const float* src = a.wpack + ((size_t)ntile * a.n_chunks + chunk) * G::B_FLOATS + 4 * lane;
        float* dst = buf + G::A_FLOATS;
        static_for<0, (PIECES + WAVES - 1) / WAVES>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const int piece = wave + WAVES * i;            // wave uniform
            if (PIECES % WAVES == 0 || piece < PIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + 256 * piece),
                    (__attribute__((address_space(3))) void*)(dst + 256 * piece), 16, 0, 0);
        });
    };
    auto load_input = [&](int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const int c = c0 + a_c4[i];
            areg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (c < c_last ? c : c_last));
        });
    };
    auto store_input = [&](float* buf, int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (a_item[i]) {
                // zero padding (halo outside the image, channels past cin) by bit mask: exact for any
                // loaded bit pattern and, unlike `keep ? v : 0`, never compiled into branches
                const bool keep = a_inb[i] && (G::KQ == 1 || c0 + a_c4[i] < a.cin_phys);
                const unsigned m = keep ? 0xffffffffu : 0u;
                float* d = buf + a_dst[i];
                d[0] = __uint_as_float(__float_as_uint(areg[i].x) & m);
                d[G::PS] = __uint_as_float(__float_as_uint(areg[i].y) & m);
                d[2 * G::PS] = __uint_as_float(__float_as_uint(areg[i].z) & m);
                d[3 * G::PS] = __uint_as_float(__float_as_uint(areg[i].w) & m);
            }
        });
    };

    if constexpr (PRIO >= 10) {
        // tuner only: de-phase co-resident workgroups (bit (PRIO - 10) of the block index sleeps first)
        if ((blockIdx.x >> (PRIO - 10)) & 1) __builtin_amdgcn_s_sleep(20);
    }
    if constexpr (DMA) {
        // two LDS buffers; filters of chunk c+1 stream in by DMA and the input tile of chunk c+1 sits in
        // registers while chunk c is multiplied; ONE barrier per chunk (hipcc drains vmcnt(0) -- i.e. the
        // DMA -- in front of __syncthreads by itself)
        dma_filters(0, smem);
        load_input(0);
        store_input(smem, 0);
        __syncthreads();
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            float* cur = smem + (chunk & 1) * G::BUF;
            float* nxt = smem + ((chunk + 1) & 1) * G::BUF;
            const bool more = chunk + 1 < a.n_chunks;
            if (more) {
                dma_filters(chunk + 1, nxt);
                load_input(chunk + 1);
            }
            compute(cur);
            if (more) store_input(nxt, chunk + 1);
            __syncthreads();
        }
    } else if constexpr (ABLATE == 2) {
        // tuner only: pure compute phase (no staging, no barriers) -- the ceiling of the MFMA loop
        load_chunk(0);
        store_chunk(smem, 0);
        __syncthreads();
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            float* b = smem;
            asm volatile("" : "+v"(b));     // opaque per iteration: the LDS reads cannot be hoisted
            compute(b);
        }
    } else if constexpr (ABLATE == 3) {
        // tuner only: barriers kept, no staging
        load_chunk(0);
        store_chunk(smem, 0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else if constexpr (ABLATE == 4) {
        // tuner only: LDS stores kept, no barriers, no global loads (racy)
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_chunk(smem, 0);
            compute(smem);
        }
    } else if constexpr (ABLATE == 1) {
        // tuner only: staging and barriers kept, global loads issued once
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_chunk(smem, 0);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else if constexpr (DB && VPIPE) {
        // Fully software-pipelined loop: two [input | filter] LDS buffers, ONE barrier per chunk, and every
        // non-MFMA job of a chunk hidden behind the MFMAs of the previous one IN THE SAME WAVE:
        //   after frequency 3 : registers -> LDS (filters of chunk c+1, input tile of chunk c+2), then the
        //                       global loads of filters c+2 / input c+3
        //   after frequency 7 : raw 4x4 patch of chunk c+1 LDS -> registers (v[0..7] of chunk c are dead)
        //   after frequency 11: its input transform (32 adds), interleaved with the last MFMAs
        // Chunk indices are clamped instead of branched on, so the loop body is one basic block; the few
        // redundant loads / stores of the last two iterations land in dead buffers.
        static_assert(G::KQ == 1 && PF > 0, "pipelined loop: one MFMA step per chunk, prefetched filter operands");
        const int last = a.n_chunks - 1;
        auto clamp = [&](int c) DCSCN_INL { return c < last ? c : last; };
        float v[16];
        load_chunk(0);
        store_chunk(smem, 0);                 // input 0, filters 0 -> buffer 0
        load_a(clamp(1));
        store_a(smem + G::BUF, clamp(1));     // input 1 -> buffer 1
        load_b(clamp(1));
        load_a(clamp(2));
        __syncthreads();
        {
            float d[4][4];
            read_raw(smem + a_lane, d);
            transform(d, v);
        }
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            float* cur = smem + (chunk & 1) * G::BUF;
            float* nxt = smem + ((chunk + 1) & 1) * G::BUF;
            float dn[4][4], vn[16];
            mfma_step(cur + b_lane, v, [&](auto f_) DCSCN_INL {
                constexpr int F = decltype(f_)::value;
                // ABLATE 21 / 22 / 23 (tuner only): no staging / no raw read + transform / neither
                if constexpr (F == 3 && ABLATE != 21 && ABLATE != 23) {
                    store_b(nxt);                          // filters c+1
                    store_a(cur, clamp(chunk + 2));        // input c+2 over input c (read during chunk c-1)
                    load_b(clamp(chunk + 2));
                    load_a(clamp(chunk + 3));
                } else if constexpr (F == 7 && ABLATE != 22 && ABLATE != 23) {
                    read_raw(nxt + a_lane, dn);
                } else if constexpr (F == 11) {
                    if constexpr (ABLATE != 22 && ABLATE != 23) transform(dn, vn);
                    else static_for<0, 16>([&](auto g_) DCSCN_INL { vn[decltype(g_)::value] = v[decltype(g_)::value]; });
                }
            });
            __syncthreads();
            static_for<0, 16>([&](auto f_) DCSCN_INL { v[decltype(f_)::value] = vn[decltype(f_)::value]; });
        }
    } else if constexpr (DB) {
        // Software-pipelined loop over two LDS buffers, ONE barrier per chunk: while the MFMAs of chunk c
        // run, the same wave writes chunk c+1 (in registers since the middle of the previous chunk) into
        // the other buffer and issues the global loads of chunk c+2 -- staging sits in the MFMA shadow
        // instead of between two barriers.
        static_assert(G::KQ == 1 && PF > 0, "pipelined loop: one MFMA step per chunk, prefetched filter operands");
        load_chunk(0);
        store_chunk(smem, 0);
        if (a.n_chunks > 1) load_chunk(1);
        __syncthreads();
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            float* cur = smem + (chunk & 1) * G::BUF;
            float* nxt = smem + ((chunk + 1) & 1) * G::BUF;
            compute_h(cur, [&](auto f_) DCSCN_INL {
                if constexpr (decltype(f_)::value == kHookF) {
                    if (chunk + 1 < a.n_chunks) {
                        store_chunk(nxt, chunk + 1);
                        if (chunk + 2 < a.n_chunks) load_chunk(chunk + 2);
                    }
                }
            });
            __syncthreads();
        }
    } else if constexpr (ABLATE == 7) {
        // tuner only: s_memtime stamps at the phase boundaries of the shipped loop; per-wave cycle sums
        // of the four phases go to the debug buffer passed in `a.alpha` (16 longs per wave)
        long long t_store = 0, t_bar1 = 0, t_comp = 0, t_bar2 = 0;
        const long long c_begin = __builtin_readcyclecounter();
        const long long r_begin = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            const long long t0 = __builtin_readcyclecounter();
            store_chunk(smem, chunk);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long t1 = __builtin_readcyclecounter();
            __syncthreads();
            const long long t2 = __builtin_readcyclecounter();
            if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
            compute(smem);
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[15][NTV - 1]));
            const long long t3 = __builtin_readcyclecounter();
            __syncthreads();
            const long long t4 = __builtin_readcyclecounter();
            t_store += t1 - t0; t_bar1 += t2 - t1; t_comp += t3 - t2; t_bar2 += t4 - t3;
        }
        if (lane == 0 && blockIdx.x < 4096) {
            long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(a.alpha)) + ((size_t)blockIdx.x * 4 + wave) * 4 + 1024;
            dbg[0] = t_store; dbg[1] = t_bar1; dbg[2] = t_comp; dbg[3] = t_bar2;
            if (wave == 0 && blockIdx.x < 512) {   // shader-clock calibration: s_memtime vs s_memrealtime
                long long* cal = reinterpret_cast<long long*>(const_cast<float*>(a.alpha)) + blockIdx.x * 2;
                cal[0] = __builtin_readcyclecounter() - c_begin;
                cal[1] = __builtin_amdgcn_s_memrealtime() - r_begin;
            }
        }
        if (tid == 0) {
            long long* life = reinterpret_cast<long long*>(const_cast<float*>(a.alpha)) + 1024 + 4096 * 16 +
                              ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            life[4] = c_begin; life[5] = __builtin_readcyclecounter();
        }
    } else if constexpr (ABLATE == 8) {
        // tuner only: the pre-r01 order (loads issued after the first barrier)
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_chunk(smem, chunk);
            __syncthreads();
            if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
            compute(smem);
            __syncthreads();
        }
    } else {
        // The loop is bound by the latency of the chunk loads (s_memtime stamps: ~17 % of a chunk is the
        // vmcnt wait in front of the LDS store), so the loads of chunk c+1 are issued as early as their
        // registers are free -- right after the LDS writes of chunk c, before the barrier.
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_chunk(smem, chunk);
            if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    }

    // ---- output transform (wave-local) + epilogue ----
    // Runs once per workgroup but is not free: the first version (per-position 64-bit index math, a
    // scalar-store fallback) cost ~21 k cycles against ~4 k per chunk. Now: vec4 stores only (the host
    // routes anything else to conv_igemm), every load issued up front, one 64-bit multiply per channel
    // tile and constant strides between the four positions of a lane's 2x2 output block.
    const int gy0 = y0 + 2 * tr;
    const int gx0 = x0 + 2 * tc;
    const int cbase = ntile * NT * 16 + 4 * lk;
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
    // the activator is a launch constant: dispatch once, not once per stored value
    auto finish = [&](auto act_c) DCSCN_INL {
    constexpr int ACT_C = decltype(act_c)::value;
    const int act_e = ACT_C >= 0 ? ACT_C : act;
    static_for<0, NTV>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        const int c = cbase + n * 16;
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
        f32x4 t0[4], t1[4];
        static_for<0, 4>([&](auto nu_) DCSCN_INL {
            constexpr int nu = decltype(nu_)::value;
            t0[nu] = acc[0 + nu][n] + acc[4 + nu][n] + acc[8 + nu][n];
            t1[nu] = acc[4 + nu][n] - acc[8 + nu][n] - acc[12 + nu][n];
        });
        f32x4 yv[2][2];
        yv[0][0] = t0[0] + t0[1] + t0[2];
        yv[0][1] = t0[1] - t0[2] - t0[3];
        yv[1][0] = t1[0] + t1[1] + t1[2];
        yv[1][1] = t1[1] - t1[2] - t1[3];
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

// WAVES waves per workgroup = a (4*WAVES) x 16 output-pixel tile; launch_bounds' second argument is
// waves per SIMD: WPS workgroups of 4 waves, or WPS/2 workgroups of 8.
template <int NT, int KC, int WPS, bool DB = false, int ABLATE = 0, int WAVES = 4, bool DMA = false, int PRIO = 0, int PF = 0, bool VPIPE = false>
__global__ __launch_bounds__(64 * WAVES, WPS) void conv_wino(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nt_valid = (blockIdx.y == gridDim.y - 1) ? a.n_full : NT;   // block uniform
    if constexpr (ABLATE == 7) {
        // tuner only: workgroup lifetime in shader cycles and 100 MHz ticks (wave 0)
        const long long c0 = __builtin_readcyclecounter();
        const long long r0 = __builtin_amdgcn_s_memrealtime();
        conv_wino_body<NT, NT, KC, DB, ABLATE, WAVES, DMA, PRIO, PF, VPIPE>(a, smem);
        const long long c1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            long long* life = reinterpret_cast<long long*>(const_cast<float*>(a.alpha)) + 1024 + 4096 * 16 +
                              ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            life[0] = r0; life[1] = __builtin_amdgcn_s_memrealtime(); life[2] = c0; life[3] = __builtin_readcyclecounter();
            life[6] = c1;
        }
        return;
    }
    if (nt_valid == NT) conv_wino_body<NT, NT, KC, DB, ABLATE, WAVES, DMA, PRIO, PF, VPIPE>(a, smem);
    else if constexpr (NT >= 2) {
        if (nt_valid == NT - 1) conv_wino_body<NT, NT - 1, KC, DB, ABLATE, WAVES, DMA, PRIO, PF, VPIPE>(a, smem);
        else if constexpr (NT >= 3) {
            if (nt_valid == NT - 2) conv_wino_body<NT, NT - 2, KC, DB, ABLATE, WAVES, DMA, PRIO, PF, VPIPE>(a, smem);
        }
    }
}

}  // namespace dcscn_lab
