// conv_wino (r01): the register-staged Winograd kernel that shipped in round 1, kept as the timing reference of
// tools/wino_tune.hip / tools/wino2_tune.hip.  The library now runs csrc/conv_wino2.hpp (LDS-DMA staged).
//
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
//   channel) and forms its 16 V_f values with 32 adds.  No transformed-input buffer exists anywhere.
// * the output transform A^T m A is wave-local too: after the K loop a lane holds m_f for its tile and 4
//   output channels for every f, so the 2x2 output pixels are 24 adds away; then bias/activator/store.
// * filters are transformed once on the host in float64 (G g G^T), rounded to f32 and packed in the LDS
//   image [f][kk][NS] per channel chunk.
// * K loop: one LDS buffer, the next chunk's global loads are issued right after this chunk's LDS writes (register
//   prefetch), two barriers per chunk; filter operands are read PF frequencies ahead of their MFMAs.
//
// This is the shipped form.  tools/conv_wino_lab.hpp carries the same kernel with every variant that was measured
// and NOT adopted (LDS double buffering, LDS-DMA filters, software-pipelined loops, 8-wave workgroups, wider filter
// reads, ...) plus the s_memtime instrumentation; profiles/r01_wino_tune_log.txt has the numbers.
//
// Numerics: F(2x2,3x3) has transform entries 0, +-1, +-1/2 only; measured error of one 196->166 layer is
// 1.8x the direct form's (2e-4 vs 1.1e-4 on outputs of magnitude 275) and the end-to-end max-abs error of
// the L12 network is unchanged at 1.6e-5 (dominated by the final add) -- inside the 1e-4 parity bar.
#pragma once
#include "../dcscn-super-resolution_amd/csrc/conv_igemm.hpp"

namespace dcscn {

template <int NT, int KC>
struct WinoGeom {
    static constexpr int THREADS = 256;
    static constexpr int TH = 16, TW = 16;               // output pixels per workgroup
    static constexpr int HTH = TH + 2, HTW = TW + 2;     // halo tile
    static constexpr int HP = HTH * HTW;
    static constexpr int PS = conv_plane_stride(HP);     // floats per channel plane of the halo tile
    static constexpr int NS = conv_ns(NT);               // floats per (f, kk) filter row
    static constexpr int KQ = KC / 4;                    // MFMA k-steps per chunk
    static constexpr int A_FLOATS = KC * PS;
    static constexpr int B_FLOATS = 16 * KC * NS;
    static constexpr int BUF = A_FLOATS + B_FLOATS;
    static constexpr int A_ITEMS = HP * KQ;              // (pixel, channel quad) float4 pieces of the halo tile
    static constexpr int A_LOADS = (A_ITEMS + THREADS - 1) / THREADS;
    static constexpr int B_VEC = B_FLOATS / 4;
    static constexpr int B_LOADS = (B_VEC + THREADS - 1) / THREADS;
};

// NT: channel tiles per group as packed (LDS image, bias indexing); NTV <= NT: tiles that are real in this
// workgroup's group (the last group of a layer may be narrower) -- compile time, so the MFMA stream stays branch
// free.  PF: how many frequencies ahead of their MFMAs the filter operands are read (hipcc on its own keeps a
// single operand pair in flight and waits lgkmcnt(0) in front of every frequency).
template <int NT, int NTV, int KC, int PF>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a, float* smem, int tile_id, int ntile) {
    using G = WinoGeom<NT, KC>;
    constexpr int THREADS = G::THREADS;
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

    // ---- staging (conv_igemm's scheme with a 16x16 pixel tile) ----
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
    const float* b_src = a.wpack + (size_t)ntile * a.n_chunks * G::B_FLOATS + 4 * tid;
    const int c_last = a.cin_phys - 4;

    f32x4 areg[G::A_LOADS];
    f32x4 breg[G::B_LOADS];

    // NOTE: the filter buffer is over-allocated by one staging sweep, so the last (partial) sweep of a chunk may be
    // loaded by every thread; only its LDS store is predicated.
    auto load_chunk = [&](int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const int c = c0 + a_c4[i];
            areg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (c < c_last ? c : c_last));
        });
        const float* bs = b_src + (size_t)chunk * G::B_FLOATS;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            breg[i] = *reinterpret_cast<const f32x4*>(bs + 4 * THREADS * i);
        });
    };
    auto store_chunk = [&](float* buf, int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (a_item[i]) {
                // zero padding (halo outside the image, channels past cin) by bit mask: exact for any loaded bit
                // pattern and, unlike `keep ? v : 0`, never compiled into branches
                const bool keep = a_inb[i] && (G::KQ == 1 || c0 + a_c4[i] < a.cin_phys);
                const unsigned m = keep ? 0xffffffffu : 0u;
                float* d = buf + a_dst[i];
                d[0] = __uint_as_float(__float_as_uint(areg[i].x) & m);
                d[G::PS] = __uint_as_float(__float_as_uint(areg[i].y) & m);
                d[2 * G::PS] = __uint_as_float(__float_as_uint(areg[i].z) & m);
                d[3 * G::PS] = __uint_as_float(__float_as_uint(areg[i].w) & m);
            }
        });
        float* bd = buf + G::A_FLOATS + 4 * tid;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (i < G::B_LOADS - 1 || tid + THREADS * i < G::B_VEC)
                *reinterpret_cast<f32x4*>(bd + 4 * THREADS * i) = breg[i];
        });
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
    const int b_lane = G::A_FLOATS + lk * G::NS + lj;

    // raw 4x4 patch of this lane's (tile, channel)
    auto read_raw = [&](const float* As, float (&d)[4][4]) DCSCN_INL {
        static_for<0, 4>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const float2 lo = *reinterpret_cast<const float2*>(As + i * G::HTW);
            const float2 hi = *reinterpret_cast<const float2*>(As + i * G::HTW + 2);
            d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
        });
    };
    // V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    auto transform = [&](const float (&d)[4][4], float (&v)[16]) DCSCN_INL {
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
    // the 16*NTV MFMAs of one k-step, filter operands read PF frequencies ahead
    auto mfma_step = [&](const float* Bs, const float (&v)[16]) DCSCN_INL {
        float wq[PF + 1][NTV];
        static_for<0, PF>([&](auto p_) DCSCN_INL {
            constexpr int pf = decltype(p_)::value;
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                wq[pf][n] = Bs[(pf * KC) * G::NS + n * 16];
            });
        });
        static_for<0, 16>([&](auto f_) DCSCN_INL {
            constexpr int f = decltype(f_)::value;
            if constexpr (f + PF < 16) {
                static_for<0, NTV>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    wq[(f + PF) % (PF + 1)][n] = Bs[((f + PF) * KC) * G::NS + n * 16];
                });
                __builtin_amdgcn_sched_group_barrier(0x100, NTV, 0);   // DS reads of f + PF
            }
            static_for<0, NTV>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[f % (PF + 1)][n], v[f], acc[f][n], 0, 0, 0);
            });
            __builtin_amdgcn_sched_group_barrier(0x8, NTV, 0);          // MFMAs of f
        });
    };
    auto compute = [&](const float* buf) DCSCN_INL {
        const float* As = buf + a_lane;
        const float* Bs = buf + b_lane;
        // one 4-channel MFMA step per iteration; NOT unrolled across steps so that only one step's raw patch /
        // transformed operands are live next to the 16*NT accumulators
#pragma unroll 1
        for (int ks = 0; ks < G::KQ; ++ks, As += 4 * G::PS, Bs += 4 * G::NS) {
            float d[4][4], v[16];
            read_raw(As, d);
            transform(d, v);
            mfma_step(Bs, v);
        }
    };

    // ---- K loop ----
    // The loads of chunk c+1 are issued as early as their registers are free -- right after the LDS writes of
    // chunk c, before the barrier (s_memtime stamps: ~17 % of a chunk was the vmcnt wait in front of the LDS store).
    load_chunk(0);
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        store_chunk(smem, chunk);
        if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
        __syncthreads();
        compute(smem);
        __syncthreads();
    }

    // ---- output transform (wave-local) + epilogue ----
    // Runs once per workgroup but is not free: the first version (per-position 64-bit index math, a per-value
    // activator switch, a scalar-store fallback) cost ~21 k cycles against ~4 k per chunk.  Now: vec4 stores only
    // (the host routes anything else to conv_igemm), every load issued up front, the activator dispatched once, one
    // 64-bit multiply per channel tile and constant strides between the four positions of a lane's 2x2 output block.
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

// launch_bounds' second argument is waves per SIMD = resident 4-wave workgroups per CU
template <int NT, int KC, int WPS, int PF = 3>
__global__ __launch_bounds__(256, WPS) void conv_wino(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware decode of the 1-D grid.  Workgroup ids go round-robin over the 8 XCDs (own L2 each); the channel
    // groups of one pixel tile all read the same input tile, so they get ids that are congruent mod 8 and close
    // together: [8 tiles of group 0][the same 8 tiles of group 1] ... -- they land on ONE XCD at about the same time
    // and all but the first find the input tile in that L2.
    // With many groups the span is limited to `group_span` of them at a time (phases over the whole image set).
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
    const int nt_valid = (ntile == G - 1) ? a.n_full /* r01: tiles in the last group */ : NT;                // block uniform
    if (nt_valid == NT) conv_wino_body<NT, NT, KC, PF>(a, smem, tile_id, ntile);
    else if constexpr (NT >= 2) {
        if (nt_valid == NT - 1) conv_wino_body<NT, NT - 1, KC, PF>(a, smem, tile_id, ntile);
        else if constexpr (NT >= 3) {
            if (nt_valid == NT - 2) conv_wino_body<NT, NT - 2, KC, PF>(a, smem, tile_id, ntile);
        }
    }
}

}  // namespace dcscn
