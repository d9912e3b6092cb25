// LAB CODE (r02, measured and NOT adopted; correct: 358 parity tests passed with it wired into the plan).
// Result on C5 (dcscn_L7_F32to8 x4 DS, 1024 patches): the narrow feature layers take EXACTLY the time they take on
// conv_igemm<1,...,DWK=3> (CNN2 0.326 vs 0.329 ms, CNN3 0.270 vs 0.273, ... CNN7 0.166 vs 0.166) although this kernel keeps a
// tile's DMA in flight under the previous tile's compute and the other parks its waves 60 % of the time -- so neither the
// kernel structure nor latency is the limiter.  What both share is the traffic pattern: a 26-channel (104-byte) slice of the
// 140-channel NHWC concat record is a partial, misaligned cache-line access per pixel for reads and writes alike, i.e. about
// twice the useful bytes at the memory controller.  The fix is a data-layout change (dense per-layer buffers + a
// multi-source NIN GEMM), not a kernel.  Wide outputs were slower here (Up-PS 32->128: 1.08 vs 0.57 ms, the 128
// accumulators of NT = 8 spill; Up-PS2 0.83 vs 0.66).  Kept for the record; include path adjusted for tools/.
//
// conv_dsp: tf.nn.separable_conv2d (3x3 depthwise, channel multiplier 1, + 1x1 pointwise; helper/tf_graph.py:155-216) for the
// NARROW layers of the depthwise-separable models (<= 32 input channels: every layer of the shipped c-DCSCN DS net but the
// 131-wide NIN GEMM) as ONE persistent, software-pipelined kernel.
//
// Why: these layers move 150 MB - 1.5 GB each and do almost no arithmetic (AI 8-40); launched one 8x16 tile per workgroup
// (conv_igemm<1,...,DWK=3>) they ran at ~1.7 TB/s with the waves parked on s_waitcnt / barriers 60-68 % of the time
// (profiles/r02_c5_*: SQ_WAIT_ANY / SQ_WAVE_CYCLES) -- every workgroup loads, waits, computes, stores, with nothing in flight
// meanwhile.  Here a workgroup is resident for the whole launch and walks over 16x16 pixel tiles:
//
//   LDS  two input stages   [18x18 halo pixels][CK channels], pixel major: the DMA (global_load_lds_dwordx4, conv_wino2.hpp
//                           glds16) copies CK*4 contiguous bytes per pixel with CK/4 adjacent lanes;
//        DW buffer          [256 pixels][CK]: the depthwise result = the pointwise GEMM's B operand;
//        pointwise filter   [CK/16][s*4+k][NS] (row (blk, s, k) = channel 16*blk + 4k + s) and depthwise filter [9][CK]:
//                           loaded ONCE per workgroup.
//   per tile t:   issue the DMA of tile t+1 into the other stage
//                 depthwise: thread = (channel quad, pixel): 9 ds_read_b128 x 4 FMAs each, taps in (dy, dx) order, filter taps in
//                            registers; halo pixels outside the image are zeros (SAME padding: their slots are cleared by the
//                            lane that would have loaded them)
//                 barrier;  pointwise GEMM on v_mfma_f32_16x16x4_f32: wave w owns rows 4w..4w+3 of the tile (4 x NT accumulator
//                            tiles); one ds_read_b128 of the DW buffer feeds four k-steps (as conv_nin)
//                 s_waitcnt vmcnt(0) (tile t+1 landed; issued a whole tile earlier), epilogue (bias, activator, depth_to_space /
//                 residual / scalar forms of conv_igemm), barrier.
#pragma once
#include "../dcscn-super-resolution_amd/csrc/conv_wino2.hpp"

namespace dcscn {

template <int NT, int CK>
struct DspGeom {
    static_assert(CK == 16 || CK == 32, "channel chunk of the narrow separable layers");
    static constexpr int THREADS = 256;
    static constexpr int Q = CK / 4;                          // 16-byte quads per pixel
    static constexpr int HT = 18;                             // halo tile edge
    static constexpr int HP = HT * HT;                        // 324 halo pixels
    static constexpr int PB = CK * 4;                         // bytes per pixel record
    static constexpr int A_SLOTS = HP * Q;
    static constexpr int A_DMA = (A_SLOTS + 63) / 64;         // wave instructions per tile
    static constexpr int A_ROUNDS = (A_DMA + 3) / 4;
    static constexpr int A_BYTES = A_DMA * 1024;
    static constexpr int D_BYTES = 256 * PB;                  // depthwise output
    static constexpr int NS = conv_ns(NT);
    static constexpr int W_FLOATS = CK * NS;                  // pointwise filter image
    static constexpr int W_BYTES = ((W_FLOATS * 4 + 1023) / 1024) * 1024;
    static constexpr int DWW_BYTES = 9 * CK * 4;              // depthwise filter [tap][CK]
    static constexpr int OFF_D = 2 * A_BYTES;
    static constexpr int OFF_W = OFF_D + D_BYTES;
    static constexpr int OFF_DWW = OFF_W + W_BYTES;
    static constexpr int OFF_BA = OFF_DWW + ((DWW_BYTES + 1023) / 1024) * 1024;   // bias [NT*16] | slope [NT*16]
    static constexpr int LDS_BYTES = OFF_BA + 1024;
    static_assert(2 * NT * 16 * 4 <= 1024, "bias / slope block");
    static constexpr int PASSES = Q;                          // depthwise passes: 256 / Q pixels per pass
};

template <int NT, int CK>
__global__ __launch_bounds__(256, 1) void conv_dsp(const ConvArgs a) {
    using G = DspGeom<NT, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;
    const int H = a.H, W = a.W;
    const int n_tiles = a.N * a.tiles_y * a.tiles_x;

    // ---- once per workgroup: clear the stages (channel-tail quads are never loaded), fetch the filters ----
    {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = tid; i < (2 * G::A_BYTES + G::D_BYTES) / 16; i += G::THREADS) *reinterpret_cast<f32x4*>(lds + 16 * i) = z;
        for (int i = tid; i < G::W_FLOATS / 4; i += G::THREADS)
            *reinterpret_cast<f32x4*>(lds + G::OFF_W + 16 * i) = *reinterpret_cast<const f32x4*>(a.wpack + 4 * i);
        for (int i = tid; i < 9 * G::Q; i += G::THREADS)
            *reinterpret_cast<f32x4*>(lds + G::OFF_DWW + 16 * i) = *reinterpret_cast<const f32x4*>(a.dww + 4 * i);
        // bias and slope stay in LDS: the tile loop must not contain a single compiler-visible VMEM load (hipcc would wait
        // vmcnt(0) for it -- and with it for the LDS-DMA of the next tile that was just issued)
        for (int i = tid; i < NT * 4; i += G::THREADS) {
            *reinterpret_cast<f32x4*>(lds + G::OFF_BA + 16 * i) = *reinterpret_cast<const f32x4*>(a.bias + 4 * i);
            *reinterpret_cast<f32x4*>(lds + G::OFF_BA + NT * 64 + 16 * i) = *reinterpret_cast<const f32x4*>(a.alpha + 4 * i);
        }
        __syncthreads();
    }
    // DMA geometry of this lane's slots (tile independent): byte offset from the halo origin, halo coordinates
    unsigned a_off[G::A_ROUNDS];
    int a_hyx[G::A_ROUNDS];                                  // hy | hx << 8, or -1: the slot does not exist / holds channels past cin
    static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        const int slot = (wave + 4 * r) * 64 + lane;
        const int hp = slot / G::Q;
        const int q = slot - hp * G::Q;
        const int hy = hp / G::HT, hx = hp - hy * G::HT;
        const bool real = wave + 4 * r < G::A_DMA && slot < G::A_SLOTS && 4 * q < a.cin_phys;
        a_hyx[r] = real ? (hy | (hx << 8)) : -1;
        a_off[r] = (unsigned)(((hy * W + hx) * a.in_stride + 4 * q) * 4);
    });
    // this thread's depthwise work: channel quad dq of pixel (pass * 256/Q + tid / Q)
    const int dq = tid & (G::Q - 1);
    const int dp0 = tid / G::Q;
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(lds + G::OFF_DWW + (t * G::Q + dq) * 16);

    // DMA of one tile's halo region into a stage; out-of-image pixels get zeros from the lane that owns the slot
    auto stage_tile = [&](int tile, unsigned stage) DCSCN_INL {
        int bid = tile;
        const int tx = bid % a.tiles_x;
        bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        const int img = bid / a.tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        const float* base = a.in + ((size_t)img * H * W + (ptrdiff_t)(y0 - 1) * W + (x0 - 1)) * a.in_stride + a.in_off;   // wave-uniform
        static_for<0, G::A_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int piece = wave + 4 * r;
            if (piece < G::A_DMA) {                                                   // wave-uniform
                const int gy = y0 - 1 + (a_hyx[r] & 0xff), gx = x0 - 1 + (a_hyx[r] >> 8);
                const bool real = a_hyx[r] >= 0;
                const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
                if (real && inb) glds16(base, a_off[r], lds0 + stage * G::A_BYTES + (unsigned)piece * 1024u);
                else if (real) *reinterpret_cast<f32x4*>(lds + stage * G::A_BYTES + 16 * (piece * 64 + lane)) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        });
    };

    typedef const volatile __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    const int b_lane = G::OFF_W + (lk * G::NS + lj) * 4;
    const int act = a.act;
    const int ps = a.ps;

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    stage_tile(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (unsigned it = 0; tile < n_tiles; ++it, tile += gridDim.x) {
        const unsigned st = it & 1;
        const int next = tile + gridDim.x;
        if (next < n_tiles) stage_tile(next, st ^ 1);

        // ---- depthwise: stage st -> DW buffer ----
        static_for<0, G::PASSES>([&](auto p_) DCSCN_INL {
            constexpr int pass = decltype(p_)::value;
            const int p = pass * (256 / G::Q) + dp0;
            const int py = p >> 4, px = p & 15;
            const unsigned src = lds0 + st * G::A_BYTES + (unsigned)((py * G::HT + px) * G::PB + dq * 16);
            f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 v = *(lds_f32x4_ptr)(uintptr_t)(src + (dy * G::HT + dx) * G::PB);
                    sum += v * wt[dy * 3 + dx];
                }
            *reinterpret_cast<f32x4*>(lds + G::OFF_D + p * G::PB + dq * 16) = sum;
        });
        __syncthreads();

        // ---- pointwise GEMM: wave w owns tile rows 4w..4w+3 ----
        f32x4 acc[4][NT];
        static_for<0, 4>([&](auto m_) DCSCN_INL {
            static_for<0, NT>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
        });
        static_for<0, CK / 16>([&](auto b_) DCSCN_INL {
            constexpr int blk = decltype(b_)::value;
            f32x4 xv[4];
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                xv[m] = *(lds_f32x4_ptr)(uintptr_t)(lds0 + G::OFF_D + (unsigned)((16 * (4 * wave + m) + lj) * G::PB + blk * 64 + lk * 16));
            });
            static_for<0, 4>([&](auto s_) DCSCN_INL {
                constexpr int s = decltype(s_)::value;
                float wv[NT];
                static_for<0, NT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    wv[n] = *reinterpret_cast<const float*>(lds + b_lane + ((blk * 16 + s * 4) * G::NS + n * 16) * 4);
                });
                static_for<0, 4>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    static_for<0, NT>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n], xv[m][s], acc[m][n], 0, 0, 0);
                    });
                });
            });
        });
        // the next tile's DMA was issued a whole tile ago; the previous tile's stores are older still
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---- epilogue (conv_igemm's): bias, activator, (depth_to_space), (residual), store ----
        int bid = tile;
        const int tx = bid % a.tiles_x;
        bid /= a.tiles_x;
        const int ty = bid % a.tiles_y;
        const int img = bid / a.tiles_y;
        const int gx = tx * 16 + lj;
        const int gy0 = ty * 16 + 4 * wave;
        const int cbase = 4 * lk;
        const int orow = W * ps;                                   // destination pixels per row
        if (gx < W) {
            auto finish = [&](auto act_c, auto vec_c) DCSCN_INL {
                constexpr int ACT_C = decltype(act_c)::value;
                constexpr bool VEC = decltype(vec_c)::value;
                const int act_e = ACT_C >= 0 ? ACT_C : act;
                static_for<0, NT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    const int c = cbase + n * 16;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(lds + G::OFF_BA + 4 * c);
                    f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (act_e == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(lds + G::OFF_BA + NT * 64 + 4 * c);
                    const bool first = c < a.split;
                    float* optr = first ? a.out0.ptr : a.out1.ptr;
                    const int ostride = first ? a.out0.stride : a.out1.stride;
                    const int ooff = first ? a.out0.off : a.out1.off;
                    const int owidth = first ? a.out0.width : a.out1.width;
                    const int cc = first ? c : c - a.split;
                    const size_t dy = (size_t)ps * orow * ostride;     // one LR row down in the destination
                    if constexpr (VEC) {
                        int ch = cc, ay = 0, bx = 0;
                        if (ps != 1) {
                            const int sub = cc / a.ps_c;
                            ch = cc - sub * a.ps_c;
                            ay = sub / ps;
                            bx = sub - ay * ps;
                        }
                        const size_t pix0 = (size_t)((img * H + gy0) * ps + ay) * orow + (size_t)(gx * ps + bx);
                        float* o0 = optr + pix0 * ostride + ooff + ch;
                        const bool live = cc < owidth;
                        static_for<0, 4>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            f32x4 v = acc[m][n] + bv;
                            v.x = activate1(v.x, av.x, act_e);
                            v.y = activate1(v.y, av.y, act_e);
                            v.z = activate1(v.z, av.z, act_e);
                            v.w = activate1(v.w, av.w, act_e);
                            if (live && gy0 + m < H) {
                                if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + (pix0 + (size_t)(m * ps) * orow) * a.res_stride + ch);
                                *reinterpret_cast<f32x4*>(o0 + m * dy) = v;
                            }
                        });
                    } else {
                        size_t pix0[4];
                        int ch[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ccr = cc + r;
                            int ay = 0, bx = 0;
                            ch[r] = ccr;
                            if (ps != 1) {
                                const int sub = ccr / a.ps_c;
                                ch[r] = ccr - sub * a.ps_c;
                                ay = sub / ps;
                                bx = sub - ay * ps;
                            }
                            pix0[r] = (size_t)((img * H + gy0) * ps + ay) * orow + (size_t)(gx * ps + bx);
                        }
                        static_for<0, 4>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            f32x4 v = acc[m][n] + bv;
                            v.x = activate1(v.x, av.x, act_e);
                            v.y = activate1(v.y, av.y, act_e);
                            v.z = activate1(v.z, av.z, act_e);
                            v.w = activate1(v.w, av.w, act_e);
                            if (gy0 + m < H) {
                                const float vr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (cc + r < owidth) {
                                        const size_t pix = pix0[r] + (size_t)(m * ps) * orow;
                                        float out = vr[r];
                                        if (a.res) out += a.res[pix * a.res_stride + ch[r]];
                                        optr[pix * ostride + ooff + ch[r]] = out;
                                    }
                                }
                            }
                        });
                    }
                });
            };
            using std::integral_constant;
            if (a.vec4) {
                if (act == ACT_ALPHA) finish(integral_constant<int, ACT_ALPHA>{}, integral_constant<bool, true>{});
                else if (act == ACT_NONE) finish(integral_constant<int, ACT_NONE>{}, integral_constant<bool, true>{});
                else finish(integral_constant<int, -1>{}, integral_constant<bool, true>{});
            } else {
                finish(integral_constant<int, -1>{}, integral_constant<bool, false>{});
            }
        }
        __syncthreads();          // DW buffer and stage st are free again; stage st^1 (waited for above) is visible to every wave
    }
}

}  // namespace dcscn
