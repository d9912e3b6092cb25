// conv_igemm: the implicit-GEMM convolution kernel template (see kernels.hip for the overview).
// Shared by kernels.hip (the shipped variant table) and tools/conv_tune.hip (variant exploration).
#pragma once
#include <type_traits>

#include "kernels.h"

namespace dcscn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

// ---------------------------------------------------------------------------------------------
// activator (helper/tf_graph.py:77-102)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float activate1(float v, float alpha, int act) {
    switch (act) {
        case ACT_ALPHA:   return v > 0.0f ? v : alpha * v;   // == relu(v) + alpha*(v-|v|)*0.5 in f32
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_TANH:    return tanhf(v);
        case ACT_SELU:    return 1.0507009873554805f * (v > 0.0f ? v : 1.6732632423543772f * (expf(v) - 1.0f));
        default:          return v;
    }
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution on v_mfma_f32_16x16x4_f32
//
// GEMM view per tap: D[cout][pixel] += W[cout][cin] * X[cin][pixel]; the MFMA "A" operand (rows) is
// the filter, the "B" operand (columns) is 16 consecutive pixels of one image row, so a lane ends up
// holding 4 consecutive output channels of one pixel = one float4 NHWC store.
//
// Workgroup = 4 waves, pixel tile = (4*MT) rows x 16 columns, all NT*16 output channels of its
// channel tile.  Wave w owns rows [w*MT, (w+1)*MT).  K is walked in chunks of KC input channels;
// for each chunk the input tile (with a 1-pixel zero halo for 3x3: SAME padding is re-applied at every
// layer) and the KS*KS*KC*NT*16 filter block are staged in LDS.  The global loads of chunk c+1 are
// issued before the MFMAs of chunk c (register prefetch); DB selects whether they are then written to a
// second LDS buffer (one barrier per chunk) or to the same buffer after a second barrier (half the LDS,
// more workgroups per CU -- the shipped choice).
//
// LDS image, per buffer:
//   A: [KC][PS]        input, channel-major planes of the halo tile (PS = 16 mod 32)
//   B: [taps][KC][NS]  filters (NS = 16 mod 32)
// Both operands are read with ds_read_b32 where lanes 0-15 walk 16 consecutive floats and lanes 16-31
// the same 16 floats of the next k-plane; the plane strides put those on the other half of the 32
// banks, so every read is conflict free.
// ---------------------------------------------------------------------------------------------
template <int KS, int MT, int NT, int KC>
struct ConvGeom {
    static constexpr int TAPS = KS * KS;
    static constexpr int HALO = KS / 2;
    static constexpr int TH = 4 * MT;
    static constexpr int TW = 16;
    static constexpr int HTH = TH + 2 * HALO;
    static constexpr int HTW = TW + 2 * HALO;
    static constexpr int HP = HTH * HTW;
    static constexpr int PS = conv_plane_stride(HP);
    static constexpr int NS = conv_ns(NT);
    static constexpr int KQ = KC / 4;
    static constexpr int A_FLOATS = KC * PS;
    static constexpr int B_FLOATS = TAPS * KC * NS;
    static constexpr int BUF = A_FLOATS + B_FLOATS;
    static constexpr int A_ITEMS = HP * KQ;
    static constexpr int A_LOADS = (A_ITEMS + 255) / 256;
    static constexpr int B_VEC = B_FLOATS / 4;
    static constexpr int B_LOADS = (B_VEC + 255) / 256;
    static_assert(KC % 4 == 0, "KC must be a multiple of the MFMA k extent");
    static_assert(BUF % 4 == 0 && A_FLOATS % 4 == 0, "LDS carve must stay 16-byte aligned");
};

// LDS regions of the fused depthwise stage (DWK x DWK depthwise in front of a 1x1 GEMM): the raw halo tile of the
// chunk, channel-major like A, and the chunk's depthwise filter [tap][KC]; they follow A and B in the buffer.
template <int MT, int KC, int DWK>
struct DwGeom {
    static constexpr int TH = 4 * MT, TW = 16;
    static constexpr int RH = TH + DWK - 1, RW = TW + DWK - 1;
    static constexpr int RHP = RH * RW;
    static constexpr int PSR = conv_plane_stride(RHP);
    static constexpr int KQ = KC / 4;
    static constexpr int R_FLOATS = KC * PSR;
    static constexpr int W_FLOATS = DWK * DWK * KC;
    static constexpr int FLOATS = DWK > 0 ? R_FLOATS + W_FLOATS : 0;
    static constexpr int R_ITEMS = RHP * KQ;
    static constexpr int R_LOADS = (R_ITEMS + 255) / 256;
    static constexpr int W_ITEMS = DWK * DWK * KQ;            // float4 pieces of the filter chunk
    static constexpr int PIX = TH * TW;                       // output pixels of the tile
    static constexpr int PAIRS = KC * PIX / 256;              // (channel, pixel) results per thread
    static_assert(DWK == 0 || (PIX <= 256 && 256 % PIX == 0 && (KC * PIX) % 256 == 0 && W_ITEMS <= 256),
                  "depthwise stage: one thread per pixel and channel stripe");
};

// Compile-time loop: the index reaches the body as a constant, so register arrays (accumulators,
// operand fragments, staging registers) are only ever indexed statically and stay in VGPRs whatever
// the optimiser's unrolling heuristics decide (a runtime-indexed f32x4 array lands in scratch).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#define DCSCN_INL __attribute__((always_inline))

template <int KS, int MT, int NT, int KC, bool DB = true, int WPS = 2, int DWK = 0>
__global__ __launch_bounds__(256, WPS) void conv_igemm(const ConvArgs a) {
    static_assert(DWK == 0 || KS == 1, "the fused depthwise stage feeds a pointwise (1x1) GEMM");
    using G = ConvGeom<KS, MT, NT, KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;   // pixel column within the 16-wide segment / filter row within a 16-tile
    const int lk = lane >> 4;   // k index within the 4-deep MFMA step / channel quad of the result

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    if (a.redo_check && (a.redo[0] == 0 || a.redo[1 + img] == 0)) return;   // float32 plan behind a split16 pass: flagged images only
    const int ntile = blockIdx.y;
    const int y0 = ty * G::TH;
    const int x0 = tx * G::TW;
    const int H = a.H, W = a.W;

    const float* in_img = a.in + (size_t)img * H * W * a.in_stride + a.in_off;

    // ---- staging descriptors (constant over the K loop) ----
    const float* a_src[G::A_LOADS];
    int a_dst[G::A_LOADS];
    int a_c4[G::A_LOADS];
    bool a_item[G::A_LOADS], a_inb[G::A_LOADS];
    static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
        constexpr int i = decltype(i_)::value;
        const int item = tid + 256 * i;
        const int hp = item / G::KQ;
        const int q = item - hp * G::KQ;
        const int hy = hp / G::HTW;
        const int hx = hp - hy * G::HTW;
        const int gy = y0 + hy - G::HALO;
        const int gx = x0 + hx - G::HALO;
        a_item[i] = item < G::A_ITEMS;
        a_inb[i] = a_item[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
        a_c4[i] = 4 * q;
        a_dst[i] = 4 * q * G::PS + hp;
        a_src[i] = in_img + ((size_t)(a_inb[i] ? gy : 0) * W + (a_inb[i] ? gx : 0)) * a.in_stride + 4 * q;
    });
    const float* b_src = a.wpack + (size_t)ntile * a.n_chunks * G::B_FLOATS + 4 * tid;

    f32x4 areg[G::A_LOADS];
    f32x4 breg[G::B_LOADS];

    auto load_chunk = [&](int chunk) DCSCN_INL {
        const int c0 = chunk * KC;
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (a_inb[i] && c0 + a_c4[i] < a.cin_phys) v = *reinterpret_cast<const f32x4*>(a_src[i] + c0);
            areg[i] = v;
        });
        const float* bs = b_src + (size_t)chunk * G::B_FLOATS;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (i < G::B_LOADS - 1 || tid + 256 * i < G::B_VEC)
                breg[i] = *reinterpret_cast<const f32x4*>(bs + 1024 * i);
        });
    };
    auto store_chunk = [&](float* buf) DCSCN_INL {
        static_for<0, G::A_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (a_item[i]) {
                float* d = buf + a_dst[i];
                d[0] = areg[i].x;
                d[G::PS] = areg[i].y;
                d[2 * G::PS] = areg[i].z;
                d[3 * G::PS] = areg[i].w;
            }
        });
        float* bd = buf + G::A_FLOATS + 4 * tid;
        static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            if (i < G::B_LOADS - 1 || tid + 256 * i < G::B_VEC)
                *reinterpret_cast<f32x4*>(bd + 1024 * i) = breg[i];
        });
    };

    f32x4 acc[MT][NT];
    static_for<0, MT>([&](auto m_) DCSCN_INL {
        static_for<0, NT>([&](auto n_) DCSCN_INL {
            acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        });
    });

    const int a_lane = lk * G::PS + wave * MT * G::HTW + lj;
    const int b_lane = G::A_FLOATS + lk * G::NS + lj;

    // one filter tap (dy, dx): KQ k-steps of MT*NT MFMAs; As / Bs already point at the tap's row
    auto compute_tap = [&](const float* As, const float* Bs, auto dx_) DCSCN_INL {
        constexpr int dx = decltype(dx_)::value;
        static_for<0, G::KQ>([&](auto ks_) DCSCN_INL {
            constexpr int ks = decltype(ks_)::value;
            float xv[MT], wv[NT];
            static_for<0, MT>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                xv[m] = As[(ks * 4) * G::PS + m * G::HTW + dx];
            });
            static_for<0, NT>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                wv[n] = Bs[(dx * KC + ks * 4) * G::NS + n * 16];
            });
            static_for<0, MT>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                static_for<0, NT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n], xv[m], acc[m][n], 0, 0, 0);
                });
            });
        });
    };
    auto compute = [&](const float* buf) DCSCN_INL {
        const float* As = buf + a_lane;
        const float* Bs = buf + b_lane;
        if constexpr (KS <= 3) {
            // fully unrolled over the taps: every LDS offset is an immediate
            static_for<0, KS>([&](auto dy_) DCSCN_INL {
                constexpr int dy = decltype(dy_)::value;
                static_for<0, KS>([&](auto dx_) DCSCN_INL { compute_tap(As + dy * G::HTW, Bs + dy * KS * KC * G::NS, dx_); });
            });
        } else {
            // 5x5 / 7x7: the tap rows are a real loop (25-49 unrolled taps of MT*NT MFMAs would only bloat the
            // kernel); accumulators stay statically indexed, only the two LDS base addresses advance
#pragma unroll 1
            for (int dy = 0; dy < KS; ++dy, As += G::HTW, Bs += KS * KC * G::NS)
                static_for<0, KS>([&](auto dx_) DCSCN_INL { compute_tap(As, Bs, dx_); });
        }
    };

    // ---- K loop ----
    if constexpr (DWK > 0) {
        // Separable conv (tf.nn.separable_conv2d, tf_graph.py:155-177): the depthwise half runs inside the workgroup.
        // Per chunk: raw halo tile + filter block + depthwise filter -> LDS; every thread then forms its (channel,
        // pixel) depthwise sums from LDS (taps in (dy, dx) order; halo pixels outside the image are zeros = SAME
        // padding) and writes them as the A operand; MFMAs.  The depthwise output never touches HBM, and each input
        // element is fetched once per workgroup (the first version gathered the 9 neighbours from global memory for
        // every staged element: CNN2 of the DS L7 model 0.55 ms vs LDS-staged 0.2x).
        using D = DwGeom<MT, KC, DWK>;
        float* Rs = smem + G::BUF;                       // [KC][PSR] raw halo tile
        float* Ws = Rs + D::R_FLOATS;                    // [tap][KC] depthwise filter chunk
        const float* r_src[D::R_LOADS];
        int r_dst[D::R_LOADS], r_c4[D::R_LOADS];
        bool r_item[D::R_LOADS], r_inb[D::R_LOADS];
        static_for<0, D::R_LOADS>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            const int item = tid + 256 * i;
            const int hp = item / D::KQ;
            const int q = item - hp * D::KQ;
            const int hy = hp / D::RW, hx = hp - hy * D::RW;
            const int gy = y0 + hy - DWK / 2, gx = x0 + hx - DWK / 2;
            r_item[i] = item < D::R_ITEMS;
            r_inb[i] = r_item[i] && gy >= 0 && gy < H && gx >= 0 && gx < W;
            r_c4[i] = 4 * q;
            r_dst[i] = 4 * q * D::PSR + hp;
            r_src[i] = in_img + ((size_t)(r_inb[i] ? gy : 0) * W + (r_inb[i] ? gx : 0)) * a.in_stride + 4 * q;
        });
        const int w_tap = tid / D::KQ, w_q = tid - w_tap * D::KQ;     // this thread's float4 of the filter chunk
        f32x4 rreg[D::R_LOADS], wreg;
        auto load_dw = [&](int chunk) DCSCN_INL {
            const int c0 = chunk * KC;
            static_for<0, D::R_LOADS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (r_inb[i] && c0 + r_c4[i] < a.cin_phys) v = *reinterpret_cast<const f32x4*>(r_src[i] + c0);
                rreg[i] = v;
            });
            wreg = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (tid < D::W_ITEMS && c0 + 4 * w_q < a.cin_phys)
                wreg = *reinterpret_cast<const f32x4*>(a.dww + (size_t)w_tap * a.cin_phys + c0 + 4 * w_q);
            const float* bs = b_src + (size_t)chunk * G::B_FLOATS;
            static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                if (i < G::B_LOADS - 1 || tid + 256 * i < G::B_VEC) breg[i] = *reinterpret_cast<const f32x4*>(bs + 1024 * i);
            });
        };
        auto store_dw = [&]() DCSCN_INL {
            static_for<0, D::R_LOADS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                if (r_item[i]) {
                    float* d = Rs + r_dst[i];
                    d[0] = rreg[i].x;
                    d[D::PSR] = rreg[i].y;
                    d[2 * D::PSR] = rreg[i].z;
                    d[3 * D::PSR] = rreg[i].w;
                }
            });
            if (tid < D::W_ITEMS) *reinterpret_cast<f32x4*>(Ws + w_tap * KC + 4 * w_q) = wreg;
            float* bd = smem + G::A_FLOATS + 4 * tid;
            static_for<0, G::B_LOADS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                if (i < G::B_LOADS - 1 || tid + 256 * i < G::B_VEC) *reinterpret_cast<f32x4*>(bd + 1024 * i) = breg[i];
            });
        };
        // depthwise: thread -> pixel p of the tile and channels c = cb, cb + 256/PIX, ...
        const int p = tid % D::PIX, cb = tid / D::PIX;
        const int prow = p / D::TW, pcol = p - prow * D::TW;
        auto depthwise = [&]() DCSCN_INL {
            static_for<0, D::PAIRS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value;
                const int c = cb + (256 / D::PIX) * i;
                const float* r = Rs + c * D::PSR + prow * D::RW + pcol;
                float sum = 0.0f;
#pragma unroll
                for (int dy = 0; dy < DWK; ++dy)
#pragma unroll
                    for (int dx = 0; dx < DWK; ++dx) sum += r[dy * D::RW + dx] * Ws[(dy * DWK + dx) * KC + c];
                smem[c * G::PS + p] = sum;
            });
        };
        load_dw(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_dw();
            __syncthreads();
            if (chunk + 1 < a.n_chunks) load_dw(chunk + 1);
            depthwise();
            __syncthreads();
            compute(smem);
            __syncthreads();
        }
    } else if constexpr (DB) {
        // LDS double buffered: one barrier per chunk
        load_chunk(0);
        store_chunk(smem);
        __syncthreads();
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            float* cur = smem + (chunk & 1) * G::BUF;
            float* nxt = smem + ((chunk + 1) & 1) * G::BUF;
            const bool more = chunk + 1 < a.n_chunks;
            if (more) load_chunk(chunk + 1);
            compute(cur);
            if (more) store_chunk(nxt);
            __syncthreads();
        }
    } else {
        // one LDS buffer, next chunk prefetched into registers during the MFMAs: half the LDS, so more
        // workgroups per CU cover each other's staging phases; two barriers per chunk
        load_chunk(0);
        for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
            store_chunk(smem);
            __syncthreads();
            if (chunk + 1 < a.n_chunks) load_chunk(chunk + 1);
            compute(smem);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, activator, (depth_to_space), (residual), store ----
    // Runs once per workgroup, but its instruction count matters against a ~4 k-cycle chunk: activator
    // and store form are launch constants and dispatched ONCE (not per stored value), the destination
    // index math is done once per channel tile, rows differ by a constant stride.
    const int gx = x0 + lj;
    const int gy0 = y0 + wave * MT;
    const int cbase = ntile * NT * 16 + 4 * lk;
    const int act = a.act;
    if (gx >= W) return;
    const int ps = a.ps;
    const int orow = W * ps;                                   // destination pixels per row
    if constexpr (KS == 5) {
        if (a.fold) {
            // Folded linear tail: conv channel n*16 + 4*lk + r is (phase n*4 + lk, border variant r).
            // Variant bit 1 = this phase's out-of-image HR row tap is dropped (phase row 0 at the top image
            // row, phase row ps-1 at the bottom one), bit 0 the same for columns.
            float* yout = a.out0.ptr;
            static_for<0, NT>([&](auto n_) DCSCN_INL {
                constexpr int n = decltype(n_)::value;
                const int phase = n * 4 + lk;
                if (phase < ps * ps) {
                    const int pa = phase / ps, pb = phase - pa * ps;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + 4 * lk);
                    const bool cb = (pb == 0 && gx == 0) || (pb == ps - 1 && gx == W - 1);
                    static_for<0, MT>([&](auto m_) DCSCN_INL {
                        constexpr int m = decltype(m_)::value;
                        const int gy = gy0 + m;
                        if (gy < H) {
                            const bool rb = (pa == 0 && gy == 0) || (pa == ps - 1 && gy == H - 1);
                            const f32x4 v = acc[m][n] + bv;
                            const float lo = cb ? v.y : v.x, hi = cb ? v.w : v.z;
                            const size_t idx = ((size_t)(img * H + gy) * ps + pa) * orow + (size_t)(gx * ps + pb);
                            float out = rb ? hi : lo;
                            if (a.res) out += a.res[idx];
                            yout[idx] = out;
                        }
                    });
                }
            });
            return;
        }
    }
    auto finish = [&](auto act_c, auto vec_c) DCSCN_INL {
        constexpr int ACT_C = decltype(act_c)::value;
        constexpr bool VEC = decltype(vec_c)::value;
        const int act_e = ACT_C >= 0 ? ACT_C : act;
        f32x4 bv_next = *reinterpret_cast<const f32x4*>(a.bias + cbase);
        f32x4 av_next = {0.0f, 0.0f, 0.0f, 0.0f};
        if (act_e == ACT_ALPHA) av_next = *reinterpret_cast<const f32x4*>(a.alpha + cbase);
        static_for<0, NT>([&](auto n_) DCSCN_INL {
            constexpr int n = decltype(n_)::value;
            const int c = cbase + n * 16;
            const f32x4 bv = bv_next, av = av_next;
            if constexpr (n + 1 < NT) {                        // next tile's bias / slope while this one is stored
                bv_next = *reinterpret_cast<const f32x4*>(a.bias + c + 16);
                if (act_e == ACT_ALPHA) av_next = *reinterpret_cast<const f32x4*>(a.alpha + c + 16);
            }
            const bool first = c < a.split;
            float* optr = first ? a.out0.ptr : a.out1.ptr;
            const int ostride = first ? a.out0.stride : a.out1.stride;
            const int ooff = first ? a.out0.off : a.out1.off;
            const int owidth = first ? a.out0.width : a.out1.width;
            const int cc = first ? c : c - a.split;
            const size_t dy = (size_t)ps * orow * ostride;     // one LR row down in the destination
            if constexpr (VEC) {
                // the 4 channels of this lane stay together: pixel (gy*ps + ay, gx*ps + bx), channels ch..ch+3
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
                static_for<0, MT>([&](auto m_) DCSCN_INL {
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
                // scalar stores: ragged widths / offsets, or depth_to_space splitting the 4 channels
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
                static_for<0, MT>([&](auto m_) DCSCN_INL {
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


}  // namespace dcscn
