// gfx950 (MI355X / CDNA4) kernels of the DCSCN forward pass.
//
// conv_igemm   -- every dense conv of the graph (tf.nn.conv2d SAME stride 1, helper/tf_graph.py:105)
//                 as an implicit GEMM on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), with
//                 bias + activator (tf_graph.py:89-94,109), the skip-concat (DCSCN.py:259: the store
//                 goes straight into the layer's channel slice of the concat buffer), depth_to_space
//                 (tf_graph.py:248) and the final residual add (DCSCN.py:325) fused into the epilogue.
// conv_cin1    -- the first feature layer (1 input channel): direct form, write bound.
// depthwise    -- depthwise half of tf.nn.separable_conv2d (tf_graph.py:161).
//
// Layout: activations are NHWC float32; every tensor slice starts on a 4-channel boundary and is
// padded to a multiple of 4 channels (padding lanes hold finite values and meet zero weights), so all
// global traffic is 16-byte vectors.
#include "kernels.h"
#include <type_traits>

namespace dcscn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// layer) and the KS*KS*KC*NT*16 filter block are staged in LDS, double buffered: global loads for
// chunk c+1 are issued before the MFMAs of chunk c and written to the other LDS buffer after them.
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

template <int KS, int MT, int NT, int KC>
__global__ __launch_bounds__(256, 2) void conv_igemm(const ConvArgs a) {
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

    auto compute = [&](const float* buf) DCSCN_INL {
        const float* As = buf + a_lane;
        const float* Bs = buf + b_lane;
        static_for<0, G::TAPS>([&](auto tap_) DCSCN_INL {
            constexpr int tap = decltype(tap_)::value;
            constexpr int dy = tap / KS, dx = tap % KS;
            static_for<0, G::KQ>([&](auto ks_) DCSCN_INL {
                constexpr int ks = decltype(ks_)::value;
                float xv[MT], wv[NT];
                static_for<0, MT>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    xv[m] = As[(ks * 4) * G::PS + (m + dy) * G::HTW + dx];
                });
                static_for<0, NT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    wv[n] = Bs[(tap * KC + ks * 4) * G::NS + n * 16];
                });
                static_for<0, MT>([&](auto m_) DCSCN_INL {
                    constexpr int m = decltype(m_)::value;
                    static_for<0, NT>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n], xv[m], acc[m][n], 0, 0, 0);
                    });
                });
            });
        });
    };

    // ---- K loop, LDS double buffered ----
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

    // ---- epilogue: bias, activator, (depth_to_space), (residual), store ----
    const int gx = x0 + lj;
    const int gy0 = y0 + wave * MT;
    const int cbase = ntile * NT * 16;
    const int act = a.act;
    if (gx >= W) return;
    static_for<0, NT>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        const int c = cbase + n * 16 + 4 * lk;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + c);
        f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
        if (act == ACT_ALPHA) av = *reinterpret_cast<const f32x4*>(a.alpha + c);
        const bool first = c < a.split;
        float* optr = first ? a.out0.ptr : a.out1.ptr;
        const int ostride = first ? a.out0.stride : a.out1.stride;
        const int ooff = first ? a.out0.off : a.out1.off;
        const int owidth = first ? a.out0.width : a.out1.width;
        const int cc = first ? c : c - a.split;
        // destination of channel cc+r: pixel (gy*ps + ay[r], gx*ps + bx[r]), channel ch[r]
        int ch[4], ay[4], bx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ccr = cc + r;
            if (a.ps == 1) {
                ch[r] = ccr; ay[r] = 0; bx[r] = 0;
            } else {
                const int sub = ccr / a.ps_c;
                ch[r] = ccr - sub * a.ps_c;
                ay[r] = sub / a.ps;
                bx[r] = sub - ay[r] * a.ps;
            }
        }
        const size_t orow = (size_t)W * a.ps;
        static_for<0, MT>([&](auto m_) DCSCN_INL {
            constexpr int m = decltype(m_)::value;
            const int gy = gy0 + m;
            if (gy < H) {
                f32x4 v = acc[m][n] + bv;
                v.x = activate1(v.x, av.x, act);
                v.y = activate1(v.y, av.y, act);
                v.z = activate1(v.z, av.z, act);
                v.w = activate1(v.w, av.w, act);
                const size_t prow = ((size_t)img * H + gy) * a.ps;
                if (a.vec4) {
                    if (cc < owidth) {
                        const size_t pix = (prow + ay[0]) * orow + (size_t)gx * a.ps + bx[0];
                        if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + pix * a.res_stride + ch[0]);
                        *reinterpret_cast<f32x4*>(optr + pix * ostride + ooff + ch[0]) = v;
                    }
                } else {
                    const float vr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (cc + r < owidth) {
                            const size_t pix = (prow + ay[r]) * orow + (size_t)gx * a.ps + bx[r];
                            float out = vr[r];
                            if (a.res) out += a.res[pix * a.res_stride + ch[r]];
                            optr[pix * ostride + ooff + ch[r]] = out;
                        }
                    }
                }
            }
        });
    });
}

// ---- variant table ---------------------------------------------------------------------------
// (mt, kc) as a function of (ks, nt): wide channel tiles keep the pixel tile small (8x16) so two
// workgroups fit a CU's LDS; narrow ones take a 16x16 pixel tile to amortise the filter reads.
__host__ __device__ constexpr int pick_mt(int ks, int nt) { return nt >= 8 ? 2 : 4; }
__host__ __device__ constexpr int pick_kc(int ks, int nt) { return ks == 1 ? 16 : (nt >= 5 ? 4 : 8); }

ConvShape conv_pick_shape(int ks, int nt) { return ConvShape{ks, pick_mt(ks, nt), nt, pick_kc(ks, nt)}; }

static size_t lds_bytes_for(int ks, int mt, int nt, int kc) {
    const int halo = ks / 2;
    const int hp = (4 * mt + 2 * halo) * (16 + 2 * halo);
    const int ps = conv_plane_stride(hp);
    const int ns = conv_ns(nt);
    return 2 * (size_t)(kc * ps + ks * ks * kc * ns) * sizeof(float);
}
size_t conv_lds_bytes(const ConvShape& s) { return lds_bytes_for(s.ks, s.mt, s.nt, s.kc); }

#define DCSCN_FOR_NT(X, KS) \
    X(KS, 1) X(KS, 2) X(KS, 3) X(KS, 4) X(KS, 5) X(KS, 6) X(KS, 7) X(KS, 8) X(KS, 9) X(KS, 10) X(KS, 11) X(KS, 12) X(KS, 13)

template <int KS, int NT>
static hipError_t conv_set_attr() {
    constexpr int MT = pick_mt(KS, NT), KC = pick_kc(KS, NT);
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm<KS, MT, NT, KC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_for(KS, MT, NT, KC));
}

hipError_t conv_init_kernels() {
    hipError_t e;
#define X(KS, NT) if ((e = conv_set_attr<KS, NT>()) != hipSuccess) return e;
    DCSCN_FOR_NT(X, 1)
    DCSCN_FOR_NT(X, 3)
#undef X
    return hipSuccess;
}

template <int KS, int NT>
static hipError_t conv_launch_one(const ConvArgs& a, int n_tiles, hipStream_t stream) {
    constexpr int MT = pick_mt(KS, NT), KC = pick_kc(KS, NT);
    const dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)n_tiles);
    hipLaunchKernelGGL((conv_igemm<KS, MT, NT, KC>), grid, dim3(256), lds_bytes_for(KS, MT, NT, KC), stream, a);
    return hipGetLastError();
}

hipError_t conv_launch(const ConvShape& s, const ConvArgs& a, int n_tiles, hipStream_t stream) {
    if (s.mt != pick_mt(s.ks, s.nt) || s.kc != pick_kc(s.ks, s.nt)) return hipErrorInvalidValue;
    switch (s.ks * 100 + s.nt) {
#define X(KS, NT) case KS * 100 + NT: return conv_launch_one<KS, NT>(a, n_tiles, stream);
        DCSCN_FOR_NT(X, 1)
        DCSCN_FOR_NT(X, 3)
#undef X
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// first layer: conv from one input channel, direct form
// ---------------------------------------------------------------------------------------------
// Tile 16x16 pixels per workgroup.  LDS: the (16+2h)^2 input halo tile, the filters [taps][cs] and
// bias/alpha.  Threads are split (pixel group) x (channel quad): `tpp` = threads per pixel, a power
// of two, so that a wave's float4 stores cover runs of consecutive channels of consecutive pixels.
template <int KS>
__global__ __launch_bounds__(256) void conv_cin1(const Cin1Args a, int tpp_log2) {
    constexpr int TAPS = KS * KS, HALO = KS / 2, T = 16, HT = T + 2 * HALO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cs = a.cs, c4n = cs >> 2;
    float* xs = smem;                       // [HT*HT] (+pad to 4)
    float* ws = smem + ((HT * HT + 3) & ~3);   // [TAPS][cs]
    float* bs = ws + TAPS * cs;             // [cs]
    float* as = bs + cs;                    // [cs]

    int bid = blockIdx.x;
    const int tiles_x = (a.W + T - 1) / T, tiles_y = (a.H + T - 1) / T;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * T, x0 = tx * T;
    const int tid = threadIdx.x;

    const float* xin = a.x + (size_t)img * a.H * a.W;
    for (int i = tid; i < HT * HT; i += 256) {
        const int hy = i / HT, hx = i - hy * HT;
        const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;
        xs[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? xin[(size_t)gy * a.W + gx] : 0.0f;
    }
    for (int i = tid; i < TAPS * cs; i += 256) ws[i] = a.w[i];
    for (int i = tid; i < cs; i += 256) {
        bs[i] = a.bias[i];
        as[i] = a.alpha[i];
    }
    __syncthreads();

    const int tpp = 1 << tpp_log2;
    const int cl = tid & (tpp - 1);
    const int pl = tid >> tpp_log2;
    const int ppi = 256 >> tpp_log2;   // pixels per iteration
    for (int c4 = cl; c4 < c4n; c4 += tpp) {
        f32x4 wr[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) wr[t] = *reinterpret_cast<const f32x4*>(ws + t * cs + 4 * c4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + 4 * c4);
        const f32x4 av = *reinterpret_cast<const f32x4*>(as + 4 * c4);
        for (int p = pl; p < T * T; p += ppi) {
            const int py = p >> 4, px = p & 15;
            const int gy = y0 + py, gx = x0 + px;
            if (gy >= a.H || gx >= a.W) continue;
            f32x4 v = bv;   // accumulation order differs from TF's; covered by the f32 tolerance
            f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const float xv = xs[(py + t / KS) * HT + px + t % KS];
                s += wr[t] * xv;
            }
            v += s;
            v.x = activate1(v.x, av.x, a.act);
            v.y = activate1(v.y, av.y, a.act);
            v.z = activate1(v.z, av.z, a.act);
            v.w = activate1(v.w, av.w, a.act);
            const size_t pix = ((size_t)img * a.H + gy) * a.W + gx;
            *reinterpret_cast<f32x4*>(a.out.ptr + pix * a.out.stride + a.out.off + 4 * c4) = v;
        }
    }
}

hipError_t cin1_launch(const Cin1Args& a, hipStream_t stream) {
    const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    const int c4n = a.cs / 4;
    int tpp_log2 = 0;
    while ((1 << tpp_log2) < c4n && tpp_log2 < 6) ++tpp_log2;
    const int halo = a.ks / 2;
    const int ht = 16 + 2 * halo;
    const size_t lds = (size_t)(((ht * ht + 3) & ~3) + (a.ks * a.ks + 2) * a.cs) * sizeof(float);
    const dim3 grid((unsigned)(a.N * tiles));
    if (a.ks == 3) hipLaunchKernelGGL(conv_cin1<3>, grid, dim3(256), lds, stream, a, tpp_log2);
    else if (a.ks == 1) hipLaunchKernelGGL(conv_cin1<1>, grid, dim3(256), lds, stream, a, tpp_log2);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// depthwise conv (channel multiplier 1), SAME, stride 1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depthwise_kernel(const DwArgs a, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % a.cout_phys);
    const long long pix = idx / a.cout_phys;
    float out = 0.0f;
    if (c < a.cin) {
        const int x = (int)(pix % a.W);
        const long long t = pix / a.W;
        const int y = (int)(t % a.H);
        const long long img = t / a.H;
        const int halo = a.ks / 2;
        const int pc = a.chan_map[c];
        const float* base = a.in + (size_t)img * a.H * a.W * a.in_stride + a.in_off + pc;
        for (int dy = 0; dy < a.ks; ++dy) {
            const int gy = y + dy - halo;
            if (gy < 0 || gy >= a.H) continue;
            for (int dx = 0; dx < a.ks; ++dx) {
                const int gx = x + dx - halo;
                if (gx < 0 || gx >= a.W) continue;
                out += base[((size_t)gy * a.W + gx) * a.in_stride] * a.w[(dy * a.ks + dx) * a.cin + c];
            }
        }
    }
    a.out[(size_t)pix * a.out_stride + c] = out;
}

hipError_t depthwise_launch(const DwArgs& a, hipStream_t stream) {
    const long long total = (long long)a.N * a.H * a.W * a.cout_phys;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(depthwise_kernel, dim3(blocks), dim3(256), 0, stream, a, total);
    return hipGetLastError();
}

}  // namespace dcscn
