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
#include "conv_variants.hpp"

namespace dcscn {

// ---- dispatch over the per-family translation units (conv_k1.hip, conv_k3.hip, ..., conv_wino2.hip) --------
ConvShape conv_pick_shape(int ks, int nt, int dwk) { return ConvShape{ks, pick_mt(ks, nt), nt, pick_kc(ks, nt), dwk}; }
size_t conv_lds_bytes(const ConvShape& s) { return lds_bytes_for(s.ks, s.mt, s.nt, s.kc, s.dwk); }
int conv_max_fused_dw_nt() { return kMaxDwNt; }
int conv_max_nt(int ks) { return ks == 7 ? kMaxK7Nt : 13; }

hipError_t conv_init_kernels() {
    hipError_t e = conv_init_k1();
    if (e == hipSuccess) e = conv_init_k3();
    if (e == hipSuccess) e = conv_init_k5();
    if (e == hipSuccess) e = conv_init_k7();
    if (e == hipSuccess) e = wino_init_kernels();
    if (e == hipSuccess) e = nin_init_kernels();
    if (e == hipSuccess) e = nin_h_init_kernels();
    if (e == hipSuccess) e = c3h_init_kernels();
    if (e == hipSuccess) e = c5h_init_kernels();
    if (e == hipSuccess) stream_init_kernels();
    return e;
}

hipError_t conv_launch(const ConvShape& s, const ConvArgs& a, int n_tiles, hipStream_t stream) {
    if (s.mt != pick_mt(s.ks, s.nt) || s.kc != pick_kc(s.ks, s.nt)) return hipErrorInvalidValue;
    if (s.dwk != 0 && (s.ks != 1 || a.dww == nullptr || a.dwk != s.dwk)) return hipErrorInvalidValue;
    if (s.ks == 1) return conv_launch_k1(s.nt, s.dwk, a, n_tiles, stream);
    if (s.ks == 3) return conv_launch_k3(s.nt, a, n_tiles, stream);
    if (s.ks == 5) return conv_launch_k5(s.nt, a, n_tiles, stream);
    if (s.ks == 7 && s.nt <= kMaxK7Nt) return conv_launch_k7(s.nt, a, n_tiles, stream);
    return hipErrorInvalidValue;
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
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (a.ks == 3) hipLaunchKernelGGL(conv_cin1<3>, grid, dim3(256), lds, stream, a, tpp_log2);
    else if (a.ks == 1) hipLaunchKernelGGL(conv_cin1<1>, grid, dim3(256), lds, stream, a, tpp_log2);
    else if (a.ks == 5) hipLaunchKernelGGL(conv_cin1<5>, grid, dim3(256), lds, stream, a, tpp_log2);
    else if (a.ks == 7) hipLaunchKernelGGL(conv_cin1<7>, grid, dim3(256), lds, stream, a, tpp_log2);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// last reconstruction conv: C input channels -> 1 output channel (+ residual)
// ---------------------------------------------------------------------------------------------
// out(q) = sum_t sum_c in(q + o_t)[c] w[t][c] is evaluated as sum_t s_t(q + o_t) with the per-pixel
// partial sums s_t(p) = <in(p), w[t]>: every input pixel of the 18x18 halo tile is read ONCE (8 lanes
// share a pixel and walk its channels in 128-byte runs), its taps' partial sums go to LDS, and each
// thread then gathers the 9 partial sums of its output pixel.  Pixels outside the image contribute 0
// (SAME zero padding).
template <int KS, int LPP>
__global__ __launch_bounds__(256) void conv_cout1(const Cout1Args a) {
    constexpr int TAPS = KS * KS, HALO = KS / 2, T = 16, HT = T + 2 * HALO, HP = HT * HT;
    constexpr int SP = (HP + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ws = smem;                         // [TAPS][cin_phys]
    float* sp = smem + TAPS * a.cin_phys;     // [TAPS][SP]

    int bid = blockIdx.x;
    const int tiles_x = (a.W + T - 1) / T, tiles_y = (a.H + T - 1) / T;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int img = bid / tiles_y;
    const int y0 = ty * T, x0 = tx * T;
    const int tid = threadIdx.x;

    for (int i = tid; i < TAPS * a.cin_phys; i += 256) ws[i] = a.w[i];
    __syncthreads();

    const int sub = tid % LPP;        // lane within the LPP-lane group that shares a pixel
    const int grp = tid / LPP;
    const int nq = a.cin_phys >> 2;
    const float* in_img = a.in + (size_t)img * a.H * a.W * a.in_stride + a.in_off;
    for (int hp = grp; hp < HP; hp += 256 / LPP) {
        const int hy = hp / HT, hx = hp - hy * HT;
        const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;
        float acc[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[t] = 0.0f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const float* px = in_img + ((size_t)gy * a.W + gx) * a.in_stride;
            for (int q = sub; q < nq; q += LPP) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * q);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(ws + t * a.cin_phys + 4 * q);
                    acc[t] += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            float s = acc[t];
            if constexpr (LPP >= 2) s += __shfl_xor(s, 1);
            if constexpr (LPP >= 4) s += __shfl_xor(s, 2);
            if constexpr (LPP >= 8) s += __shfl_xor(s, 4);
            if (sub == 0) sp[t * SP + hp] = s;
        }
    }
    __syncthreads();

    const int py = tid >> 4, pxl = tid & 15;
    const int gy = y0 + py, gx = x0 + pxl;
    if (gy < a.H && gx < a.W) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) s += sp[t * SP + (py + t / KS) * HT + pxl + t % KS];
        s = s * a.scale + a.bias;
        const size_t pix = ((size_t)img * a.H + gy) * a.W + gx;
        if (a.res) s += a.res[pix * a.res_stride];
        a.out[pix * a.out_stride] = s;
    }
}

hipError_t cout1_launch(const Cout1Args& a, hipStream_t stream) {
    const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    const int halo = a.ks / 2;
    const int hp = (16 + 2 * halo) * (16 + 2 * halo);
    const size_t lds = (size_t)(a.ks * a.ks * a.cin_phys + a.ks * a.ks * ((hp + 3) & ~3)) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.N * tiles));
    const bool narrow = a.cin_phys <= 8;      // one lane per pixel when a pixel is at most two float4
    if (a.ks == 3) {
        if (narrow) hipLaunchKernelGGL((conv_cout1<3, 1>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((conv_cout1<3, 8>), grid, dim3(256), lds, stream, a);
    } else if (a.ks == 1) {
        if (narrow) hipLaunchKernelGGL((conv_cout1<1, 1>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((conv_cout1<1, 8>), grid, dim3(256), lds, stream, a);
    } else if (a.ks == 5) {
        if (narrow) hipLaunchKernelGGL((conv_cout1<5, 1>), grid, dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((conv_cout1<5, 8>), grid, dim3(256), lds, stream, a);
    } else {
        return hipErrorInvalidValue;
    }
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
