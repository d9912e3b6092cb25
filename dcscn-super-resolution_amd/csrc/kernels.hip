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
#include "p16.hpp"

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
    if (e == hipSuccess) e = c3e_init_kernels();
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
    if (a.redo_check && (a.redo[0] == 0 || a.redo[1 + img] == 0)) return;   // float32 plan behind a split16 pass: flagged images only

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
    const bool o16 = a.out.p16.base != nullptr;   // P16 destination (p16.hpp)
    float chk = 0.0f;
    if (o16) {
        // a thread owns a channel OCTET of a pixel: its (hi | lo) units are one 32-byte store, four adjacent lanes fill a 128-byte record.
        // tpp16 threads per pixel (a power of two >= octets), the rest of the geometry as below
        const int octs = a.out.p16.octs;
        int t16 = 0;
        while ((1 << t16) < octs && t16 < 6) ++t16;
        const int tpp16 = 1 << t16, cl8 = tid & (tpp16 - 1), pl8 = tid >> t16, ppi8 = 256 >> t16;
        const float m1 = opaque_minus_one();
        const h2 zero2 = p16_opaque_zero2();
        for (int c8 = cl8; c8 < octs; c8 += tpp16) {
            const bool two = 2 * c8 + 1 < c4n;         // the last octet of an odd quad count: its upper half is written as zeros
            f32x4 wr[TAPS][2];
            const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                wr[t][0] = *reinterpret_cast<const f32x4*>(ws + t * cs + 8 * c8);
                wr[t][1] = two ? *reinterpret_cast<const f32x4*>(ws + t * cs + 8 * c8 + 4) : z4;
            }
            const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bs + 8 * c8), bv1 = two ? *reinterpret_cast<const f32x4*>(bs + 8 * c8 + 4) : z4;
            const f32x4 av0 = *reinterpret_cast<const f32x4*>(as + 8 * c8), av1 = two ? *reinterpret_cast<const f32x4*>(as + 8 * c8 + 4) : z4;
            const int chunk = c8 >> 2, rem = octs - 4 * chunk;
            const int rec = rem >= 4 ? 128 : 32 * rem;
            char* plane = a.out.p16.base + (long long)chunk * a.out.p16.plane + 128 + (c8 & 3) * 32;
            for (int p = pl8; p < T * T; p += ppi8) {
                const int py = p >> 4, px = p & 15;
                const int gy = y0 + py, gx = x0 + px;
                if (gy >= a.H || gx >= a.W) continue;
                f32x4 s0 = z4, s1 = z4;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float xv = xs[(py + t / KS) * HT + px + t % KS];
                    s0 += wr[t][0] * xv;
                    s1 += wr[t][1] * xv;
                }
                f32x4 v0 = bv0, v1 = bv1;
                v0 += s0; v1 += s1;                    // (the float32 path's association: bias + (sum over taps))
                v0.x = activate1(v0.x, av0.x, a.act); v0.y = activate1(v0.y, av0.y, a.act); v0.z = activate1(v0.z, av0.z, a.act); v0.w = activate1(v0.w, av0.w, a.act);
                if (two) { v1.x = activate1(v1.x, av1.x, a.act); v1.y = activate1(v1.y, av1.y, a.act); v1.z = activate1(v1.z, av1.z, a.act); v1.w = activate1(v1.w, av1.w, a.act); }
                else v1 = z4;
                h8 hi, lo;
                split8(v0, v1, m1, hi, lo);
                const u32x4 hu = __builtin_bit_cast(u32x4, hi);
                chk = p16_check(p16_check(p16_check(p16_check(chk, hu.x, zero2), hu.y, zero2), hu.z, zero2), hu.w, zero2);
                const size_t pix = ((size_t)img * a.H + gy) * a.W + gx;
                *reinterpret_cast<h8*>(plane + pix * rec) = hi;
                *reinterpret_cast<h8*>(plane + pix * rec + 16) = lo;
            }
        }
    } else
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
    if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }     // an output beyond the f16 range: the image goes to the float32 plan
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
    if (a.redo_check && (a.redo[0] == 0 || a.redo[1 + img] == 0)) return;   // float32 plan behind a split16 pass: flagged images only

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
    if (a.redo_check && (a.redo[0] == 0 || a.redo[1 + pix / ((long long)a.H * a.W)] == 0)) return;   // float32 plan: flagged images only
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

// ---------------------------------------------------------------------------------------------
// debug: poison what a kernel must not depend on (LDS and register contents left by whoever ran before)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pass_begin_kernel(int32_t* redo, int n, const unsigned long long* zrec, int nz) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) redo[i] = 0;
    if (i < nz * 8) reinterpret_cast<u32x4*>((uintptr_t)zrec[i >> 3])[i & 7] = u32x4{0u, 0u, 0u, 0u};
}
hipError_t pass_begin_launch(int32_t* redo, int n, const unsigned long long* zrec, int nz, hipStream_t stream) {
    const int work = n > nz * 8 ? n : nz * 8;
    hipLaunchKernelGGL(pass_begin_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, redo, n, zrec, zrec ? nz : 0);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void poison_lds(int bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < bytes / 4; i += 256) smem[i] = __uint_as_float(0x7fc0dea0u + (i & 15));
    __syncthreads();
    __builtin_amdgcn_s_sleep(64);                          // stay resident long enough for the grid to cover every CU
}
__global__ __launch_bounds__(256, 2) void poison_vgprs(float* sink) {
    unsigned p = 0x7fc0beefu;
    asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\tv_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\tv_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\tv_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\tv_mov_b32 v67, %0\n\tv_mov_b32 v68, %0\n\tv_mov_b32 v69, %0\n\tv_mov_b32 v70, %0\n\tv_mov_b32 v71, %0\n\tv_mov_b32 v72, %0\n\tv_mov_b32 v73, %0\n\tv_mov_b32 v74, %0\n\tv_mov_b32 v75, %0\n\tv_mov_b32 v76, %0\n\tv_mov_b32 v77, %0\n\tv_mov_b32 v78, %0\n\tv_mov_b32 v79, %0\n\tv_mov_b32 v80, %0\n\tv_mov_b32 v81, %0\n\tv_mov_b32 v82, %0\n\tv_mov_b32 v83, %0\n\tv_mov_b32 v84, %0\n\tv_mov_b32 v85, %0\n\tv_mov_b32 v86, %0\n\tv_mov_b32 v87, %0\n\tv_mov_b32 v88, %0\n\tv_mov_b32 v89, %0\n\tv_mov_b32 v90, %0\n\tv_mov_b32 v91, %0\n\tv_mov_b32 v92, %0\n\tv_mov_b32 v93, %0\n\tv_mov_b32 v94, %0\n\tv_mov_b32 v95, %0\n\tv_mov_b32 v96, %0\n\tv_mov_b32 v97, %0\n\tv_mov_b32 v98, %0\n\tv_mov_b32 v99, %0\n\tv_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, %0\n\tv_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\tv_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\tv_mov_b32 v108, %0\n\tv_mov_b32 v109, %0\n\tv_mov_b32 v110, %0\n\tv_mov_b32 v111, %0\n\tv_mov_b32 v112, %0\n\tv_mov_b32 v113, %0\n\tv_mov_b32 v114, %0\n\tv_mov_b32 v115, %0\n\tv_mov_b32 v116, %0\n\tv_mov_b32 v117, %0\n\tv_mov_b32 v118, %0\n\tv_mov_b32 v119, %0\n\tv_mov_b32 v120, %0\n\tv_mov_b32 v121, %0\n\tv_mov_b32 v122, %0\n\tv_mov_b32 v123, %0\n\tv_mov_b32 v124, %0\n\tv_mov_b32 v125, %0\n\tv_mov_b32 v126, %0\n\tv_mov_b32 v127, %0\n\tv_mov_b32 v128, %0\n\tv_mov_b32 v129, %0\n\tv_mov_b32 v130, %0\n\tv_mov_b32 v131, %0\n\tv_mov_b32 v132, %0\n\tv_mov_b32 v133, %0\n\tv_mov_b32 v134, %0\n\tv_mov_b32 v135, %0\n\tv_mov_b32 v136, %0\n\tv_mov_b32 v137, %0\n\tv_mov_b32 v138, %0\n\tv_mov_b32 v139, %0\n\tv_mov_b32 v140, %0\n\tv_mov_b32 v141, %0\n\tv_mov_b32 v142, %0\n\tv_mov_b32 v143, %0\n\tv_mov_b32 v144, %0\n\tv_mov_b32 v145, %0\n\tv_mov_b32 v146, %0\n\tv_mov_b32 v147, %0\n\tv_mov_b32 v148, %0\n\tv_mov_b32 v149, %0\n\tv_mov_b32 v150, %0\n\tv_mov_b32 v151, %0\n\tv_mov_b32 v152, %0\n\tv_mov_b32 v153, %0\n\tv_mov_b32 v154, %0\n\tv_mov_b32 v155, %0\n\tv_mov_b32 v156, %0\n\tv_mov_b32 v157, %0\n\tv_mov_b32 v158, %0\n\tv_mov_b32 v159, %0\n\tv_mov_b32 v160, %0\n\tv_mov_b32 v161, %0\n\tv_mov_b32 v162, %0\n\tv_mov_b32 v163, %0\n\tv_mov_b32 v164, %0\n\tv_mov_b32 v165, %0\n\tv_mov_b32 v166, %0\n\tv_mov_b32 v167, %0\n\tv_mov_b32 v168, %0\n\tv_mov_b32 v169, %0\n\tv_mov_b32 v170, %0\n\tv_mov_b32 v171, %0\n\tv_mov_b32 v172, %0\n\tv_mov_b32 v173, %0\n\tv_mov_b32 v174, %0\n\tv_mov_b32 v175, %0\n\tv_mov_b32 v176, %0\n\tv_mov_b32 v177, %0\n\tv_mov_b32 v178, %0\n\tv_mov_b32 v179, %0\n\tv_mov_b32 v180, %0\n\tv_mov_b32 v181, %0\n\tv_mov_b32 v182, %0\n\tv_mov_b32 v183, %0\n\tv_mov_b32 v184, %0\n\tv_mov_b32 v185, %0\n\tv_mov_b32 v186, %0\n\tv_mov_b32 v187, %0\n\tv_mov_b32 v188, %0\n\tv_mov_b32 v189, %0\n\tv_mov_b32 v190, %0\n\tv_mov_b32 v191, %0\n\tv_mov_b32 v192, %0\n\tv_mov_b32 v193, %0\n\tv_mov_b32 v194, %0\n\tv_mov_b32 v195, %0\n\tv_mov_b32 v196, %0\n\tv_mov_b32 v197, %0\n\tv_mov_b32 v198, %0\n\tv_mov_b32 v199, %0\n\tv_mov_b32 v200, %0\n\tv_mov_b32 v201, %0\n\tv_mov_b32 v202, %0\n\tv_mov_b32 v203, %0\n\tv_mov_b32 v204, %0\n\tv_mov_b32 v205, %0\n\tv_mov_b32 v206, %0\n\tv_mov_b32 v207, %0\n\tv_mov_b32 v208, %0\n\tv_mov_b32 v209, %0\n\tv_mov_b32 v210, %0\n\tv_mov_b32 v211, %0\n\tv_mov_b32 v212, %0\n\tv_mov_b32 v213, %0\n\tv_mov_b32 v214, %0\n\tv_mov_b32 v215, %0\n\tv_mov_b32 v216, %0\n\tv_mov_b32 v217, %0\n\tv_mov_b32 v218, %0\n\tv_mov_b32 v219, %0\n\tv_mov_b32 v220, %0\n\tv_mov_b32 v221, %0\n\tv_mov_b32 v222, %0\n\tv_mov_b32 v223, %0\n\tv_mov_b32 v224, %0\n\tv_mov_b32 v225, %0\n\tv_mov_b32 v226, %0\n\tv_mov_b32 v227, %0\n\tv_mov_b32 v228, %0\n\tv_mov_b32 v229, %0\n\tv_mov_b32 v230, %0\n\tv_mov_b32 v231, %0\n\tv_mov_b32 v232, %0\n\tv_mov_b32 v233, %0\n\tv_mov_b32 v234, %0\n\tv_mov_b32 v235, %0\n\tv_mov_b32 v236, %0\n\tv_mov_b32 v237, %0\n\tv_mov_b32 v238, %0\n\tv_mov_b32 v239, %0\n\tv_mov_b32 v240, %0\n\tv_mov_b32 v241, %0\n\tv_mov_b32 v242, %0\n\tv_mov_b32 v243, %0\n\tv_mov_b32 v244, %0\n\tv_mov_b32 v245, %0\n\tv_mov_b32 v246, %0\n\tv_mov_b32 v247, %0\n\tv_mov_b32 v248, %0\n\tv_mov_b32 v249, %0\n\tv_mov_b32 v250, %0\n\tv_mov_b32 v251, %0\n\tv_mov_b32 v252, %0\n\tv_mov_b32 v253, %0\n\tv_mov_b32 v254, %0\n\tv_mov_b32 v255, %0" ::"s"(p) : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    if (sink && p == 1u) sink[0] = 0.0f;
    __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(256) void digest_words(const unsigned* p, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += (unsigned long long)p[i] * ((i * 0x9E3779B97F4A7C15ull) | 1ull);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
hipError_t debug_digest_launch(const void* p, size_t n_words, unsigned long long* out, hipStream_t stream) {
    hipLaunchKernelGGL(digest_words, dim3(2048), dim3(256), 0, stream, static_cast<const unsigned*>(p), n_words, out);
    return hipGetLastError();
}
hipError_t debug_poison_launch(int what, hipStream_t stream) {
    static bool attr = false;
    const int bytes = 160 * 1024;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (what & 1) hipLaunchKernelGGL(poison_lds, dim3(1024), dim3(256), bytes, stream, bytes);
    if (what & 2) hipLaunchKernelGGL(poison_vgprs, dim3(4096), dim3(256), 0, stream, (float*)nullptr);
    return hipGetLastError();
}

}  // namespace dcscn
