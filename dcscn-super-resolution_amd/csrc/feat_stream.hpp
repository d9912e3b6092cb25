// Row-streamed feature extractor of the separable (depthwise_separable=True) narrow nets: CNN1 .. CNNL, A1 || B1 and B2
// (DCSCN.py:240-291 with tf_graph.py:155-177 build_depthwise_separable_conv) in ONE launch, intermediate tensors in LDS.
//
// Why: layer by layer these nets move 2-3.6 KB per LR pixel through HBM for 36-132 B of compulsory I/O (input Y, the
// 32-channel Concat2).  Here a workgroup owns a 48-pixel-wide column strip of an image and walks down its rows; every
// layer keeps only a THREE-ROW ring of its output in LDS, which is all the next 3x3 layer needs:
//
//   wave 0            CNN1: depthwise 3x3 on Y (global, prefetched one step ahead) -> pointwise 1 -> C1, VALU only
//   waves 1 .. L-1    CNN2 .. CNNL: depthwise 3x3 (VALU, from the predecessor's ring) -> pointwise GEMM on the MFMA
//                     pipe (filters re-read from L1 / L2 every row: 4 KB) -> bias, PReLU -> own ring
//   wave L            B2: the same, from the B1 ring to global Concat2[0 : nb)
//   waves L+1 .. 2L   A1 || B1: H_concat never exists.  The 1x1 layers are linear in the concat, so the contribution of
//                     feature layer l to row g is accumulated as soon as that row is in layer l's ring; the L rows in
//                     flight are spread over L waves (row g -> wave (g/2) mod L, two rows per wave in registers), each of
//                     which therefore does exactly one (layer, row) contribution per step.
//
// One "step" = every role advances one row: compute phase (reads rings, results stay in registers), barrier, write phase
// (ring slot of the row that just left the read window), barrier.  Layer l runs 2 rows behind layer l-1, so the pipeline
// is 2L+1 steps deep; a workgroup therefore streams its jobs back to back, one zero row between them, which is also the
// SAME padding of both neighbours.  Columns left / right of the image and rows of the separator are forced to zero in
// every ring (SAME padding at every layer); strips of images wider than 48 and row blocks of tall images overlap by the
// receptive field (L+1) and only their interior is stored.
//
// MFMA operand layout (v_mfma_f32_16x16x4_f32, A = filters, B = 16 pixels): lane (j = lane & 15, q = lane >> 4) reads ONE
// ds_read_b128 = channels 4q' .. 4q'+3 (q' = 4 * chunk + q) of pixel j and uses its four floats as the B operand of four
// k-steps; k-step s of a chunk therefore covers channels {16 chunk + 4q + s}, and the filters are packed to match.
// Ring rows are [pixel -1 .. 48][units] float4 with an ODD number of units per pixel: the 16 lanes of a ds_read_b128
// phase hit 16 different 16-byte bank groups.
#pragma once
#include "conv_igemm.hpp"

namespace dcscn {

typedef const __attribute__((address_space(3))) f32x4* stream_lds_rd;
typedef __attribute__((address_space(3))) f32x4* stream_lds_wr;

__device__ __forceinline__ void stream_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct StreamRow {
    int img, r;           // image, image row (may lie outside [0, H): zero row)
    int sx;               // image column of computed column 0
    int ux0, ux1;         // stored columns [ux0, ux1)
    bool zero;            // separator or outside the image: every ring gets zeros, nothing is stored
    bool store;           // row belongs to the block's interior
};

__device__ __forceinline__ StreamRow stream_row(const StreamArgs& a, int j0, int g) {
    StreamRow o;
    const int per = a.rows_c + 1;
    const int job = j0 + g / per, i = g % per;
    const int per_img = a.n_strips * a.n_blocks;
    o.img = job / per_img;
    const int rem = job % per_img;
    const int blk = rem / a.n_strips, strip = rem % a.n_strips;
    const int yb = a.n_blocks == 1 ? 0 : blk * a.useful_h - a.halo;
    o.r = yb + i;
    o.zero = i == a.rows_c || o.r < 0 || o.r >= a.H;
    const int y0 = a.n_blocks == 1 ? 0 : blk * a.useful_h;
    const int y1 = a.n_blocks == 1 ? a.H : min(a.H, y0 + a.useful_h);
    o.store = !o.zero && o.r >= y0 && o.r < y1;
    o.sx = a.n_strips == 1 ? 0 : strip * a.useful_w - a.halo;
    o.ux0 = a.n_strips == 1 ? 0 : strip * a.useful_w;
    o.ux1 = a.n_strips == 1 ? a.W : min(a.W, o.ux0 + a.useful_w);
    return o;
}

__device__ __forceinline__ f32x4 stream_prelu(f32x4 v, f32x4 b, f32x4 al) {
    v += b;
    v.x = v.x > 0.0f ? v.x : al.x * v.x;
    v.y = v.y > 0.0f ? v.y : al.y * v.y;
    v.z = v.z > 0.0f ? v.z : al.z * v.z;
    v.w = v.w > 0.0f ? v.w : al.w * v.w;
    return v;
}

// ---- CNN1: Y -> depthwise 3x3 (one channel) -> pointwise 1 -> C1, bias, PReLU ------------------------------------
__device__ __forceinline__ void stream_first_role(const StreamArgs& a, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    float w9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w9[i] = a.blob[a.first_w + i];
    f32x4 pw[2], bs[2], al[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        pw[n] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + 12 + n * 16 + 4 * q);
        bs[n] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + 44 + n * 16 + 4 * q);
        al[n] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + 76 + n * 16 + 4 * q);
    }
    // rows g-1, g, g+1 in registers and row g+2 in flight; a zero row IS the SAME padding of its neighbours
    float xw[4][kStreamMT][3];
    auto load_row = [&](int gs, float (&dst)[kStreamMT][3]) DCSCN_INL {
        const bool in = gs >= 0 && gs < rows;
        const StreamRow ri = stream_row(a, j0, in ? gs : 0);
        const bool live = in && !ri.zero;
        const float* row = a.x + ((size_t)ri.img * a.H + (live ? ri.r : 0)) * a.W;
#pragma unroll
        for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int cx = ri.sx + 16 * m + j + dx - 1;
                dst[m][dx] = live && cx >= 0 && cx < a.W ? row[cx] : 0.0f;
            }
    };
#pragma unroll
    for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xw[1][m][dx] = 0.0f;
    load_row(0, xw[2]);
    load_row(1, xw[3]);
    for (int t = 0; t < T; ++t) {
        const int g = t;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) xw[s][m][dx] = xw[s + 1][m][dx];
        load_row(g + 2, xw[3]);
        const bool live = g < rows;
        float d[kStreamMT];
        bool ok[kStreamMT];
        if (live) {
            const StreamRow ri = stream_row(a, j0, g);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                d[m] = 0.0f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) d[m] = fmaf(w9[dy * 3 + dx], xw[dy][m][dx], d[m]);
                const int cx = ri.sx + 16 * m + j;
                ok[m] = !ri.zero && cx >= 0 && cx < a.W;
            }
        }
        stream_barrier();
        if (live) {
            const unsigned slot = (unsigned)(g % 3);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (n * 4 + q < a.first_out.quads) {
                        const f32x4 r = stream_prelu(pw[n] * d[m], bs[n], al[n]);
                        *(stream_lds_wr)(uintptr_t)(lds0 + a.first_out.off +
                                                   ((slot * kStreamRowPx + 16 * m + j + 1) * a.first_out.units + n * 4 + q) * 16u) =
                            ok[m] ? r : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
        }
        stream_barrier();
    }
}

// ---- CNN2 .. CNNL, B2: depthwise 3x3 from the predecessor's ring -> pointwise GEMM -> bias, PReLU -----------------
__device__ __forceinline__ void stream_conv_role(const StreamArgs& a, const StreamConv& c, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    const int in_chunks = (c.in.quads + 3) >> 2;
    const int out_tiles = (c.out.quads + 3) >> 2;
    const unsigned in_px = (unsigned)c.in.units * 16u, in_row = (unsigned)kStreamRowPx * in_px;
    for (int t = 0; t < T; ++t) {
        const int g = t - c.lag;
        const bool live = g >= 0 && g < rows;
        f32x4 acc[kStreamMT][2];
#pragma unroll
        for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        StreamRow ri{};
        if (live) {
            ri = stream_row(a, j0, g);
            if (!ri.zero) {
                unsigned rowb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowb[dy] = lds0 + c.in.off + (unsigned)((g + 2 + dy) % 3) * in_row;   // rows g-1, g, g+1
                static_for<0, 2>([&](auto ch_) DCSCN_INL {
                    constexpr int ch = decltype(ch_)::value;
                    if (ch < in_chunks) {
                        const int quad = ch * 4 + q;
                        const bool qv = quad < c.in.quads;
                        const unsigned qoff = (unsigned)(qv ? quad : 0) * 16u;
                        // pointwise filter of the chunk: 2 KB per wave and step from L1 / L2 (resident it costs 16 VGPRs too many)
                        float wp[4][2];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int n = 0; n < 2; ++n) wp[s][n] = a.blob[c.wp + ((ch * 4 + s) * 2 + n) * 64 + lane];
                        f32x4 dw[9];
#pragma unroll
                        for (int k = 0; k < 9; ++k)
                            dw[k] = *(stream_lds_rd)(uintptr_t)(lds0 + c.dww + (unsigned)(k * c.in.quads) * 16u + qoff);
                        static_for<0, kStreamMT>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) {
                                    const f32x4 xv = *(stream_lds_rd)(uintptr_t)(rowb[dy] + (unsigned)(16 * m + j + dx) * in_px + qoff);
                                    d += dw[dy * 3 + dx] * xv;
                                }
                            if (!qv) d = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[s][0], d[s], acc[m][0], 0, 0, 0);
                                if (out_tiles > 1) acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[s][1], d[s], acc[m][1], 0, 0, 0);
                            }
                        });
                    }
                });
            }
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                const int cx = ri.sx + 16 * m + j;
                const bool ok = !ri.zero && cx >= 0 && cx < a.W;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(a.blob + c.ba + n * 16 + 4 * q);        // L1 hits: once per row
                    const f32x4 al = *reinterpret_cast<const f32x4*>(a.blob + c.ba + 32 + n * 16 + 4 * q);
                    const f32x4 r = stream_prelu(acc[m][n], bs, al);
                    acc[m][n] = ok ? r : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
            if (c.to_global && ri.store) {
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) {
                    const int cx = ri.sx + 16 * m + j;
                    if (cx >= ri.ux0 && cx < ri.ux1) {
                        float* o = a.out + (((size_t)ri.img * a.H + ri.r) * a.W + cx) * a.out_stride;
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            if (n * 4 + q < c.out.quads) *reinterpret_cast<f32x4*>(o + (n * 4 + q) * 4) = acc[m][n];
                    }
                }
            }
        }
        stream_barrier();
        if (live && !c.to_global) {
            const unsigned slot = (unsigned)(g % 3);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (n * 4 + q < c.out.quads)
                        *(stream_lds_wr)(uintptr_t)(lds0 + c.out.off + ((slot * kStreamRowPx + 16 * m + j + 1) * c.out.units + n * 4 + q) * 16u) = acc[m][n];
        }
        stream_barrier();
    }
}

// ---- A1 || B1: one (feature layer, row) contribution per step and wave --------------------------------------------
__device__ __forceinline__ void stream_nin_role(const StreamArgs& a, int w, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    const int L = a.L;
    f32x4 acc[2][kStreamMT][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[p][m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int t = 0; t < T; ++t) {
        // row g = t + 1 - 2l receives layer l's contribution at step t; (g >> 1) mod L == w picks this wave's l
        const int hl = (t + 1) >> 1;
        int l = ((hl - w) % L + L) % L;
        if (l == 0) l = L;
        const int g = t + 1 - 2 * l;
        const bool live = g >= 0 && g < rows;
        const bool last = l == L;
        StreamRow ri{};
        f32x4 b1v[kStreamMT];
        if (live) {
            ri = stream_row(a, j0, g);
            const StreamNinSrc& s = a.nin[l - 1];
            auto body = [&](auto p_) DCSCN_INL {
                constexpr int p = decltype(p_)::value;
                if (!ri.zero) {
                    const unsigned rowb = lds0 + s.ring.off + (unsigned)(g % 3) * (unsigned)(kStreamRowPx * s.ring.units) * 16u;
                    for (int ch = 0; ch < s.chunks; ++ch) {
                        const int quad = ch * 4 + q;
                        const bool qv = quad < s.ring.quads;
                        const unsigned qoff = (unsigned)(qv ? quad : 0) * 16u;
                        const f32x4 w0 = *(stream_lds_rd)(uintptr_t)(lds0 + s.w + (unsigned)((ch * 2 + 0) * 64 + lane) * 16u);
                        const f32x4 w1 = *(stream_lds_rd)(uintptr_t)(lds0 + s.w + (unsigned)((ch * 2 + 1) * 64 + lane) * 16u);
                        const int steps = ch == s.chunks - 1 ? s.last_steps : 4;
                        static_for<0, kStreamMT>([&](auto m_) DCSCN_INL {
                            constexpr int m = decltype(m_)::value;
                            f32x4 xv = *(stream_lds_rd)(uintptr_t)(rowb + (unsigned)((16 * m + j + 1) * s.ring.units) * 16u + qoff);
                            if (!qv) xv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (k < steps) {
                                    acc[p][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[k], xv[k], acc[p][m][0], 0, 0, 0);
                                    acc[p][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[k], xv[k], acc[p][m][1], 0, 0, 0);
                                }
                        });
                    }
                }
                if (last) {
                    // bias, PReLU; A1 -> Concat2 (global) now, the B1 quads (tile 0, nb <= 16) wait for the write phase
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 bs = *reinterpret_cast<const f32x4*>(a.blob + a.nin_ba + n * 16 + 4 * q);
                        const f32x4 al = *reinterpret_cast<const f32x4*>(a.blob + a.nin_ba + 32 + n * 16 + 4 * q);
                        const int quad = n * 4 + q;
#pragma unroll
                        for (int m = 0; m < kStreamMT; ++m) {
                            const int cx = ri.sx + 16 * m + j;
                            const bool ok = !ri.zero && cx >= 0 && cx < a.W;
                            const f32x4 r = stream_prelu(acc[p][m][n], bs, al);
                            const f32x4 v = ok ? r : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                            acc[p][m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                            if (n == 0) b1v[m] = v;
                            if (ri.store && cx >= ri.ux0 && cx < ri.ux1 && quad >= a.nb_quads && quad * 4 < a.out_stride)
                                *reinterpret_cast<f32x4*>(a.out + (((size_t)ri.img * a.H + ri.r) * a.W + cx) * a.out_stride + quad * 4) = v;
                        }
                    }
                }
            };
            if (g & 1) body(std::integral_constant<int, 1>{});
            else body(std::integral_constant<int, 0>{});
        }
        stream_barrier();
        if (live && last) {
            const unsigned slot = (unsigned)(g % 3);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
                if (q < a.nb_quads)       // B1 lives in tile 0 (nb <= 16)
                    *(stream_lds_wr)(uintptr_t)(lds0 + a.b1.off + ((slot * kStreamRowPx + 16 * m + j + 1) * a.b1.units + q) * 16u) = b1v[m];
        }
        stream_barrier();
    }
}

__global__ __launch_bounds__(1024) void feat_stream(const StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    {
        f32x4* s4 = reinterpret_cast<f32x4*>(smem);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = tid; i < a.ring_bytes / 16; i += blockDim.x) s4[i] = z;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.blob + a.ldsw_src);
        for (int i = tid; i < a.ldsw_bytes / 16; i += blockDim.x) s4[a.ring_bytes / 16 + i] = src[i];
        __syncthreads();
    }
    const int j0 = blockIdx.x * a.jobs_per_wg;
    const int j1 = min(a.n_jobs, j0 + a.jobs_per_wg);
    const int rows = (j1 - j0) * (a.rows_c + 1);
    const int T = rows + a.total_lag;
    if (wave == 0) stream_first_role(a, lds0, j0, rows, T, lane);
    else if (wave <= a.n_conv) stream_conv_role(a, a.conv[wave - 1], lds0, j0, rows, T, lane);
    else stream_nin_role(a, wave - 1 - a.n_conv, lds0, j0, rows, T, lane);
}

}  // namespace dcscn
