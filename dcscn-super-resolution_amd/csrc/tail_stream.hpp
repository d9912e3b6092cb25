// Row-streamed x4 upsampler + reconstruction of the separable narrow nets: Up-PS (separable 3x3, C -> 4C) + depth_to_space(2),
// Up-PS2 (separable 3x3, C -> 4) + depth_to_space(2), R-CNN1 (separable 3x3, 1 -> 1) + the bicubic residual
// (DCSCN.py:293-325 with tf_graph.py:155-177, 239-249) in ONE launch.  Layer by layer the C-channel tensor at twice the
// resolution (512 B per LR pixel) is written and read back through HBM -- two thirds of the tail's time; here it lives in
// a six-row LDS ring.
//
// Same machinery as feat_stream.hpp (48-column strips, one LR row per step, compute phase / barrier / write phase /
// barrier, jobs streamed back to back behind one zero row), with these roles:
//
//   wave 0        loads Concat2 row t+1 into registers while row t waits for its write phase -> IN ring (3 rows)
//   waves 1..3    Up-PS on 16 pixels each: depthwise from IN (one pixel per lane), pointwise C -> 4C (MFMA, 8 channel tiles),
//                 bias -> depth_to_space -> rows 2g, 2g+1 of the U ring (6 rows of 96 pixels x C channels)
//   waves 4..7    Up-PS2 for one of the two new U rows x one 48-pixel half: depthwise from U, pointwise C -> 4 (MFMA, one
//                 tile of which 4 columns are real), bias -> depth_to_space -> 2 rows x 96 pixels of the V ring (1 channel,
//                 12 rows of 192 pixels)
//   waves 8..9    R-CNN1 on two of the four new HR rows each: 3x3 on V, the pointwise scalar, + x2 -> y (global)
//
// Lags: Up-PS runs 2 LR rows behind the loader, Up-PS2 2 behind Up-PS, R-CNN1 2 behind Up-PS2.
#pragma once
#include "feat_stream.hpp"

namespace dcscn {

constexpr int kTailURowPx = 2 * kStreamPX + 2;      // U ring row: pixels -1 .. 96
constexpr int kTailUSlots = 6;
constexpr int kTailVRow = 4 * kStreamPX + 4;        // V ring row in floats: 2 zero floats, 192 pixels, 2 zero floats
constexpr int kTailVSlots = 12;

__device__ __forceinline__ StreamArgs tail_geometry(const TailArgs& a) {
    // stream_row only looks at the job geometry
    StreamArgs g{};
    g.H = a.H; g.W = a.W;
    g.n_strips = a.n_strips; g.useful_w = a.useful_w; g.halo = a.halo;
    g.n_blocks = a.n_blocks; g.useful_h = a.useful_h; g.rows_c = a.rows_c;
    return g;
}

// ---- loader: Concat2 rows -> IN ring -------------------------------------------------------------------------------
__device__ __forceinline__ void tail_load_role(const TailArgs& a, const StreamArgs& geo, unsigned lds0, int j0, int rows, int T, int lane) {
    constexpr int ITEMS = (kStreamPX * 8 + 63) / 64;      // float4 items per lane (up to 8 quads per pixel)
    const int n_items = kStreamPX * a.in.quads;
    StreamCursor cur;
    f32x4 now[ITEMS], nxt[ITEMS];
    auto load_row = [&](int gs, f32x4 (&dst)[ITEMS]) DCSCN_INL {
        const bool in = gs >= 0 && gs < rows;
        const StreamRow ri = stream_row(geo, j0, cur, in ? gs : 0);
        const bool live = in && !ri.zero;
        const float* row = a.c2 + ((size_t)ri.img * a.H + (live ? ri.r : 0)) * a.W * a.c2_stride;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = lane + 64 * i;
            const int px = it / a.in.quads, quad = it - px * a.in.quads;
            const int cx = ri.sx + px;
            dst[i] = live && it < n_items && cx >= 0 && cx < a.W && STREAM_ABL != 10 && STREAM_ABL != 13 && STREAM_ABL != 14 ? *reinterpret_cast<const f32x4*>(row + (size_t)cx * a.c2_stride + 4 * quad) : kStreamZero;
        }
    };
    load_row(0, now);
    for (int t = 0; t < T; ++t) {
        load_row(t + 1, nxt);
        stream_barrier();
        if (t < rows) {
            const unsigned slot = (unsigned)(t % 3);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int it = lane + 64 * i;
                const int px = it / a.in.quads, quad = it - px * a.in.quads;
                if (it < n_items) stream_st(lds0 + a.in.off + ((slot * kStreamRowPx + px + 1) * a.in.units + quad) * 16u, now[i]);
            }
        }
        stream_barrier();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) now[i] = nxt[i];
    }
}

// ---- Up-PS on 16 pixels, all four sub-pixel phases ------------------------------------------------------------------------
// (splitting by phase instead would make four waves repeat the same depthwise; here a lane owns ONE pixel, its window is 3
// reads per row, and the wave runs the whole [4C x C] pointwise: 8 channel tiles of accumulators)
template <int QI, bool F16>
__device__ __forceinline__ void tail_up1_role(const TailArgs& a, const StreamArgs& geo, int seg, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    constexpr unsigned in_px = (unsigned)(QI | 1) * 16u, in_row = (unsigned)kStreamRowPx * in_px;
    const unsigned u_px = (unsigned)a.u.units * 16u, u_row = (unsigned)kTailURowPx * u_px;
    const int px = 16 * seg + j;
    const int tiles = a.u.quads > 4 ? 2 : 1;          // channel tiles per phase
    StreamCursor cur;
    for (int t = 0; t < T; ++t) {
        const int g = t - 2;
        const bool live = g >= 0 && g < rows;
        f32x4 acc[1][8];
        bool keep = false;                  // false: zero row or a column outside the image -> zeros
        bool inside = false;                // wave uniform: the whole strip lies inside the image (no per-lane select needed)
        if (live && STREAM_ABL != 11 && STREAM_ABL != 13 && STREAM_ABL != 14) {
            const StreamRow ri = stream_row(geo, j0, cur, g);
            if (!ri.zero) {
                unsigned rowb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowb[dy] = lds0 + a.in.off + (unsigned)((g + 2 + dy) % 3) * in_row + (unsigned)px * in_px;
                f32x4 bs[8];
                if constexpr (!F16) {
#pragma unroll
                    for (int n = 0; n < 8; ++n) bs[n] = stream_ld(lds0 + a.a_bias + (unsigned)(n * 4 + q) * 16u);
                }
                stream_dw_pw<QI, 8, 1, F16>(acc, bs, lds0, rowb, a.a_dww, a.a_wp, q, lane, lds0 + a.a_bias + (unsigned)q * 16u);
                if constexpr (F16) {
#pragma unroll
                    for (int n = 0; n < 8; ++n) acc[0][n] = acc[0][n] * a.a_inv;
                }
                const int cx = ri.sx + px;
                keep = cx >= 0 && cx < a.W;
                inside = ri.sx >= 0 && ri.sx + kStreamPX <= a.W;
            }
        }
        stream_barrier();
        if (live) {
            // conv channel 16n + 4q + e = phase * C + c (depth_to_space: phase = 2 dy + dx)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int ph = tiles == 2 ? n >> 1 : n, quad = (tiles == 2 ? (n & 1) * 4 : 0) + q;
                const unsigned slot = (unsigned)((2 * g + (ph >> 1)) % kTailUSlots);
                if (quad < a.u.quads && ph < 4) {
                    const unsigned dst = lds0 + a.u.off + slot * u_row + (unsigned)(2 * px + (ph & 1) + 1) * u_px + (unsigned)quad * 16u;
                    if (inside) stream_st(dst, acc[0][n]);
                    else stream_st(dst, keep ? acc[0][n] : kStreamZero);
                }
            }
        }
        stream_barrier();
    }
}

// ---- Up-PS2 on U row 2g + r2, pixels 48 * half .. + 47 ------------------------------------------------------------------
template <int QU, bool F16>
__device__ __forceinline__ void tail_up2_role(const TailArgs& a, const StreamArgs& geo, int r2, int half, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    constexpr unsigned u_px = (unsigned)(QU | 1) * 16u, u_row = (unsigned)kTailURowPx * u_px;
    StreamCursor cur;
    for (int t = 0; t < T; ++t) {
        const int g = t - 4;
        const bool live = g >= 0 && g < rows;
        f32x4 acc[kStreamMT][1];
        bool okm[kStreamMT] = {false, false, false};
        bool inside = false;                // wave uniform: the strip lies inside the image
        if (live && STREAM_ABL != 12 && STREAM_ABL != 13 && STREAM_ABL != 14) {
            const StreamRow ri = stream_row(geo, j0, cur, g);
            if (!ri.zero) {
                const int ur = 2 * g + r2;
                unsigned rowb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
                    rowb[dy] = lds0 + a.u.off + (unsigned)((ur - 1 + dy + kTailUSlots) % kTailUSlots) * u_row + (unsigned)(kStreamPX * half + 3 * j) * u_px;
                const f32x4 bs[1] = {F16 ? kStreamZero : stream_ld(lds0 + a.b_bias)};
                stream_dw_pw<QU, 1, kStreamMT, F16>(acc, bs, lds0, rowb, a.b_dww, a.b_wp, q, lane, lds0 + a.b_bias);
                if constexpr (F16) {
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) acc[m][0] = acc[m][0] * a.b_inv;
                }
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) {
                    const int cx2 = 2 * ri.sx + kStreamPX * half + 3 * j + m;
                    okm[m] = cx2 >= 0 && cx2 < 2 * a.W;
                }
                inside = ri.sx >= 0 && ri.sx + kStreamPX <= a.W;
            }
        }
        stream_barrier();
        if (live && q == 0) {
            // depth_to_space(2) of the 4 channels: (v.x v.y / v.z v.w) -> HR rows 4g + 2 r2 + {0, 1}, pixels 2 px2 + {0, 1}
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) f32x2* lds_f2;
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                const int px4 = 2 * (kStreamPX * half + 3 * j + m);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const unsigned slot = (unsigned)((4 * g + 2 * r2 + dy) % kTailVSlots);
                    f32x2 v = dy == 0 ? f32x2{acc[m][0].x, acc[m][0].y} : f32x2{acc[m][0].z, acc[m][0].w};
                    if (!inside && !okm[m]) v = f32x2{0.0f, 0.0f};
                    *(lds_f2)(uintptr_t)(lds0 + a.v_off + (slot * kTailVRow + px4 + 2) * 4u) = v;
                }
            }
        }
        stream_barrier();
    }
}

// ---- R-CNN1 + residual on HR rows 4g + 2c + {0, 1} -----------------------------------------------------------------------
template <bool F16, bool GATE>
__device__ __forceinline__ void tail_rec_role(const TailArgs& a, const StreamArgs& geo, int c, unsigned lds0, int j0, int rows, int T, int lane) {
    typedef const __attribute__((address_space(3))) float* lds_f1;
    StreamCursor cur, pcur;
    const int W4 = 4 * a.W;
    // the bicubic pixels of the NEXT step's rows are fetched one step ahead (a global load issued and consumed inside
    // one step put its whole latency, ~4000 cycles, on the step's critical path)
    float res[2][3], nres[2][3];
    auto fetch = [&](int gs, float (&dst)[2][3]) DCSCN_INL {
        const bool in = gs >= 0 && gs < rows;
        const StreamRow ri = stream_row(geo, j0, pcur, in ? gs : 0);
        const int cx0 = 4 * ri.sx + 3 * lane;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int cx = cx0 + k;
                const bool ok = in && ri.store && cx >= 4 * ri.ux0 && cx < 4 * ri.ux1 && STREAM_ABL != 9 && STREAM_ABL != 13;
                dst[e][k] = ok ? a.x2[((size_t)ri.img * 4 * a.H + 4 * ri.r + 2 * c + e) * W4 + cx] : 0.0f;
            }
    };
    fetch(-6, res);
    for (int t = 0; t < T; ++t) {
        const int g = t - 6;
        const bool live = g >= 0 && g < rows;
        fetch(g + 1, nres);
        if (live) {
            const StreamRow ri = stream_row(geo, j0, cur, g);
            if (ri.store && STREAM_ABL != 9 && STREAM_ABL != 13 && (!GATE || a.redo[1 + ri.img] != 0)) {
                const int vr0 = 4 * g + 2 * c;                  // first output row in V-row numbering
                const int cx0 = 4 * ri.sx + 3 * lane;           // first of the lane's three HR columns
                float v[4][5];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const unsigned slot = (unsigned)((vr0 - 1 + rr + kTailVSlots) % kTailVSlots);
#pragma unroll
                    for (int k = 0; k < 5; ++k) v[rr][k] = *(lds_f1)(uintptr_t)(lds0 + a.v_off + (slot * kTailVRow + 3 * lane + k + 1) * 4u);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float s = 0.0f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) s = fmaf(a.c_w[dy * 3 + dx], v[e + dy][k + dx], s);
                        const int cx = cx0 + k;
                        if (cx >= 4 * ri.ux0 && cx < 4 * ri.ux1) {
                            const float out = s * a.c_scale + res[e][k];
                            if (F16 && !(fabsf(out) <= 3.0e38f) && a.redo) { a.redo[0] = 1; a.redo[1 + ri.img] = 1; }    // not finite: the image goes to the float32 plan
                            a.y[((size_t)ri.img * 4 * a.H + 4 * ri.r + 2 * c + e) * W4 + cx] = out;
                        }
                    }
            }
        }
        stream_barrier();
        stream_barrier();
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 3; ++k) res[e][k] = nres[e][k];
    }
}

// F16: the two pointwise GEMMs on the f16 matrix pipe (feat_stream.hpp: stream_dw_pw); false: f32 MFMAs, and the float32 plan (redo_check)
template <bool F16, bool GATE = false>
__global__ __launch_bounds__(640) void tail_stream(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (GATE && a.redo[0] == 0) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    {
        f32x4* s4 = reinterpret_cast<f32x4*>(smem);
        for (int i = tid; i < a.ring_bytes / 16; i += blockDim.x) s4[i] = kStreamZero;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.blob);
        for (int i = tid; i < a.ldsw_bytes / 16; i += blockDim.x) s4[a.ring_bytes / 16 + i] = src[i];
        __syncthreads();
    }
    const StreamArgs geo = tail_geometry(a);
    const int j0 = blockIdx.x * a.jobs_per_wg;
    const int j1 = min(a.n_jobs, j0 + a.jobs_per_wg);
    const int rows = (j1 - j0) * (a.rows_c + 1);
    const int T = rows + 6;
    // wave -> role so that the four SIMDs (wave w runs on SIMD w & 3; MFMA and VALU time of a SIMD's waves add up) carry about
    // the same estimated cycles per row: SIMD0 = two Up-PS2 waves + the loader, SIMD1 = one Up-PS wave + both R-CNN1 waves,
    // SIMD2 and SIMD3 = one Up-PS + one Up-PS2 wave each.  0 loader, 1..3 Up-PS, 4..7 Up-PS2, 8..9 R-CNN1.
    constexpr int kRole[10] = {4, 1, 2, 3, 5, 8, 6, 7, 0, 9};
    const int role = kRole[wave];
    if (role == 0) tail_load_role(a, geo, lds0, j0, rows, T, lane);
    else if (role <= 3) tail_up1_role<8, F16>(a, geo, role - 1, lds0, j0, rows, T, lane);      // instantiated for 32 -> 4 x 32 -> 4 channels (api.hip: fuse_tail_stream)
    else if (role <= 7) tail_up2_role<8, F16>(a, geo, (role - 4) >> 1, (role - 4) & 1, lds0, j0, rows, T, lane);
    else tail_rec_role<F16, GATE>(a, geo, role - 8, lds0, j0, rows, T, lane);
}

}  // namespace dcscn
