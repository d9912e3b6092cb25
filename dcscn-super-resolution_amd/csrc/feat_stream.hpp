// Row-streamed feature extractor of the separable (depthwise_separable=True) narrow nets: CNN1 .. CNNL, A1 || B1 and B2
// (DCSCN.py:240-291 with tf_graph.py:155-177 build_depthwise_separable_conv) in ONE launch, intermediate tensors in LDS.
//
// Why: layer by layer these nets move 2-3.6 KB per LR pixel through HBM for 36-132 B of compulsory I/O (input Y, the
// 32-channel Concat2).  Here a workgroup owns a 48-pixel-wide column strip of an image and walks down its rows; every
// layer keeps only a THREE-ROW ring of its output in LDS, which is all the next 3x3 layer needs:
//
// (which wave plays which role is a table, StreamArgs::role, filled so that the MFMA counts of the four SIMDs are even)
//   1 wave            CNN1: depthwise 3x3 on Y (global, prefetched one step ahead) -> pointwise 1 -> C1, VALU only
//   L-1 waves         CNN2 .. CNNL: depthwise 3x3 (VALU, from the predecessor's ring) -> pointwise GEMM on the MFMA
//                     pipe (filters in LDS, the bias is the first MFMA's C operand) -> PReLU -> own ring
//   1 wave            B2: the same, from the B1 ring to global Concat2[0 : nb)
//   L waves           A1 || B1: H_concat never exists.  The 1x1 layers are linear in the concat, so the contribution of
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
// ds_read_b128 = channels 4q' .. 4q'+3 (q' = 4 * chunk + q) of a pixel and uses its four floats as the B operand of four
// k-steps; k-step s of a chunk therefore covers channels {16 chunk + 4q + s}, and the filters are packed to match.
// Column j of pixel tile m is pixel 3j + m, not 16m + j: a lane's three pixels are neighbours, so the 3x3 depthwise
// window of all three needs 5 reads per row instead of 9.  Ring rows are [pixel -1 .. 48][units] float4 with an odd number
// of units per pixel.  (ds_read_b128 lane groups mix two lane quarters, MI355X_MICROARCH.md, so these reads are still 2-way
// conflicted; a planar [quad][pixel] layout that is conflict free was measured at the same time -- DESIGN.md 3.7.)
//
// What bounds it: f32 MFMA does not overlap with VALU instructions of other waves on a SIMD (tools/mfma_valu_overlap.hip),
// so a row costs 32 cycles per MFMA PLUS ~4.5 per VALU instruction per SIMD; the code below is written to keep the
// non-FMA instruction count down (immediate LDS offsets, compile-time ring geometry, masks only at image edges).
#pragma once
#include "conv_igemm.hpp"

#ifndef STREAM_ABL
#define STREAM_ABL 0      // tools/stream_abl.sh: timing-only builds with one cost removed (results wrong by design); 1 = no MFMA in the
                          // separable convs (4 VALU FMAs per tile instead), 15 = no MFMA anywhere (A1 || B1 waves too): a bound for ANY faster matrix instruction,
                          // 16 = PROJECTION of a split16 variant: per two 16-channel chunks the 8 NT v_mfma_f32_16x16x4_f32 are replaced by the hi / lo
                          // split of the B values (12 VALU per pixel tile) and 3 NT v_mfma_f32_16x16x32_f16 on stand-in A operands (finite bit patterns)
#endif
#include "split16.hpp"

namespace dcscn {

typedef const __attribute__((address_space(3))) f32x4* stream_lds_rd;
typedef __attribute__((address_space(3))) f32x4* stream_lds_wr;

__device__ __forceinline__ void stream_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef STREAM_DBG
// timing probe: workgroup 0 records the shader clock at four points of steps 64..127 per wave
#define STREAM_STAMP(a, t, k)                                                                                     \
    do {                                                                                                          \
        if ((a).dbg && blockIdx.x == 0 && (t) >= 64 && (t) < 128 && (threadIdx.x & 63) == 0)                        \
            (a).dbg[((threadIdx.x >> 6) * 64 + ((t) - 64)) * 4 + (k)] = clock64();                                 \
    } while (0)
#else
#define STREAM_STAMP(a, t, k) do { } while (0)
#endif
__device__ __forceinline__ f32x4 stream_ld(unsigned addr) { return *(stream_lds_rd)(uintptr_t)addr; }
__device__ __forceinline__ void stream_st(unsigned addr, f32x4 v) { *(stream_lds_wr)(uintptr_t)addr = v; }

// Decodes stream rows (all values are wave uniform).  A job = (image, row block, column strip); its rows are followed
// by one separator row.  Rows are visited in increasing order, so the divisions run once per job.
struct StreamCursor {
    int base = 1 << 30;   // stream index of the cached job's first row
    int img = 0, yb = 0, y0 = 0, y1 = 0, sx = 0, ux0 = 0, ux1 = 0;
};
struct StreamRow {
    int img, r;           // image, image row (may lie outside [0, H): zero row)
    int sx;               // image column of computed column 0
    int ux0, ux1;         // stored columns [ux0, ux1)
    bool zero;            // separator or outside the image: every ring gets zeros, nothing is stored
    bool store;           // row belongs to the block's interior
};

__device__ __forceinline__ StreamRow stream_row(const StreamArgs& a, int j0, StreamCursor& c, int g) {
    const int per = a.rows_c + 1;
    if (g < c.base || g >= c.base + per) {
        const int jl = g / per;
        c.base = jl * per;
        const int job = j0 + jl;
        const int per_img = a.n_strips * a.n_blocks;
        c.img = job / per_img;
        const int rem = job - c.img * per_img;
        const int blk = rem / a.n_strips, strip = rem - blk * a.n_strips;
        const bool one_b = a.n_blocks == 1, one_s = a.n_strips == 1;
        c.yb = one_b ? 0 : blk * a.useful_h - a.halo;
        c.y0 = one_b ? 0 : blk * a.useful_h;
        c.y1 = one_b ? a.H : min(a.H, c.y0 + a.useful_h);
        c.sx = one_s ? 0 : strip * a.useful_w - a.halo;
        c.ux0 = one_s ? 0 : strip * a.useful_w;
        c.ux1 = one_s ? a.W : min(a.W, c.ux0 + a.useful_w);
    }
    StreamRow o;
    const int i = g - c.base;
    o.img = c.img;
    o.r = c.yb + i;
    o.zero = i == a.rows_c || o.r < 0 || o.r >= a.H;
    o.store = !o.zero && o.r >= c.y0 && o.r < c.y1;
    o.sx = c.sx;
    o.ux0 = c.ux0;
    o.ux1 = c.ux1;
    return o;
}

// bias + PReLU as v + (alpha - 1) * min(v, 0): 4 v_min + 2 v_pk_fma instead of 4 compares, 4 selects and 2 multiplies
// (`am1` = alpha - 1, precomputed on the host; differs from alpha * v by one rounding of a product that is then added to v)
__device__ __forceinline__ float stream_min0(float v) {
    // (a hand-written `v_min_f32 n, 0, v` would save the canonicalising v_max fminf() costs -- but the compiler does not see an
    // asm statement's MFMA -> VALU read hazard, and the v_min read stale accumulators: measured, 3e-4 relative error)
    return fminf(v, 0.0f);
}
__device__ __forceinline__ f32x4 stream_prelu(f32x4 v, f32x4 am1) {
    const f32x4 n = {stream_min0(v.x), stream_min0(v.y), stream_min0(v.z), stream_min0(v.w)};
    return v + am1 * n;
}

constexpr f32x4 kStreamZero = {0.0f, 0.0f, 0.0f, 0.0f};

// A value stored to global memory by an F16 kernel is not finite (an activation beyond the f16 range somewhere upstream of it): the image
// goes to the float32 plan (exec.hip: run_forward).  `z` = opaque_zero().
__device__ __forceinline__ void stream_flag(int32_t* redo, int img, const f32x4 v, float z) {
    const float c = nonfinite_acc(0.0f, v, z);
    if (c != c && redo) { redo[0] = 1; redo[1 + img] = 1; }
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
    // rows g-1, g, g+1 in registers and row g+2 in flight; a zero row IS the SAME padding of its neighbours.
    // A lane holds columns 3j-1 .. 3j+3 of each row: the windows of its pixels 3j, 3j+1, 3j+2.
    float xw[4][5];
    StreamCursor lc, cc;
    auto load_row = [&](int gs, float (&dst)[5]) DCSCN_INL {
        const bool in = gs >= 0 && gs < rows;
        const StreamRow ri = stream_row(a, j0, lc, in ? gs : 0);
        const bool live = in && !ri.zero;
        const float* row = a.x + ((size_t)ri.img * a.H + (live ? ri.r : 0)) * a.W;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int cx = ri.sx + 3 * j + k - 1;
            dst[k] = live && cx >= 0 && cx < a.W && STREAM_ABL != 7 && STREAM_ABL != 8 ? row[cx] : 0.0f;
        }
    };
#pragma unroll
    for (int k = 0; k < 5; ++k) xw[1][k] = 0.0f;
    load_row(0, xw[2]);
    load_row(1, xw[3]);
    for (int t = 0; t < T; ++t) {
        STREAM_STAMP(a, t, 0);
        const int g = t;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int k = 0; k < 5; ++k) xw[s][k] = xw[s + 1][k];
        load_row(g + 2, xw[3]);
        const bool live = g < rows;
        // (everything but the stores happens in the compute phase: VALU work in the write phase delays the SIMD's other waves)
        f32x4 v[kStreamMT][2];
        if (live) {
            const StreamRow ri = stream_row(a, j0, cc, g);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                float d = 0.0f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) d = fmaf(w9[dy * 3 + dx], xw[dy][m + dx], d);
                const int cx = ri.sx + 3 * j + m;
                const bool ok = !ri.zero && cx >= 0 && cx < a.W;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 r = stream_prelu(pw[n] * d + bs[n], al[n]);
                    v[m][n] = ok ? r : kStreamZero;
                }
            }
        }
        STREAM_STAMP(a, t, 1);
        stream_barrier();
        STREAM_STAMP(a, t, 2);
        if (live) {
            const unsigned wb = lds0 + a.first_out.off + (((unsigned)(g % 3) * kStreamRowPx + 3 * j + 1) * a.first_out.units + q) * 16u;
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (n * 4 + q < a.first_out.quads) stream_st(wb + (unsigned)(m * a.first_out.units + n * 4) * 16u, v[m][n]);
        }
        STREAM_STAMP(a, t, 3);
        stream_barrier();
    }
}

// K-step bookkeeping of a 16-channel chunk that holds QL (1..4) valid channel quads.  A lane (q = lane >> 4) always
// reads one float4 = one quad; with 4 or 3 valid quads lane q feeds element s of quad q into k-step s (4 k-steps, the
// lanes of a missing quad multiply zero filter rows).  With 2 valid quads the lanes q = 2, 3 read quads 0, 1 again and
// feed their elements 2, 3: all 8 channels in 2 k-steps.  With 1 valid quad every lane reads it and feeds element q:
// 1 k-step.  The filter images are packed to match (api.hip: stream_chunk_channel).
template <int QL> struct StreamChunk {
    static constexpr int STEPS = QL >= 3 ? 4 : QL;
    static __device__ __forceinline__ int quad(int q) { return QL >= 3 ? (q < QL ? q : QL - 1) : QL == 2 ? (q & 1) : 0; }
    static __device__ __forceinline__ float pick(const f32x4& d, int s, int q) {
        if (QL >= 3) return d[s];
        if (QL == 2) return q >= 2 ? d[s + 2] : d[s];
        return q == 0 ? d[0] : q == 1 ? d[1] : q == 2 ? d[2] : d[3];
    }
};

// The depthwise 3x3 + pointwise core shared by every separable layer of the streamed kernels: rowb[dy] = LDS address of
// (row dy, the lane's first window pixel, quad 0); the lane's three pixels are window positions 0..2, 1..3, 2..4.
// acc[m][n] += sum over chunks / k-steps of pointwise[n] x depthwise(pixel m).  QUADS = channel quads of the input ring
// (compile time: every LDS offset below is an immediate).
// `init[n]` (the bias) is the C operand of each accumulator's FIRST MFMA: no zero fill, no bias add afterwards -- on this
// chip every VALU instruction costs matrix-pipe time too (tools/mfma_valu_overlap.hip: they do not overlap).
//
// F16 (r05): the pointwise GEMM on v_mfma_f32_16x16x32_f16 at f32 accuracy (split16.hpp): the depthwise
// outputs of TWO 16-channel chunks are one K = 32 B operand (lane q: the 4 channels of quad q of each chunk), split into f16 (hi, lo) in
// registers; the filters are f16 (hi, lo) fragments scaled by 2^e (pack.hip: stream_f16_image), three products per accumulator -- 3 NT MT
// instructions of 16 cycles per chunk pair instead of 8 NT MT of 32, and beside the f16 instruction the depthwise VALU work of the SIMD's
// other waves overlaps (profiles/r03_pipe_probe.txt).  An odd last chunk takes the K = 16 form.  `init` holds the bias times 2^e; the
// caller multiplies the accumulators by 2^-e.  Every chunk is treated as four quads: lanes of a missing quad read a valid one (K::quad) and
// meet zero filter rows.
// F16: `init` is not read -- the bias (times 2^e) is fetched from LDS at the first MFMA (init_addr = address of tile 0's float4 of this
// lane, tile n 64 bytes further): eight registers fewer across the depthwise part.
template <int QUADS, int NT, int MT = kStreamMT, bool F16 = false>
__device__ __forceinline__ void stream_dw_pw(f32x4 (&acc)[MT][NT], const f32x4 (&init)[NT], unsigned lds0, const unsigned (&rowb)[3], int dww, int wpo,
                                             int q, int lane, unsigned init_addr = 0) {
    constexpr int CH = (QUADS + 3) / 4;
    constexpr unsigned PX = (unsigned)(QUADS | 1) * 16u;
    f32x4 dpair[F16 ? MT : 1];                                // F16: the first chunk of a pair waits here for the second
#if STREAM_ABL == 16
    f32x4 dkeep[MT];
#endif
    static_for<0, CH>([&](auto ch_) DCSCN_INL {
        constexpr int ch = decltype(ch_)::value;
        constexpr int QL = ch == CH - 1 ? QUADS - 4 * (CH - 1) : 4;
        using K = StreamChunk<QL>;
        const unsigned qoff = (unsigned)(ch * 4 + K::quad(q)) * 16u;
        f32x4 wp[F16 ? 1 : NT];
        if constexpr (!F16) {
#pragma unroll
            for (int n = 0; n < NT; ++n) wp[n] = stream_ld(lds0 + wpo + (unsigned)((ch * NT + n) * 64 + lane) * 16u);
        }
        f32x4 d[MT];
        const unsigned dwb = lds0 + dww + qoff;
        // rows are double buffered: the reads of row dy + 1 are in flight while row dy is multiplied (the compiler
        // barriers keep it from hoisting all three rows -- 96 VGPRs -- or none)
        f32x4 dw[2][3], xv[2][MT + 2];
        auto fetch = [&](auto dy_, auto b_) DCSCN_INL {
            constexpr int dy = decltype(dy_)::value, b = decltype(b_)::value;
            const unsigned xb = rowb[dy] + qoff;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) dw[b][dx] = STREAM_ABL == 4 ? f32x4{1.0f, 2.0f, 3.0f, (float)dx} : stream_ld(dwb + (unsigned)((dy * 3 + dx) * QUADS) * 16u);
#pragma unroll
            for (int k = 0; k < MT + 2; ++k) xv[b][k] = STREAM_ABL == 3 ? f32x4{(float)k, (float)lane, 1.0f, 2.0f} : stream_ld(xb + (unsigned)k * PX);
        };
        auto mult = [&](auto b_, auto first_) DCSCN_INL {
            constexpr int b = decltype(b_)::value;
            constexpr bool first = decltype(first_)::value;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (first && dx == 0) d[m] = dw[b][dx] * xv[b][m + dx];
                    else if (STREAM_ABL == 2) { if (dx == 1) d[m] += xv[b][m + dx] + dw[b][dx]; }
                    else d[m] += dw[b][dx] * xv[b][m + dx];
                }
            asm volatile("" ::: "memory");
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if constexpr (F16 && MT > 1) {
            // single-buffered rows: 32 registers fewer than the prefetch of row dy + 1 (this kernel sits at 128); beside f16 MFMAs the SIMD's
            // other waves fill the LDS latency
            fetch(I0{}, I0{});
            mult(I0{}, std::true_type{});
            fetch(I1{}, I0{});
            mult(I0{}, std::false_type{});
            fetch(I2{}, I0{});
            mult(I0{}, std::false_type{});
        } else {
        fetch(I0{}, I0{});
        fetch(I1{}, I1{});
        mult(I0{}, std::true_type{});
        fetch(I2{}, I0{});
        mult(I1{}, std::false_type{});
        mult(I0{}, std::false_type{});
        }
        if (QL < 3) {
            // keep the packed depthwise: without this the compiler sinks pick()'s lane-dependent selects into the 9 x (MT + 2)
            // window values and runs the taps unpacked
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(d[m]));
        }
        // (lanes of a missing quad hold the depthwise of a real one: finite, times zero filter rows)
        if constexpr (F16) {
            const float m1 = opaque_minus_one();
            if constexpr ((ch & 1) == 0 && ch != CH - 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) dpair[m] = d[m];
            } else if constexpr ((ch & 1) == 1) {
                constexpr bool first = ch == 1;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    h8 bh, bl;
                    split8(dpair[m], d[m], m1, bh, bl);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const h8 ah = __builtin_bit_cast(h8, stream_ld(lds0 + wpo + (unsigned)(((ch - 1) * NT + 2 * n) * 64 + lane) * 16u));
                        const h8 al = __builtin_bit_cast(h8, stream_ld(lds0 + wpo + (unsigned)(((ch - 1) * NT + 2 * n + 1) * 64 + lane) * 16u));
                        f32x4 c0 = first ? stream_ld(init_addr + (unsigned)n * 64u) : acc[m][n];
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c0, 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c0, 0, 0, 0);
                    }
                }
            } else {                                          // an odd last chunk: K = 16, the lane's slot holds [hi 0-3 | lo 0-3]
                constexpr bool first = ch == 0;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    h4 bh, bl;
                    split4(d[m], m1, bh, bl);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const f32x4 wv = stream_ld(lds0 + wpo + (unsigned)((ch * NT + n) * 64 + lane) * 16u);
                        const u32x4 wu = __builtin_bit_cast(u32x4, wv);
                        const h4 ah = __builtin_bit_cast(h4, u32x2{wu.x, wu.y}), al = __builtin_bit_cast(h4, u32x2{wu.z, wu.w});
                        f32x4 c0 = first ? stream_ld(init_addr + (unsigned)n * 64u) : acc[m][n];
                        c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c0, 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c0, 0, 0, 0);
                    }
                }
            }
        } else {
#if STREAM_ABL == 16
        if constexpr ((ch & 1) == 0 && ch != CH - 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m) dkeep[m] = d[m];
        } else {
            const float m1 = opaque_minus_one();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                h8 bh, bl;
                if constexpr ((ch & 1) != 0) split8(dkeep[m], d[m], m1, bh, bl); else split8(d[m], d[m], m1, bh, bl);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const u32x4 wu = __builtin_bit_cast(u32x4, wp[n]) & 0x3fff3fffu;      // a finite f16 pattern (timing only)
                    const h8 ah = __builtin_bit_cast(h8, wu);
                    f32x4 c0 = ch <= 1 ? init[n] : acc[m][n];
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c0, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c0, 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c0, 0, 0, 0);
                }
            }
        }
#else
#pragma unroll
        for (int s = 0; s < K::STEPS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float bv = K::pick(d[m], s, q);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (STREAM_ABL == 1 || STREAM_ABL == 15) { if (s == 0) acc[m][n] = (ch == 0 ? init[n] : acc[m][n]) + d[m] * wp[n]; }
                    else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[n][s], bv, ch == 0 && s == 0 ? init[n] : acc[m][n], 0, 0, 0);
                }
            }
#endif
        }
    });
}

// ---- CNN2 .. CNNL, B2: depthwise 3x3 from the predecessor's ring -> pointwise GEMM -> bias, PReLU -----------------
// QUADS = channel quads of the input, NT = 16-channel tiles of the output (compile time: no branches between the MFMAs)
template <int QUADS, int NT, bool F16, bool GATE>
__device__ __forceinline__ void stream_conv_role(const StreamArgs& a, const StreamConv& c, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    const float zflag = opaque_zero();
    constexpr unsigned in_px = (unsigned)(QUADS | 1) * 16u, in_row = (unsigned)kStreamRowPx * in_px;
    StreamCursor cur;
    for (int t = 0; t < T; ++t) {
        STREAM_STAMP(a, t, 0);
        const int g = t - c.lag;
        const bool live = g >= 0 && g < rows;
        f32x4 acc[kStreamMT][NT];
        bool zero_row = true;               // separator / outside the image: the ring gets zeros
        if (live && STREAM_ABL != 5 && STREAM_ABL != 8) {
            const StreamRow ri = stream_row(a, j0, cur, g);
            zero_row = ri.zero;
            if (!ri.zero) {
                unsigned rowb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rowb[dy] = lds0 + c.in.off + (unsigned)((g + 2 * c.in.slots - 1 + dy) % c.in.slots) * in_row + (unsigned)(3 * j) * in_px;   // rows g-1, g, g+1
                f32x4 bs[NT];
                if constexpr (!F16) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) bs[n] = stream_ld(lds0 + c.ba + (unsigned)(n * 4 + q) * 16u);
                }
                stream_dw_pw<QUADS, NT, kStreamMT, F16>(acc, bs, lds0, rowb, c.dww, c.wp, q, lane, lds0 + c.ba + (unsigned)q * 16u);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 al = stream_ld(lds0 + c.ba + 128u + (unsigned)(n * 4 + q) * 16u);
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) acc[m][n] = stream_prelu(F16 ? acc[m][n] * c.inv : acc[m][n], al);
                }
                if (ri.sx < 0 || ri.sx + kStreamPX > a.W) {          // the strip sticks out of the image: SAME padding is zero
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) {
                        const int cx = ri.sx + 3 * j + m;
                        if (cx < 0 || cx >= a.W) {
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc[m][n] = kStreamZero;
                        }
                    }
                }
                if (c.to_global && ri.store && (!GATE || a.redo[1 + ri.img] != 0)) {     // (GATE = the float32 plan: flagged images only)
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) {
                        const int cx = ri.sx + 3 * j + m;
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            if (cx >= ri.ux0 && cx < ri.ux1 && n * 4 + q < c.out.quads) {
                                if (F16) stream_flag(a.redo, ri.img, acc[m][n], zflag);
                                *reinterpret_cast<f32x4*>(a.out + (((size_t)ri.img * a.H + ri.r) * a.W + cx) * a.out_stride + (n * 4 + q) * 4) = acc[m][n];
                            }
                    }
                }
            }
        }
        STREAM_STAMP(a, t, 1);
        stream_barrier();
        STREAM_STAMP(a, t, 2);
        if (live && !c.to_global) {
            const unsigned wb = lds0 + c.out.off + (((unsigned)(g % 3) * kStreamRowPx + 3 * j + 1) * c.out.units + q) * 16u;
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    if (n * 4 + q < c.out.quads) {
                        if (zero_row) stream_st(wb + (unsigned)(m * c.out.units + n * 4) * 16u, kStreamZero);       // wave uniform
                        else stream_st(wb + (unsigned)(m * c.out.units + n * 4) * 16u, acc[m][n]);
                    }
        }
        STREAM_STAMP(a, t, 3);
        stream_barrier();
    }
}

// ---- A1 || B1: one (feature layer, row) contribution per step and wave --------------------------------------------
template <bool F16, bool GATE>
__device__ __forceinline__ void stream_nin_role(const StreamArgs& a, int w, unsigned lds0, int j0, int rows, int T, int) {
    const int L = a.L;
    const float zflag = opaque_zero();
    f32x4 acc[2][kStreamMT][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int m = 0; m < kStreamMT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[p][m][n] = kStreamZero;
    StreamCursor cur[2];       // the wave alternates between an even and an odd row; each advances by 2L rows at a time

    // the step loop is unrolled by two so that the row parity (which accumulator set) is a compile-time constant: a run-time
    // choice costs ~45 register moves per step, and VALU time is matrix time here.  l advances by one every second step.
    int l = ((0 - w) % L + L) % L;
    if (l == 0) l = L;
    auto step = [&](auto p_, int t) DCSCN_INL {
        constexpr int p = decltype(p_)::value;
        STREAM_STAMP(a, t, 0);
        // (opaque per step: otherwise every lane-derived address of every branch is hoisted out of the loop and spilled)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int j = lane & 15, q = lane >> 4;
        // row g = t + 1 - 2l receives layer l's contribution at step t; (g >> 1) mod L == w picks this wave's l
        const int g = t + 1 - 2 * l;                   // parity of g = parity of t + 1 = p
        const bool live = g >= 0 && g < rows;
        const bool last = l == L;
        if (live && STREAM_ABL != 6 && STREAM_ABL != 8) {
            const StreamNinSrc& s = a.nin[l - 1];
            {
                const StreamRow ri = stream_row(a, j0, cur[p], g);
                if (!ri.zero) {
                    const unsigned rowb = lds0 + s.ring.off + ((unsigned)(g % 3) * kStreamRowPx + 3 * j + 1) * (unsigned)s.ring.units * 16u;
#if STREAM_ABL == 16
                    f32x4 xkeep[kStreamMT];
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) xkeep[m] = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
#endif
                    if constexpr (F16) {
                        // chunk pairs on v_mfma_f32_16x16x32_f16, an odd last chunk on the K = 16 form (stream_dw_pw, F16); a short last
                        // chunk's missing quads read a valid one against zero filter rows
                        const float m1 = opaque_minus_one();
                        const int lql = s.last_ql;
                        const int lcq = lql >= 3 ? min(q, lql - 1) : lql == 2 ? (q & 1) : 0;
#pragma unroll 1
                        for (int cp = 0; 2 * cp + 1 < s.chunks; ++cp) {
                            const unsigned q0 = (unsigned)(8 * cp + q) * 16u, q1 = (unsigned)(8 * cp + 4 + (2 * cp + 1 == s.chunks - 1 ? lcq : q)) * 16u;
                            const unsigned wb = lds0 + s.w + (unsigned)(cp * 4 * 64 + lane) * 16u;
#pragma unroll
                            for (int m = 0; m < kStreamMT; ++m) {
                                const f32x4 x0 = stream_ld(rowb + (unsigned)(m * s.ring.units) * 16u + q0), x1 = stream_ld(rowb + (unsigned)(m * s.ring.units) * 16u + q1);
                                h8 bh, bl;
                                split8(x0, x1, m1, bh, bl);
                                // (the fragments of one tile at a time, re-read per pixel tile: LDS reads are free beside f16 MFMAs, registers are not)
#pragma unroll
                                for (int n = 0; n < 2; ++n) {
                                    const h8 ah = __builtin_bit_cast(h8, stream_ld(wb + (unsigned)n * 2048u)), al = __builtin_bit_cast(h8, stream_ld(wb + (unsigned)n * 2048u + 1024u));
                                    acc[p][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[p][m][n], 0, 0, 0);
                                    acc[p][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[p][m][n], 0, 0, 0);
                                    acc[p][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[p][m][n], 0, 0, 0);
                                    asm volatile("" ::: "memory");
                                }
                            }
                        }
                        if (s.chunks & 1) {
                            const int ch = s.chunks - 1;
                            const unsigned qo = (unsigned)(4 * ch + lcq) * 16u;
                            const u32x4 w0 = __builtin_bit_cast(u32x4, stream_ld(lds0 + s.w + (unsigned)((ch * 2 + 0) * 64 + lane) * 16u));
                            const u32x4 w1 = __builtin_bit_cast(u32x4, stream_ld(lds0 + s.w + (unsigned)((ch * 2 + 1) * 64 + lane) * 16u));
                            const h4 a0h = __builtin_bit_cast(h4, u32x2{w0.x, w0.y}), a0l = __builtin_bit_cast(h4, u32x2{w0.z, w0.w});
                            const h4 a1h = __builtin_bit_cast(h4, u32x2{w1.x, w1.y}), a1l = __builtin_bit_cast(h4, u32x2{w1.z, w1.w});
#pragma unroll
                            for (int m = 0; m < kStreamMT; ++m) {
                                h4 bh, bl;
                                split4(stream_ld(rowb + (unsigned)(m * s.ring.units) * 16u + qo), m1, bh, bl);
                                f32x4 c0 = acc[p][m][0], c1 = acc[p][m][1];
                                c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a0l, bh, c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a1l, bh, c1, 0, 0, 0);
                                c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a0h, bl, c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a1h, bl, c1, 0, 0, 0);
                                acc[p][m][0] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0h, bh, c0, 0, 0, 0);
                                acc[p][m][1] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1h, bh, c1, 0, 0, 0);
                            }
                        }
                    } else
#pragma unroll 1
                    for (int ch = 0; ch < s.chunks; ++ch) {
                        // StreamChunk<ql> with a run-time ql (one code path: this role is short of registers, not of VALU slots)
                        const int ql = ch == s.chunks - 1 ? s.last_ql : 4;
                        const int steps = ql >= 3 ? 4 : ql;
                        const int cq = ql >= 3 ? min(q, ql - 1) : ql == 2 ? (q & 1) : 0;
                        const int eoff = ql == 2 ? 2 * (q >> 1) : ql == 1 ? q : 0;
                        const unsigned qoff = (unsigned)(ch * 4 + cq) * 16u;
                        const f32x4 w0 = stream_ld(lds0 + s.w + (unsigned)((ch * 2 + 0) * 64 + lane) * 16u);
                        const f32x4 w1 = stream_ld(lds0 + s.w + (unsigned)((ch * 2 + 1) * 64 + lane) * 16u);
                        f32x4 xv[kStreamMT];
#pragma unroll
                        for (int m = 0; m < kStreamMT; ++m) xv[m] = stream_ld(rowb + (unsigned)(m * s.ring.units) * 16u + qoff);
                        if (ql <= 2) {
                            // element eoff + k feeds k-step k: rotate the float4 by eoff
#pragma unroll
                            for (int m = 0; m < kStreamMT; ++m) {
                                const f32x4 x = xv[m];
                                const float e0 = eoff == 0 ? x.x : eoff == 1 ? x.y : eoff == 2 ? x.z : x.w;
                                const float e1 = eoff == 0 ? x.y : x.w;           // only ql == 2 has a second k-step (eoff 0 or 2)
                                xv[m].x = e0;
                                xv[m].y = e1;
                            }
                        }
#if STREAM_ABL == 16
                        if ((ch & 1) == 0 && ch != s.chunks - 1) {
#pragma unroll
                            for (int m = 0; m < kStreamMT; ++m) xkeep[m] = xv[m];
                        } else {
                            const float m1 = opaque_minus_one();
                            const h8 a0 = __builtin_bit_cast(h8, __builtin_bit_cast(u32x4, w0) & 0x3fff3fffu);
                            const h8 a1 = __builtin_bit_cast(h8, __builtin_bit_cast(u32x4, w1) & 0x3fff3fffu);
#pragma unroll
                            for (int m = 0; m < kStreamMT; ++m) {
                                h8 bh, bl;
                                split8(xkeep[m], xv[m], m1, bh, bl);
                                f32x4 c0 = acc[p][m][0], c1 = acc[p][m][1];
                                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bh, c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh, c1, 0, 0, 0);
                                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bl, c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl, c1, 0, 0, 0);
                                acc[p][m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bh, c0, 0, 0, 0);
                                acc[p][m][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh, c1, 0, 0, 0);
                            }
                        }
#else
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < steps) {
#pragma unroll
                                for (int m = 0; m < kStreamMT; ++m) {
                                    if (STREAM_ABL == 15) { if (k == 0) { acc[p][m][0] += w0 * xv[m]; acc[p][m][1] += w1 * xv[m]; } }
                                    else {
                                    acc[p][m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[k], xv[m][k], acc[p][m][0], 0, 0, 0);
                                    acc[p][m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[k], xv[m][k], acc[p][m][1], 0, 0, 0);
                                    }
                                }
                            }
#endif
                    }
                }
                if (last) {
                    const bool gate = ri.store && (!GATE || a.redo[1 + ri.img] != 0);     // (GATE = the float32 plan: flagged images only)
                    // bias, PReLU; A1 -> Concat2 (global), the B1 quads (tile 0, nb <= 16) -> the B1 ring, which has a FOURTH slot so
                    // that the row can be written while B2 reads the three before it (no registers held across the barrier)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 bs = stream_ld(lds0 + a.nin_ba + (unsigned)(n * 4 + q) * 16u);
                        const f32x4 al = stream_ld(lds0 + a.nin_ba + 128u + (unsigned)(n * 4 + q) * 16u);
                        const int quad = n * 4 + q;
#pragma unroll
                        for (int m = 0; m < kStreamMT; ++m) {
                            const int cx = ri.sx + 3 * j + m;
                            const bool ok = !ri.zero && cx >= 0 && cx < a.W;
                            const f32x4 r = stream_prelu(F16 ? acc[p][m][n] * a.nin_inv + bs : acc[p][m][n] + bs, al);
                            const f32x4 v = ok ? r : kStreamZero;
                            acc[p][m][n] = kStreamZero;
                            if (n == 0 && q < a.nb_quads)
                                stream_st(lds0 + a.b1.off + (((unsigned)(g & 3) * kStreamRowPx + 3 * j + m + 1) * a.b1.units + q) * 16u, v);
                            if (gate && cx >= ri.ux0 && cx < ri.ux1 && quad >= a.nb_quads && quad * 4 < a.out_stride) {
                                if (F16) stream_flag(a.redo, ri.img, v, zflag);
                                *reinterpret_cast<f32x4*>(a.out + (((size_t)ri.img * a.H + ri.r) * a.W + cx) * a.out_stride + quad * 4) = v;
                            }
                        }
                    }
                }
            }
        }
        STREAM_STAMP(a, t, 1);
        stream_barrier();
        STREAM_STAMP(a, t, 2);
        STREAM_STAMP(a, t, 3);
        stream_barrier();
    };
    for (int t = 0; t < T; t += 2) {
        step(std::integral_constant<int, 1>{}, t);
        if (t + 1 < T) {
            l = l == L ? 1 : l + 1;
            step(std::integral_constant<int, 0>{}, t + 1);
        }
    }
}

// F16 = the pointwise GEMMs and A1 || B1 on the f16 matrix pipe (stream_dw_pw); false = v_mfma_f32_16x16x4_f32 (split16 = 0).
// GATE = the float32 plan of flagged images (F16 = false): leaves at once unless the pass flagged an image, stores only flagged images --
// its own instantiation, so that the kernels of every pass carry none of it (they sit at their 128-register cap)
template <bool F16, bool GATE = false>
__global__ __launch_bounds__(1024) void feat_stream(const StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (GATE && a.redo[0] == 0) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    {
        f32x4* s4 = reinterpret_cast<f32x4*>(smem);
        for (int i = tid; i < a.ring_bytes / 16; i += blockDim.x) s4[i] = kStreamZero;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.blob + a.ldsw_src);
        for (int i = tid; i < a.ldsw_bytes / 16; i += blockDim.x) s4[a.ring_bytes / 16 + i] = src[i];
        __syncthreads();
    }
    const int j0 = blockIdx.x * a.jobs_per_wg;
    const int j1 = min(a.n_jobs, j0 + a.jobs_per_wg);
    const int rows = (j1 - j0) * (a.rows_c + 1);
    const int T = rows + a.total_lag;
    const int role = a.role[wave];
    if (role == 0) stream_first_role(a, lds0, j0, rows, T, lane);
    else if (role < 16) {
        const StreamConv& c = a.conv[role - 1];
        // the instantiated (input quads, output tiles) pairs -- api.hip: stream_conv_supported
        const bool nt2 = c.out.quads > 4;
        switch (c.in.quads) {
            case 1: stream_conv_role<1, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 2: stream_conv_role<2, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 3: stream_conv_role<3, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 4: stream_conv_role<4, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 5: stream_conv_role<5, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 6: stream_conv_role<6, 2, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            case 7: stream_conv_role<7, 2, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
            default: if (nt2) stream_conv_role<8, 2, F16, GATE>(a, c, lds0, j0, rows, T, lane); else stream_conv_role<8, 1, F16, GATE>(a, c, lds0, j0, rows, T, lane); break;
        }
    } else stream_nin_role<F16, GATE>(a, role - 16, lds0, j0, rows, T, lane);
}

}  // namespace dcscn
