// feat3_stream: the feature extractor of the NON-separable narrow nets (c-DCSCN: CNN1 .. CNNL, 3x3 SAME convs + bias + PReLU of at most 32
// channels, DCSCN.py:240-256 with tf_graph.py:104-153) as ONE row-streamed launch -- VERDICT r03 item 4 / r04 item 3.
//
// Layer by layer these nets are bound by HBM round trips and launch overheads (seven launches of 0.08 - 0.21 ms for ~0.1 ms of matrix
// work); here a workgroup owns a 48-pixel column strip and walks down its rows like feat_stream.hpp (same jobs, strips, row blocks,
// separator rows), every layer keeps a four-slot row ring of its output in LDS (three rows are read while the fourth is written: one barrier
// per step), and every layer's rows also go to
// global memory ONCE, for the 1x1 GEMM A1 || B1 that follows (as P16 tensors or float32, whatever plan_p16 gave the tensor).
//
//   * rings hold P16 units (p16.hpp): per pixel and channel octet [hi 8 halfs | lo 8 halfs], written by the producing wave's epilogue
//     (p16_unit: split + v_permlane16_swap) -- a consumer's B operand is two ds_read_b128, no VALU.
//   * a conv is a K = 9 * Cin implicit GEMM on v_mfma_f32_16x16x32_f16 with the (tap, octet) pairs packed four to an instruction (conv3_h's
//     packed tail, for every chunk): lane group q of step s multiplies pair 4 s + q; ceil(9 * octets / 4) = 3 / 5 / 7 / 9 steps.
//     Three products per accumulator (split16.hpp), filters f16 (hi, lo) scaled by 2^e, bias * 2^e as the first C operand.
//   * a wave = a conv (one or two 16-channel output tiles); its filter fragments -- at most 9 steps x 2 tiles x (hi, lo) = 144 registers of the
//     256 an 8-wave workgroup leaves each wave -- live in REGISTERS for the whole launch (the dense filters of the net are 140 KB as fragments: they do not fit beside the rings, and nothing else needs them).
//     Pixel tile m of lane column j is pixel 3 j + m, as in feat_stream.
//   * wave -> SIMD placement balances the MFMA counts (pack.hip: pack_feat3_stream); CNN1 (one input channel) is VALU work on its own wave.
//
// The float32 plan of a flagged image, and split16 = 0, run the layers one by one on their own kernels (exec.hip: Op::fused).
#pragma once
#include "feat_stream.hpp"
#include "p16.hpp"

#ifndef S3_ABL
#define S3_ABL 0           // tools/s3_abl.sh: timing-only builds (results wrong by design): 1 no MFMAs, 2 no ring reads, 4 no global stores, 8 no CNN1 arithmetic,
                           // 16 no epilogue arithmetic (PReLU / split / swap), 32 no ring stores, 64 no workgroup barrier, 128 no loads of the input image
#endif

namespace dcscn {

// (ring stores and the step's barrier through wrappers: the timing-only builds of tools/s3_abl.sh take them out)
__device__ __forceinline__ void s3_st(unsigned addr, f32x4 v) {
    if constexpr ((S3_ABL & 32) != 0) asm volatile("" ::"v"(v), "v"(addr));
    else stream_st(addr, v);
}
__device__ __forceinline__ void s3_barrier() {
    if constexpr ((S3_ABL & 64) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else stream_barrier();
}

__device__ __forceinline__ StreamArgs s3_geometry(const Stream3Args& a) {
    StreamArgs g{};                                            // stream_row only looks at the job geometry
    g.H = a.H; g.W = a.W;
    g.n_strips = a.n_strips; g.useful_w = a.useful_w; g.halo = a.halo;
    g.n_blocks = a.n_blocks; g.useful_h = a.useful_h; g.rows_c = a.rows_c;
    return g;
}

// A layer's row in its global tensor.  At most 32 channels = one P16 chunk, so the record size is wave uniform: `base` = the record of the
// strip's column 0 in this image row (one scalar 64-bit computation per row), a lane adds (3 j + m) * rec + its unit's 16 n' bytes -- no
// per-store address arithmetic (r05's first build spent ~1.5 k of 6.6 k cycles per step on 64-bit multiplies in front of its stores).
struct S3RowOut {
    char* base;        // P16: record of (img, r, sx); float32: &ptr[pixel (img, r, sx)][0]
    unsigned rec;      // bytes per pixel
    bool p16;
};
__device__ __forceinline__ S3RowOut s3_row_out(const S3Out& o, const StreamRow& ri, int H, int W) {
    const long long pix = ((long long)ri.img * H + ri.r) * W + ri.sx;
    S3RowOut r;
    r.p16 = o.p16.base != nullptr;
    r.rec = r.p16 ? (unsigned)p16_rec_bytes(o.p16.octs, 0) : (unsigned)o.stride * 4u;
    r.base = r.p16 ? o.p16.base + 128 + pix * (long long)r.rec : reinterpret_cast<char*>(o.ptr) + pix * (long long)r.rec;
    return r;
}
// tile n, lane group q: P16 unit 4 n + q of the record (octet 2 n + (q >> 1), part q & 1); float32: channels 16 n + 4 q ..  A destination stores the
// writer's conv channels [lo, hi) only (B2 writes octet 0 of Concat2, the A1 || B1 waves octets 1 .. 3); none: both pointers null
__device__ __forceinline__ void s3_store_global(const S3Out& o, const S3RowOut& ro, int col, int n, int q, const u32x4 unit, const f32x4 v) {
    if constexpr ((S3_ABL & 4) != 0) return;
    const unsigned off = (unsigned)col * ro.rec + (unsigned)(n * 64 + q * 16);
    if (ro.p16) {
        const int c8 = 8 * (2 * n + (q >> 1));
        if (c8 >= o.lo && c8 < o.hi) *reinterpret_cast<u32x4*>(ro.base + off) = unit;
    } else if (o.ptr != nullptr) {
        const int c4 = 16 * n + 4 * q;
        if (c4 >= o.lo && c4 < o.hi) *reinterpret_cast<f32x4*>(ro.base + off) = v;
    }
}

// ---- CNN1: Y -> 3x3 conv to C1 <= 32 channels, bias, PReLU; VALU only ---------------------------------------------------
__device__ __forceinline__ void s3_first_role(const Stream3Args& a, const StreamArgs& geo, unsigned lds0, int j0, int rows, int T, int lane) {
    const int j = lane & 15, q = lane >> 4;
    f32x4 w9[2][9], bs[2], al[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[n][t] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + t * 32 + n * 16 + 4 * q);
        bs[n] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + 288 + n * 16 + 4 * q);
        al[n] = *reinterpret_cast<const f32x4*>(a.blob + a.first_w + 320 + n * 16 + 4 * q);
    }
    const float m1 = opaque_minus_one();
    const h2 zero2 = p16_opaque_zero2();
    const unsigned out_px = (unsigned)a.first_out.px, out_row = (unsigned)kStreamRowPx * out_px;
    // rows g-1 .. g+3 of the input in a rotating five-row window (slot = row mod 5, compile-time: the step loop is unrolled by five): the
    // row loaded at step t is first used at step t + 2 -- with one step of distance (r05's first build) every step waited for a global load
    float xw[5][5];
    StreamCursor lc, cc;
    auto load_row = [&](int gs, float (&dst)[5]) DCSCN_INL {
        const bool in = gs >= 0 && gs < rows;
        const StreamRow ri = stream_row(geo, j0, lc, in ? gs : 0);
        const bool live = in && !ri.zero;
        const float* row = a.x + ((size_t)ri.img * a.H + (live ? ri.r : 0)) * a.W;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int cx = ri.sx + 3 * j + k - 1;
            if constexpr ((S3_ABL & 128) != 0) dst[k] = (float)(cx & 7); else
            dst[k] = live && cx >= 0 && cx < a.W ? row[cx] : 0.0f;
        }
    };
#pragma unroll
    for (int k = 0; k < 5; ++k) xw[4][k] = 0.0f;              // row -1
    load_row(0, xw[0]);
    load_row(1, xw[1]);
    load_row(2, xw[2]);
#ifdef S3_DBG
    long long dbg_c = 0, dbg_b = 0, dbg_t = __builtin_readcyclecounter();
#endif
    auto step = [&](auto p_, int t) DCSCN_INL {
        constexpr int p = decltype(p_)::value;                // t mod 5
        const int g = t;
        load_row(g + 3, xw[(p + 3) % 5]);
        const bool live = g < rows;
        if (live) {
            const StreamRow ri = stream_row(geo, j0, cc, g);
            const bool to_global = a.out[0].ptr != nullptr || a.out[0].p16.base != nullptr;    // (wave uniform)
            S3RowOut ro{};
            if (to_global) ro = s3_row_out(a.out[0], ri, a.H, a.W);
            float chk = 0.0f;
            const unsigned wb = lds0 + a.first_out.off + (unsigned)(g & 3) * out_row + (unsigned)(3 * j + 1) * out_px + (unsigned)q * 16u;
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                const int cx = ri.sx + 3 * j + m;
                const bool ok = !ri.zero && cx >= 0 && cx < a.W;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 s = kStreamZero;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) if (!(S3_ABL & 8) || (dy == 1 && dx == 1)) s += w9[n][dy * 3 + dx] * xw[(p + 4 + dy) % 5][m + dx];
                    f32x4 v = stream_prelu(bs[n] + s, al[n]);
                    v = ok ? v : kStreamZero;
                    const u32x4 unit = p16_unit(v, m1, chk, zero2);
                    if (to_global && ri.store && cx >= ri.ux0 && cx < ri.ux1) s3_store_global(a.out[0], ro, 3 * j + m, n, q, unit, v);
                    // (four ring slots: row g goes to its slot while CNN2 reads rows g-3 .. g-1; ONE barrier per step)
                    if (2 * n + (q >> 1) < a.first_out.octs) s3_st(wb + (unsigned)m * out_px + (unsigned)n * 64u, __builtin_bit_cast(f32x4, unit));
                }
            }
            if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + ri.img] = 1; }
        }
#ifdef S3_DBG
        const long long tb = __builtin_readcyclecounter();
        dbg_c += tb - dbg_t;
#endif
        s3_barrier();
#ifdef S3_DBG
        dbg_t = __builtin_readcyclecounter();
        dbg_b += dbg_t - tb;
#endif
    };
    for (int t = 0; t < T; t += 5) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) step(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < T) step(std::integral_constant<int, 3>{}, t + 3);
        if (t + 4 < T) step(std::integral_constant<int, 4>{}, t + 4);
    }
#ifdef S3_DBG
    if (a.dbg && blockIdx.x == 0 && lane == 0) { long long* d = a.dbg + (threadIdx.x >> 6) * 4; d[0] = dbg_c; d[1] = dbg_b; d[2] = T; }
#endif
}

// ---- CNN2 .. CNNL (and B2): a 3x3 conv (NT 16-channel output tiles) from the predecessor's ring ---------------------------------
// OCTS = channel octets of the input ring (compile time: the step count, every LDS offset an immediate or one register per step).
// One wave per conv: the B operands of a step are read ONCE for all its output tiles -- the kernel is bound by LDS read bandwidth (a
// ds_read_b128 is 1 KB: ~250 of them per row step and workgroup), not by the MFMAs; the first build (one wave per output tile) read
// them twice.  Split in a load part (filter fragments -> registers, for the whole launch) and a step part so that one wave can run
// several light convs (s3_trio_role).
template <int OCTS, int NT>
struct S3ConvRegs {
    static constexpr int STEPS = (9 * OCTS + 3) / 4;
    h8 fh[STEPS][NT], fl[STEPS][NT];                           // [step][tile][hi | lo][64 lanes][8 halfs] in the blob
    f32x4 bs[NT], am1[NT];                                     // bias * 2^e, slope - 1
    unsigned soff[OCTS == 1 ? STEPS : 1];
    unsigned dyp;
    StreamCursor cur;
};
template <int OCTS, int NT>
__device__ __forceinline__ void s3_conv_load(const Stream3Args& a, int ci, int lane, S3ConvRegs<OCTS, NT>& r) {
    const S3Conv& c = a.conv[ci];
    constexpr int STEPS = S3ConvRegs<OCTS, NT>::STEPS;
    constexpr unsigned IN_PX = (unsigned)(2 * OCTS + 1) * 16u;
    const int q = lane >> 4;
    const char* wsrc = reinterpret_cast<const char*>(a.blob + c.w_off);
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            r.fh[s][n] = *reinterpret_cast<const h8*>(wsrc + ((size_t)((s * NT + n) * 2 + 0) * 64 + lane) * 16);
            r.fl[s][n] = *reinterpret_cast<const h8*>(wsrc + ((size_t)((s * NT + n) * 2 + 1) * 64 + lane) * 16);
        }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        r.bs[n] = *reinterpret_cast<const f32x4*>(a.blob + c.ba_off + n * 16 + 4 * q);           // bias * 2^e
        r.am1[n] = *reinterpret_cast<const f32x4*>(a.blob + c.ba_off + 32 + n * 16 + 4 * q);     // slope - 1
    }
    // step s, lane group q: pair p = 4 s + q = (tap, octet) with tap = p / OCTS.  For OCTS >= 2 the four lane groups of a step see at most two
    // taps -- tap0 = 4 s / OCTS below the lane-group threshold `thr`, tap0 + 1 from it on -- so row, column and octet offset are compile-time
    // constants selected by one compare; pairs past the last (tap 9) read valid units of tap 8 against zero filters.  OCTS = 1: four taps per
    // step, one offset register per step.
    r.dyp = 0;
    if constexpr (OCTS == 1) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            int tap = 4 * s + q;
            tap = tap > 8 ? 8 : tap;
            const int dy = tap / 3, dx = tap - 3 * dy;
            r.soff[s] = (unsigned)dx * IN_PX;
            r.dyp |= (unsigned)dy << (2 * s);
        }
    } else r.soff[0] = (unsigned)q * 32u;
}
// one step of the conv: stream row g = t - lag from the input ring's rows g - 1 .. g + 1 -> its output ring (four slots: row g goes to its slot
// while the next layer reads rows g-3 .. g-1 -- ONE barrier per step, issued by the caller) and its global tensor
// (Issuing the epilogue of row g - 1 between the MFMA groups of row g -- a software pipeline with the layers' lags spaced by three -- was
// built and measured in r05: 0.83 ms against 0.66, the heaviest wave 7.8 k cycles per step instead of 6.0 k: the ring stores and the B-operand
// reads share one in-order LGKM counter, and the waits for the one also wait for the other.)
template <int OCTS, int NT>
__device__ __forceinline__ void s3_conv_step(const Stream3Args& a, const StreamArgs& geo, int ci, unsigned lds0, int j0, int rows, int t, int lane,
                                             S3ConvRegs<OCTS, NT>& r, float m1, h2 zero2) {
    const S3Conv& c = a.conv[ci];
    const S3Out& og = a.out[ci + 1];
    constexpr int STEPS = S3ConvRegs<OCTS, NT>::STEPS;
    constexpr unsigned IN_PX = (unsigned)(2 * OCTS + 1) * 16u, IN_ROW = (unsigned)kStreamRowPx * IN_PX;
    const int j = lane & 15, q = lane >> 4;
    const unsigned out_px = (unsigned)c.out.px, out_row = (unsigned)kStreamRowPx * out_px;
    const int g = t - c.lag;
    const bool live = g >= 0 && g < rows;
    u32x4 unit[kStreamMT][NT];
    bool zero_row = true;
    if (live) {
        const StreamRow ri = stream_row(geo, j0, r.cur, g);
        zero_row = ri.zero;
        if (!ri.zero) {
            unsigned rb[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rb[dy] = lds0 + c.in.off + (unsigned)((g + 3 + dy) & 3) * IN_ROW + (unsigned)(3 * j) * IN_PX;   // rows g-1, g, g+1
            f32x4 acc[kStreamMT][NT];
            static_for<0, STEPS>([&](auto s_) DCSCN_INL {
                constexpr int s = decltype(s_)::value;
                unsigned base;
                if constexpr (OCTS == 1) {
                    const unsigned dy = (r.dyp >> (2 * s)) & 3u;
                    base = (dy == 0 ? rb[0] : dy == 1 ? rb[1] : rb[2]) + r.soff[s];
                } else {
                    constexpr int p0 = 4 * s, tap0 = p0 / OCTS, thr = OCTS * (tap0 + 1) - p0;       // lane groups q >= thr are on tap0 + 1
                    constexpr int ta = tap0 > 8 ? 8 : tap0, tb = tap0 + 1 > 8 ? 8 : tap0 + 1;
                    constexpr int offa = (ta % 3) * (int)IN_PX + (p0 - OCTS * tap0) * 32, offb = (tb % 3) * (int)IN_PX + (p0 - OCTS * (tap0 + 1)) * 32;
                    if constexpr (thr > 3) base = rb[ta / 3] + (unsigned)offa + r.soff[0];
                    else base = (q >= thr ? rb[tb / 3] + (unsigned)offb : rb[ta / 3] + (unsigned)offa) + r.soff[0];
                }
                // the three pixel tiles' products interleaved: an accumulator is touched every third MFMA (back to back they wait for each other)
                h8 xh[kStreamMT], xl[kStreamMT];
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) {
                    if constexpr ((S3_ABL & 2) != 0) { u32x4 z = {0x3c003c00u, 0x3c003c00u, base, 0x3c003c00u}; asm volatile("" : "+v"(z)); xh[m] = xl[m] = __builtin_bit_cast(h8, z); } else {
                    xh[m] = __builtin_bit_cast(h8, stream_ld(base + (unsigned)m * IN_PX));
                    xl[m] = __builtin_bit_cast(h8, stream_ld(base + (unsigned)m * IN_PX + 16u));
                    }
                }
                if constexpr ((S3_ABL & 1) != 0) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m = 0; m < kStreamMT; ++m) { if (s == 0) acc[m][n] = r.bs[n]; f32x4 tt = acc[m][n]; const h8 fa = r.fh[s][n], fb = r.fl[s][n], xa = xh[m], xb = xl[m]; asm volatile("" : "+v"(tt) : "v"(xa), "v"(xb), "v"(fa), "v"(fb)); acc[m][n] = tt; }
                } else
#pragma unroll
                for (int n = 0; n < NT; ++n) {
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.fl[s][n], xh[m], s == 0 ? r.bs[n] : acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.fh[s][n], xl[m], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < kStreamMT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.fh[s][n], xh[m], acc[m][n], 0, 0, 0);
                }
            });
            float chk = 0.0f;
            const bool to_global = og.ptr != nullptr || og.p16.base != nullptr;      // (wave uniform; nin.on: only B2 has a global tensor)
            S3RowOut ro{};
            if (to_global) ro = s3_row_out(og, ri, a.H, a.W);
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                const int cx = ri.sx + 3 * j + m;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 v = (S3_ABL & 16) ? acc[m][n] : stream_prelu(acc[m][n] * c.inv, r.am1[n]);
                    v = cx >= 0 && cx < a.W ? v : kStreamZero;      // SAME padding: columns outside the image are zero in every ring
                    if constexpr ((S3_ABL & 16) != 0) unit[m][n] = __builtin_bit_cast(u32x4, v); else
                    unit[m][n] = p16_unit(v, m1, chk, zero2);
                    if (to_global && ri.store && cx >= ri.ux0 && cx < ri.ux1) s3_store_global(og, ro, 3 * j + m, n, q, unit[m][n], v);
                }
            }
            if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + ri.img] = 1; }
        }
    }
    if (live && c.out.px > 0) {
        const unsigned wb = lds0 + c.out.off + (unsigned)(g & 3) * out_row + (unsigned)(3 * j + 1) * out_px + (unsigned)q * 16u;
#pragma unroll
        for (int n = 0; n < NT; ++n)
            if (2 * n + (q >> 1) < c.out.octs) {
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) s3_st(wb + (unsigned)m * out_px + (unsigned)n * 64u, zero_row ? kStreamZero : __builtin_bit_cast(f32x4, unit[m][n]));
            }
    }
}
template <int OCTS, int NT>
__device__ __forceinline__ void s3_conv_role(const Stream3Args& a, const StreamArgs& geo, int ci, unsigned lds0, int j0, int rows, int T, int lane) {
    S3ConvRegs<OCTS, NT> r;
    s3_conv_load<OCTS, NT>(a, ci, lane, r);
    const float m1 = opaque_minus_one();
    const h2 zero2 = p16_opaque_zero2();
#ifdef S3_DBG
    long long dbg_c = 0, dbg_b = 0, dbg_t = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < T; ++t) {
        s3_conv_step<OCTS, NT>(a, geo, ci, lds0, j0, rows, t, lane, r, m1, zero2);
#ifdef S3_DBG
        const long long tb = __builtin_readcyclecounter();
        dbg_c += tb - dbg_t;
#endif
        s3_barrier();
#ifdef S3_DBG
        dbg_t = __builtin_readcyclecounter();
        dbg_b += dbg_t - tb;
#endif
    }
#ifdef S3_DBG
    if (a.dbg && blockIdx.x == 0 && lane == 0) { long long* d = a.dbg + (threadIdx.x >> 6) * 4; d[0] = dbg_c; d[1] = dbg_b; d[2] = T; }
#endif
}

// ---- two light convs in ONE wave (nin.on: A1 || B1 takes two waves of the eight): they compute different stream rows of the same step, one
// after the other.  c-DCSCN: (CNN6, CNN7) = 45 + 45 MFMAs and (CNN5, B2) = 63 + 27 against CNN2's 162 ------------------------------------
template <int O1, int O2>
__device__ __forceinline__ void s3_pair_role(const Stream3Args& a, const StreamArgs& geo, int c1, int c2, unsigned lds0, int j0, int rows, int T, int lane) {
    S3ConvRegs<O1, 1> r1;
    S3ConvRegs<O2, 1> r2;
    s3_conv_load<O1, 1>(a, c1, lane, r1);
    s3_conv_load<O2, 1>(a, c2, lane, r2);
    const float m1 = opaque_minus_one();
    const h2 zero2 = p16_opaque_zero2();
#ifdef S3_DBG
    long long dbg_c = 0, dbg_b = 0, dbg_t = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < T; ++t) {
        s3_conv_step<O1, 1>(a, geo, c1, lds0, j0, rows, t, lane, r1, m1, zero2);
        s3_conv_step<O2, 1>(a, geo, c2, lds0, j0, rows, t, lane, r2, m1, zero2);
#ifdef S3_DBG
        const long long tb = __builtin_readcyclecounter();
        dbg_c += tb - dbg_t;
#endif
        s3_barrier();
#ifdef S3_DBG
        dbg_t = __builtin_readcyclecounter();
        dbg_b += dbg_t - tb;
#endif
    }
#ifdef S3_DBG
    if (a.dbg && blockIdx.x == 0 && lane == 0) { long long* d = a.dbg + (threadIdx.x >> 6) * 4; d[0] = dbg_c; d[1] = dbg_b; d[2] = T; }
#endif
}

// ---- A1 || B1 (nin.on): the 1x1 GEMM over the concat of all L layers, accumulated as the layers' rows appear -- H_concat never exists, no
// feature map goes to HBM.  Output channels [B1 (8) | A1 (24)] = two 16-channel tiles, ONE wave per tile: layer i's row g is
// in its ring from step g + 2 i + 1 on, so the accumulators of a row are live for R = 2 L - 1 steps -- R rows x 3 pixel tiles x 4 registers (156 for
// L = 7) -- with the slot of a row a compile-time constant of (step mod R, layer): the step loop is unrolled R times.  A layer is one K = 32 step (its
// <= 4 octets; lane groups past the last octet re-read the last one against zero filters).  The row layer L - 1 completes is finished: scale,
// PReLU, split; channels 0 .. 7 (B1) go to the B1 ring for B2 (s3_trio_role), channels 8 .. 31 (A1) to Concat2 behind B2's octet.
template <int L>
__device__ __forceinline__ void s3_nin_role(const Stream3Args& a, const StreamArgs& geo, int n, unsigned lds0, int j0, int rows, int T, int lane) {
    constexpr int R = 2 * L - 1;
    const int j = lane & 15, q = lane >> 4;
#ifndef S3_NIN_PRIO
#define S3_NIN_PRIO 1
#endif
    // this wave's step is seven short dependent blocks (ring reads -> nine MFMAs): behind its SIMD partner's bursts of 27 - 54 MFMAs each of them
    // would wait; with priority its few MFMAs go first and the partner loses nothing it can measure
    if (S3_NIN_PRIO) asm volatile("s_setprio 1");
    h8 fh[L], fl[L];
    {
        const char* wsrc = reinterpret_cast<const char*>(a.blob + a.nin.w_off);
#pragma unroll
        for (int i = 0; i < L; ++i) {
            fh[i] = *reinterpret_cast<const h8*>(wsrc + ((size_t)((i * 2 + n) * 2 + 0) * 64 + lane) * 16);
            fl[i] = *reinterpret_cast<const h8*>(wsrc + ((size_t)((i * 2 + n) * 2 + 1) * 64 + lane) * 16);
        }
    }
    // (156 accumulator + 56 fragment registers: everything else is re-derived per step -- ring addresses from the lane, bias / slope from the blob)
    const unsigned ba = lds0 + (unsigned)a.ring_bytes + (unsigned)(n * 16 + 4 * q) * 4u;    // bias * 2^e, slope - 1 at + 128 bytes: copied behind the rings at kernel start
    const float m1 = opaque_minus_one();
    const h2 zero2 = p16_opaque_zero2();
    const unsigned b1_px = (unsigned)a.nin.b1.px, b1_row = (unsigned)kStreamRowPx * b1_px;
    f32x4 acc[R][kStreamMT];
    StreamCursor cur;
#ifdef S3_DBG
    long long dbg_c = 0, dbg_b = 0, dbg_t = __builtin_readcyclecounter();
#endif
    auto step = [&](auto p_, int t) DCSCN_INL {
        constexpr int p = decltype(p_)::value;                     // t mod R
        static_for<0, L>([&](auto i_) DCSCN_INL {
            constexpr int i = decltype(i_)::value;
            constexpr int s = ((p - 2 * i - 1) % R + R) % R;       // accumulator slot of the row layer i contributes to at this step
            const int g = t - 2 * i - 1;
            if (g >= 0 && g < rows) {                               // (wave uniform)
                const S3Ring& rg = i == 0 ? a.first_out : a.conv[i > 0 ? i - 1 : 0].out;
                int le = lane;
                asm volatile("" : "+v"(le));                         // (keeps the per-layer addresses out of the registers between steps)
                const int je = le & 15, qe = le >> 4;
                const int qq = qe < rg.octs ? qe : rg.octs - 1;      // (octets past the layer's last: any valid unit of the pixel, the filter rows are zero)
                const unsigned px = (unsigned)rg.px;
                const unsigned b = lds0 + (unsigned)rg.off + ((unsigned)(g & 3) * (unsigned)kStreamRowPx + (unsigned)(3 * je + 1)) * px + (unsigned)qq * 32u;
                h8 xh[kStreamMT], xl[kStreamMT];
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) {
                    xh[m] = __builtin_bit_cast(h8, stream_ld(b + (unsigned)m * px));
                    xl[m] = __builtin_bit_cast(h8, stream_ld(b + (unsigned)m * px + 16u));
                }
                f32x4 bs = kStreamZero;
                if constexpr (i == 0) bs = stream_ld(ba);
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], xh[m], i == 0 ? bs : acc[s][m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], xl[m], acc[s][m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < kStreamMT; ++m) acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], xh[m], acc[s][m], 0, 0, 0);
            }
        });
        constexpr int sl = ((p - 2 * (L - 1) - 1) % R + R) % R;
        const int g = t - 2 * (L - 1) - 1;                          // the row the last layer just completed
        if (g >= 0 && g < rows) {
            const StreamRow ri = stream_row(geo, j0, cur, g);
            const S3RowOut ro = s3_row_out(a.out2, ri, a.H, a.W);
            float chk = 0.0f;
            const f32x4 am1 = stream_ld(ba + 128u);
            const unsigned wb = lds0 + (unsigned)a.nin.b1.off + (unsigned)(g & 3) * b1_row + (unsigned)(3 * j + 1) * b1_px + (unsigned)q * 16u;
#pragma unroll
            for (int m = 0; m < kStreamMT; ++m) {
                const int cx = ri.sx + 3 * j + m;
                f32x4 v = stream_prelu(acc[sl][m] * a.nin.inv, am1);
                v = !ri.zero && cx >= 0 && cx < a.W ? v : kStreamZero;      // SAME padding for B2: zero rows and columns outside the image are zeros in the ring
                const u32x4 unit = p16_unit(v, m1, chk, zero2);
                if (!ri.zero && ri.store && cx >= ri.ux0 && cx < ri.ux1) s3_store_global(a.out2, ro, 3 * j + m, n, q, unit, v);
                if (n == 0 && q < 2) s3_st(wb + (unsigned)m * b1_px, __builtin_bit_cast(f32x4, unit));      // B1 = channels 0 .. 7: the octet's hi and lo units
            }
            if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + ri.img] = 1; }
        }
#ifdef S3_DBG
        const long long tb = __builtin_readcyclecounter();
        dbg_c += tb - dbg_t;
#endif
        s3_barrier();
#ifdef S3_DBG
        dbg_t = __builtin_readcyclecounter();
        dbg_b += dbg_t - tb;
#endif
    };
    for (int t = 0; t < T; t += R)
        static_for<0, R>([&](auto p_) DCSCN_INL {
            constexpr int p = decltype(p_)::value;
            if (t + p < T) step(p_, t + p);
        });
#ifdef S3_DBG
    if (a.dbg && blockIdx.x == 0 && lane == 0) { long long* d = a.dbg + (threadIdx.x >> 6) * 4; d[0] = dbg_c; d[1] = dbg_b; d[2] = T; }
#endif
}

// one workgroup = n_waves <= 8 waves (CNN1 + one per conv; pack.hip: the role table), one per CU (the rings take most of the LDS)
__global__ __launch_bounds__(512) void feat3_stream(const Stream3Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    {
        f32x4* s4 = reinterpret_cast<f32x4*>(smem);
        for (int i = tid; i < a.ring_bytes / 16; i += blockDim.x) s4[i] = kStreamZero;
        if (a.nin.on && tid < 16) s4[a.ring_bytes / 16 + tid] = reinterpret_cast<const f32x4*>(a.blob + a.nin.ba_off)[tid];     // A1 || B1: bias * 2^e [32], slope - 1 [32]
        __syncthreads();
    }
    const StreamArgs geo = s3_geometry(a);
    const int j0 = blockIdx.x * a.jobs_per_wg;
    const int j1 = min(a.n_jobs, j0 + a.jobs_per_wg);
    const int rows = (j1 - j0) * (a.rows_c + 1);
    const int T = rows + a.total_lag;
    const int ci = a.role_conv[wave];
    if (ci < 0) s3_first_role(a, geo, lds0, j0, rows, T, lane);
    else if (ci >= kS3RolePair) {
        // (instantiated for the c-DCSCN shape; graph.hip: fuse_feat3_stream checks: pair 0 = conv[L - 3], conv[L - 2] reading two octets each,
        //  pair 1 = conv[L - 4] reading three octets and B2 = conv[L - 1] reading B1's one)
        if (ci == kS3RolePair) s3_pair_role<2, 2>(a, geo, a.L - 3, a.L - 2, lds0, j0, rows, T, lane);
        else s3_pair_role<3, 1>(a, geo, a.L - 4, a.L - 1, lds0, j0, rows, T, lane);
#ifndef S3_NO_NIN
    } else if (ci >= kS3RoleNin) s3_nin_role<7>(a, geo, ci - kS3RoleNin, lds0, j0, rows, T, lane);
#else
    } else if (ci >= kS3RoleNin) { }
#endif
    else {
        const bool two = a.conv[ci].tiles == 2;
        switch (a.conv[ci].in.octs) {
            case 1: if (two) s3_conv_role<1, 2>(a, geo, ci, lds0, j0, rows, T, lane); else s3_conv_role<1, 1>(a, geo, ci, lds0, j0, rows, T, lane); break;
            case 2: if (two) s3_conv_role<2, 2>(a, geo, ci, lds0, j0, rows, T, lane); else s3_conv_role<2, 1>(a, geo, ci, lds0, j0, rows, T, lane); break;
            case 3: if (two) s3_conv_role<3, 2>(a, geo, ci, lds0, j0, rows, T, lane); else s3_conv_role<3, 1>(a, geo, ci, lds0, j0, rows, T, lane); break;
            default: if (two) s3_conv_role<4, 2>(a, geo, ci, lds0, j0, rows, T, lane); else s3_conv_role<4, 1>(a, geo, ci, lds0, j0, rows, T, lane); break;
        }
    }
}

}  // namespace dcscn
