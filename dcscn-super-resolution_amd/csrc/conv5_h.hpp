// conv5_h: the folded linear tail (graph.hip: fold_linear_tail -- Up-PS conv + depth_to_space + last reconstruction conv,
// DCSCN.py:293-323, as ONE 5x5 conv whose 4 s^2 output channels are (sub-pixel phase, border variant)) on
// v_mfma_f32_16x16x32_f16 at f32 accuracy (split16.hpp).  conv3_h.hpp's scheme with the geometry of a 5x5 window:
//
// * workgroup = 4 waves = 16 x 16 LR pixels x NT * 16 channels (NT = ceil(4 s^2 / 16): 1 / 3 / 4 for x2 / x3 / x4), wave w owns
//   rows 4w..4w+3; halo tile 20 x 20 pixels, written to LDS as conv3_h's B-operand image (same unit swizzle: conflict free for
//   all five tap columns), 51,200 bytes per 32-channel chunk.
// * With one or a few output tiles a tap is only 12 NT MFMAs per wave -- too little to pay a barrier for -- so the ring's unit is a
//   COLUMN of five taps: two slots of 5 * NT * 2 KB, the next column's 10 NT pieces fetched by LDS-DMA while this one computes,
//   one `vmcnt(0)` + barrier per column (60 NT MFMAs per wave).  Down a column the B rows slide as in conv3_h: 8 + 4 * 2 reads.
// * LDS = 50 + NT * 20 KB: two workgroups per CU at NT = 1 (x2), one above.
// * epilogue: conv_igemm's fold epilogue (phase = channel / 4, variant picked by the pixel's position on the image border,
//   + x2, scalar store to y), accumulators * 2^-e; non-finite accumulators raise the image's redo flag (split16.hpp).
#pragma once
#include "conv3_h.hpp"

namespace dcscn {

template <int NT>
struct C5HGeom {
    static constexpr int THREADS = 256;
    static constexpr int KC = 32;
    static constexpr int TH = 16, TW = 16;
    static constexpr int HT = 20;                             // halo tile edge
    static constexpr int HP = HT * HT;
    static constexpr int PIX_BYTES = 128;
    static constexpr int ROW_BYTES = HT * PIX_BYTES;
    static constexpr int IN_BYTES = HP * PIX_BYTES;           // 51200
    static constexpr int IN_ITEMS = HP * 8;
    static constexpr int IN_ROUNDS = (IN_ITEMS + THREADS - 1) / THREADS;   // 13
    static constexpr int F_TAP_BYTES = NT * 2048;
    static constexpr int F_COL_BYTES = 5 * F_TAP_BYTES;
    static constexpr int F_PIECES = 10 * NT;                  // 1 KB DMA pieces of a column
    static constexpr int F_ROUNDS = (F_PIECES + 3) / 4;
    static constexpr int F_BASE = IN_BYTES;
    static constexpr int BA_BASE = IN_BYTES + 2 * F_COL_BYTES;   // the composite's bias terms (NT * 16 floats), staged at workgroup start
    static constexpr int LDS_BYTES = BA_BASE + NT * 64;
};

// IN16: the input is a P16 tensor (a.in16, p16.hpp), staged as in conv3_h<.., IN16>: one ready (hi | lo) unit per item, no conversion
template <int NT, bool IN16 = false>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void conv5_h(const ConvArgs a) {
    using G = C5HGeom<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem_c5h[];
    char* smem = smem_c5h;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15;
    const int lk = lane >> 4;

    const int tile_id = blockIdx.x;
    int bid = tile_id;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int y0 = ty * G::TH;
    const int x0 = tx * G::TW;
    const int H = a.H, W = a.W;
    // origin of the halo tile; only in-image addresses are dereferenced (out-of-image items read the tile's own first pixel)
    const float* a_base = IN16 ? nullptr : a.in + (size_t)img * H * W * a.in_stride + a.in_off + ((ptrdiff_t)(y0 - 2) * W + (x0 - 2)) * a.in_stride;
    const int pix0 = (img * H + y0 - 2) * W + x0 - 2;         // IN16: flat pixel index of the halo tile's origin (may be negative)

    // ---- staging of the input image: item = r * 256 + tid = (halo pixel hp = r * 32 + (tid >> 3), channel quad tid & 7) ----
    const int cq = tid & 7;
    unsigned ok_mask = 0;
    {
        int hrow = (tid >> 3) >= G::HT ? 1 : 0, hcol = (tid >> 3) - G::HT * hrow;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int gy = y0 - 2 + hrow, gx = x0 - 2 + hcol;
            const bool ok = r * 32 + (tid >> 3) < G::HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            ok_mask |= ok ? (1u << r) : 0u;
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    }
    const bool all_in = __builtin_amdgcn_readfirstlane((int)(y0 >= 2 && x0 >= 2 && y0 + G::TH + 2 <= H && x0 + G::TW + 2 <= W)) != 0;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* f_base = reinterpret_cast<const char*>(a.wpack16);
    const unsigned f_off = (unsigned)(lane * 16);

    if (tid < NT * 4) *reinterpret_cast<f32x4*>(smem + G::BA_BASE + tid * 16) = reinterpret_cast<const f32x4*>(a.bias)[tid];   // no global round trip in the epilogue

    f32x4 gin[G::IN_ROUNDS];
    auto load_in = [&](int chunk) DCSCN_INL {
        if constexpr (IN16) {
            const int rem = a.in16.octs - 4 * chunk;           // octets of this chunk (block uniform)
            const int rec = rem >= 4 ? 128 : 32 * rem;
            const char* base = a.in16.base + (long long)chunk * a.in16.plane;
            int hp0 = tid >> 3;
            asm volatile("" : "+v"(hp0));
            int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
            static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
                constexpr int r = decltype(r_)::value;
                const int kq = ((cq >> 1) - (hcol >> 1)) & 3;  // the unit c3h_unit puts at slot cq of this halo column
                const int part = (cq ^ kq ^ hcol) & 1;
                const bool ok = ((ok_mask >> r) & 1u) && kq < rem;
                const unsigned off = ok ? 128u + (unsigned)(pix0 + hrow * W + hcol) * (unsigned)rec + (unsigned)((2 * kq + part) * 16) : (unsigned)(cq * 16);
                gin[r] = *reinterpret_cast<const f32x4*>(base + (size_t)off);
                hcol += 32 - G::HT; hrow += 1;
                if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
            });
            return;
        }
        const int c0 = chunk * G::KC + cq * 4;
        const unsigned coff = (unsigned)((c0 < a.cin_phys ? c0 : 0) * 4);
        const char* base = reinterpret_cast<const char*>(a_base);
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hrow = hp0 >= G::HT ? 1 : 0, hcol = hp0 - G::HT * hrow;
        const int stride4 = a.in_stride * 4;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int pix = ((ok_mask >> r) & 1u) ? hrow * W + hcol : 2 * W + 2;
            gin[r] = *reinterpret_cast<const f32x4*>(base + (size_t)((unsigned)(pix * stride4) + coff));
            hcol += 32 - G::HT; hrow += 1;
            if (hcol >= G::HT) { hcol -= G::HT; hrow += 1; }
        });
    };
    const float m1 = opaque_minus_one();
    auto convert_in = [&](auto r_, int chunk) DCSCN_INL {
        constexpr int r = decltype(r_)::value;
        if constexpr (IN16) return;
        f32x4 x = gin[r];
        const bool whole = all_in && (chunk + 1) * G::KC <= a.cin_phys;
        if (!whole) {
            const bool ok = chunk * G::KC + cq * 4 < a.cin_phys && ((ok_mask >> r) & 1u);
            x.x = ok ? x.x : 0.0f; x.y = ok ? x.y : 0.0f; x.z = ok ? x.z : 0.0f; x.w = ok ? x.w : 0.0f;
        }
        h4 hi, lo;
        split4(x, m1, hi, lo);
        const u32x2 hu = __builtin_bit_cast(u32x2, hi), lu = __builtin_bit_cast(u32x2, lo);
        gin[r] = __builtin_bit_cast(f32x4, u32x4{hu.x, hu.y, lu.x, lu.y});
    };
    auto store_in = [&]() DCSCN_INL {
        int hp0 = tid >> 3;
        asm volatile("" : "+v"(hp0));
        int hcol = hp0 >= G::HT ? hp0 - G::HT : hp0;
        static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int hp = r * 32 + hp0;
            const int kq = cq >> 1;
            const int off = hp * G::PIX_BYTES + c3h_unit(hcol, kq, 0) * 16 + (cq & 1) * 8;
            const u32x4 v = __builtin_bit_cast(u32x4, gin[r]);
            if constexpr (IN16) {
                if (r < G::IN_ROUNDS - 1 || hp < G::HP) *reinterpret_cast<u32x4*>(smem + hp * G::PIX_BYTES + cq * 16) = v;
            } else
            if (r < G::IN_ROUNDS - 1 || hp < G::HP) {
                *reinterpret_cast<u32x2*>(smem + off) = u32x2{v.x, v.y};
                *reinterpret_cast<u32x2*>(smem + (off ^ 16)) = u32x2{v.z, v.w};
            }
            hcol += 32 - G::HT;
            if (hcol >= G::HT) hcol -= G::HT;
        });
    };
    // the 10 NT pieces of tap column kx of a chunk (image: [chunk][tap = ky * 5 + kx][n][hi | lo][1 KB]) -> ring slot
    auto dma_col = [&](int chunk, int kx, int slot) DCSCN_INL {
        static_for<0, G::F_ROUNDS>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int p = (wave + 4 * r) % G::F_PIECES;       // waves without a piece of their own repeat one
            const int ky = p / (2 * NT), rem = p - ky * (2 * NT);
            glds16(f_base + ((size_t)(chunk * 25 + ky * 5 + kx) * (2 * NT) + rem) * 1024, f_off,
                   lds0 + G::F_BASE + (unsigned)slot * G::F_COL_BYTES + (unsigned)p * 1024u);
        });
    };

    f32x4 acc[4][NT];
    static_for<0, 4>([&](auto m_) DCSCN_INL {
        static_for<0, NT>([&](auto n_) DCSCN_INL { acc[decltype(m_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
    });
    auto b_col = [&](auto kx_) DCSCN_INL {
        constexpr int kx = decltype(kx_)::value;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int hx = (l & 15) + kx;
        return (4 * wave * G::HT + hx) * G::PIX_BYTES + c3h_unit(hx, l >> 4, 0) * 16;
    };
    const int a_lane = G::F_BASE + lane * 16;

    const int n_chunks = a.n_chunks;
    dma_col(0, 0, 0);
#ifndef C5H_ABL
#define C5H_ABL 0          // timing-only builds (results wrong by design): 1 no fetch of the first chunk's image (the prologue a persistent kernel would hide)
#endif
    if constexpr ((C5H_ABL & 1) == 0) load_in(0);
    else static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL { gin[decltype(r_)::value] = f32x4{1.0f, 2.0f, 3.0f, (float)tid}; });
    static_for<0, G::IN_ROUNDS>([&](auto r_) DCSCN_INL { convert_in(r_, 0); });
    store_in();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const bool more = chunk + 1 < n_chunks;                // block uniform
        const int nchunk = more ? chunk + 1 : chunk;
        static_for<0, 5>([&](auto kx_) DCSCN_INL {
            constexpr int kx = decltype(kx_)::value;
            const int slot = (chunk + kx) & 1;                 // (chunk * 5 + kx) & 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this column's pieces (issued a column ago) and, at kx 1, the input values
            __syncthreads();                                   // ... of every wave; all waves are past the other slot's column
            dma_col(kx + 1 < 5 ? chunk : nchunk, (kx + 1) % 5, slot ^ 1);   // past the end: a re-fetch nobody reads
            if constexpr (kx == 0) load_in(nchunk);
            const int b_hi = b_col(std::integral_constant<int, kx>{});
            const char* fcol = smem + a_lane + slot * G::F_COL_BYTES;
            h8 xh[4], xl[4];
            static_for<0, 5>([&](auto ky_) DCSCN_INL {
                constexpr int ky = decltype(ky_)::value;
                static_for<(ky == 0 ? 0 : 3), 4>([&](auto m_) DCSCN_INL {
                    constexpr int row = ky + decltype(m_)::value;
                    xh[row & 3] = *reinterpret_cast<const h8*>(smem + b_hi + row * G::ROW_BYTES);
                    xl[row & 3] = *reinterpret_cast<const h8*>(smem + (b_hi ^ 16) + row * G::ROW_BYTES);
                });
                static_for<0, NT>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    const h8 wh = *reinterpret_cast<const h8*>(fcol + ky * G::F_TAP_BYTES + (2 * n) * 1024);
                    const h8 wl = *reinterpret_cast<const h8*>(fcol + ky * G::F_TAP_BYTES + (2 * n + 1) * 1024);
                    static_for<0, 4>([&](auto m_) DCSCN_INL {
                        constexpr int m = decltype(m_)::value;
                        constexpr int q = (ky + m) & 3;
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[q], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[q], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[q], acc[m][n], 0, 0, 0);
                    });
                });
            });
            // the next chunk's input values become (hi, lo) pairs four items per column from the second column on
            if constexpr (kx >= 1)
                static_for<4 * (kx - 1), (4 * kx < G::IN_ROUNDS ? 4 * kx : G::IN_ROUNDS)>([&](auto r_) DCSCN_INL { convert_in(r_, nchunk); });
        });
        if (more) {
            __syncthreads();                                  // every wave is past its last read of this chunk's image
            store_in();                                       // made visible by the barrier in front of the next column
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the re-fetch of the last column

    // ---- fold epilogue (conv_igemm.hpp): conv channel n * 16 + 4 lk + r = (phase n * 4 + lk, border variant r) ----
    const int gx = x0 + lj;
    const int ps = a.ps;
    const int orow = W * ps;
    const float inv = a.inv_scale;
    const float zero = opaque_zero();
    float chk = 0.0f;
    float* yout = a.out0.ptr;
    if constexpr (NT == 1) {
        if (a.fold == 2) {
            // whole-tail fold (graph.hip: fold_whole_tail): conv channel 4 lk + r = sub-pixel phase (pa, pb) = (p / ps, p % ps) of the ps x ps block of LR
            // pixel (gy, gx), interior kernel; the pixels of the image's border ring are computed by fold_border (their kernels differ)
            const bool col_in = gx > 0 && gx < W - 1;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + lk * 16);
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                const int gy = y0 + 4 * wave + m;
                if (col_in && gy > 0 && gy < H - 1) {
                    chk = nonfinite_acc(chk, acc[m][0], zero);
                    const f32x4 v = acc[m][0] * inv + bv;
                    if (ps == 4) {                             // phases 4 lk .. 4 lk + 3 = HR row 4 gy + lk, columns 4 gx .. 4 gx + 3: one 16-byte store
                        const size_t idx = ((size_t)(img * H + gy) * 4 + lk) * orow + (size_t)gx * 4;
                        f32x4 o = v;
                        if (a.res) o = o + *reinterpret_cast<const f32x4*>(a.res + idx);
                        *reinterpret_cast<f32x4*>(yout + idx) = o;
                    } else {
                        const float vr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int p = 4 * lk + r;
                            if (p < ps * ps) {
                                const int pa = p / ps, pb = p - pa * ps;
                                const size_t idx = ((size_t)(img * H + gy) * ps + pa) * orow + (size_t)(gx * ps + pb);
                                float out = vr[r];
                                if (a.res) out += a.res[idx];
                                yout[idx] = out;
                            }
                        }
                    }
                }
            });
            if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }
            return;
        }
    }
    static_for<0, NT>([&](auto n_) DCSCN_INL {
        constexpr int n = decltype(n_)::value;
        const int phase = n * 4 + lk;
        if (phase < ps * ps && gx < W) {
            const int pa = phase / ps, pb = phase - pa * ps;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + G::BA_BASE + (n * 4 + lk) * 16);
            const bool cb = (pb == 0 && gx == 0) || (pb == ps - 1 && gx == W - 1);
            static_for<0, 4>([&](auto m_) DCSCN_INL {
                constexpr int m = decltype(m_)::value;
                const int gy = y0 + 4 * wave + m;
                if (gy < H) {
                    chk = nonfinite_acc(chk, acc[m][n], zero);
                    const bool rb = (pa == 0 && gy == 0) || (pa == ps - 1 && gy == H - 1);
                    const f32x4 v = acc[m][n] * inv + bv;
                    const float lo = cb ? v.y : v.x, hi = cb ? v.w : v.z;
                    const size_t idx = ((size_t)(img * H + gy) * ps + pa) * orow + (size_t)(gx * ps + pb);
                    float out = rb ? hi : lo;
                    if (a.res) out += a.res[idx];
                    yout[idx] = out;
                }
            });
        }
    });
    if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }     // the image goes to the float32 plan (exec.hip)
}

// fold_border: the border ring of a whole-tail fold (ConvArgs::fold == 2).  On the first / last row and column of an image the composite kernel
// differs from the interior one -- the zero padding of the intermediate maps (the shuffled maps at 2x / 4x resolution, the HR map the last conv
// reads) drops taps there, and with them the bias terms they carry -- so every (vy, vx) in {interior, first, last, both}^2 has its own 5x5 kernel
// and bias (pack.hip: pack_foldx).  One wave = one JOB = 16 pixels that share a variant, the 16 columns of the MFMA's B operand:
//   * 16 consecutive pixels of the first / last row (columns 1 .. W - 2) or of the first / last column (rows 1 .. H - 2) of one image: the
//     5 x 20-pixel window the 25 taps read is staged ONCE in the wave's own 12.5 KB of LDS (conv3_h's unit swizzle: conflict-free reads for
//     all five shifts), a tap's B operand is two ds_read_b128 -- the first build fetched every tap's pixels from the tensor: 5 x the bytes
//     through L2, 46 of its 98 us per 1024 patches (tools/fb_abl.sh);
//   * the same corner of 16 consecutive images (no window to share: every lane fetches its taps straight from the tensor).
// A fragments come from the variant's pack_conv16 image (L2 resident), a tap row at a time; 25 taps x chunks x 3 MFMAs per job; no
// workgroup barrier (a wave's LDS traffic is in order).  8 % of the pixels of a 48 x 48 patch, 1.5 % of a 256 x 256 image.
#ifndef FB_ABL
#define FB_ABL 0      // timing-only builds (tools/fb_abl.sh; results wrong by design): 1 no pixel fetches, 2 no filter fetches, 4 no stores / residual reads, 8 no MFMAs
#endif
constexpr int kFbWinBytes = 5 * 20 * 128;      // a job's window: 5 lines across x 20 pixels along x one 128-byte record
struct FoldBorderJobs {             // job list of a launch (host and device)
    int segs_w, segs_h, rows_n, cols_n, j_img, cb;
    long long img_jobs, total;
};
__host__ __device__ inline FoldBorderJobs fold_border_jobs(int N, int H, int W) {
    FoldBorderJobs j;
    j.segs_w = W > 2 ? (W - 2 + 15) / 16 : 0; j.segs_h = H > 2 ? (H - 2 + 15) / 16 : 0;
    j.rows_n = H > 1 ? 2 : 1; j.cols_n = W > 1 ? 2 : 1;
    j.j_img = j.rows_n * j.segs_w + j.cols_n * j.segs_h;
    j.cb = (N + 15) / 16;
    j.img_jobs = (long long)N * j.j_img;
    j.total = j.img_jobs + (long long)j.rows_n * j.cols_n * j.cb;
    return j;
}

template <bool IN16>
__global__ __launch_bounds__(256, 2) void fold_border(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_fb[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lj = lane & 15, lk = lane >> 4;
    const int H = a.H, W = a.W, N = a.N, ps = a.ps;
    const FoldBorderJobs jb = fold_border_jobs(N, H, W);
    const long long job = (long long)blockIdx.x * 4 + wave;
    if (job >= jb.total) return;
    // the lane's pixel (img, y, x); row / column jobs: (by, bx) = the pixel of lane column 0, along_x = the 16 pixels run along x
    int img, y, x, vy, vx, by = 0, bx = 0;
    bool valid, along_x = true;
    const bool corner = job >= jb.img_jobs;                   // (wave uniform)
    if (!corner) {
        img = (int)(job / jb.j_img);
        int r = (int)(job - (long long)img * jb.j_img);
        if (r < jb.rows_n * jb.segs_w) {
            const int which = r / jb.segs_w, seg = r - which * jb.segs_w;
            by = which ? H - 1 : 0; bx = 1 + 16 * seg;
            y = by; x = bx + lj; valid = x <= W - 2;
            vy = H == 1 ? 3 : which ? 2 : 1; vx = 0;
        } else {
            r -= jb.rows_n * jb.segs_w;
            const int which = r / jb.segs_h, seg = r - which * jb.segs_h;
            bx = which ? W - 1 : 0; by = 1 + 16 * seg;
            x = bx; y = by + lj; valid = y <= H - 2;
            vx = W == 1 ? 3 : which ? 2 : 1; vy = 0;
            along_x = false;
        }
    } else {
        const int r = (int)(job - jb.img_jobs);
        const int c = r / jb.cb, batch = r - c * jb.cb;
        const int cy = c / jb.cols_n, cx = c - cy * jb.cols_n;
        y = cy ? H - 1 : 0; x = cx ? W - 1 : 0;
        img = 16 * batch + lj; valid = img < N;
        vy = H == 1 ? 3 : cy ? 2 : 1; vx = W == 1 ? 3 : cx ? 2 : 1;
    }
    const int v = __builtin_amdgcn_readfirstlane(vy * 4 + vx);
    const int n_chunks = a.n_chunks;
    const char* fv = reinterpret_cast<const char*>(a.wpack16) + (size_t)v * n_chunks * (25 * 2048) + lane * 16;
    const float m1 = opaque_minus_one();
    char* win = smem_fb + wave * kFbWinBytes;
    int rd[5];                                                 // window reads: the lane's unit of position lj + shift of line 0
#pragma unroll
    for (int sh = 0; sh < 5; ++sh) rd[sh] = (lj + sh) * 128 + c3h_unit(lj + sh, lk, 0) * 16;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const char* fc = fv + (size_t)chunk * (25 * 2048);
        int rem = 4;
        unsigned rec = 0;
        const char* base16 = nullptr;
        if constexpr (IN16) {
            rem = a.in16.octs - 4 * chunk;
            rec = rem >= 4 ? 128u : 32u * (unsigned)rem;
            base16 = a.in16.base + (long long)chunk * a.in16.plane;
        }
        // the (hi | lo) units of octet kq of pixel (gi, yy, xx); ok = false: zeros
        auto fetch16 = [&](int gi, int yy, int xx, int kq, bool ok, h8& hi, h8& lo) DCSCN_INL {
            if constexpr ((FB_ABL & 1) != 0) { u32x4 z = {0x3c003c00u, (unsigned)(ok ? xx : yy), 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(z)); hi = lo = __builtin_bit_cast(h8, z); return; }
            if constexpr (IN16) {
                // (pixels outside the image and octets past the tensor's last read the plane's zero record)
                const unsigned off = ok && kq < rem ? 128u + (unsigned)((gi * H + yy) * W + xx) * rec + (unsigned)kq * 32u : (unsigned)kq * 32u;
                hi = *reinterpret_cast<const h8*>(base16 + (size_t)off);
                lo = *reinterpret_cast<const h8*>(base16 + (size_t)off + 16);
            } else {
                const int c0 = chunk * 32 + 8 * kq;
                const float* p = a.in + ((size_t)((ok ? gi : 0) * H + (ok ? yy : 0)) * W + (ok ? xx : 0)) * a.in_stride + a.in_off + c0;
                f32x4 q0 = {0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0;
                if (ok && c0 < a.cin_phys) q0 = *reinterpret_cast<const f32x4*>(p);
                if (ok && c0 + 4 < a.cin_phys) q1 = *reinterpret_cast<const f32x4*>(p + 4);
                split8(q0, q1, m1, hi, lo);
            }
        };
        if (!corner) {
            // ---- window -> LDS: item t = (window pixel t >> 2 = (line across, position along), octet t & 3); 400 items, 7 per lane ----
            if (chunk > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the previous chunk's reads are done before its window is overwritten)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int t = i * 64 + lane;
                const int wp = t >> 2, kq = t & 3;
                const int ac = wp / 20, al = wp - ac * 20;
                const int yy = along_x ? by + ac - 2 : by + al - 2, xx = along_x ? bx + al - 2 : bx + ac - 2;
                const bool ok = t < 400 && yy >= 0 && yy < H && xx >= 0 && xx < W;
                h8 hi, lo;
                fetch16(img, yy, xx, kq, ok, hi, lo);
                if (t < 400) {
                    const int u = c3h_unit(al, kq, 0);
                    *reinterpret_cast<h8*>(win + wp * 128 + u * 16) = hi;
                    *reinterpret_cast<h8*>(win + wp * 128 + (u ^ 1) * 16) = lo;
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");              // a wave's LDS accesses execute in order: no barrier
        }
        // a LINE of the window at a time (line a across, shift s along: tap (a, s) for row jobs, (s, a) for column jobs): its 5 filter fragments
        // (and, corner jobs, 5 pixel fetches) are issued together, then the 15 MFMAs
#pragma unroll
        for (int la = 0; la < 5; ++la) {
            h8 wh[5], wl[5], xh[5], xl[5];
#pragma unroll
            for (int sh = 0; sh < 5; ++sh) {
                if constexpr ((FB_ABL & 2) != 0) { u32x4 z = {0x3c003c00u, (unsigned)lane, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(z)); wh[sh] = wl[sh] = __builtin_bit_cast(h8, z); continue; }
                const int tap = along_x ? la * 5 + sh : sh * 5 + la;                 // (wave uniform)
                wh[sh] = *reinterpret_cast<const h8*>(fc + tap * 2048);
                wl[sh] = *reinterpret_cast<const h8*>(fc + tap * 2048 + 1024);
            }
            if (corner) {
#pragma unroll
                for (int sh = 0; sh < 5; ++sh) {
                    const int yy = y + la - 2, xx = x + sh - 2;
                    fetch16(img, yy, xx, lk, valid && yy >= 0 && yy < H && xx >= 0 && xx < W, xh[sh], xl[sh]);
                }
            } else {
#pragma unroll
                for (int sh = 0; sh < 5; ++sh) {
                    xh[sh] = *reinterpret_cast<const h8*>(win + rd[sh] + la * (20 * 128));
                    xl[sh] = *reinterpret_cast<const h8*>(win + (rd[sh] ^ 16) + la * (20 * 128));
                }
            }
#pragma unroll
            for (int sh = 0; sh < 5; ++sh) {
                if constexpr ((FB_ABL & 8) != 0) { f32x4 t = acc; const h8 a0 = wl[sh], a1 = wh[sh], b0 = xh[sh], b1 = xl[sh]; asm volatile("" : "+v"(t) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)); acc = t; continue; }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[sh], xh[sh], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sh], xl[sh], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[sh], xh[sh], acc, 0, 0, 0);
            }
        }
    }
    if (!valid) return;
    // lane (lj, lk): phases 4 lk .. 4 lk + 3 of its pixel
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + v * 16 + 4 * lk);
    const f32x4 o4 = acc * a.inv_scale + bv;
    const float zero = opaque_zero();
    const float chk = nonfinite_acc(0.0f, acc, zero);
    const int orow = W * ps;
    float* yout = a.out0.ptr;
    if (ps == 4) {                                             // phases 4 lk .. 4 lk + 3 = HR row 4 y + lk, columns 4 x .. 4 x + 3: one 16-byte store
        const size_t idx = ((size_t)(img * H + y) * 4 + lk) * orow + (size_t)x * 4;
        f32x4 o = o4;
        if constexpr ((FB_ABL & 4) != 0) { if (o.x == 1.2345e-30f) *reinterpret_cast<f32x4*>(yout + idx) = o; } else {
        if (a.res) o = o + *reinterpret_cast<const f32x4*>(a.res + idx);
        *reinterpret_cast<f32x4*>(yout + idx) = o;
        }
    } else {
        const float vr[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 4 * lk + r;
            if (p < ps * ps) {
                const int pa = p / ps, pb = p - pa * ps;
                const size_t idx = ((size_t)(img * H + y) * ps + pa) * orow + (size_t)(x * ps + pb);
                float out = vr[r];
                if constexpr ((FB_ABL & 4) != 0) { if (out == 1.2345e-30f) yout[idx] = out; continue; }
                if (a.res) out += a.res[idx];
                yout[idx] = out;
            }
        }
    }
    if (chk != chk && a.redo) { a.redo[0] = 1; a.redo[1 + img] = 1; }
}

}  // namespace dcscn
