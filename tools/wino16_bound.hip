// a TIMING-ONLY bound (results meaningless by design) for F(2x2,3x3) Winograd on the f16 matrix pipe with split16
// operands, for CNN2 of the bench net (196 -> 166, 1024 x 48 x 48) -- the number behind DESIGN.md 3.2's "does not pay".
//
// What a conv_wino_h would have to do per 16 x 16 pixel tile (= 64 Winograd tiles = 4 MFMA column tiles) and 32-channel chunk, for ONE
// channel group of 6 output tiles (CNN2 has two: 6 + 5):
//   MFMA phase   16 frequencies x 6 cout tiles x 4 column tiles x 3 products = 1152 v_mfma_f32_16x16x32_f16 (direct form: 9 x 6 x 16 x 3 = 2592)
//                A fragments: 16 frequencies x 6 tiles x (hi, lo) = 192 KB per chunk from L2 by LDS-DMA (direct: 9 x 6 x 2 = 108 KB) -- and each
//                fragment feeds 4 x 3 = 12 MFMAs of the workgroup, where a direct tap's feeds 16 x 3 = 48: a quarter of the matrix work per
//                filter byte, per ring slot and per barrier
//                accumulators: 16 frequencies x 24 (cout, column) tiles x 4 registers = 1536 per lane position: 192 per wave of an 8-wave
//                workgroup (3 tile pairs per wave: one column tile x three cout tiles), which fixes the operand reuse: per frequency and wave
//                2 B + 6 A fragment reads for 9 MFMAs (direct, NT 6: 14 reads for 72)
//   transform    64 tiles x 32 channels x (reconstruct hi + lo, B^T d B, split the 16 frequency values) ~ 72 - 104 VALU lane-ops per tile-channel,
//                the 16 x 64 x 32 transformed (hi, lo) values = 128 KB per chunk -- they only fit in LDS four frequencies at a time
//   epilogue     A^T m A: 24 adds per output tile-channel, then conv3_h8's
// This file builds ONLY the MFMA phase, with everything else free: B operands sit in LDS (never written), no image staging, no transform,
// no epilogue, no HBM traffic at all.  If even that is not >= 20 % under conv3_h8's time for the same group work, the rest cannot help.
//
//   filter ring: 3 slots of FS frequencies (FS x 12 KB), LDS-DMA two slots ahead, one workgroup barrier per slot -- conv3_h's tap protocol
//   with counted vmcnt waits; B operands: FS-independent 64 KB region (2 x 4 frequencies x 4 column tiles x (hi, lo) KB), read in place.
//   FS = 1, 2 fit beside the B region (36 / 72 KB); FS = 4 (144 KB) only with the B region shrunk to 16 KB -- built to show the trend.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -Xclang -target-feature -Xclang -packed-fp32-ops tools/wino16_bound.hip -o tools/wino16_bound
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../dcscn-super-resolution_amd/csrc/conv3_h8.hpp"

using namespace dcscn;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int CT = 6;                       // cout tiles of the group
constexpr int FREQ_BYTES = CT * 2048;       // A fragments of one frequency: [tile][hi | lo][64 lanes][16 bytes]

// FS = frequencies per ring slot / barrier; BKB = KB of the (static) B region
// MIX bit 0: the input transform's instruction mix for the NEXT four frequencies inside every step -- per lane 4 tile-channels: reconstruct hi + lo of
//   the 4 x 4 patch (48 ops per tile-channel, once per chunk), B^T d B (32 adds), split of the 16 values (24): 104 x 4 = 416 VALU per lane and chunk,
//   32 ds_read_b64 of the raw (hi | lo) image and 32 ds_write_b64 of transformed operands -- issued as real instructions on dummy registers;
// bit 1: the raw image by LDS-DMA from a large buffer (41 pieces of 1 KB per chunk and workgroup, distinct addresses: HBM traffic);
// bit 2: an epilogue per item: A^T m A (24 adds per output tile-channel), scale + PReLU + split (about 10 per value), 12 16-byte stores per lane
template <int FS, int BKB, int MIX = 0>
__global__ __launch_bounds__(512, 2) void wino_mfma_phase(const char* filt, float* out, int items, int steps_per_item, const char* img, char* dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = FS * FREQ_BYTES;
    constexpr int F_BASE = BKB * 1024;
    constexpr int RAW_BASE = F_BASE + 3 * SLOT;                 // MIX bit 1: one raw image buffer of 41 KB behind the ring (a real kernel wants two)
    constexpr int PIECES = FS * CT * 2;                         // 1 KB DMA pieces of a slot
    constexpr int F = (PIECES + 7) / 8;                         // DMA instructions per wave and slot
    constexpr int LDS_TOP = RAW_BASE + ((MIX & 2) ? 41 * 1024 : 0) + ((MIX & 1) ? 8192 : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = wave & 3, cgrp = wave >> 2;                // column tile, and which three cout tiles
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned f_off = (unsigned)(lane * 16);
    // finite, non-trivial B operands (random bits keep the pipe's power state honest)
    for (int i = tid; i < BKB * 64; i += 512) {
        unsigned s = (unsigned)i * 2654435761u;
        u32x4 v;
        s = s * 1664525u + 1013904223u; v.x = (s & 0x03ff03ffu) | 0x38003800u;
        s = s * 1664525u + 1013904223u; v.y = (s & 0x03ff03ffu) | 0x38003800u;
        s = s * 1664525u + 1013904223u; v.z = (s & 0x03ff03ffu) | 0x38003800u;
        s = s * 1664525u + 1013904223u; v.w = (s & 0x03ff03ffu) | 0x38003800u;
        *reinterpret_cast<u32x4*>(smem + i * 16) = v;
    }
    __syncthreads();
    auto dma_slot = [&](int step, int slot) DCSCN_INL {
        const char* src = filt + (size_t)(step & 63) * SLOT;    // 64 slots of filter data, cycled: L2 resident like a layer's filters
        static_for<0, F>([&](auto r_) DCSCN_INL {
            constexpr int r = decltype(r_)::value;
            const int piece = (wave + 8 * r) % PIECES;
            glds16c(src + piece * 1024, f_off, lds0 + F_BASE + slot * SLOT + (unsigned)piece * 1024u);
        });
    };
    f32x4 acc[16][3];
    static_for<0, 16>([&](auto f_) DCSCN_INL { static_for<0, 3>([&](auto n_) DCSCN_INL { acc[decltype(f_)::value][decltype(n_)::value] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }); });
    const int total = items * steps_per_item;
    dma_slot(0, 0);
    dma_slot(1, 1);
    for (int step0 = 0; step0 < total; step0 += 16 / FS) {
        static_for<0, 16 / FS>([&](auto s_) DCSCN_INL {
            constexpr int s = decltype(s_)::value;                // this step covers frequencies s * FS .. + FS - 1 of the chunk
            const int step = step0 + s;
            // counted wait: this slot's pieces were issued two steps ago; younger: that step's image piece, the last step's F pieces and image piece
            constexpr int SC = 16 / FS;
            constexpr int img1 = (MIX & 2) && ((s + SC - 1) % SC) * 8 < 41 ? 1 : 0, img2 = (MIX & 2) && ((s + SC - 2) % SC) * 8 < 41 ? 1 : 0;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(F + img1 + img2) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            c3p_barrier();
            __builtin_amdgcn_sched_barrier(0);
            dma_slot(step + 2, (step + 2) % 3);
            constexpr int STEPS_CHUNK = 16 / FS;
            if constexpr ((MIX & 2) != 0) {
                // 41 image pieces per chunk over STEPS_CHUNK steps x 8 waves: one per wave and step while they last (waves without one repeat piece 40)
                constexpr int first = s * 8;
                if constexpr (first < 41) {
                    const int piece = first + wave < 41 ? first + wave : 40;
                    const size_t chunk_id = (size_t)blockIdx.x * 4096 + (size_t)(step / STEPS_CHUNK);
                    glds16c(img + (chunk_id % 6144) * 43008 + piece * 1024, f_off, lds0 + RAW_BASE + (unsigned)piece * 1024u);
                }
            }
            if constexpr ((MIX & 1) != 0) {
                constexpr int NV = 416 / STEPS_CHUNK, NL = 32 / STEPS_CHUNK;      // VALU ops, ds_read_b64 and ds_write_b64 of this step
                float t0 = 1.0f, t1 = 2.0f, t2 = 3.0f, t3 = 4.0f;
                u32x2 rv[NL];
                const unsigned ra = lds0 + (unsigned)(lane * 8 + (s & 3) * 4096);
#pragma unroll
                for (int k = 0; k < NL; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(rv[k]) : "v"(ra + (unsigned)k * 512u));
                // 4 independent chains; a third of the ops are conversions (v_cvt_f32_f16 / v_cvt_pk_f16_f32 / v_fma_mix), the rest adds
#pragma unroll
                for (int k = 0; k < NV / 4; ++k) {
                    if (k % 3 == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_f32_f16 %1, %1\n\tv_fma_mix_f32 %2, %2, %3, %2\n\tv_cvt_f32_f16 %3, %3" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                    else asm volatile("v_add_f32 %0, %0, %1\n\tv_sub_f32 %1, %1, %2\n\tv_add_f32 %2, %2, %3\n\tv_sub_f32 %3, %3, %0" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned wa_ = lds0 + (unsigned)(LDS_TOP - 8192 + lane * 8);
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    u32x2 w = rv[k];
                    w.x ^= __builtin_bit_cast(unsigned, t0);
                    // (a dummy 8 KB window at the top of LDS: nothing the MFMAs read is overwritten)
                    asm volatile("ds_write_b64 %0, %1" :: "v"(wa_ + (unsigned)k * 512u), "v"(w) : "memory");
                }
            }
            const char* fs = smem + F_BASE + (step % 3) * SLOT + lane * 16 + cgrp * 3 * 2048;
            static_for<0, FS>([&](auto i_) DCSCN_INL {
                constexpr int i = decltype(i_)::value, f = s * FS + i;
                // B fragments of frequency f for this wave's column tile: the region holds 2 x 4 frequencies (the double buffer of a real kernel)
                const char* bsrc = smem + ((f % (BKB / 8)) * 4 + ncol) * 2048 + lane * 16;
                const h8 xh = *reinterpret_cast<const h8*>(bsrc);
                const h8 xl = *reinterpret_cast<const h8*>(bsrc + 1024);
                static_for<0, 3>([&](auto n_) DCSCN_INL {
                    constexpr int n = decltype(n_)::value;
                    const h8 wh = *reinterpret_cast<const h8*>(fs + i * FREQ_BYTES + (2 * n) * 1024);
                    const h8 wl = *reinterpret_cast<const h8*>(fs + i * FREQ_BYTES + (2 * n + 1) * 1024);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[f][n], 0, 0, 0);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[f][n], 0, 0, 0);
                    acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[f][n], 0, 0, 0);
                });
            });
            if constexpr ((MIX & 4) != 0) {
                if ((step + 1) % steps_per_item == 0) {
                    // A^T m A per (cout, column) tile pair and register position: 16 frequency values -> 2 x 2 outputs (24 adds), then scale, PReLU and a
                    // split-like pack (6 ops per value), 4 stores of 16 bytes per pair to distinct addresses
                    char* o = dst + ((size_t)blockIdx.x * 64 + (size_t)((step / steps_per_item) & 63)) * 98304 + (size_t)tid * 16;
                    static_for<0, 3>([&](auto n_) DCSCN_INL {
                        constexpr int n = decltype(n_)::value;
                        f32x4 r[4][4];
                        static_for<0, 4>([&](auto i_) DCSCN_INL {          // rows: t = m0 + m1 + m2, m1 - m2 - m3 on each of the 4 columns
                            constexpr int i = decltype(i_)::value;
                            r[0][i] = acc[i][n] + acc[4 + i][n] + acc[8 + i][n];
                            r[1][i] = acc[4 + i][n] - acc[8 + i][n] - acc[12 + i][n];
                        });
                        f32x4 y[4];
                        y[0] = r[0][0] + r[0][1] + r[0][2]; y[1] = r[0][1] - r[0][2] - r[0][3];
                        y[2] = r[1][0] + r[1][1] + r[1][2]; y[3] = r[1][1] - r[1][2] - r[1][3];
                        static_for<0, 4>([&](auto i_) DCSCN_INL {
                            constexpr int i = decltype(i_)::value;
                            f32x4 v = y[i] * 0.25f + f32x4{0.1f, 0.2f, 0.3f, 0.4f};
                            v.x = v.x > 0.0f ? v.x : 0.2f * v.x; v.y = v.y > 0.0f ? v.y : 0.2f * v.y; v.z = v.z > 0.0f ? v.z : 0.2f * v.z; v.w = v.w > 0.0f ? v.w : 0.2f * v.w;
                            h4 hi, lo;
                            split4(v, -1.0f, hi, lo);
                            const u32x2 hu = __builtin_bit_cast(u32x2, hi), lu = __builtin_bit_cast(u32x2, lo);
                            *reinterpret_cast<u32x4*>(o + (n * 4 + i) * 8192) = u32x4{hu.x, hu.y, lu.x, lu.y};
                        });
                        static_for<0, 16>([&](auto f_) DCSCN_INL { acc[decltype(f_)::value][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; });
                    });
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<0, 16>([&](auto f_) DCSCN_INL { static_for<0, 3>([&](auto n_) DCSCN_INL { sum += acc[decltype(f_)::value][decltype(n_)::value]; }); });
    out[(size_t)blockIdx.x * 512 + tid] = sum.x + sum.y + sum.z + sum.w;
}

static const char* g_img = nullptr;
static char* g_dst = nullptr;
template <int FS, int BKB, int MIX = 0>
static float run(const char* d_f, float* d_o, int wgs, int items, int chunks_x4) {
    auto k = wino_mfma_phase<FS, BKB, MIX>;
    constexpr int lds = BKB * 1024 + 3 * FS * FREQ_BYTES + ((MIX & 2) ? 41 * 1024 : 0) + ((MIX & 1) ? 8192 : 0);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int steps_per_item = chunks_x4 * 4 / FS;               // chunks_x4 = quarter chunks (4 frequencies each)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), lds, 0, d_f, d_o, items, steps_per_item, g_img, g_dst);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = ms < best ? ms : best;
    }
    CK(hipGetLastError());
    return best;
}

int main() {
    const size_t fbytes = (size_t)64 * 4 * FREQ_BYTES;
    std::vector<unsigned> h(fbytes / 4);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x03ff03ffu) | 0x38003800u; }
    char* d_f; float* d_o;
    CK(hipMalloc(&d_f, fbytes)); CK(hipMalloc(&d_o, 256 * 512 * 4));
    CK(hipMemcpy(d_f, h.data(), fbytes, hipMemcpyHostToDevice));
    // CNN2, one channel group of 6 tiles: 9216 pixel tiles over 256 CUs = 36 items; 196 channels = 6 chunks + one octet (its 16 frequencies
    // packed four to an instruction = a quarter chunk): 25 quarter chunks per item
    const int items = 36, q = 25;
    const double mfmas = 256.0 * items * q * 4 * 6 * 4 * 3;      // whole launch
    printf("# Winograd F(2x2,3x3) MFMA phase only, CNN2 group of 6 cout tiles (the layer has 6 + 5: x 11/6 for the layer), 36 items x 25 quarter chunks per CU\n");
    printf("# floor at 16 cycles per MFMA and 1.76 GHz (conv3_h8's clock on this layer): %.3f ms for the group\n", mfmas / 1024.0 * 16 / 1.76e9 * 1e3);
    const float t1 = run<1, 64>(d_f, d_o, 256, items, q);
    printf("FS 1 (one frequency per barrier,  ring 36 KB + B 64 KB)    %.3f ms  -> layer (x 11/6) %.3f ms\n", t1, t1 * 11.0 / 6.0);
    const float t2 = run<2, 64>(d_f, d_o, 256, items, q);
    printf("FS 2 (two frequencies per barrier, ring 72 KB + B 64 KB)   %.3f ms  -> layer (x 11/6) %.3f ms\n", t2, t2 * 11.0 / 6.0);
    const float t4 = run<4, 16>(d_f, d_o, 256, items, q);
    printf("FS 4 (four per barrier, ring 144 KB: B region cut to 16 KB -- no room for a real kernel's operands)  %.3f ms  -> layer %.3f ms\n", t4, t4 * 11.0 / 6.0);
    // (B region 32 KB here: 2 x 2 frequencies -- B 32 + ring 72 + one raw image 41 + the 8 KB dummy write window = 153 KB; a second raw image does not fit)
    // the rest of a real kernel, as instruction mix (header): raw image by LDS-DMA from 264 MB of distinct addresses, transform mix, epilogue
    char* d_img;
    CK(hipMalloc(&d_img, (size_t)6144 * 43008 + 65536)); CK(hipMemset(d_img, 0x3a, (size_t)6144 * 43008 + 65536));
    CK(hipMalloc(&g_dst, (size_t)256 * 64 * 98304)); g_img = d_img;
    const float m1 = run<2, 32, 1>(d_f, d_o, 256, items, q);
    printf("FS 2 + transform mix (416 VALU, 32 ds_read_b64, 32 ds_write_b64 per lane and chunk)            %.3f ms  -> layer %.3f ms\n", m1, m1 * 11.0 / 6.0);
    const float m3 = run<2, 32, 3>(d_f, d_o, 256, items, q);
    printf("FS 2 + transform mix + image staging by LDS-DMA (41 KB per chunk from HBM)                     %.3f ms  -> layer %.3f ms\n", m3, m3 * 11.0 / 6.0);
    const float m7 = run<2, 32, 7>(d_f, d_o, 256, items, q);
    printf("FS 2 + transform mix + image staging + epilogue (output transform, PReLU, split, 12 stores)    %.3f ms  -> layer %.3f ms\n", m7, m7 * 11.0 / 6.0);
    const float m6 = run<2, 32, 6>(d_f, d_o, 256, items, q);
    printf("FS 2 + image staging + epilogue, NO transform mix                                              %.3f ms  -> layer %.3f ms\n", m6, m6 * 11.0 / 6.0);
    printf("# conv3_h8 on CNN2 (whole kernel, both groups, P16 staging, epilogue, HBM traffic): 3.09 - 3.20 ms (profiles/r06_per_launch.txt)\n");
    return 0;
}
