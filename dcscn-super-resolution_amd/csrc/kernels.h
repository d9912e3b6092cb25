// Internal interface between the plan/ABI layer (api.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace dcscn {

// Activation codes as the kernels see them (build_activator, helper/tf_graph.py:77-102).
// prelu / relu / leaky_relu all become ACT_ALPHA with a per-channel negative slope.
enum { ACT_NONE = 0, ACT_ALPHA = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_SELU = 4 };

// Pre-split activation tensor ("P16", p16.hpp): what a split16 kernel consumes, stored once in the form it consumes it.  Per 32-channel
// chunk one plane [zero record of 128 bytes][pixel records]; a record holds, per channel octet of the chunk, the 8 `hi` halfs and the
// 8 `lo` halfs of split16.hpp as two 16-byte units: 128 bytes for a full chunk, 32 * octets for the last one.  Same 4 bytes per value
// as float32, every record of a full chunk is one aligned 128-byte line.
struct P16Desc {
    char*     base;   // plane of chunk 0 (nullptr: the tensor is float32 NHWC)
    long long plane;  // bytes between the planes of consecutive chunks
    int32_t   octs;   // channel octets of the tensor = ceil(channels / 8); chunk c holds octets [4c, 4c + 4)
    int32_t   pad_;
};
__host__ __device__ constexpr int p16_rec_bytes(int octs, int chunk) { return octs - 4 * chunk >= 4 ? 128 : 32 * (octs - 4 * chunk); }
__host__ __device__ constexpr long long p16_plane_bytes(long long npix) { return ((npix + 1) * 128 + 255) & ~255LL; }
__host__ __device__ constexpr long long p16_tensor_bytes(long long npix, int octs) {
    return ((octs + 3) / 4 - 1) * p16_plane_bytes(npix) + ((128 + npix * p16_rec_bytes(octs, (octs + 3) / 4 - 1) + 255) & ~255LL);
}
// pixels per pass a P16 tensor can hold: record offsets are 32-bit (128 bytes * (pixel + 1))
constexpr long long kP16MaxPixels = (1LL << 25) - 2;

struct OutDesc {
    float*  ptr;      // base of the destination tensor (NHWC, possibly a wider concat buffer)
    int32_t stride;   // floats per pixel in the destination
    int32_t off;      // first channel of the slice written
    int32_t width;    // number of conv output channels stored through this descriptor
    P16Desc p16;      // base != nullptr: the destination is a P16 tensor (off = first channel, a multiple of 16); ptr / stride unused
};

struct NinSrcQuad {           // conv_nin multi-source input: one 16-byte channel quad of the virtual concat
    unsigned long long ptr;   // address of this quad of pixel 0
    unsigned stride;          // bytes between pixels of the source tensor
    unsigned valid;           // 0: padding quad past the last channel
};

// Arguments of the implicit-GEMM convolution (tf.nn.conv2d SAME stride 1 + bias + activator,
// helper/tf_graph.py:104-153; optional depth_to_space and residual add folded into the store).
struct ConvArgs {
    const float* in;          // NHWC source, [n, H, W, in_stride]
    P16Desc in16;             // base != nullptr: the source is a P16 tensor read from channel 0 (split16 kernels only); in / in_stride unused
    int32_t in_stride;
    int32_t in_off;           // first channel of the slice read (multiple of 4)
    int32_t cin_phys;         // physical channels read (multiple of 4)
    int32_t n_chunks;         // ceil(cin_phys / KC)
    const float* wpack;       // [n_tiles][n_chunks][taps * KC * NS] repacked filters
    const float* bias;        // [n_tiles * NT * 16], zero padded
    const float* alpha;       // same length (negative slope); ignored unless act == ACT_ALPHA
    int32_t act;
    int32_t N, H, W;          // batch and spatial size the conv runs at
    int32_t tiles_x, tiles_y;
    OutDesc out0, out1;       // conv channel c goes to out0 when c < split, else to out1 (c - split)
    int32_t split;            // multiple of 16; >= padded channel count when there is one output
    int32_t ps;               // depth_to_space block (1 = none), tf_graph.py:248
    int32_t ps_c;             // channels after depth_to_space
    int32_t vec4;             // 1: every 4-channel group may be stored as one float4
    const float* res;         // optional residual ([n, H*ps, W*ps, res_stride]) added before the store
    int32_t res_stride;
    // fused depthwise stage of tf.nn.separable_conv2d (1x1 kernels only): the staged input element is
    // sum_t in(p + o_t)[c] * dww[t][c] instead of in(p)[c]
    const float* dww;         // [dwk*dwk][cin_phys], physical channel order, zero padded (or nullptr)
    int32_t dwk;              // depthwise kernel size: 0 (none), 1 or 3
    int32_t n_full;           // conv_wino2: channel groups [0, n_full) hold NT 16-channel tiles, the others NT - 1
    int32_t n_groups;         // conv_wino2: channel groups of the launch (the kernel decodes (pixel tile, group) from a 1-D grid)
    int32_t group_span;       // conv_wino2: how many consecutive groups share the XCD-interleaved id range (divides the work in phases)
    // Folded linear tail (5x5 kernels): the launch computes, for every LR pixel, `ps*ps` sub-pixel phases
    // x 4 border variants of the composite [pixel-shuffler conv -> depth_to_space -> 3x3 conv to 1 channel];
    // conv channel v = phase * 4 + variant.  The epilogue picks the variant of each phase from the pixel's
    // position and stores one value per HR pixel into out0 (stride 1), plus `res`.
    // fold == 2 (graph.hip: fold_whole_tail): conv channel = sub-pixel phase of the WHOLE tail (ps = total scale, ps^2 <= 16: one tile); the launch
    // stores the pixels that are NOT on the image's border ring with the interior kernel, c5h_border_launch computes the ring.
    int32_t fold;
    const void* srctab;       // conv_nin, multi-source input: device array of NinSrcQuad, 4 * n_chunks entries (nullptr: `in` is one tensor)
    // split16 kernels (split16.hpp: f32-accurate contraction on the f16 matrix pipe)
    const void* wpack16;      // filters as f16 (hi, lo) fragments (split16_pack.hpp), scaled by 2^e
    float inv_scale;          // 2^-e
    int32_t* redo;            // [0] = pass flag, [1 + image]: a split16 kernel raises both when an image cannot stay on the f16 pipe (a
                              // non-finite accumulator = an input beyond the f16 range, or a P16 output beyond it)
    int32_t tail_octs;        // conv3_h: 0, or 1 / 2 / 3 = channel octets of the packed last chunk (c3h_tail_octs)
    int32_t redo_check;       // float32 kernels: 1 = the float32 plan behind a split16 pass -- only the units of flagged images are computed
    int32_t nt_pack;          // conv3_h8: channel tiles per group in the wpack16 image (the half of a workgroup may take fewer)
};

struct ConvShape {            // kernel variant picked by the plan
    int ks, mt, nt, kc;
    int dwk = 0;              // fused depthwise kernel size (ks == 1 only)
    int nin = 0;              // 1: LDS-DMA staged 1x1 GEMM conv_nin (ks == 1, nt <= 6, kc == 16, 256 flat pixels per workgroup)
    int wino = 0;             // 1: Winograd F(2x2,3x3) kernel conv_wino2 (ks == 3, nt <= 3, kc == 8, 16x16 pixel tiles)
};

// Geometry helpers shared by the weight packer (host) and the kernels (device).
__host__ __device__ constexpr int conv_ns(int nt) { return (nt & 1) ? nt * 16 : nt * 16 + 16; }
__host__ __device__ constexpr int conv_plane_stride(int halo_pixels) {
    // smallest value >= halo_pixels that is 16 (mod 32): two k-planes read by one 32-lane LDS group
    // then fall on disjoint bank halves
    return ((halo_pixels + 15) / 32) * 32 + 16;
}

// Picks (mt, nt, kc) for a conv with `cout_padded16` output channels per tile.
ConvShape conv_pick_shape(int ks, int nt, int dwk = 0);
size_t conv_lds_bytes(const ConvShape& s);
// One-time: raise the dynamic-LDS limit of every instantiated kernel. Returns hipSuccess or error.
hipError_t conv_init_kernels();
hipError_t conv_launch(const ConvShape& s, const ConvArgs& a, int n_tiles, hipStream_t stream);
// Winograd F(2x2,3x3) variant of a 3x3 conv (conv_wino2.hpp): `nt` channel tiles of 16 per group (1..3), `n_groups`
// groups of which the first args.n_full hold nt tiles and the others nt - 1 (tiles are spread evenly), args.wpack in the
// [group][chunk of 8 channels][16 f][(c & 1) * 4 + (c >> 1)][conv_ns(nt)] image -- the exact LDS image of a chunk.
constexpr int kWinoKC = 8;
constexpr int kWinoMaxNT = 3;
hipError_t wino_init_kernels();
hipError_t wino_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream);
// 1x1 convs as a GEMM over the flat pixel list (conv_nin.hpp): `nt` channel tiles of 16 per group (1..6), groups as for
// wino_launch (args.n_full wide ones), args.wpack in the [group][chunk of 16 channels][(c & 3) * 4 + (c >> 2)][conv_ns(nt)] image.
constexpr int kNinKC = 16;
constexpr int kNinMaxNT = 6;
hipError_t nin_init_kernels();
hipError_t nin_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream);
// ---- split16 kernels (split16.hpp): f32-accurate contractions on the f16 matrix pipe ----
// conv_nin_h.hpp: conv_nin's job with 32-channel chunks on v_mfma_f32_16x16x32_f16; args.wpack16 = pack_conv16 image with one tap
// (groups as for nin_launch), args.n_chunks = ceil(cin_phys / 32), args.inv_scale, args.redo (one flag per 256 pixels);
// a multi-source table holds 8 entries per chunk and every entry points at readable memory (invalid ones with stride 0).
constexpr int kNinHKC = 32;
hipError_t nin_h_init_kernels();
hipError_t nin_h_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream);
// conv3_h.hpp: 3x3 conv + bias + activator (+ depth_to_space) as a direct implicit GEMM; `nt` tiles per group (1..5), groups as for
// wino_launch; args.wpack16 = pack_conv16 image with 9 taps, args.n_chunks = ceil(cin_phys / 32), args.redo (one flag per 16x16 tile)
constexpr int kC3hKC = 32;
constexpr int kC3hMaxNT = 6;
// conv3_h's packed last chunk: when the last 32-channel chunk holds at most 8 / 16 / 24 physical channels (1 / 2 / 3 octets), its
// (tap, octet) pairs are packed four to a K = 32 instruction -- ceil(9 octets / 4) = 3 / 5 / 7 MFMA steps instead of 9.
// Returns the octets (0 = plain chunk); needs a full chunk in front of it.
inline int c3h_tail_octs(int cin_phys) {
    const int n_chunks = (cin_phys + kC3hKC - 1) / kC3hKC, tail = cin_phys - (n_chunks - 1) * kC3hKC;
    return n_chunks < 2 || tail > 24 ? 0 : (tail + 7) / 8;
}
inline int c3h_tail_steps(int octs) { return (9 * octs + 3) / 4; }
hipError_t c3h_init_kernels();
hipError_t c3h_launch(int nt, const ConvArgs& a, int n_groups, hipStream_t stream);
// conv3_h8 (conv3_h8.hpp): conv3_h's launches with exactly two channel groups as ONE persistent 8-wave workgroup per CU -- the pixel
// tile's input image staged once for both groups, the halves running their load and compute parts in opposite order between one barrier per tap; same wpack16 image, same arguments, bit-identical results
hipError_t c3e_init_kernels();
bool c3e_eligible(int nt, const ConvArgs& a, int n_groups);
hipError_t c3e_launch(int nt, const ConvArgs& a, int n_groups, int n_cus, hipStream_t stream);
// conv5_h (conv5_h.hpp): the folded 5x5 tail on the f16 pipe; nt = ceil(4 ps^2 / 16) in {1, 3, 4}, one channel group, args as conv_launch's
// fold launch plus args.wpack16 = pack_conv16 image with 25 taps, args.n_chunks = ceil(cin_phys / 32), args.inv_scale, args.redo
hipError_t c5h_init_kernels();
hipError_t c5h_launch(int nt, const ConvArgs& args, hipStream_t stream);
// the border ring of a fold == 2 launch (conv5_h.hpp: fold_border): args.wpack16 = 16 pack_conv16 images [variant = 4 vy + vx][chunk][25 taps][hi | lo],
// args.bias = [16 variants][16 phases]; vy / vx: 0 interior, 1 first row / column, 2 last, 3 both (a one-pixel axis)
constexpr int kFoldVariants = 16;
hipError_t c5h_border_launch(const ConvArgs& args, hipStream_t stream);

// ---- row-streamed feature extractor of the separable narrow nets (feat_stream.hpp) ----
constexpr int kStreamPX = 48;                  // computed columns per strip: three 16-pixel MFMA tiles
constexpr int kStreamRowPx = kStreamPX + 2;    // + one zero column on each side
constexpr int kStreamMaxL = 7;                 // feature layers (2L + 1 waves <= 16)
constexpr int kStreamMT = kStreamPX / 16;

struct StreamRing {
    int32_t off;      // LDS byte offset of [slots][kStreamRowPx][units] float4
    int32_t units;    // 16-byte units per pixel (odd)
    int32_t quads;    // units that hold channels: pad4(C) / 4
    int32_t slots;    // rows in the ring: 3 (written after the step's barrier) or 4 (written while its readers run)
};
struct StreamConv {   // separable 3x3 layer of the stream
    StreamRing in, out;
    int32_t lag;          // computes stream row t - lag at step t
    int32_t dww;          // LDS byte offset of the depthwise filter [9][in.quads] float4
    int32_t wp;           // LDS byte offset of the pointwise filter [chunk][tile][64 lanes] float4 (k-steps 0..3)
    int32_t ba;           // LDS byte offset of bias[32], slope[32]
    int32_t to_global;    // 1: the layer stores to `out` channels [0, 4 * out.quads) instead of a ring (B2)
    float inv;            // F16 kernels: 2^-e of the pointwise filter's scale (the bias in the f16 image is multiplied by 2^e)
};
struct StreamNinSrc {     // one feature layer as a K-slice of A1 || B1
    StreamRing ring;
    int32_t chunks;       // 16-channel chunks
    int32_t w;            // LDS byte offset of [chunk][2 tiles][64 lanes] float4 (k-steps 0..3)
    int32_t last_ql;      // valid channel quads of the last chunk (1..4): feat_stream.hpp StreamChunk
};
struct StreamArgs {
    const float* x;       // [N, H, W] luma
    float* out;           // Concat2 [N, H, W, out_stride]: B2 at channel 0, A1 at channel 4 * nb_quads
    const float* blob;    // packed parameters (api.hip: pack_feat_stream)
    int32_t out_stride;
    int32_t N, H, W;
    int32_t n_strips, useful_w, halo;      // strip s computes columns [s * useful_w - halo, .. + 48), stores [s * useful_w, (s + 1) * useful_w)
    int32_t n_blocks, useful_h, rows_c;    // row block b computes rows_c rows from b * useful_h - halo (0 when n_blocks == 1)
    int32_t n_jobs, jobs_per_wg;
    int32_t L, n_conv, total_lag;
    long long* dbg;                        // timing probe (STREAM_DBG builds, tools/stream_abl.sh): [wave][step][4] shader clocks of workgroup 0
    int8_t role[16];                       // wave -> 0: CNN1, 1 .. n_conv: conv[role - 1], 16 + i: A1 || B1 row slot i (balanced over the 4 SIMDs)
    int32_t ring_bytes, ldsw_bytes, ldsw_src;   // LDS image: [0, ring_bytes) zero, then ldsw_bytes copied from blob + ldsw_src
    int32_t first_w;                       // blob offset of CNN1: depthwise[9 (+3)], pointwise[32], bias[32], slope[32]
    StreamRing first_out;
    StreamConv conv[kStreamMaxL];          // CNN2 .. CNNL, B2
    StreamNinSrc nin[kStreamMaxL];
    StreamRing b1;
    int32_t nin_ba;                        // LDS byte offset of bias[32], slope[32] of [B1 | A1]
    int32_t nb_quads;                      // channel quads of the B1 part
    float nin_inv;                         // F16 kernel: 2^-e of the A1 || B1 filters' scale
    int32_t* redo;                         // [0] pass flag, [1 + image] (split16.hpp): raised by the F16 kernel on a non-finite output
    int32_t redo_check;                    // float32 kernel as the float32 plan: only the flagged images are stored
};


// ---- row-streamed x4 upsampler + reconstruction of the separable narrow nets (tail_stream.hpp) ----
struct TailArgs {
    const float* c2;      // Concat2 [N, H, W, c2_stride]
    const float* x2;      // bicubic image [N, 4H, 4W]
    float* y;             // [N, 4H, 4W]
    const float* blob;    // LDS image of the filters (api.hip: pack_tail_stream)
    int32_t c2_stride;
    int32_t N, H, W;
    int32_t n_strips, useful_w, halo;
    int32_t n_blocks, useful_h, rows_c;
    int32_t n_jobs, jobs_per_wg;
    StreamRing in, u;     // IN: 3 rows x 50 pixels of Concat2; U: 6 rows x 98 pixels of the first depth_to_space output
    int32_t v_off;        // V: 12 rows x 196 floats, the second depth_to_space output (1 channel)
    int32_t ring_bytes, ldsw_bytes;
    int32_t a_dww, a_wp, a_bias;    // Up-PS: depthwise [9][in.quads] float4, pointwise [4 phases][2][2][64] float4, bias [4 phases][8] float4
    int32_t b_dww, b_wp, b_bias;    // Up-PS2: depthwise [9][u.quads] float4, pointwise [2][1][64] float4, bias float4
    float c_w[9];         // R-CNN1 depthwise filter
    float c_scale;        // R-CNN1 pointwise scalar
    float a_inv, b_inv;   // F16 kernel: 2^-e of the Up-PS / Up-PS2 pointwise filters' scales
    int32_t* redo;        // as StreamArgs
    int32_t redo_check;
};
// f16 = the F16 instantiation (a.blob = the f16 image of the filters: pack.hip)
hipError_t tail_launch(const TailArgs& a, int grid, bool f16, hipStream_t stream);
void stream_init_kernels();
hipError_t stream_launch(const StreamArgs& a, int grid, bool f16, hipStream_t stream);

// ---- row-streamed feature extractor of the NON-separable narrow nets (feat3_stream.hpp) ----
constexpr int kS3MaxL = 8;         // feature layers
constexpr int kS3MaxWaves = 8;     // CNN1 + one per conv
constexpr int kS3RolePair = 32, kS3RoleNin = 16;      // role codes beyond the conv indices (feat3_stream.hpp)
struct S3Ring { int32_t off, px, octs; };              // LDS byte offset of [4 slots][kStreamRowPx][px bytes] P16 units; px = (2 octs + 1) * 16 (0: no ring)
struct S3Out {                     // a global tensor of the launch: P16 (p16.base != nullptr), float32 NHWC (ptr != nullptr) or none (both null: not stored)
    P16Desc p16;
    float* ptr;
    int32_t stride, width;         // floats per pixel, stored channels (multiple of 4)
    int32_t lo, hi;                // conv channels [lo, hi) of the writer are stored, at the same channel index of the tensor (multiples of 8; hi <= width)
};
struct S3Nin {                     // A1 || B1 accumulated inside the launch (feat3_stream.hpp: s3_nin_role), B2 behind it
    int32_t on;                    // 0: the layers' rows go to global memory and conv_nin_h reads them (r05)
    int32_t w_off;                 // blob offset (floats) of the fragments [layer][tile][hi | lo][64 lanes][8 halfs]: K = the layer's <= 4 octets, zero beyond; x 2^e
    int32_t ba_off;                // blob offset of bias * 2^e [32], slope - 1 [32], channel order [B1 (8) | A1 (24)]
    float inv;                     // 2^-e
    S3Ring b1;                     // ring of B1 rows (one octet)
};
struct S3Conv {
    S3Ring in, out;
    int32_t lag;                   // computes stream row t - lag at step t
    int32_t w_off;                 // blob offset (floats) of the filter fragments [step][tile][hi | lo][64 lanes][8 halfs], scaled by 2^e
    int32_t ba_off;                // blob offset of bias * 2^e [32], slope - 1 [32]
    int32_t tiles;                 // 16-channel output tiles
    float inv;                     // 2^-e
};
struct Stream3Args {
    const float* x;                // [N, H, W] luma
    const float* blob;             // packed parameters (pack.hip: pack_feat3_stream)
    int32_t N, H, W;
    int32_t n_strips, useful_w, halo, n_blocks, useful_h, rows_c, n_jobs, jobs_per_wg;      // as StreamArgs
    int32_t L, total_lag, n_waves;
    int8_t role_conv[kS3MaxWaves]; // wave -> -1: CNN1, 0 .. L - 2: that conv (0 = CNN2); with nin.on also kS3RolePair + 0 (conv[L - 3] and conv[L - 2] in one wave),
                                   // kS3RolePair + 1 (conv[L - 4] and B2 = conv[L - 1]) and kS3RoleNin + n (A1 || B1, output tile n)
    int8_t role_tile[kS3MaxWaves]; // (unused)
    int32_t first_w;               // blob offset of CNN1: filter [9][32], bias [32], slope - 1 [32]
    S3Ring first_out;
    S3Conv conv[kS3MaxL];
    S3Out out[kS3MaxL];            // out[0] = CNN1's tensor, out[i + 1] = conv i's (nin.on: none for the layers, out[L] = Concat2 channels [0, 8) for B2)
    S3Nin nin;
    S3Out out2;                    // nin.on: Concat2 [B2 | A1], channels [8, 32) written by the A1 || B1 waves
    int32_t ring_bytes;
    int32_t* redo;                 // [0] pass flag, [1 + image] (split16.hpp)
    long long* dbg;                // S3_DBG builds (tools/s3_abl.sh probe): per wave of workgroup 0: [0] cycles in compute, [1] cycles waiting at the barrier, [2] steps
};
hipError_t stream3_launch(const Stream3Args& a, int grid, hipStream_t stream);

// widest channel tile (in units of 16) the fused-depthwise pointwise kernels are instantiated for
int conv_max_fused_dw_nt();

// self-ensemble gather / float64 reduce (ensemble.hip); images of one call all have h*w pixels, types 4-7 transposed
hipError_t ensemble_gather_launch(const float* in, float* out, int h, int w, int n, hipStream_t stream);
hipError_t ensemble_reduce_launch(const float* y, double* out, int h, int w, int n, hipStream_t stream);

// colour conversions of helper/utilty.py:142-193 in float64 (color.hip); device pointers
hipError_t rgb_to_y_launch(const uint8_t* rgb, double* y64, float* y32, long long n, hipStream_t stream);
hipError_t rgb_to_ycbcr_launch(const uint8_t* rgb, double* out, long long n, hipStream_t stream);
hipError_t y_cbcr_to_rgb_launch(const double* y64, const float* y32, const double* cbcr, const uint8_t* rgb8, double* out, long long n,
                                hipStream_t stream);

// Pillow-compatible bicubic resize of 1-channel float images (resample.hip)
int resample_coeffs(int in_size, int out_size, std::vector<int>* bounds, std::vector<double>* kk);   // returns ksize
hipError_t resample_h_launch(const float* in, float* out, const int* bounds, const double* kk, int ksize,
                             long long rows, int w, int ow, hipStream_t stream);
hipError_t resample_v_launch(const float* in, float* out, const int* bounds, const double* kk, int ksize,
                             int n_img, int h, int oh, int w, hipStream_t stream);
// widest channel tile (units of 16) of the conv_igemm<ks, ...> family
int conv_max_nt(int ks);
// LDS bytes of conv_cin1 / conv_cout1 for a kernel size (both must fit 64 KB)
inline size_t cin1_lds_bytes(int ks, int cs) { const int ht = 16 + 2 * (ks / 2); return (size_t)(((ht * ht + 3) & ~3) + (ks * ks + 2) * cs) * sizeof(float); }
inline size_t cout1_lds_bytes(int ks, int cin_phys) { const int ht = 16 + 2 * (ks / 2); return (size_t)(ks * ks * cin_phys + ks * ks * ((ht * ht + 3) & ~3)) * sizeof(float); }

// First layer: 3x3 (or 1x1) conv from ONE input channel, direct form (write-bound).
struct Cin1Args {
    const float* x;           // [n, H, W] (channel stride 1)
    const float* w;           // [ks*ks][cs] zero padded
    const float* bias;        // [cs]
    const float* alpha;       // [cs]
    int32_t act;
    int32_t ks;
    int32_t N, H, W;
    int32_t cs;               // stored channels (multiple of 4)
    OutDesc out;              // (out.p16.base != nullptr: a P16 tensor, p16.hpp)
    int32_t* redo;            // [0] pass flag, [1 + image]: raised when a P16 output leaves the f16 range; read when redo_check is set
    int32_t redo_check;       // 1: float32 plan behind a split16 pass -- only the flagged images are computed
};
hipError_t cin1_launch(const Cin1Args& a, hipStream_t stream);

// Last reconstruction conv: k x k conv to ONE output channel, no activator, plus the residual add
// (R-CNN + x2, DCSCN.py:318-325).  HBM bound (AI 4.5): reads the input once, writes one float per pixel.
struct Cout1Args {
    const float* in; int32_t in_stride, in_off, cin_phys;
    const float* w;           // [ks*ks][cin_phys], zero padded, physical channel order
    float scale;              // out = scale * conv + bias (+ res): the 1->1 pointwise half of a separable conv
    float bias;
    int32_t ks;
    int32_t N, H, W;
    float* out; int32_t out_stride;
    const float* res; int32_t res_stride;   // optional residual, same pixel indexing as out
    const int32_t* redo; int32_t redo_check;   // 1: float32 plan behind a split16 pass -- only images with redo[1 + image] set (redo[0]: any)
};
hipError_t cout1_launch(const Cout1Args& a, hipStream_t stream);

// Depthwise k x k SAME, channel multiplier 1 (first half of tf.nn.separable_conv2d, tf_graph.py:161).
struct DwArgs {
    const float* in; int32_t in_stride, in_off;
    const int32_t* chan_map;  // [cin] logical -> physical channel (relative to in_off), device memory
    const float* w;           // [ks*ks][cin]
    int32_t ks, cin, cout_phys;
    int32_t N, H, W;
    float* out; int32_t out_stride;   // writes channels [0, cout_phys): logical then zero padding
    const int32_t* redo; int32_t redo_check;   // as Cout1Args
};
hipError_t depthwise_launch(const DwArgs& a, hipStream_t stream);
// start of a pass that runs split16 launches: clears the n redo flags and the 128-byte zero records of the P16 planes (p16.hpp; zrec =
// device array of their nz addresses, may be null)
hipError_t pass_begin_launch(int32_t* redo, int n, const unsigned long long* zrec, int nz, hipStream_t stream);
// debug (option "debug_poison"): fill the LDS (what & 1) / the vector registers (what & 2) of every CU with NaN patterns
hipError_t debug_poison_launch(int what, hipStream_t stream);
// debug (option "debug_digest"): *out += position-weighted checksum of n 32-bit words
hipError_t debug_digest_launch(const void* p, size_t n_words, unsigned long long* out, hipStream_t stream);

}  // namespace dcscn
