// Internal interface of the plan / ABI layer: the handle, the launch plan (Op) and the functions its translation units share.
//   graph.hip  build_graph (DCSCN.py:222-325 as a list of launches), graph rewrites (folded tail, streamed nets, dense features)
//   pack.hip   filter repacking into the LDS images of the kernels (finalize_op and the streamed-kernel blobs)
//   exec.hip   workspace, launches, sub-batching, spatial tiling, resize
//   api.hip    the extern "C" surface of include/dcscn.h
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dcscn.h"
#include "kernels.h"

namespace dcscn_impl {
using namespace dcscn;

inline int pad4(int c) { return (c + 3) & ~3; }
inline int pad16(int c) { return (c + 15) & ~15; }

enum { EXT_X = -1, EXT_X2 = -2, EXT_Y = -3 };
enum OpKind { OP_CONV = 0, OP_CIN1 = 1, OP_DW = 2, OP_COUT1 = 3, OP_STREAM = 4, OP_TAIL = 5, OP_STREAM3 = 6, OP_FOLDX = 7 };

struct TensorSpec {
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool set = false;
};

struct WsBuf {
    int stride = 0;       // floats per pixel
    int res = 1;          // pixels per LR pixel along one axis
    size_t offset = 0;    // byte offset inside the arena for the current layout
    // P16 (kernels.h: P16Desc; graph.hip: plan_p16): every producer can store and every consumer can read the pre-split form, so the
    // tensor lives in it while the split16 launches run; `p16` = it does in the current carve (options split16 / p16)
    bool p16_ok = false, p16 = false;
    int octs = 0;         // channel octets = ceil(stride / 8)
    long long plane = 0;  // bytes between chunk planes in the current carve
    size_t bytes = 0;     // bytes of the tensor in the current carve
};

// one source block of a launch's filter matrix: conv channels [dst, dst + cout) come from `w`
struct ColSeg {
    int w = -1, b = -1, alpha = -1;   // tensor indices (-1 = absent)
    int cout = 0;                     // output channels taken from the tensors ...
    int col0 = 0;                     // ... starting at this one (a layer split over two launches)
    int dst = 0;
    int dw1 = -1;                     // 1x1 depthwise filter [1, 1, cin, 1] of a separable 1x1 conv, folded into the
                                      // pointwise weights when they are packed: sum_c (x_c d_c) p_co = sum_c x_c (d_c p_co)
};

struct Op {
    OpKind kind = OP_CONV;
    std::string name;
    int ks = 3, cin = 0, cout = 0, res = 1;
    int act = ACT_NONE;
    float const_alpha = 0.0f;       // relu / leaky_relu slope when there is no alpha tensor
    // input
    int in_buf = EXT_X, in_off = 0, cin_phys = 0;
    int in_stride_override = 0;     // > 0: pixel stride of the input differs from its buffer's
    std::vector<int> chan_map;      // logical input channel -> physical channel relative to in_off
    // filter sources
    std::vector<ColSeg> segs;
    int dw_w = -1;                  // depthwise filter tensor (OP_DW, fused-depthwise OP_CONV, OP_COUT1 of a separable conv)
    int dwk = 0;                    // fused depthwise kernel size of an OP_CONV (0 = plain conv)
    float out_scale = 1.0f;         // OP_COUT1: pointwise scalar of a separable 1->1 conv
    int tconv_s = 0;                // > 0: the op is tf.nn.conv2d_transpose with this stride, run as its
                                    // equivalent 3x3 conv to s*s*C channels + depth_to_space (see add_tconv)
    int fold_s = 0;                 // > 0: folded linear tail (see fold_linear_tail): pixel-shuffler block
    int fold_c = 0;                 //      channels after depth_to_space
    int fold_rw = -1;               //      filter tensor of the last reconstruction conv [3, 3, C, 1]
                                    // OP_FOLDX (fold_whole_tail): fold_s = the net's scale; `fused` = the launches it replaces (the float32 plan, split16 = 0)
    // output
    int out_buf[2] = {EXT_Y, EXT_Y}, out_off[2] = {0, 0}, out_width[2] = {0, 0};
    int split = 1 << 30;
    int ps = 1, ps_c = 0;
    bool residual = false;
    bool vec4 = true;
    // conv_igemm variant
    ConvShape shape{3, 2, 1, 4};
    int n_tiles = 1, n_chunks = 0, ctot = 0;
    int n_full = 0;                 // Winograd: groups [0, n_full) hold shape.nt channel tiles, the others shape.nt - 1
    // accounting
    int64_t macs = 0, bytes = 0;
    // device copies
    float* d_w = nullptr;
    float* d_bias = nullptr;
    float* d_alpha = nullptr;
    int32_t* d_map = nullptr;
    float* d_dww = nullptr;
    // multi-source input (densify_features): the K axis is the concatenation of these dense tensors
    std::vector<std::pair<int, int>> multi;   // (buffer, physical channels = pad4)
    std::vector<NinSrcQuad> h_srctab;         // host copy of the quad table, refilled whenever the arena is re-carved
    NinSrcQuad* d_srctab = nullptr;
    // OP_STREAM (stream_features): the launches this op replaces, kept for their tensor indices, and the kernel plan
    std::vector<Op> fused;
    StreamArgs stream{};
    TailArgs tail{};
    Stream3Args stream3{};                    // OP_STREAM3 (fuse_feat3_stream): CNN1 .. CNNL of a non-separable narrow net as one launch; `fused` = the
                                              // layers' own launches: the float32 plan of a flagged image and the split16 = 0 path run those
    std::vector<int> extra_out;               // OP_STREAM3: the workspace tensors it writes (one per layer)
    int halo = -1;                            // >= 0: receptive-field radius of the op in ITS pixels (else ks / 2)
    // split16 variant of the launch (split16.hpp: the contraction on the f16 matrix pipe at f32 accuracy), taken when the handle's
    // "split16" option is on; the f32 launch then runs behind it as the fallback of the units it flags
    struct Split16 {
        bool on = false;
        int nt = 0, n_tiles = 0, n_full = 0, n_chunks = 0;   // channel groups (conv3_h: its own plan; conv_nin_h: the f32 launch's) and 32-channel chunks
        float inv_scale = 1.0f;                   // 2^-e of the filter scale
        void* d_w = nullptr;                      // pack_conv16 image
        float* d_bias = nullptr;                  // conv3_h: bias / slope in ITS padded group layout (conv_nin_h shares the f32 launch's)
        float* d_alpha = nullptr;
        int tail_octs = 0;                        // conv3_h: channel octets of the packed last chunk (kernels.h: c3h_tail_octs)
        bool rerun = false;                       // plan_p16: the launch belongs to the float32 plan of a flagged image (it is a split16 launch,
                                                  // writes a P16 tensor, or reads what such a launch writes); set for any kind of op
        bool in16_ok = false;                     // plan_p16: all the op's inputs are P16-capable tensors and its split16 kernel can read them
        std::vector<NinSrcQuad> h_tab16;          // conv_nin_h with P16 sources: one entry per channel octet of the K axis (refilled per carve)
        NinSrcQuad* d_tab16 = nullptr;
    } h16;
};

}  // namespace dcscn_impl

using namespace dcscn_impl;

struct dcscn_ctx {
    dcscn_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool finalized = false;

    std::vector<int> sched;
    std::vector<TensorSpec> tensors;
    std::map<std::string, int> tensor_index;
    std::vector<dcscn_layer_info> layers;
    std::vector<WsBuf> bufs;
    std::vector<Op> ops;

    // workspace
    void* arena = nullptr;
    size_t arena_bytes = 0;
    int lay_n = 0, lay_h = 0, lay_w = 0;     // shape the current carve was made for
    // host-path staging
    float* io_x = nullptr; float* io_x2 = nullptr; float* io_y = nullptr;
    size_t io_x_cap = 0, io_y_cap = 0;
    // bicubic resize (resample.hip): Pillow coefficient tables per (in, out) size, and the intermediate image
    struct ResampleTable { int ksize = 0; int* d_bounds = nullptr; double* d_kk = nullptr; };
    std::map<std::pair<int, int>, ResampleTable> resample_tables;
    float* rs_tmp = nullptr; size_t rs_tmp_cap = 0;
    // self-ensemble (ensemble.hip): flipped copies, their outputs, float64 mean (as 2 floats per double)
    float* ens_x = nullptr; float* ens_x2 = nullptr; float* ens_y = nullptr; float* ens_out = nullptr;
    size_t ens_x_cap = 0, ens_x2_cap = 0, ens_y_cap = 0, ens_out_cap = 0;
    float* rs_in = nullptr; float* rs_out = nullptr; size_t rs_in_cap = 0, rs_out_cap = 0;
    // colour path (color.hip): uint8 RGB in, float64 planes, float32 Y; capacities in floats
    float* col_rgb = nullptr; float* col_d = nullptr; float* col_d2 = nullptr; float* col_y32 = nullptr;
    size_t col_rgb_cap = 0, col_d_cap = 0, col_d2_cap = 0, col_y32_cap = 0;
    // spatial tiling of images larger than one pass (run_tiled): gathered tile batch
    float* tile_x = nullptr; float* tile_x2 = nullptr; float* tile_y = nullptr;
    size_t tile_x_cap = 0, tile_y_cap = 0;
    std::vector<void*> device_allocs;

    // LR pixels per pass through the layer chain.  Big passes keep >= ~10 rounds of workgroups per
    // launch on the 256 CUs (a 128-patch pass left a 10-25 % tail); bounded by workspace_budget.
    int64_t sub_batch_pixels = 4 << 20;
    int64_t workspace_budget = (int64_t)48 << 30;   // clamped to a share of the free device memory in dcscn_create
    bool budget_user_set = false;
    hipEvent_t done_ev = nullptr;            // recorded behind the last forward, on the stream it ran on
    std::vector<hipEvent_t> host_ev;         // dcscn_forward: one per chunk of the host-buffer pipeline
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    bool profile = false;
    unsigned long long* d_digest = nullptr;   // debug_digest: one arena checksum per op of the last pass + one of y (dcscn_debug_digests)
    int debug_digest = 0;
    bool conv3_h8 = true;                    // two-group 3x3 layers on conv3_h8 (option "conv3_h8"); off: conv3_h everywhere
    int n_cus = 256;                         // compute units of the device (persistent launches)
    int debug_poison = 0;                    // debug: LDS (bit 0) / VGPRs (bit 1) of every CU are filled with NaN patterns in front of every launch
    // option "graph_replay": a forward whose arguments repeat (same pointers, shape, stream) is captured into a hipGraph the second
    // time it is seen and replayed from then on -- one graph launch instead of ~30 kernel launches per pass
    bool graph_replay = false;
    struct GraphKey {
        const void *x = nullptr, *x2 = nullptr, *y = nullptr;
        void* stream = nullptr;
        int n = 0, H = 0, W = 0, split16 = 0, nb = 0, h8 = 0, p16 = 0;
        unsigned long long carve = 0;
        bool operator==(const GraphKey& o) const {
            return x == o.x && x2 == o.x2 && y == o.y && stream == o.stream && n == o.n && H == o.H && W == o.W && split16 == o.split16 && nb == o.nb &&
                   h8 == o.h8 && p16 == o.p16 && carve == o.carve;
        }
    };
    GraphKey graph_seen, graph_key;          // the previous forward's arguments; the arguments graph_exec was captured with
    hipGraphExec_t graph_exec = nullptr;
    bool winograd = true;                    // 3x3 convs as Winograd F(2x2,3x3) where it pays
    bool stream_tail = true;                 // the x4 tail of the same nets as one launch (fuse_tail_stream)
    bool stream_nin = true;                  // ... with A1 || B1 and B2 inside that launch where the shape allows (option "stream_nin"; 0: the r05 plan)
    bool stream_dense = true;                // non-separable narrow nets: CNN1 .. CNNL as one row-streamed launch (fuse_feat3_stream; option "stream_dense")
    bool stream_features = true;             // separable narrow nets: CNN1 .. B2 as one row-streamed launch (fuse_feat_stream)
    int split16_mask = 3;                    // debugging aid (option "split16" 2 / 3): bit 0 = conv3_h, bit 1 = conv_nin_h
    bool split16 = true;                     // eligible contractions on the f16 matrix pipe (conv3_h, conv_nin_h); option "split16" 0 = pure f32 kernels
    size_t redo_off = 0, redo_ints = 0;      // redo flags of a pass inside the arena (byte offset, count = 1 + images): [0] any, [1 + image]
    bool p16 = true;                         // option "p16": tensors between split16 launches are kept pre-split (p16.hpp) where plan_p16 allows
    bool p16_now = false;                    // the current carve holds them so (split16 on for both kernel families and p16)
    bool any_p16 = false;                    // plan_p16 found at least one such tensor
    int p16_max_res = 1;                     // largest resolution factor among them (bounds the pixels of a pass: kP16MaxPixels)
    bool dense_features = true;              // per-layer feature buffers + multi-source NIN GEMM instead of one concat tensor (densify_features)
    int concat_buf = -1;                     // build_graph: the skip-concat buffer, its slices (offset, logical width)
    std::vector<std::pair<int, int>> concat_slices;
    uint64_t carve_gen = 0, tables_gen = 0;  // arena carve generation / generation the multi-source tables were filled for
    bool nin = true;                         // wide 1x1 convs on the LDS-DMA staged GEMM (conv_nin); option "nin_gemm" 0 = conv_igemm
    bool fold_force = false;                 // "fold_linear_tail" 2: fold even where the composite does more work than the layers
    bool fold_tail = true;                   // graph rewrite of the linear tail, see fold_linear_tail(); option "fold_linear_tail" 0 = layer by layer
    bool fold_whole = true;                  // x3 / x4: the WHOLE tail (every shuffler stage + the last conv) as one 5x5 conv of the LR map with per-border-position
                                             // kernels (fold_whole_tail; option "fold_whole_tail", 0 = the r05 plans)
    bool spatial_tiling = true;              // images larger than one pass are cut into haloed windows (run_tiled)
    std::vector<unsigned long long> h_zrec;  // addresses of the zero records of the P16 planes in the current carve; device copy
    unsigned long long* d_zrec = nullptr;
    size_t zrec_cap = 0;
    std::vector<hipEvent_t> ev;              // event pool: 2 per launch
    size_t ev_used = 0;                      // events recorded since the last dcscn_get_profile
    std::vector<int> ev_op;                  // launch index of each recorded pair (ops.size() = the float32 plan behind a pass)
    int ev_forwards = 0;                     // forwards recorded since the last dcscn_get_profile
    std::vector<double> prof_ms;
};

namespace dcscn_impl {

extern thread_local std::string g_global_error;
void set_global_error(const char* fmt, ...);
int fail(dcscn_ctx* h, int code, const char* fmt, ...);

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, DCSCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

inline float* buf_ptr(dcscn_ctx* h, int id) { return reinterpret_cast<float*>(static_cast<char*>(h->arena) + h->bufs[id].offset); }

// graph.hip
void filter_schedule(int layers, int filters, int min_filters, double gamma, std::vector<int>& out);
int new_buf(dcscn_ctx* h, int stride, int res);
int kernel_act(int activator, float* const_alpha);
int build_graph(dcscn_ctx* h);
int op_tiles16(const Op& op);
bool nin_eligible(const dcscn_ctx* h, const Op& op);
bool wino_eligible(const dcscn_ctx* h, const Op& op);
bool h16_direct_eligible(const dcscn_ctx* h, const Op& op);
bool fold_linear_tail(dcscn_ctx* h);
bool fold_whole_tail(dcscn_ctx* h);
int stream_chunk_channel(int quads, int ch, int q, int s);
bool stream_conv_supported(int in_quads, int out_tiles);
void fuse_feat_stream(dcscn_ctx* h);
void fuse_tail_stream(dcscn_ctx* h);
void fuse_feat3_stream(dcscn_ctx* h);
void densify_features(dcscn_ctx* h);
void plan_p16(dcscn_ctx* h);
inline bool p16_active(const dcscn_ctx* h) { return h->p16 && h->any_p16 && h->split16 && h->split16_mask == 3; }
// pack.hip
int upload(dcscn_ctx* h, const void* host, size_t bytes, void** dev);
int finalize_op(dcscn_ctx* h, Op& op);
int pack_feat_stream(dcscn_ctx* h, Op& op);
int pack_tail_stream(dcscn_ctx* h, Op& op);
int pack_feat3_stream(dcscn_ctx* h, Op& op);
int pack_foldx(dcscn_ctx* h, Op& op);
// exec.hip
int ensure_workspace(dcscn_ctx* h, int nb, int H, int W, hipStream_t stream);
// redo = false: the launch of the pass (split16 kernels where the handle's options allow); true: the op's float32 launch gated by the
// pass's redo flags -- only the images a split16 launch flagged are computed (exec.hip: run_forward)
int launch_op(dcscn_ctx* h, const Op& op, int nb, int H, int W, const float* x, const float* x2, float* y, hipStream_t stream, bool redo = false);
bool op_on_split16(const dcscn_ctx* h, const Op& op);
bool op_takes_h8(const dcscn_ctx* h, const Op& op);
int halo_lr_pixels(const dcscn_ctx* h);
int run_forward(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int H, int W, hipStream_t stream);
int run_tiled(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int H, int W, int64_t pass_pixels, hipStream_t stream);
int resample_table(dcscn_ctx* h, int in_size, int out_size, const dcscn_ctx::ResampleTable** out);
int grow(dcscn_ctx* h, float** p, size_t* cap, size_t floats, hipStream_t stream);
int resize_device(dcscn_ctx* h, const float* in, float* out, int n, int H, int W, int OH, int OW, hipStream_t stream);

}  // namespace dcscn_impl
