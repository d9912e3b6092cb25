// C ABI (include/dcscn.h) and execution plan of the DCSCN forward pass on one MI355X.
//
// dcscn_create restates SuperResolution.build_graph (DCSCN.py:222-325) as a list of graph layers and a
// list of kernel launches over a small set of workspace tensors:
//
//   CONCAT  [n, H, W, sum(pad4(filters_i))]  every feature layer stores straight into its channel
//                                            slice, so tf.concat (DCSCN.py:259) costs nothing
//   T1      B1 output;  T2 = Concat2 = [B2 | A1] (DCSCN.py:281), or the "C" layer's output
//   UPk     depth_to_space outputs (the shuffle happens in the producing conv's store)
//   Rk      extra reconstruction layers;  DW  scratch of the depthwise half of separable convs
//
// All slices start on a 4-channel boundary and are padded to 4 channels; the consumer's repacked
// filter has zero rows for padding channels.
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

using namespace dcscn;

namespace {

thread_local std::string g_global_error;

void set_global_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_global_error = buf;
}

inline int pad4(int c) { return (c + 3) & ~3; }
inline int pad16(int c) { return (c + 15) & ~15; }

enum { EXT_X = -1, EXT_X2 = -2, EXT_Y = -3 };
enum OpKind { OP_CONV = 0, OP_CIN1 = 1, OP_DW = 2, OP_COUT1 = 3, OP_STREAM = 4, OP_TAIL = 5 };

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
    int halo = -1;                            // >= 0: receptive-field radius of the op in ITS pixels (else ks / 2)
};

}  // namespace

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
    bool winograd = true;                    // 3x3 convs as Winograd F(2x2,3x3) where it pays
    bool stream_tail = true;                 // the x4 tail of the same nets as one launch (fuse_tail_stream)
    bool stream_features = true;             // separable narrow nets: CNN1 .. B2 as one row-streamed launch (fuse_feat_stream)
    bool dense_features = true;              // per-layer feature buffers + multi-source NIN GEMM instead of one concat tensor (densify_features)
    int concat_buf = -1;                     // build_graph: the skip-concat buffer, its slices (offset, logical width)
    std::vector<std::pair<int, int>> concat_slices;
    uint64_t carve_gen = 0, tables_gen = 0;  // arena carve generation / generation the multi-source tables were filled for
    bool nin = true;                         // wide 1x1 convs on the LDS-DMA staged GEMM (conv_nin); option "nin_gemm" 0 = conv_igemm
    bool fold_force = false;                 // "fold_linear_tail" 2: fold even where the composite does more work than the layers
    bool fold_tail = true;                   // graph rewrite of the linear tail, see fold_linear_tail(); option "fold_linear_tail" 0 = layer by layer
    bool spatial_tiling = true;              // images larger than one pass are cut into haloed windows (run_tiled)
    std::vector<hipEvent_t> ev;              // event pool: 2 per launch
    size_t ev_used = 0;                      // events recorded since the last dcscn_get_profile
    int ev_forwards = 0;                     // forwards recorded since the last dcscn_get_profile
    std::vector<double> prof_ms;
};

namespace {

int fail(dcscn_ctx* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf;
    g_global_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, DCSCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

#pragma clang fp contract(off)
void filter_schedule(int layers, int filters, int min_filters, double gamma, std::vector<int>& out) {
    // DCSCN.py:232,240-244 -- evaluated in double exactly as CPython does
    out.clear();
    int n = filters;
    for (int i = 0; i < layers; ++i) {
        if (min_filters != 0 && i > 0) {
            const double x1 = (double)i / (double)(layers - 1);
            const double y1 = std::pow(x1, 1.0 / gamma);
            const double v = (double)(filters - min_filters) * (1.0 - y1) + (double)min_filters;
            n = (int)v;
        }
        out.push_back(n);
    }
}

int add_tensor(dcscn_ctx* h, const std::string& name, std::vector<int64_t> shape) {
    TensorSpec t;
    t.name = name;
    t.shape = std::move(shape);
    h->tensors.push_back(std::move(t));
    h->tensor_index[name] = (int)h->tensors.size() - 1;
    return (int)h->tensors.size() - 1;
}

int new_buf(dcscn_ctx* h, int stride, int res) {
    WsBuf b;
    b.stride = stride;
    b.res = res;
    h->bufs.push_back(b);
    return (int)h->bufs.size() - 1;
}

int kernel_act(int activator, float* const_alpha) {
    *const_alpha = 0.0f;
    switch (activator) {
        case DCSCN_ACT_NONE: return ACT_NONE;
        case DCSCN_ACT_PRELU: return ACT_ALPHA;
        case DCSCN_ACT_RELU: return ACT_ALPHA;
        case DCSCN_ACT_LEAKY_RELU: *const_alpha = 0.1f; return ACT_ALPHA;   // tf.maximum(x, 0.1 x)
        case DCSCN_ACT_SIGMOID: return ACT_SIGMOID;
        case DCSCN_ACT_TANH: return ACT_TANH;
        case DCSCN_ACT_SELU: return ACT_SELU;
        default: return -1;
    }
}

struct Src {            // where a layer reads its input
    int buf = EXT_X;
    int off = 0;
    int cin = 0;        // logical channels
    int cin_phys = 0;   // physical channels spanned (multiple of 4 unless external)
    std::vector<int> map;
    int res = 1;
};

Src identity_src(int buf, int off, int cin, int res) {
    Src s;
    s.buf = buf;
    s.off = off;
    s.cin = cin;
    s.cin_phys = pad4(cin);
    s.map.resize(cin);
    for (int i = 0; i < cin; ++i) s.map[i] = i;
    s.res = res;
    return s;
}

struct Dst {
    int buf = EXT_Y, off = 0, width = 0;
    int ps = 1, ps_c = 0;
    bool residual = false;
};

// Adds one graph conv layer (tf_graph.py build_conv / build_depthwise_separable_conv) and the
// launch(es) that execute it. `short_name` is the layer name used for the prelu variable.
void add_conv(dcscn_ctx* h, const std::string& var, const std::string& short_name, const Src& src, int ks,
              int cout, bool bias, int activator, bool ds, const Dst& dst, int* dw_buf) {
    const int cin = src.cin;
    dcscn_layer_info li{};
    snprintf(li.name, sizeof li.name, "%s", var.c_str());
    li.kernel_size = ks;
    li.in_channels = cin;
    li.out_channels = cout;
    li.depthwise_separable = ds;
    li.has_bias = bias;
    li.activator = activator;
    li.resolution = src.res;
    const int64_t r2 = (int64_t)src.res * src.res;
    li.macs_per_lr_pixel = r2 * (ds ? (int64_t)ks * ks * cin + (int64_t)cin * cout : (int64_t)ks * ks * cin * cout);
    h->layers.push_back(li);

    ColSeg seg;
    seg.cout = cout;
    seg.dst = 0;
    int t_dw = -1;
    if (ds) {
        t_dw = add_tensor(h, var + "/depthwise_W", {ks, ks, cin, 1});
        seg.w = add_tensor(h, var + "/pointwise_W", {1, 1, cin, cout});
    } else {
        seg.w = add_tensor(h, var + "/conv_W", {ks, ks, cin, cout});
    }
    if (bias) seg.b = add_tensor(h, var + "/conv_B", {cout});
    if (activator == DCSCN_ACT_PRELU) seg.alpha = add_tensor(h, var + "/prelu/" + short_name + "_prelu", {cout});

    // A separable conv with a 1x1 depthwise half (A1 / B1 of the DS models, tf_graph.py:155-177) is a plain 1x1 conv
    // whose weights carry the per-channel scale: no depthwise stage at all, and A1 / B1 can share one launch.
    const bool fold_dw1 = ds && ks == 1 && src.buf >= 0;
    if (fold_dw1) {
        seg.dw1 = t_dw;
        ds = false;
    }

    Op op;
    op.name = var;
    op.res = src.res;
    op.cout = cout;
    op.act = kernel_act(activator, &op.const_alpha);
    op.segs.push_back(seg);
    op.out_buf[0] = dst.buf;
    op.out_off[0] = dst.off;
    op.out_width[0] = dst.width;
    op.ps = dst.ps;
    op.ps_c = dst.ps_c;
    op.residual = dst.residual;
    const int out_stride = dst.buf >= 0 ? h->bufs[dst.buf].stride : 1;
    op.vec4 = out_stride % 4 == 0 && dst.off % 4 == 0 && dst.width % 4 == 0 && (dst.ps == 1 || dst.ps_c % 4 == 0) &&
              !dst.residual;
    const int64_t out_bytes = 4 * r2 * dst.width;

    if (ds && src.buf >= 0 && cin == 1 && cout == 1 && dst.buf == EXT_Y && !bias && activator == DCSCN_ACT_NONE &&
        cout1_lds_bytes(ks, src.cin_phys) <= 64 * 1024) {
        // separable 1 -> 1 conv (R-CNN of the c-DCSCN DS models): depthwise sum, times the pointwise
        // scalar, plus the residual -- one launch of the single-output kernel
        op.kind = OP_COUT1;
        op.ks = ks;
        op.cin = 1;
        op.in_buf = src.buf;
        op.in_off = src.off;
        op.cin_phys = src.cin_phys;
        op.chan_map = src.map;
        op.dw_w = t_dw;
        op.macs = li.macs_per_lr_pixel;
        op.bytes = 4 * r2 * src.cin_phys + out_bytes + (dst.residual ? 4 * r2 : 0);
    } else if (ds && src.buf >= 0 && ks == 3) {
        // depthwise half fused into the staging of the pointwise GEMM: its output never touches HBM
        // (instantiated for 1x1 / 3x3 depthwise filters; --cnn_size=5/7 separable models take the two-launch form below)
        op.kind = OP_CONV;
        op.ks = 1;
        op.dwk = ks;
        op.dw_w = t_dw;
        op.cin = cin;
        op.in_buf = src.buf;
        op.in_off = src.off;
        op.cin_phys = src.cin_phys;
        op.chan_map = src.map;
        op.macs = li.macs_per_lr_pixel;
        op.bytes = 4 * r2 * src.cin_phys + out_bytes + (dst.residual ? 4 * r2 : 0);
    } else if (ds) {
        // first layer (reads the 1-channel external input), or a 5x5 / 7x7 depthwise filter: depthwise half ->
        // DW scratch (logical channel order, zero padded to 4), then the pointwise GEMM
        if (*dw_buf < 0) *dw_buf = new_buf(h, 4, 1);
        Op dw;
        dw.kind = OP_DW;
        dw.name = var + "/depthwise";
        dw.ks = ks;
        dw.cin = cin;
        dw.cout = cin;
        dw.res = src.res;
        dw.in_buf = src.buf;
        dw.in_off = src.off;
        dw.cin_phys = pad4(cin);
        dw.chan_map = src.map;
        dw.dw_w = t_dw;
        dw.out_buf[0] = *dw_buf;
        dw.macs = r2 * (int64_t)ks * ks * cin;
        dw.bytes = 4 * r2 * (cin + pad4(cin));
        h->ops.push_back(dw);

        op.kind = OP_CONV;
        op.ks = 1;
        op.cin = cin;
        op.in_buf = *dw_buf;
        op.in_off = 0;
        op.in_stride_override = pad4(cin);
        op.cin_phys = pad4(cin);
        op.chan_map.resize(cin);
        for (int i = 0; i < cin; ++i) op.chan_map[i] = i;
        op.macs = r2 * (int64_t)cin * cout;
        op.bytes = 4 * r2 * pad4(cin) + out_bytes;
    } else if (src.buf == EXT_X) {
        op.kind = OP_CIN1;
        op.ks = ks;
        op.cin = 1;
        op.in_buf = EXT_X;
        op.macs = r2 * (int64_t)ks * ks * cout;
        op.bytes = 4 * r2 + out_bytes;
    } else {
        const bool to_y = cout == 1 && dst.buf == EXT_Y && !bias && activator == DCSCN_ACT_NONE && ks <= 5 &&
                          cout1_lds_bytes(ks, src.cin_phys) <= 64 * 1024;
        op.kind = to_y ? OP_COUT1 : OP_CONV;
        op.ks = ks;
        op.cin = cin;
        op.in_buf = src.buf;
        op.in_off = src.off;
        op.cin_phys = src.cin_phys;
        op.chan_map = src.map;
        op.macs = li.macs_per_lr_pixel;
        op.bytes = 4 * r2 * src.cin_phys + out_bytes + (dst.residual ? 4 * r2 : 0);
    }
    h->ops.push_back(op);
}

// build_transposed_conv (tf_graph.py:219-236): tf.nn.conv2d_transpose(x, W[k,k,C,C], stride s, SAME) with
// k = 2s - s%2, no bias, no activator.  Output pixel (s*h0 + a, s*w0 + b) only receives input pixels
// (h0 + dy, w0 + dx) with dy, dx in {-1, 0, 1}: filter tap ky = a + pt - s*dy (pt = (k - s) / 2) when that
// lies in [0, k).  So the op IS a 3x3 SAME conv from C to s*s*C channels followed by depth_to_space(s):
//   W3[dy+1][dx+1][ic][(a*s + b)*C + oc] = W[a + pt - s*dy][b + pt - s*dx][oc][ic]   (0 where out of range)
// and runs on the same kernels as the pixel shuffler (products identical, the added terms are exact zeros).
void add_tconv(dcscn_ctx* h, const Src& src, int s) {
    const int C = src.cin;
    const int k = 2 * s - s % 2;
    dcscn_layer_info li{};
    snprintf(li.name, sizeof li.name, "Up-TCNN");
    li.kernel_size = k;
    li.in_channels = C;
    li.out_channels = C;
    li.resolution = src.res;
    const int64_t r2 = (int64_t)src.res * src.res;
    li.macs_per_lr_pixel = r2 * k * k * C * (int64_t)C;
    h->layers.push_back(li);

    Op op;
    op.kind = OP_CONV;
    op.name = "Up-TCNN";
    op.ks = 3;
    op.cin = C;
    op.cout = C;
    op.res = src.res;
    op.act = ACT_NONE;
    op.tconv_s = s;
    ColSeg seg;
    seg.w = add_tensor(h, "Up-TCNN/Tconv_W", {k, k, C, C});
    seg.cout = s * s * C;
    op.segs.push_back(seg);
    op.in_buf = src.buf;
    op.in_off = src.off;
    op.cin_phys = src.cin_phys;
    op.chan_map = src.map;
    const int ub = new_buf(h, pad4(C), src.res * s);
    op.out_buf[0] = ub;
    op.out_off[0] = 0;
    op.out_width[0] = s * s * C;
    op.ps = s;
    op.ps_c = C;
    op.vec4 = C % 4 == 0;
    op.macs = li.macs_per_lr_pixel;
    op.bytes = 4 * r2 * (src.cin_phys + (int64_t)s * s * C);
    h->ops.push_back(op);
}

int build_graph(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    const bool ds = c.depthwise_separable != 0;
    const int k = c.cnn_size;
    int dw_buf = -1;

    filter_schedule(c.layers, c.filters, c.min_filters, c.filters_decay_gamma, h->sched);
    std::vector<int> slice_off(c.layers);
    int concat_stride = 0, total = 0;
    for (int i = 0; i < c.layers; ++i) {
        if (h->sched[i] <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "feature layer %d has %d filters", i + 1, h->sched[i]);
        slice_off[i] = concat_stride;
        concat_stride += pad4(h->sched[i]);
        total += h->sched[i];
    }
    const int concat = new_buf(h, concat_stride, 1);
    h->concat_buf = concat;
    for (int i = 0; i < c.layers; ++i) h->concat_slices.push_back({slice_off[i], h->sched[i]});

    // feature extraction, DCSCN.py:240-256
    Src src;
    src.buf = EXT_X;
    src.cin = c.channels;
    src.cin_phys = c.channels;
    src.map = {0};
    src.res = 1;
    for (int i = 0; i < c.layers; ++i) {
        char nm[32];
        snprintf(nm, sizeof nm, "CNN%d", i + 1);
        Dst d;
        d.buf = concat;
        d.off = slice_off[i];
        d.width = pad4(h->sched[i]);
        add_conv(h, nm, nm, src, k, h->sched[i], true, c.activator, ds, d, &dw_buf);
        src = identity_src(concat, slice_off[i], h->sched[i], 1);
    }
    Src cat;   // H_concat as an input
    cat.buf = concat;
    cat.off = 0;
    cat.cin = total;
    cat.cin_phys = concat_stride;
    cat.res = 1;
    for (int i = 0; i < c.layers; ++i)
        for (int j = 0; j < h->sched[i]; ++j) cat.map.push_back(slice_off[i] + j);

    // reconstruction, DCSCN.py:262-291
    if (c.use_nin) {
        const int na = c.nin_filters, nb = c.nin_filters2;
        if (na <= 0 || nb <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "nin_filters / nin_filters2 must be positive");
        const int t1 = new_buf(h, pad4(nb), 1);
        const int t2 = new_buf(h, pad4(nb) + pad4(na), 1);
        Dst da, db;
        da.buf = t2; da.off = pad4(nb); da.width = pad4(na);
        db.buf = t1; db.off = 0; db.width = pad4(nb);
        add_conv(h, "A1", "A1", cat, 1, na, true, c.activator, ds, da, &dw_buf);
        add_conv(h, "B1", "B1", cat, 1, nb, true, c.activator, ds, db, &dw_buf);
        {
            // A1 and B1 read the same 1301-wide concat: run them as ONE GEMM with conv channels
            // [B1 | pad to 16 | A1] and two destinations (halves the concat traffic).
            Op b1 = h->ops.back();
            h->ops.pop_back();
            Op a1 = h->ops.back();
            h->ops.pop_back();
            Op f = a1;
            f.name = "B1+A1";
            f.cout = na + nb;
            f.segs.clear();
            ColSeg sb = b1.segs[0];
            sb.dst = 0;
            ColSeg sa = a1.segs[0];
            sa.dst = pad16(nb);
            f.segs.push_back(sb);
            f.segs.push_back(sa);
            f.split = pad16(nb);
            f.out_buf[0] = t1; f.out_off[0] = 0; f.out_width[0] = pad4(nb);
            f.out_buf[1] = t2; f.out_off[1] = pad4(nb); f.out_width[1] = pad4(na);
            f.macs = a1.macs + b1.macs;
            f.bytes = 4 * (int64_t)concat_stride + 4 * (pad4(na) + pad4(nb));
            h->ops.push_back(f);
        }
        Dst d2;
        d2.buf = t2; d2.off = 0; d2.width = pad4(nb);
        add_conv(h, "B2", "B2", identity_src(t1, 0, nb, 1), 3, nb, true, c.activator, ds, d2, &dw_buf);
        src = Src();
        src.buf = t2;
        src.off = 0;
        src.cin = na + nb;
        src.cin_phys = pad4(nb) + pad4(na);
        src.res = 1;
        for (int j = 0; j < nb; ++j) src.map.push_back(j);                 // Concat2 = [B2, A1]
        for (int j = 0; j < na; ++j) src.map.push_back(pad4(nb) + j);
    } else if (c.legacy_no_c) {
        src = cat;
    } else {
        const int t2 = new_buf(h, pad4(c.filters), 1);
        Dst d;
        d.buf = t2; d.off = 0; d.width = pad4(c.filters);
        add_conv(h, "C", "C", cat, 1, c.filters, true, c.activator, ds, d, &dw_buf);
        src = identity_src(t2, 0, c.filters, 1);
    }

    // upsampling, DCSCN.py:293-311 + tf_graph.py:219-249
    if (c.pixel_shuffler) {
        const int ps_out = c.pixel_shuffler_filters != 0 ? c.pixel_shuffler_filters : src.cin;
        struct Stage { const char* name; int s; int cout; };
        std::vector<Stage> stages;
        if (c.scale == 4) {
            stages.push_back({"Up-PS", 2, src.cin});
            stages.push_back({"Up-PS2", 2, ps_out});
        } else {
            stages.push_back({"Up-PS", c.scale, ps_out});
        }
        for (const Stage& st : stages) {
            const int ub = new_buf(h, pad4(st.cout), src.res * st.s);
            Dst d;
            d.buf = ub; d.off = 0; d.width = st.s * st.s * st.cout;
            d.ps = st.s; d.ps_c = st.cout;
            const std::string var = std::string(st.name) + "/" + st.name + "_CNN";
            add_conv(h, var, std::string(st.name) + "_CNN", src, k, st.s * st.s * st.cout, true, DCSCN_ACT_NONE, ds, d, &dw_buf);
            src = identity_src(ub, 0, st.cout, src.res * st.s);
        }
    } else {
        add_tconv(h, src, c.scale);
        src = identity_src(h->ops.back().out_buf[0], 0, src.cin, src.res * c.scale);
    }

    // reconstruction convs at HR, DCSCN.py:313-323
    const int rl = std::max(c.reconstruct_layers, 1);
    for (int i = 0; i < rl - 1; ++i) {
        char nm[32];
        snprintf(nm, sizeof nm, "R-CNN%d", i + 1);
        const int rb = new_buf(h, pad4(c.reconstruct_filters), src.res);
        Dst d;
        d.buf = rb; d.off = 0; d.width = pad4(c.reconstruct_filters);
        add_conv(h, nm, nm, src, k, c.reconstruct_filters, true, c.activator, false, d, &dw_buf);
        src = identity_src(rb, 0, c.reconstruct_filters, src.res);
    }
    {
        char nm[32];
        snprintf(nm, sizeof nm, "R-CNN%d", rl);
        Dst d;
        d.buf = EXT_Y; d.off = 0; d.width = 1;
        d.residual = true;                                                  // y_ = R-CNN + x2, DCSCN.py:325
        add_conv(h, nm, nm, src, k, 1, false, DCSCN_ACT_NONE, ds, d, &dw_buf);
    }
    if (src.res != c.scale) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: output resolution %d != scale %d", src.res, c.scale);
    for (const Op& op : h->ops)
        if (op.kind == OP_CIN1 && cin1_lds_bytes(op.ks, pad4(op.cout)) > 64 * 1024)
            return fail(h, DCSCN_ERR_UNSUPPORTED, "first layer %dx%d with %d filters needs more than 64 KB of LDS", op.ks, op.ks, op.cout);

    // size the depthwise scratch: widest separable input at its resolution (per pixel: stride floats)
    if (dw_buf >= 0) {
        // one stride per resolution would waste nothing, but a single shared tensor is simpler: give it
        // the largest per-LR-pixel footprint by choosing res = 1 and stride = max(res^2 * pad4(cin)).
        int best = 4;
        for (const Op& op : h->ops)
            if (op.kind == OP_DW) best = std::max(best, op.res * op.res * pad4(op.cin));
        h->bufs[dw_buf].stride = best;
        h->bufs[dw_buf].res = 1;
    }
    return DCSCN_OK;
}

// ---- Winograd plan ------------------------------------------------------------------------------------
int op_tiles16(const Op& op) {
    int ctot = 0;
    for (const ColSeg& s : op.segs) ctot = std::max(ctot, s.dst + s.cout);
    return (ctot + 15) / 16;
}

// Winograd F(2x2,3x3) (conv_wino2) for 3x3 convs with enough input channels to amortise the transforms (measured on
// MI355X: 1.25-1.35x over conv_igemm from 57 input channels up, still 1.3x at 22-26; the last, single-tile layers of the c-DCSCN models stay on the direct
// kernel).  A layer's 16-channel tiles are spread evenly over ceil(tiles / 3) channel groups (10 tiles = 3+3+2+2): a
// group's cost is only partly its MFMA count (the input tile and its transform are per group), so a 1-tile group costs
// ~70 % of a 3-tile one.  (op.vec4: the Winograd epilogue only has the 16-byte store form.)
// 1x1 convs wide enough to be worth the LDS-DMA GEMM (conv_nin): plain conv + bias + activator into one or two NHWC
// slices; everything with a fused depthwise stage, depth_to_space, a residual or scalar stores stays on conv_igemm.
bool nin_eligible(const dcscn_ctx* h, const Op& op) {
    return h->nin && op.kind == OP_CONV && op.ks == 1 && op.dwk == 0 && op.ps == 1 && !op.residual && op.vec4 && op.fold_s == 0 &&
           op.tconv_s == 0 && op.cin_phys >= 32 && op.in_stride_override == 0;
}

// (>= 24 input channels: measured on the c-DCSCN L7 net, 26 -> 22 and 22 -> 18 take 0.28 / 0.22 ms here against 0.36 / 0.30 ms
// on the direct kernel; below that the output is a single channel tile and the direct kernel wins)
bool wino_eligible(const dcscn_ctx* h, const Op& op) {
    const int tiles16 = op_tiles16(op);
    return h->winograd && op.kind == OP_CONV && op.vec4 && op.ks == 3 && op.dwk == 0 && op.cin_phys >= 24 &&
           op.segs.size() == 1 && op.tconv_s == 0 && tiles16 >= 2;
}

// ---- optional graph rewrite: the linear tail as one conv ----------------------------------------
//
// The last pixel-shuffler stage (3x3 conv + bias, NO activator, DCSCN.py:293-311), depth_to_space and the
// last reconstruction conv (3x3 to 1 channel, no bias, no activator, DCSCN.py:319-323) are all linear, so
// their composition is ONE convolution of the low-resolution map: HR pixel (s y + a, s x + b) is a 5x5
// conv of the LR neighbourhood of (y, x) with a kernel that depends on the sub-pixel phase (a, b) only:
//
//   out(sy+a, sx+b) = sum_{dy,dx} sum_c Wr[dy][dx][c] U_c(sy+a+dy, sx+b+dx),   U_c(Y, X) = UpConv(Y div s, X div s)[((Y mod s) s + X mod s) C + c]
//
// except that the reconstruction conv zero-pads the HR map: a tap that leaves the image is dropped, which
// changes the composite kernel (and its bias term) on the border rows / columns of that phase.  Per phase
// only one row tap (dy = -1 for a = 0, dy = +1 for a = s-1) and one column tap can leave, so 4 "border
// variants" per phase cover every case; the launch computes all of them (conv channel = phase * 4 +
// variant; the 16-wide MFMA channel tile is padded anyway) and the epilogue keeps the one that applies.
// 25 * Cin * 4 s^2 MACs per LR pixel replace 9 * Cin * s^2 C + 9 s^2 C (C = 96, s = 2: 38 k instead of 335 k),
// and the s^2 C-channel HR map is never written.  The result equals the layer-by-layer graph in exact
// arithmetic; in f32 it differs by re-association (composite weights are formed in float64 and rounded
// once).  On by default where the composite is less work than the layers (option "fold_linear_tail": 0 = the reference's
// layers one by one, 2 = fold even where it is more work).
bool fold_linear_tail(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    if (!c.pixel_shuffler || c.depthwise_separable || c.cnn_size != 3 || c.reconstruct_layers > 1) return false;
    if (h->ops.size() < 2) return false;
    const Op r = h->ops[h->ops.size() - 1];
    const Op u = h->ops[h->ops.size() - 2];
    if (u.kind != OP_CONV || u.ps < 2 || u.ps > 4 || u.segs.size() != 1 || u.dwk != 0 || u.tconv_s != 0 || u.act != ACT_NONE) return false;
    const bool r_ok = (r.kind == OP_COUT1 && r.dw_w < 0) || (r.kind == OP_CONV && r.cout == 1 && r.dwk == 0);
    if (!r_ok || !r.residual || r.segs.size() != 1 || r.segs[0].b >= 0 || r.act != ACT_NONE || r.ks != 3) return false;
    if (r.in_buf != u.out_buf[0] || r.cin != u.ps_c || (u.ps * u.ps + 3) / 4 > 4) return false;
    // worth it only where the composite does less work: 25 taps x (4 s^2 variants padded to 16-channel tiles) per input
    // channel against the shuffler conv's 9 s^2 C (the c-DCSCN nets shuffle to ONE channel: 400 vs 36 -- measured 0.53 ms
    // folded against 0.44 ms layer by layer)
    if (!h->fold_force && 25 * pad16(4 * u.ps * u.ps) >= 9 * u.ps * u.ps * u.ps_c) return false;
    Op f = u;
    f.name = u.name + "+" + r.name + " (folded)";
    f.ks = 5;
    f.cout = 4 * u.ps * u.ps;
    f.segs[0].cout = f.cout;
    f.segs[0].dst = 0;
    f.fold_s = u.ps;
    f.fold_c = u.ps_c;
    f.fold_rw = r.segs[0].w;
    f.out_buf[0] = f.out_buf[1] = EXT_Y;
    f.out_off[0] = f.out_off[1] = 0;
    f.out_width[0] = 1;
    f.out_width[1] = 0;
    f.split = 1 << 30;
    f.residual = true;
    f.vec4 = false;
    f.macs = u.macs + r.macs;                       // algorithmic work of the layers it replaces
    const int64_t hr2 = (int64_t)u.res * u.ps * u.res * u.ps;
    f.bytes = 4 * (int64_t)u.res * u.res * u.cin_phys + 8 * hr2;
    const int dead = u.out_buf[0];
    h->ops.pop_back();
    h->ops.pop_back();
    bool used = false;
    for (const Op& o : h->ops) used = used || o.in_buf == dead || o.out_buf[0] == dead || o.out_buf[1] == dead;
    if (!used && dead >= 0) h->bufs[dead].stride = 0;   // the shuffled HR map no longer exists
    h->ops.push_back(f);
    return true;
}

// ---- weight repack -----------------------------------------------------------------------------

int upload(dcscn_ctx* h, const void* host, size_t bytes, void** dev) {
    HIP_TRY(h, hipMalloc(dev, bytes));
    h->device_allocs.push_back(*dev);
    HIP_TRY(h, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return DCSCN_OK;
}

int pack_feat_stream(dcscn_ctx* h, Op& op);
int pack_tail_stream(dcscn_ctx* h, Op& op);

int finalize_op(dcscn_ctx* h, Op& op) {
    if (op.kind == OP_STREAM) return pack_feat_stream(h, op);
    if (op.kind == OP_TAIL) return pack_tail_stream(h, op);
    if (op.kind == OP_DW) {
        const TensorSpec& w = h->tensors[op.dw_w];          // [k, k, cin, 1] -> [taps][cin]
        int rc = upload(h, w.data.data(), w.data.size() * sizeof(float), (void**)&op.d_w);
        if (rc) return rc;
        std::vector<int32_t> map(op.chan_map.begin(), op.chan_map.end());
        return upload(h, map.data(), map.size() * sizeof(int32_t), (void**)&op.d_map);
    }

    const int taps = op.ks * op.ks;
    if (op.kind == OP_COUT1) {
        const ColSeg& s = op.segs[0];
        const bool separable = op.dw_w >= 0;
        const TensorSpec& tw = h->tensors[separable ? op.dw_w : s.w];   // [k, k, cin, 1]
        if (separable) op.out_scale = h->tensors[s.w].data[0];       // pointwise [1, 1, 1, 1]
        const int cin = (int)op.chan_map.size();
        std::vector<float> w((size_t)taps * op.cin_phys, 0.0f);
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < cin; ++ci) {
                float v = tw.data[(size_t)t * cin + ci];
                if (s.dw1 >= 0) v = h->tensors[s.dw1].data[ci] * v;          // folded 1x1 depthwise half (ColSeg::dw1)
                w[(size_t)t * op.cin_phys + op.chan_map[ci]] = v;
            }
        return upload(h, w.data(), w.size() * sizeof(float), (void**)&op.d_w);
    }
    if (op.kind == OP_CIN1) {
        const ColSeg& s = op.segs[0];
        const int cs = op.out_width[0];
        std::vector<float> w((size_t)taps * cs, 0.0f), b(cs, 0.0f), al(cs, op.const_alpha);
        const TensorSpec& tw = h->tensors[s.w];             // [k, k, 1, cout]
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < s.cout; ++c) w[(size_t)t * cs + c] = tw.data[(size_t)t * s.cout + c];
        if (s.b >= 0) std::copy(h->tensors[s.b].data.begin(), h->tensors[s.b].data.end(), b.begin());
        if (s.alpha >= 0) std::copy(h->tensors[s.alpha].data.begin(), h->tensors[s.alpha].data.end(), al.begin());
        for (int c = s.cout; c < cs; ++c) al[c] = 0.0f;
        int rc = upload(h, w.data(), w.size() * sizeof(float), (void**)&op.d_w);
        if (!rc) rc = upload(h, b.data(), b.size() * sizeof(float), (void**)&op.d_bias);
        if (!rc) rc = upload(h, al.data(), al.size() * sizeof(float), (void**)&op.d_alpha);
        return rc;
    }

    // transposed conv: materialise the equivalent 3x3 filter [3][3][C][s*s*C] (see add_tconv)
    TensorSpec derived;
    if (op.tconv_s > 0) {
        const TensorSpec& t = h->tensors[op.segs[0].w];      // [k, k, out C, in C]
        const int sc = op.tconv_s, kk = (int)t.shape[0], C = (int)t.shape[2], pt = (kk - sc) / 2, co = sc * sc * C;
        derived.data.assign((size_t)9 * C * co, 0.0f);
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                for (int a2 = 0; a2 < sc; ++a2)
                    for (int b2 = 0; b2 < sc; ++b2) {
                        const int ky = a2 + pt - sc * dy, kx = b2 + pt - sc * dx;
                        if (ky < 0 || ky >= kk || kx < 0 || kx >= kk) continue;
                        for (int ic = 0; ic < C; ++ic)
                            for (int oc = 0; oc < C; ++oc)
                                derived.data[(((size_t)(dy + 1) * 3 + (dx + 1)) * C + ic) * co + (size_t)(a2 * sc + b2) * C + oc] =
                                    t.data[(((size_t)ky * kk + kx) * C + oc) * C + ic];
                    }
    }
    // folded linear tail: composite 5x5 filter [5][5][cin][phase * 4 + variant] and its bias, in float64
    std::vector<float> derived_bias;
    if (op.fold_s > 0) {
        const int sc = op.fold_s, C = op.fold_c, cin = (int)op.chan_map.size(), V = 4 * sc * sc, UC = sc * sc * C;
        const TensorSpec& wu = h->tensors[op.segs[0].w];     // [3, 3, cin, s*s*C]
        const TensorSpec& wr = h->tensors[op.fold_rw];       // [3, 3, C, 1]
        const float* bu = op.segs[0].b >= 0 ? h->tensors[op.segs[0].b].data.data() : nullptr;
        std::vector<double> wacc((size_t)25 * cin * V, 0.0), bacc(V, 0.0);
        auto fdiv = [](int x, int d) { return x >= 0 ? x / d : -((-x + d - 1) / d); };
        for (int pa = 0; pa < sc; ++pa)
            for (int pb = 0; pb < sc; ++pb)
                for (int var = 0; var < 4; ++var) {
                    const int v = (pa * sc + pb) * 4 + var;
                    const bool rbit = var & 2, cbit = var & 1;
                    for (int dy = -1; dy <= 1; ++dy) {
                        if (rbit && ((pa == 0 && dy == -1) || (pa == sc - 1 && dy == 1))) continue;   // tap above / below the image
                        const int oy = fdiv(pa + dy, sc), a2 = pa + dy - oy * sc;
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (cbit && ((pb == 0 && dx == -1) || (pb == sc - 1 && dx == 1))) continue;
                            const int ox = fdiv(pb + dx, sc), b2 = pb + dx - ox * sc;
                            for (int cc = 0; cc < C; ++cc) {
                                const double wrv = wr.data[((size_t)(dy + 1) * 3 + (dx + 1)) * C + cc];
                                const int ch = (a2 * sc + b2) * C + cc;
                                if (bu) bacc[v] += wrv * bu[ch];
                                for (int ey = -1; ey <= 1; ++ey)
                                    for (int ex = -1; ex <= 1; ++ex) {
                                        const size_t tap5 = (size_t)(oy + ey + 2) * 5 + (ox + ex + 2);
                                        const float* wsrc = &wu.data[((size_t)(ey + 1) * 3 + (ex + 1)) * cin * UC + ch];
                                        double* wdst = &wacc[tap5 * cin * V + v];
                                        for (int k = 0; k < cin; ++k) wdst[(size_t)k * V] += wrv * wsrc[(size_t)k * UC];
                                    }
                            }
                        }
                    }
                }
        derived.data.resize(wacc.size());
        for (size_t i = 0; i < wacc.size(); ++i) derived.data[i] = (float)wacc[i];
        derived_bias.resize(V);
        for (int v = 0; v < V; ++v) derived_bias[v] = (float)bacc[v];
    }
    const TensorSpec* w_override = (op.tconv_s > 0 || op.fold_s > 0) ? &derived : nullptr;

    // OP_CONV: dense [tap][k_phys][conv channel] -> [n_tile][chunk][tap][kk][NS]
    int ctot = 0;
    for (const ColSeg& s : op.segs) ctot = std::max(ctot, s.dst + s.cout);
    const int tiles16 = (ctot + 15) / 16;
    if (nin_eligible(h, op)) {
        op.n_tiles = (tiles16 + kNinMaxNT - 1) / kNinMaxNT;                   // channel groups
        const int nt = (tiles16 + op.n_tiles - 1) / op.n_tiles;
        op.n_full = tiles16 - op.n_tiles * (nt - 1);
        op.shape = ConvShape{1, 4, nt, kNinKC, 0, 1, 0};
        op.ctot = op.n_tiles * nt * 16;
        const int kc = kNinKC;
        op.n_chunks = (op.cin_phys + kc - 1) / kc;
        const int ns = conv_ns(nt);
        const size_t chunk_floats = (size_t)kc * ns;
        std::vector<float> pack((size_t)op.n_tiles * op.n_chunks * chunk_floats, 0.0f);
        std::vector<float> bias(op.ctot, 0.0f), alpha(op.ctot, 0.0f);
        auto padded = [&](int cc) {
            const int t = cc / 16;
            const int wide = op.n_full * nt;
            const int g = t < wide ? t / nt : op.n_full + (t - wide) / (nt - 1);
            const int tg = t < wide ? t % nt : (t - wide) % (nt - 1);
            return (g * nt + tg) * 16 + cc % 16;
        };
        const int cin = (int)op.chan_map.size();
        for (const ColSeg& sg : op.segs) {
            const TensorSpec& tw = h->tensors[sg.w];                          // [1, 1, cin, cout]
            const int wcols = (int)tw.shape.back();
            for (int ci = 0; ci < cin; ++ci) {
                const int kp = op.chan_map[ci];
                const int chunk = kp / kc, c16 = kp % kc;
                const int row = (c16 & 3) * 4 + (c16 >> 2);                   // k-step c16 & 3, MFMA k index c16 >> 2
                const float dscale = sg.dw1 >= 0 ? h->tensors[sg.dw1].data[ci] : 1.0f;     // folded 1x1 depthwise
                const float* wrow = &tw.data[(size_t)ci * wcols + sg.col0];
                for (int co = 0; co < sg.cout; ++co) {
                    const int pc = padded(sg.dst + co);
                    const int grp = pc / (nt * 16), jn = pc % (nt * 16);
                    pack[((size_t)grp * op.n_chunks + chunk) * chunk_floats + (size_t)row * ns + jn] = sg.dw1 >= 0 ? dscale * wrow[co] : wrow[co];
                }
            }
            for (int co = 0; co < sg.cout; ++co) {
                const int pc = padded(sg.dst + co);
                if (sg.b >= 0) bias[pc] = h->tensors[sg.b].data[sg.col0 + co];
                alpha[pc] = sg.alpha >= 0 ? h->tensors[sg.alpha].data[sg.col0 + co] : op.const_alpha;
            }
        }
        int rcn = upload(h, pack.data(), pack.size() * sizeof(float), (void**)&op.d_w);
        if (!rcn) rcn = upload(h, bias.data(), bias.size() * sizeof(float), (void**)&op.d_bias);
        if (!rcn) rcn = upload(h, alpha.data(), alpha.size() * sizeof(float), (void**)&op.d_alpha);
        return rcn;
    }
    if (wino_eligible(h, op)) {
        op.n_tiles = (tiles16 + kWinoMaxNT - 1) / kWinoMaxNT;                 // channel groups
        const int nt = (tiles16 + op.n_tiles - 1) / op.n_tiles;               // tiles of the wide groups
        op.n_full = tiles16 - op.n_tiles * (nt - 1);                          // how many groups are wide; the others hold nt - 1
        op.shape = ConvShape{3, 4, nt, kWinoKC, 0, 0, 1};
        op.ctot = op.n_tiles * nt * 16;
        const int kc = kWinoKC;
        op.n_chunks = (op.cin_phys + kc - 1) / kc;
        const int ns = conv_ns(nt);
        const size_t chunk_floats = (size_t)16 * kc * ns;
        std::vector<float> pack((size_t)op.n_tiles * op.n_chunks * chunk_floats, 0.0f);
        std::vector<float> bias(op.ctot, 0.0f), alpha(op.ctot, 0.0f);
        const ColSeg& sg = op.segs[0];
        const TensorSpec& tw = w_override ? *w_override : h->tensors[sg.w];   // [3, 3, cin, cout]
        const int cin = (int)op.chan_map.size();
        const int wcols = w_override ? sg.cout : (int)tw.shape.back();
        // conv channel -> slot of the padded [group][nt * 16] layout (bias, slope and filter columns)
        auto padded = [&](int cc) {
            const int t = cc / 16;
            const int wide = op.n_full * nt;                                   // tiles held by the wide groups
            const int g = t < wide ? t / nt : op.n_full + (t - wide) / (nt - 1);
            const int tg = t < wide ? t % nt : (t - wide) % (nt - 1);
            return (g * nt + tg) * 16 + cc % 16;
        };
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int ci = 0; ci < cin; ++ci) {
            const int kp = op.chan_map[ci];
            const int chunk = kp / kc, c8 = kp % kc;
            const int row = (c8 & 1) * 4 + (c8 >> 1);                          // k-step c8 & 1, MFMA k index c8 >> 1
            for (int co = 0; co < sg.cout; ++co) {
                double g[3][3], gg[4][3];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) g[i][j] = tw.data[((size_t)(i * 3 + j) * cin + ci) * wcols + sg.col0 + co];
                for (int xi = 0; xi < 4; ++xi)                      // G g
                    for (int j = 0; j < 3; ++j) gg[xi][j] = G[xi][0] * g[0][j] + G[xi][1] * g[1][j] + G[xi][2] * g[2][j];
                const int pc = padded(sg.dst + co);
                const int grp = pc / (nt * 16), jn = pc % (nt * 16);
                for (int xi = 0; xi < 4; ++xi)
                    for (int nu = 0; nu < 4; ++nu) {                // (G g) G^T, float64, rounded once
                        const double u = gg[xi][0] * G[nu][0] + gg[xi][1] * G[nu][1] + gg[xi][2] * G[nu][2];
                        pack[((size_t)grp * op.n_chunks + chunk) * chunk_floats + ((size_t)(xi * 4 + nu) * kc + row) * ns + jn] = (float)u;
                    }
            }
        }
        for (int co = 0; co < sg.cout; ++co) {
            const int pc = padded(sg.dst + co);
            if (sg.b >= 0) bias[pc] = h->tensors[sg.b].data[sg.col0 + co];
            alpha[pc] = sg.alpha >= 0 ? h->tensors[sg.alpha].data[sg.col0 + co] : op.const_alpha;
        }
        int rcw = upload(h, pack.data(), pack.size() * sizeof(float), (void**)&op.d_w);
        if (!rcw) rcw = upload(h, bias.data(), bias.size() * sizeof(float), (void**)&op.d_bias);
        if (!rcw) rcw = upload(h, alpha.data(), alpha.size() * sizeof(float), (void**)&op.d_alpha);
        return rcw;
    }
    const int max_nt = op.dwk ? conv_max_fused_dw_nt() : conv_max_nt(op.ks);
    op.n_tiles = (tiles16 + max_nt - 1) / max_nt;
    const int nt = (tiles16 + op.n_tiles - 1) / op.n_tiles;
    op.shape = conv_pick_shape(op.ks, nt, op.dwk);
    if (op.dwk) {
        const TensorSpec& td = h->tensors[op.dw_w];          // [k, k, cin, 1] -> [taps][cin_phys] physical
        const int dtaps = op.dwk * op.dwk, cin = (int)op.chan_map.size();
        std::vector<float> dww((size_t)dtaps * op.cin_phys, 0.0f);
        for (int t = 0; t < dtaps; ++t)
            for (int ci = 0; ci < cin; ++ci) dww[(size_t)t * op.cin_phys + op.chan_map[ci]] = td.data[(size_t)t * cin + ci];
        int rc0 = upload(h, dww.data(), dww.size() * sizeof(float), (void**)&op.d_dww);
        if (rc0) return rc0;
    }
    op.ctot = op.n_tiles * nt * 16;
    const int kc = op.shape.kc;
    op.n_chunks = (op.cin_phys + kc - 1) / kc;
    const int ns = conv_ns(nt);
    const size_t chunk_floats = (size_t)taps * kc * ns;
    std::vector<float> pack((size_t)op.n_tiles * op.n_chunks * chunk_floats, 0.0f);
    std::vector<float> bias(op.ctot, 0.0f), alpha(op.ctot, 0.0f);
    for (const ColSeg& s : op.segs) {
        const TensorSpec& tw = w_override ? *w_override : h->tensors[s.w];   // [ks, ks, cin, cout] (or [1,1,cin,cout])
        const int cin = (int)op.chan_map.size();
        const int wcols = w_override ? s.cout : (int)tw.shape.back();
        for (int t = 0; t < taps; ++t)
            for (int ci = 0; ci < cin; ++ci) {
                const int kp = op.chan_map[ci];
                const int chunk = kp / kc, kk = kp % kc;
                const float* wrow = &tw.data[((size_t)t * cin + ci) * wcols + s.col0];
                const float dscale = s.dw1 >= 0 ? h->tensors[s.dw1].data[ci] : 1.0f;     // folded 1x1 depthwise
                for (int co = 0; co < s.cout; ++co) {
                    const int cc = s.dst + co;
                    const int tile = cc / (nt * 16), j = cc % (nt * 16);
                    pack[((size_t)tile * op.n_chunks + chunk) * chunk_floats + ((size_t)t * kc + kk) * ns + j] =
                        s.dw1 >= 0 ? dscale * wrow[co] : wrow[co];
                }
            }
        for (int co = 0; co < s.cout; ++co) {
            if (op.fold_s > 0) bias[s.dst + co] = derived_bias[co];
            else if (s.b >= 0) bias[s.dst + co] = h->tensors[s.b].data[s.col0 + co];
            alpha[s.dst + co] = s.alpha >= 0 ? h->tensors[s.alpha].data[s.col0 + co] : op.const_alpha;
        }
    }
    int rc = upload(h, pack.data(), pack.size() * sizeof(float), (void**)&op.d_w);
    if (!rc) rc = upload(h, bias.data(), bias.size() * sizeof(float), (void**)&op.d_bias);
    if (!rc) rc = upload(h, alpha.data(), alpha.size() * sizeof(float), (void**)&op.d_alpha);
    return rc;
}

// ---- workspace ---------------------------------------------------------------------------------

// (Re)carves the arena for passes of nb images of H x W.  `stream` is the stream the coming forward runs on: the clear
// of the new carve is enqueued there, behind an event wait on the previous forward (which may have run on another
// stream and may still be in flight) -- nothing is cleared or re-carved underneath live kernels.
int ensure_workspace(dcscn_ctx* h, int nb, int H, int W, hipStream_t stream) {
    if (h->arena && nb <= h->lay_n && H == h->lay_h && W == h->lay_w) return DCSCN_OK;
    std::vector<size_t> offsets(h->bufs.size());
    size_t total = 0;
    for (size_t i = 0; i < h->bufs.size(); ++i) {
        const WsBuf& b = h->bufs[i];
        offsets[i] = total;
        const size_t bytes = (size_t)nb * H * b.res * W * b.res * b.stride * sizeof(float);
        total += (bytes + 255) & ~(size_t)255;
    }
    total = std::max<size_t>(total, 256);
    if (total > h->arena_bytes) {
        // the old arena is freed: everything that uses it must have finished
        if (h->has_last) HIP_TRY(h, hipStreamSynchronize(h->last_stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->arena) HIP_TRY(h, hipFree(h->arena));
        h->arena = nullptr;
        h->arena_bytes = 0;
        h->lay_n = h->lay_h = h->lay_w = 0;
        hipError_t e = hipMalloc(&h->arena, total);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, DCSCN_ERR_NOMEM, "workspace of %zu bytes: %s", total, hipGetErrorString(e));
        }
        h->arena_bytes = total;
    } else if (h->has_last && h->last_stream != stream) {
        HIP_TRY(h, hipStreamWaitEvent(stream, h->done_ev, 0));
    }
    for (size_t i = 0; i < h->bufs.size(); ++i) h->bufs[i].offset = offsets[i];
    // padding channels that no kernel writes (depth_to_space outputs with C % 4 != 0) must hold
    // finite values: clear the bytes of the new carve
    HIP_TRY(h, hipMemsetAsync(h->arena, 0, total, stream));
    h->carve_gen += 1;
    h->lay_n = nb;
    h->lay_h = H;
    h->lay_w = W;
    return DCSCN_OK;
}

inline float* buf_ptr(dcscn_ctx* h, int id) { return reinterpret_cast<float*>(static_cast<char*>(h->arena) + h->bufs[id].offset); }

int launch_op(dcscn_ctx* h, const Op& op, int nb, int H, int W, const float* x, const float* x2, float* y,
              hipStream_t stream) {
    const int Hr = H * op.res, Wr = W * op.res;
    if (op.kind == OP_TAIL) {
        TailArgs a = op.tail;
        a.c2 = buf_ptr(h, op.in_buf);
        a.c2_stride = h->bufs[op.in_buf].stride;
        a.x2 = x2;
        a.y = y;
        a.blob = op.d_w;
        a.N = nb; a.H = H; a.W = W;
        a.halo = 2;
        if (W <= kStreamPX) { a.n_strips = 1; a.useful_w = W; }
        else { a.useful_w = kStreamPX - 2 * a.halo; a.n_strips = (W + a.useful_w - 1) / a.useful_w; }
        const int64_t cols = (int64_t)nb * a.n_strips;
        const int want = (int)std::max<int64_t>(1, (512 + cols - 1) / cols);
        a.useful_h = std::max(32, (H + want - 1) / want);
        a.n_blocks = (H + a.useful_h - 1) / a.useful_h;
        a.rows_c = a.n_blocks == 1 ? H : a.useful_h + 2 * a.halo;
        a.n_jobs = (int)(cols * a.n_blocks);
        a.jobs_per_wg = (a.n_jobs + 255) / 256;
        const int grid = (a.n_jobs + a.jobs_per_wg - 1) / a.jobs_per_wg;
        HIP_TRY(h, tail_launch(a, grid, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_STREAM) {
        StreamArgs a = op.stream;
        a.x = x;
        a.out = buf_ptr(h, op.out_buf[0]);
        a.out_stride = h->bufs[op.out_buf[0]].stride;
        a.blob = op.d_w;
        a.N = nb; a.H = H; a.W = W;
        a.halo = a.L + 1;
        // column strips of 48 computed pixels; row blocks only where whole images do not fill the chip
        if (W <= kStreamPX) { a.n_strips = 1; a.useful_w = W; }
        else { a.useful_w = kStreamPX - 2 * a.halo; a.n_strips = (W + a.useful_w - 1) / a.useful_w; }
        const int64_t cols = (int64_t)nb * a.n_strips;
        const int want = (int)std::max<int64_t>(1, (512 + cols - 1) / cols);
        a.useful_h = std::max(32, (H + want - 1) / want);
        a.n_blocks = (H + a.useful_h - 1) / a.useful_h;
        a.rows_c = a.n_blocks == 1 ? H : a.useful_h + 2 * a.halo;
        a.n_jobs = (int)(cols * a.n_blocks);
        a.jobs_per_wg = (a.n_jobs + 255) / 256;
        const int grid = (a.n_jobs + a.jobs_per_wg - 1) / a.jobs_per_wg;
        static long long* dbg = nullptr;
        if (getenv("DCSCN_STREAM_DBG")) {
            if (!dbg) HIP_TRY(h, hipMalloc((void**)&dbg, 16 * 64 * 4 * sizeof(long long)));
            HIP_TRY(h, hipMemsetAsync(dbg, 0, 16 * 64 * 4 * sizeof(long long), stream));
            a.dbg = dbg;
        }
        HIP_TRY(h, stream_launch(a, grid, stream));
        if (a.dbg) {
            std::vector<long long> host(16 * 64 * 4);
            HIP_TRY(h, hipStreamSynchronize(stream));
            HIP_TRY(h, hipMemcpy(host.data(), dbg, host.size() * sizeof(long long), hipMemcpyDeviceToHost));
            FILE* f = fopen(getenv("DCSCN_STREAM_DBG"), "w");
            if (f) {
                for (int w = 0; w < 2 * a.L + 1; ++w)
                    for (int t = 0; t < 64; ++t)
                        fprintf(f, "%d %d %d %lld %lld %lld %lld\n", w, (int)a.role[w], t, host[(w * 64 + t) * 4], host[(w * 64 + t) * 4 + 1], host[(w * 64 + t) * 4 + 2], host[(w * 64 + t) * 4 + 3]);
                fclose(f);
            }
        }
        return DCSCN_OK;
    }
    if (op.kind == OP_DW) {
        DwArgs a{};
        a.in = op.in_buf == EXT_X ? x : buf_ptr(h, op.in_buf);
        a.in_stride = op.in_buf == EXT_X ? 1 : h->bufs[op.in_buf].stride;
        a.in_off = op.in_off;
        a.chan_map = op.d_map;
        a.w = op.d_w;
        a.ks = op.ks;
        a.cin = op.cin;
        a.cout_phys = pad4(op.cin);
        a.N = nb; a.H = Hr; a.W = Wr;
        a.out = buf_ptr(h, op.out_buf[0]);
        a.out_stride = pad4(op.cin);
        HIP_TRY(h, depthwise_launch(a, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_COUT1) {
        Cout1Args a{};
        a.in = buf_ptr(h, op.in_buf);
        a.in_stride = h->bufs[op.in_buf].stride;
        a.in_off = op.in_off;
        a.cin_phys = op.cin_phys;
        a.w = op.d_w;
        a.scale = op.out_scale;
        a.bias = 0.0f;
        a.ks = op.ks;
        a.N = nb; a.H = Hr; a.W = Wr;
        a.out = y;
        a.out_stride = 1;
        a.res = op.residual ? x2 : nullptr;
        a.res_stride = 1;
        HIP_TRY(h, cout1_launch(a, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_CIN1) {
        Cin1Args a{};
        a.x = x;
        a.w = op.d_w; a.bias = op.d_bias; a.alpha = op.d_alpha;
        a.act = op.act;
        a.ks = op.ks;
        a.N = nb; a.H = Hr; a.W = Wr;
        a.cs = op.out_width[0];
        a.out.ptr = buf_ptr(h, op.out_buf[0]);
        a.out.stride = h->bufs[op.out_buf[0]].stride;
        a.out.off = op.out_off[0];
        a.out.width = op.out_width[0];
        HIP_TRY(h, cin1_launch(a, stream));
        return DCSCN_OK;
    }
    ConvArgs a{};
    a.in = buf_ptr(h, op.in_buf);
    // the depthwise scratch is re-strided per use (pad4(cin) of the separable conv that filled it)
    a.in_stride = op.in_stride_override > 0 ? op.in_stride_override : h->bufs[op.in_buf].stride;
    a.in_off = op.in_off;
    a.cin_phys = op.cin_phys;
    a.n_chunks = op.n_chunks;
    a.wpack = op.d_w; a.bias = op.d_bias; a.alpha = op.d_alpha;
    a.act = op.act;
    a.N = nb; a.H = Hr; a.W = Wr;
    a.tiles_x = (Wr + 15) / 16;
    a.tiles_y = (Hr + 4 * op.shape.mt - 1) / (4 * op.shape.mt);
    a.n_full = op.n_full;
    for (int i = 0; i < 2; ++i) {
        OutDesc& o = i == 0 ? a.out0 : a.out1;
        const int id = op.out_buf[i];
        o.ptr = id == EXT_Y ? y : buf_ptr(h, id);
        o.stride = id == EXT_Y ? 1 : h->bufs[id].stride;
        o.off = op.out_off[i];
        o.width = op.out_width[i];
    }
    a.split = op.split;
    a.ps = op.ps;
    a.ps_c = op.ps == 1 ? 1 : op.ps_c;
    a.vec4 = op.vec4 ? 1 : 0;
    a.res = op.residual ? x2 : nullptr;
    a.res_stride = 1;
    a.dww = op.d_dww;
    a.dwk = op.dwk;
    a.fold = op.fold_s > 0 ? 1 : 0;
    a.srctab = op.multi.empty() ? nullptr : op.d_srctab;
    if (op.shape.nin) HIP_TRY(h, nin_launch(op.shape.nt, a, op.n_tiles, stream));
    else if (op.shape.wino) HIP_TRY(h, wino_launch(op.shape.nt, a, op.n_tiles, stream));
    else HIP_TRY(h, conv_launch(op.shape, a, op.n_tiles, stream));
    return DCSCN_OK;
}

// Receptive-field radius of y_ in LR pixels: every launch widens it by floor(k/2) pixels of ITS resolution.
// (Summing over all launches over-counts the parallel A1 / B1->B2 branches by nothing: 1x1 convs add 0.)
int halo_lr_pixels(const dcscn_ctx* h) {
    double r = 0.0;
    for (const Op& op : h->ops) {
        const int k = op.kind == OP_CONV && op.dwk ? op.dwk : op.ks;
        r += (double)(op.halo >= 0 ? op.halo : k / 2) / op.res;
    }
    return (int)std::ceil(r - 1e-9);
}

int run_forward(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int H, int W, hipStream_t stream);

// An image that does not fit one pass of the layer chain (workspace budget / sub_batch_pixels) is cut into
// equally shaped windows that overlap by twice the receptive-field radius R; the windows run as an ordinary
// batch and every output pixel is taken from a window in which it lies >= R pixels away from any window edge
// that is not also an image edge.  There the value is the same function of the same inputs as in the untiled
// pass (SAME zero padding only ever acts at true image borders), so no per-layer masking is needed.
int run_tiled(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int H, int W, int64_t pass_pixels,
              hipStream_t stream) {
    const int R = halo_lr_pixels(h), s = h->cfg.scale;
    // window shape: as square as the pass allows, never wider / taller than the image
    int Ht = (int)std::min<int64_t>(H, std::max<int64_t>(1, (int64_t)std::sqrt((double)pass_pixels)));
    int Wt = (int)std::min<int64_t>(W, std::max<int64_t>(1, pass_pixels / Ht));
    if (Wt == W) Ht = (int)std::min<int64_t>(H, pass_pixels / Wt);
    if ((Ht < H && Ht <= 2 * R) || (Wt < W && Wt <= 2 * R))
        return fail(h, DCSCN_ERR_NOMEM, "image %dx%d needs spatial tiling, but a pass of %lld LR pixels is too small for windows "
                    "with a %d-pixel halo; raise sub_batch_pixels / workspace_budget_bytes", H, W, (long long)pass_pixels, R);
    auto starts = [&](int full, int win) {
        std::vector<int> v;
        if (win >= full) { v.push_back(0); return v; }
        const int stride = win - 2 * R;
        for (int a = 0; a + win < full; a += stride) v.push_back(a);
        v.push_back(full - win);
        return v;
    };
    const std::vector<int> ys = starts(H, Ht), xs = starts(W, Wt);
    const size_t tiles = (size_t)n * ys.size() * xs.size();
    const size_t lr = tiles * Ht * Wt, hr = lr * s * s;
    if (lr > h->tile_x_cap || hr > h->tile_y_cap) {
        HIP_TRY(h, hipStreamSynchronize(stream));
        for (float** p : {&h->tile_x, &h->tile_x2, &h->tile_y}) {
            if (*p) HIP_TRY(h, hipFree(*p));
            *p = nullptr;
        }
        h->tile_x_cap = h->tile_y_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->tile_x, lr * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->tile_x2, hr * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->tile_y, hr * sizeof(float)));
        h->tile_x_cap = lr;
        h->tile_y_cap = hr;
    }
    size_t t = 0;
    for (int img = 0; img < n; ++img)
        for (int wy : ys)
            for (int wx : xs) {
                HIP_TRY(h, hipMemcpy2DAsync(h->tile_x + t * Ht * Wt, (size_t)Wt * sizeof(float),
                                            x + ((size_t)img * H + wy) * W + wx, (size_t)W * sizeof(float),
                                            (size_t)Wt * sizeof(float), Ht, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(h, hipMemcpy2DAsync(h->tile_x2 + t * Ht * Wt * s * s, (size_t)Wt * s * sizeof(float),
                                            x2 + ((size_t)img * H * s + (size_t)wy * s) * W * s + (size_t)wx * s, (size_t)W * s * sizeof(float),
                                            (size_t)Wt * s * sizeof(float), (size_t)Ht * s, hipMemcpyDeviceToDevice, stream));
                ++t;
            }
    int rc = run_forward(h, h->tile_x, h->tile_x2, h->tile_y, (int)tiles, Ht, Wt, stream);
    if (rc) return rc;
    // scatter: window i owns [a_i + (a_i > 0 ? R : 0), a_{i+1} + R) -- up to the next window's first owned pixel
    auto owned = [&](const std::vector<int>& st, size_t i, int full, int win, int* lo, int* hi) {
        *lo = st[i] + (st[i] > 0 ? R : 0);
        *hi = i + 1 < st.size() ? st[i + 1] + R : full;
        (void)win;
    };
    t = 0;
    for (int img = 0; img < n; ++img)
        for (size_t iy = 0; iy < ys.size(); ++iy)
            for (size_t ix = 0; ix < xs.size(); ++ix) {
                int y0, y1, x0, x1;
                owned(ys, iy, H, Ht, &y0, &y1);
                owned(xs, ix, W, Wt, &x0, &x1);
                if (y1 > y0 && x1 > x0) {
                    const float* src = h->tile_y + t * Ht * Wt * s * s + ((size_t)(y0 - ys[iy]) * s) * Wt * s + (size_t)(x0 - xs[ix]) * s;
                    float* dst = y + ((size_t)img * H * s + (size_t)y0 * s) * W * s + (size_t)x0 * s;
                    HIP_TRY(h, hipMemcpy2DAsync(dst, (size_t)W * s * sizeof(float), src, (size_t)Wt * s * sizeof(float),
                                                (size_t)(x1 - x0) * s * sizeof(float), (size_t)(y1 - y0) * s,
                                                hipMemcpyDeviceToDevice, stream));
                }
                ++t;
            }
    return DCSCN_OK;
}

int run_forward(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int H, int W, hipStream_t stream) {
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_forward before dcscn_finalize");
    if (n < 0 || H <= 0 || W <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape n=%d h=%d w=%d", n, H, W);
    if (n == 0) return DCSCN_OK;
    if (!x || !x2 || !y) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t per_image = (int64_t)H * W;
    int64_t ws_per_lr_pixel = 0;      // workspace bytes per LR pixel
    for (const WsBuf& b : h->bufs) ws_per_lr_pixel += (int64_t)b.res * b.res * b.stride * (int64_t)sizeof(float);
    const int64_t pass_pixels = std::min<int64_t>(h->sub_batch_pixels, h->workspace_budget / std::max<int64_t>(ws_per_lr_pixel, 1));
    // sub_batch_pixels is a soft knob (a pass holds at least one image); the workspace budget is the hard one
    const int64_t budget_pixels = h->workspace_budget / std::max<int64_t>(ws_per_lr_pixel, 1);
    if (per_image > budget_pixels && h->spatial_tiling) return run_tiled(h, x, x2, y, n, H, W, budget_pixels, stream);
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>(n, pass_pixels / per_image));
    // two forwards of one handle share the arena: a forward on another stream than the previous one waits for it
    if (h->has_last && h->last_stream != stream) HIP_TRY(h, hipStreamWaitEvent(stream, h->done_ev, 0));
    int rc = ensure_workspace(h, nb, H, W, stream);
    while (rc == DCSCN_ERR_NOMEM && nb > 1) {            // less free memory than the budget assumed: smaller passes
        nb = (nb + 1) / 2;
        rc = ensure_workspace(h, nb, H, W, stream);
    }
    if (rc) return rc;
    if (h->tables_gen != h->carve_gen) {
        // the multi-source tables hold arena addresses: refill them behind the re-carve, on the launch stream
        for (Op& op : h->ops) {
            if (op.multi.empty()) continue;
            size_t q = 0;
            for (const auto& sg : op.multi) {
                const char* base = reinterpret_cast<const char*>(buf_ptr(h, sg.first));
                const unsigned stride = (unsigned)(h->bufs[sg.first].stride * sizeof(float));
                for (int c4 = 0; c4 < sg.second / 4 && q < op.h_srctab.size(); ++c4, ++q)
                    op.h_srctab[q] = NinSrcQuad{(unsigned long long)(uintptr_t)(base + 16 * c4), stride, 1u};
            }
            for (; q < op.h_srctab.size(); ++q) op.h_srctab[q] = NinSrcQuad{0, 0, 0};
            HIP_TRY(h, hipMemcpyAsync(op.d_srctab, op.h_srctab.data(), op.h_srctab.size() * sizeof(NinSrcQuad), hipMemcpyHostToDevice, stream));
        }
        h->tables_gen = h->carve_gen;
    }
    const int s = h->cfg.scale;
    const int batches = (n + nb - 1) / nb;
    const int nops = (int)h->ops.size();
    // profile mode: one event pair per launch, kept for every forward since the last dcscn_get_profile
    size_t ev_base = 0;
    if (h->profile) {
        ev_base = h->ev_used;
        const size_t need = ev_base + (size_t)batches * nops * 2;
        while (h->ev.size() < need) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
        h->ev_used = need;
        h->ev_forwards += 1;
    }
    for (int b = 0; b < batches; ++b) {
        const int b0 = b * nb;
        const int cnt = std::min(nb, n - b0);
        const float* xb = x + (size_t)b0 * H * W;
        const float* x2b = x2 + (size_t)b0 * H * s * W * s;
        float* yb = y + (size_t)b0 * H * s * W * s;
        for (int i = 0; i < nops; ++i) {
            if (h->profile) HIP_TRY(h, hipEventRecord(h->ev[ev_base + ((size_t)b * nops + i) * 2], stream));
            rc = launch_op(h, h->ops[i], cnt, H, W, xb, x2b, yb, stream);
            if (rc) return rc;
            if (h->profile) HIP_TRY(h, hipEventRecord(h->ev[ev_base + ((size_t)b * nops + i) * 2 + 1], stream));
        }
    }
    HIP_TRY(h, hipEventRecord(h->done_ev, stream));
    h->last_stream = stream;
    h->has_last = true;
    return DCSCN_OK;
}

// ---- Pillow-compatible bicubic resize on the device (resample.hip) -------------------------------------
int resample_table(dcscn_ctx* h, int in_size, int out_size, const dcscn_ctx::ResampleTable** out) {
    auto key = std::make_pair(in_size, out_size);
    auto it = h->resample_tables.find(key);
    if (it == h->resample_tables.end()) {
        std::vector<int> bounds;
        std::vector<double> kk;
        dcscn_ctx::ResampleTable t;
        t.ksize = resample_coeffs(in_size, out_size, &bounds, &kk);
        int rc = upload(h, bounds.data(), bounds.size() * sizeof(int), (void**)&t.d_bounds);
        if (!rc) rc = upload(h, kk.data(), kk.size() * sizeof(double), (void**)&t.d_kk);
        if (rc) return rc;
        it = h->resample_tables.emplace(key, t).first;
    }
    *out = &it->second;
    return DCSCN_OK;
}

int grow(dcscn_ctx* h, float** p, size_t* cap, size_t floats, hipStream_t stream) {
    if (floats <= *cap) return DCSCN_OK;
    HIP_TRY(h, hipStreamSynchronize(stream));
    if (*p) HIP_TRY(h, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void**)p, floats * sizeof(float));
    if (e != hipSuccess) return fail(h, DCSCN_ERR_NOMEM, "buffer of %zu floats: %s", floats, hipGetErrorString(e));
    *cap = floats;
    return DCSCN_OK;
}

// [n, H, W] -> [n, OH, OW], device pointers; horizontal pass first, as Pillow (a pass whose size does not
// change is skipped there too, so it adds no rounding)
int resize_device(dcscn_ctx* h, const float* in, float* out, int n, int H, int W, int OH, int OW, hipStream_t stream) {
    if (n <= 0) return DCSCN_OK;
    const float* src = in;
    if (OW != W) {
        const dcscn_ctx::ResampleTable* t;
        int rc = resample_table(h, W, OW, &t);
        if (rc) return rc;
        float* dst = out;
        if (OH != H) {
            rc = grow(h, &h->rs_tmp, &h->rs_tmp_cap, (size_t)n * H * OW, stream);
            if (rc) return rc;
            dst = h->rs_tmp;
        }
        HIP_TRY(h, resample_h_launch(src, dst, t->d_bounds, t->d_kk, t->ksize, (long long)n * H, W, OW, stream));
        src = dst;
    }
    if (OH != H) {
        const dcscn_ctx::ResampleTable* t;
        int rc = resample_table(h, H, OH, &t);
        if (rc) return rc;
        HIP_TRY(h, resample_v_launch(src, out, t->d_bounds, t->d_kk, t->ksize, n, H, OH, OW, stream));
    } else if (OW == W) {
        HIP_TRY(h, hipMemcpyAsync(out, in, (size_t)n * H * W * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    return DCSCN_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int dcscn_abi_version(void) { return DCSCN_ABI_VERSION; }

const char* dcscn_last_global_error(void) { return g_global_error.c_str(); }

int dcscn_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_global_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -DCSCN_ERR_HIP;
    }
    return n;
}

int dcscn_filter_schedule(int layers, int filters, int min_filters, double gamma, int32_t* out) {
    if (!out || layers <= 0 || gamma <= 0.0) {
        set_global_error("dcscn_filter_schedule: bad arguments");
        return DCSCN_ERR_INVALID_ARG;
    }
    std::vector<int> s;
    filter_schedule(layers, filters, std::min(filters, min_filters), gamma, s);
    for (int i = 0; i < layers; ++i) out[i] = s[i];
    return DCSCN_OK;
}

int dcscn_create(const dcscn_config* cfg, int device, dcscn_handle* out) {
    if (!cfg || !out) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "dcscn_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(dcscn_config))
        return fail(nullptr, DCSCN_ERR_INVALID_ARG, "dcscn_create: struct_size %d != %zu", cfg->struct_size, sizeof(dcscn_config));
    dcscn_config c = *cfg;
    c.min_filters = std::min(c.filters, c.min_filters);                     // DCSCN.py:36
    c.reconstruct_layers = std::max(c.reconstruct_layers, 1);               // DCSCN.py:42
    if (c.scale < 2 || c.scale > 4) return fail(nullptr, DCSCN_ERR_UNSUPPORTED, "scale %d (supported: 2, 3, 4)", c.scale);
    if (c.layers < 1 || c.layers > 256 || c.filters < 1) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "bad layers/filters");
    if (c.layers > 1 && !(c.filters_decay_gamma > 0.0)) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "filters_decay_gamma must be > 0");
    if (c.cnn_size != 1 && c.cnn_size != 3 && c.cnn_size != 5 && c.cnn_size != 7)
        return fail(nullptr, DCSCN_ERR_UNSUPPORTED, "cnn_size %d (supported: 1, 3, 5, 7)", c.cnn_size);
    if (c.channels != 1) return fail(nullptr, DCSCN_ERR_UNSUPPORTED, "channels %d (the reference itself only supports 1)", c.channels);
    if (c.batch_norm) return fail(nullptr, DCSCN_ERR_UNSUPPORTED, "batch_norm is not implemented");
    float dummy;
    if (kernel_act(c.activator, &dummy) < 0) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "Not implemented activator:%d", c.activator);
    if (c.legacy_no_c && c.use_nin) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "legacy_no_c requires use_nin = 0");
    if (c.reconstruct_layers > 1 && c.reconstruct_filters < 1) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "reconstruct_filters must be positive");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, DCSCN_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, DCSCN_ERR_INVALID_ARG, "device %d out of range [0, %d)", device, ndev);

    dcscn_ctx* h = new (std::nothrow) dcscn_ctx();
    if (!h) return fail(nullptr, DCSCN_ERR_NOMEM, "out of host memory");
    h->cfg = c;
    h->device = device;
    int rc = DCSCN_OK;
    do {
        if ((e = hipSetDevice(device)) != hipSuccess) { rc = fail(h, DCSCN_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e)); break; }
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { rc = fail(h, DCSCN_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); break; }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            rc = fail(h, DCSCN_ERR_HIP, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
            break;
        }
        if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) { rc = fail(h, DCSCN_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); break; }
        if ((e = hipEventCreateWithFlags(&h->done_ev, hipEventDisableTiming)) != hipSuccess) { rc = fail(h, DCSCN_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)); break; }
        if ((e = conv_init_kernels()) != hipSuccess) { rc = fail(h, DCSCN_ERR_HIP, "kernel attribute setup: %s", hipGetErrorString(e)); break; }
        {   // default workspace budget: at most 60 % of what is free now (other ranks may share the device)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
                h->workspace_budget = std::min<int64_t>(h->workspace_budget, std::max<int64_t>((int64_t)(free_b / 10 * 6), (int64_t)256 << 20));
        }
        rc = build_graph(h);
    } while (0);
    if (rc != DCSCN_OK) {
        g_global_error = h->error;
        dcscn_destroy(h);
        return rc;
    }
    *out = h;
    return DCSCN_OK;
}

int dcscn_num_tensors(dcscn_handle h) { return h ? (int)h->tensors.size() : -DCSCN_ERR_INVALID_ARG; }

int dcscn_tensor_info(dcscn_handle h, int index, char* name, int name_capacity, int64_t* shape, int* rank) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (index < 0 || index >= (int)h->tensors.size() || !name || name_capacity <= 0 || !shape || !rank)
        return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_tensor_info: bad argument");
    const TensorSpec& t = h->tensors[index];
    snprintf(name, (size_t)name_capacity, "%s", t.name.c_str());
    *rank = (int)t.shape.size();
    for (int i = 0; i < 4; ++i) shape[i] = i < *rank ? t.shape[i] : 1;
    return DCSCN_OK;
}

int dcscn_set_tensor(dcscn_handle h, const char* name, const float* data, const int64_t* shape, int rank) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!name || !data || !shape) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_set_tensor: null argument");
    if (h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_set_tensor after dcscn_finalize");
    auto it = h->tensor_index.find(name);
    if (it == h->tensor_index.end()) return fail(h, DCSCN_ERR_SHAPE, "variable '%s' is not part of this graph", name);
    TensorSpec& t = h->tensors[it->second];
    bool same = rank == (int)t.shape.size();
    for (int i = 0; same && i < rank; ++i) same = shape[i] == t.shape[i];
    if (!same) {
        std::string want, got;
        for (int64_t d : t.shape) want += std::to_string(d) + ",";
        for (int i = 0; i < rank && i < 8; ++i) got += std::to_string(shape[i]) + ",";
        return fail(h, DCSCN_ERR_SHAPE, "variable '%s': shape [%s] does not match graph shape [%s]", name, got.c_str(), want.c_str());
    }
    size_t count = 1;
    for (int64_t d : t.shape) count *= (size_t)d;
    for (size_t i = 0; i < count; ++i)
        if (!std::isfinite(data[i])) return fail(h, DCSCN_ERR_INVALID_ARG, "variable '%s' holds a non-finite value", name);
    t.data.assign(data, data + count);
    t.set = true;
    return DCSCN_OK;
}

namespace {
// ---- row-streamed feature extractor (feat_stream.hpp) ---------------------------------------------------
// Channel that lane group q (= lane >> 4) feeds into k-step s of 16-channel chunk ch, for an input ring of `quads` channel
// quads (feat_stream.hpp: StreamChunk); -1 = none (the filter row stays zero).
int stream_chunk_channel(int quads, int ch, int q, int s) {
    const int chunks = (quads + 3) / 4;
    const int ql = ch == chunks - 1 ? quads - 4 * (chunks - 1) : 4;
    if (ql >= 3) return q < ql ? 16 * ch + 4 * q + s : -1;
    if (ql == 2) return s < 2 ? 16 * ch + 4 * (q & 1) + 2 * (q >> 1) + s : -1;
    return s == 0 ? 16 * ch + q : -1;
}
// the (input quads, output tiles) pairs stream_conv_role is instantiated for (feat_stream.hpp: feat_stream)
bool stream_conv_supported(int in_quads, int out_tiles) {
    if (in_quads <= 5) return out_tiles == 1;
    if (in_quads <= 7) return out_tiles == 2;
    return true;
}

// The separable narrow nets (depthwise_separable, <= 7 feature layers of <= 32 filters, NIN of <= 32 channels): the
// launches CNN1/depthwise, CNN1 .. CNNL, B1+A1, B2 become ONE launch that keeps every intermediate tensor in LDS.
void fuse_feat_stream(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    const int L = c.layers;
    if (!h->stream_features || !c.depthwise_separable || c.cnn_size != 3 || !c.use_nin || L < 2 || L > kStreamMaxL) return;
    if (c.nin_filters2 > 16 || pad4(c.nin_filters) + pad4(c.nin_filters2) > 32) return;
    for (int i = 0; i < L; ++i)
        if (h->sched[i] > 32) return;
    // expected launch sequence
    const size_t n_rep = (size_t)L + 3;
    if (h->ops.size() < n_rep) return;
    auto is_ds3 = [](const Op& o) { return o.kind == OP_CONV && o.dwk == 3 && o.ks == 1 && o.segs.size() == 1 && o.act == ACT_ALPHA && o.ps == 1 && o.res == 1; };
    const Op& dw1 = h->ops[0];
    const Op& c1 = h->ops[1];
    if (dw1.kind != OP_DW || dw1.ks != 3 || dw1.in_buf != EXT_X || c1.kind != OP_CONV || c1.ks != 1 || c1.cin != 1 || c1.act != ACT_ALPHA || c1.segs.size() != 1) return;
    for (int i = 1; i < L; ++i)
        if (!is_ds3(h->ops[1 + i]) || h->ops[1 + i].cout != h->sched[i]) return;
    const Op& nin = h->ops[L + 1];
    const Op& b2 = h->ops[L + 2];
    if (nin.kind != OP_CONV || nin.ks != 1 || nin.dwk != 0 || nin.segs.size() != 2 || nin.act != ACT_ALPHA || !is_ds3(b2)) return;
    if (b2.out_buf[0] != nin.out_buf[1] || b2.out_off[0] != 0 || nin.out_off[1] != pad4(c.nin_filters2)) return;

    for (int i = 0; i < L; ++i) {
        const int cin = i == L - 1 ? c.nin_filters2 : h->sched[i], cout = i == L - 1 ? c.nin_filters2 : h->sched[i + 1];
        if (!stream_conv_supported(pad4(cin) / 4, (cout + 15) / 16)) return;
    }
    // LDS budget: rings + the filters that are indexed by a run-time layer (A1 || B1 slices, depthwise)
    auto units = [](int ch) { const int q = pad4(ch) / 4; return q | 1; };
    size_t lds = 0;
    for (int i = 0; i < L; ++i) lds += (size_t)3 * kStreamRowPx * units(h->sched[i]) * 16;
    lds += (size_t)4 * kStreamRowPx * units(c.nin_filters2) * 16;
    for (int i = 0; i < L; ++i) lds += (size_t)((h->sched[i] + 15) / 16) * 2 * 64 * 16;
    for (int i = 0; i + 1 < L; ++i) lds += (size_t)9 * (pad4(h->sched[i]) / 4) * 16;
    lds += (size_t)9 * (pad4(c.nin_filters2) / 4) * 16;
    for (int i = 0; i < L; ++i) {                     // pointwise filters [chunk][tile][64] float4, bias + slope
        const int cin = i == L - 1 ? c.nin_filters2 : h->sched[i], cout = i == L - 1 ? c.nin_filters2 : h->sched[i + 1];
        lds += (size_t)((cin + 15) / 16) * ((cout + 15) / 16) * 64 * 16 + 256;
    }
    lds += 256;
    if (lds > 160 * 1024) return;

    Op f;
    f.kind = OP_STREAM;
    f.name = "CNN1.." + b2.name + " (streamed)";
    f.ks = 3;
    f.cin = 1;
    f.cout = c.nin_filters + c.nin_filters2;
    f.res = 1;
    f.act = ACT_ALPHA;
    f.in_buf = EXT_X;
    f.out_buf[0] = f.out_buf[1] = b2.out_buf[0];
    f.out_width[0] = h->bufs[b2.out_buf[0]].stride;
    f.halo = L + 1;
    for (size_t i = 0; i < n_rep; ++i) {
        f.macs += h->ops[i].macs;
        f.fused.push_back(h->ops[i]);
    }
    f.bytes = 4 + 4 * (int64_t)h->bufs[b2.out_buf[0]].stride;
    const int t1 = nin.out_buf[0], cat = h->concat_buf;
    h->ops.erase(h->ops.begin(), h->ops.begin() + n_rep);
    h->ops.insert(h->ops.begin(), f);
    for (int dead : {t1, cat, dw1.out_buf[0]}) {
        bool used = false;
        for (const Op& o : h->ops) used = used || o.in_buf == dead || o.out_buf[0] == dead || o.out_buf[1] == dead;
        if (!used && dead >= 0) h->bufs[dead].stride = 0;
    }
    h->concat_buf = -1;                                    // nothing left for densify_features
}

// The x4 tail of the same nets: Up-PS, Up-PS2 (each separable 3x3 + depth_to_space(2)) and the separable 1 -> 1 R-CNN1 with
// the residual add become ONE launch (tail_stream.hpp); the C-channel tensor at 2x resolution stays in LDS.
void fuse_tail_stream(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    if (!h->stream_tail || !c.depthwise_separable || c.cnn_size != 3 || c.scale != 4 || !c.pixel_shuffler || h->ops.size() < 3) return;
    const size_t n = h->ops.size();
    const Op& u1 = h->ops[n - 3];
    const Op& u2 = h->ops[n - 2];
    const Op& rc = h->ops[n - 1];
    auto is_up = [](const Op& o) { return o.kind == OP_CONV && o.dwk == 3 && o.ks == 1 && o.segs.size() == 1 && o.act == ACT_NONE && o.ps == 2 && o.tconv_s == 0 && o.fold_s == 0; };
    if (!is_up(u1) || !is_up(u2) || u1.res != 1 || u2.res != 2) return;
    // tail_stream is instantiated for 8 channel quads in and out of Up-PS (every shipped separable checkpoint: A1 || B2 = 32
    // channels, pixel shuffler to 32); other widths keep the layer-by-layer tail
    if (u1.cin != 32 || u1.ps_c != 32) return;
    if (u2.cin != u1.ps_c || u2.ps_c != 1 || u2.in_buf != u1.out_buf[0]) return;
    for (int i = 0; i < u1.cin; ++i)
        if (u1.chan_map[i] != i) return;
    if (u1.in_off != 0) return;
    if (rc.kind != OP_COUT1 || rc.dw_w < 0 || rc.ks != 3 || !rc.residual || rc.in_buf != u2.out_buf[0] || rc.res != 4) return;

    Op f;
    f.kind = OP_TAIL;
    f.name = u1.name.substr(0, u1.name.find('/')) + ".." + rc.name + " (streamed)";
    f.ks = 3;
    f.cin = u1.cin;
    f.cout = 1;
    f.res = 1;
    f.in_buf = u1.in_buf;
    f.cin_phys = u1.cin_phys;
    f.residual = true;
    f.halo = 2;
    f.macs = u1.macs + u2.macs + rc.macs;
    f.bytes = 4 * (int64_t)u1.cin_phys + 4 * 16 * 2;
    f.fused = {u1, u2, rc};
    const int dead[2] = {u1.out_buf[0], u2.out_buf[0]};
    h->ops.erase(h->ops.end() - 3, h->ops.end());
    h->ops.push_back(f);
    for (int d : dead)
        if (d >= 0) h->bufs[d].stride = 0;
}

int pack_tail_stream(dcscn_ctx* h, Op& op) {
    const Op& u1 = op.fused[0];
    const Op& u2 = op.fused[1];
    const Op& rc = op.fused[2];
    const int cin = u1.cin, C = u1.ps_c;
    TailArgs& a = op.tail;
    a = TailArgs{};
    auto tens = [&](int id) -> const std::vector<float>& { return h->tensors[id].data; };
    int lds = 0;
    a.in.quads = cin / 4; a.in.units = a.in.quads | 1; a.in.slots = 3; a.in.off = lds;
    lds += 3 * kStreamRowPx * a.in.units * 16;
    a.u.quads = C / 4; a.u.units = a.u.quads | 1; a.u.slots = 6; a.u.off = lds;
    lds += 6 * (2 * kStreamPX + 2) * a.u.units * 16;
    a.v_off = lds;
    lds += 12 * (4 * kStreamPX + 4) * 4;
    a.ring_bytes = lds;
    std::vector<float> blob;
    auto region = [&](size_t floats) { const size_t base = blob.size(); blob.resize(base + floats, 0.0f); lds += (int)floats * 4; return base; };
    // Up-PS
    a.a_dww = lds;
    {
        const size_t base = region((size_t)9 * a.in.quads * 4);
        const std::vector<float>& dw = tens(u1.dw_w);                 // [3, 3, cin, 1]
        for (int k = 0; k < 9; ++k)
            for (int ci = 0; ci < cin; ++ci) blob[base + (size_t)k * a.in.quads * 4 + ci] = dw[(size_t)k * cin + ci];
    }
    a.a_wp = lds;
    {
        // [chunk][channel tile 0..7][lane] float4 over the 4C conv channels (tile = 2 * phase + half when C > 16)
        const size_t base = region((size_t)2 * 8 * 64 * 4);
        const ColSeg& sg = u1.segs[0];
        const std::vector<float>& pw = tens(sg.w);                    // [1, 1, cin, 4C]: column phase * C + c
        const int tiles = C > 16 ? 2 : 1;
        for (int ch = 0; ch < 2; ++ch)
            for (int n = 0; n < 8; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int st = 0; st < 4; ++st) {
                        const int ci = stream_chunk_channel(cin / 4, ch, lane >> 4, st);
                        const int ph = n / tiles, cc = 16 * (n % tiles) + (lane & 15);
                        if (ci >= 0 && ci < cin && ph < 4 && cc < C) blob[base + (((size_t)ch * 8 + n) * 64 + lane) * 4 + st] = pw[(size_t)ci * 4 * C + ph * C + cc];
                    }
    }
    a.a_bias = lds;
    {
        const size_t base = region(8 * 16);                           // [channel tile][16]
        const ColSeg& sg = u1.segs[0];
        const int tiles = C > 16 ? 2 : 1;
        for (int ph = 0; ph < 4; ++ph)
            for (int cc = 0; cc < C; ++cc) blob[base + (ph * tiles + cc / 16) * 16 + cc % 16] = sg.b >= 0 ? tens(sg.b)[ph * C + cc] : 0.0f;
    }
    // Up-PS2
    a.b_dww = lds;
    {
        const size_t base = region((size_t)9 * a.u.quads * 4);
        const std::vector<float>& dw = tens(u2.dw_w);                 // [3, 3, C, 1]
        for (int k = 0; k < 9; ++k)
            for (int ci = 0; ci < C; ++ci) blob[base + (size_t)k * a.u.quads * 4 + ci] = dw[(size_t)k * C + ci];
    }
    a.b_wp = lds;
    {
        const size_t base = region((size_t)2 * 64 * 4);
        const std::vector<float>& pw = tens(u2.segs[0].w);            // [1, 1, C, 4]
        for (int ch = 0; ch < 2; ++ch)
            for (int lane = 0; lane < 64; ++lane)
                for (int st = 0; st < 4; ++st) {
                    const int ci = stream_chunk_channel(C / 4, ch, lane >> 4, st), co = lane & 15;
                    if (ci >= 0 && ci < C && co < 4) blob[base + ((size_t)ch * 64 + lane) * 4 + st] = pw[(size_t)ci * 4 + co];
                }
    }
    a.b_bias = lds;
    {
        const size_t base = region(4);
        const ColSeg& sg = u2.segs[0];
        for (int co = 0; co < 4; ++co) blob[base + co] = sg.b >= 0 ? tens(sg.b)[co] : 0.0f;
    }
    a.ldsw_bytes = lds - a.ring_bytes;
    if (lds > 160 * 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: tail_stream needs %d bytes of LDS", lds);
    for (int k = 0; k < 9; ++k) a.c_w[k] = tens(rc.dw_w)[k];         // [3, 3, 1, 1]
    a.c_scale = tens(rc.segs[0].w)[0];                               // [1, 1, 1, 1]
    return upload(h, blob.data(), blob.size() * sizeof(float), (void**)&op.d_w);
}

int pack_feat_stream(dcscn_ctx* h, Op& op) {
    const dcscn_config& c = h->cfg;
    const int L = c.layers, nb = c.nin_filters2, na = c.nin_filters;
    StreamArgs& a = op.stream;
    a = StreamArgs{};
    a.L = L;
    a.n_conv = L;                      // CNN2 .. CNNL and B2
    a.total_lag = 2 * L + 1;
    a.nb_quads = pad4(nb) / 4;
    auto ring = [&](int ch, int slots, int* off) {
        StreamRing r;
        r.quads = pad4(ch) / 4;
        r.units = r.quads | 1;
        r.slots = slots;
        r.off = *off;
        *off += slots * kStreamRowPx * r.units * 16;
        return r;
    };
    int lds = 0;
    std::vector<StreamRing> fr(L);
    for (int i = 0; i < L; ++i) fr[i] = ring(h->sched[i], 3, &lds);
    a.b1 = ring(nb, 4, &lds);
    a.first_out = fr[0];
    a.ring_bytes = lds;

    std::vector<float> blob;
    auto tens = [&](int id) -> const std::vector<float>& { return h->tensors[id].data; };
    // --- LDS image: A1 || B1 slices, then the depthwise filters ---
    const Op& nin = op.fused[L + 1];
    const ColSeg& sb = nin.segs[0];
    const ColSeg& sa = nin.segs[1];
    int cbase = 0;
    for (int i = 0; i < L; ++i) {
        const int C = h->sched[i];
        StreamNinSrc& s = a.nin[i];
        s.ring = fr[i];
        s.chunks = (C + 15) / 16;
        s.last_ql = s.ring.quads - 4 * (s.chunks - 1);
        s.w = lds;
        const size_t base = blob.size();
        blob.resize(base + (size_t)s.chunks * 2 * 64 * 4, 0.0f);
        for (int ch = 0; ch < s.chunks; ++ch)
            for (int n = 0; n < 2; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int k = 0; k < 4; ++k) {
                        const int ci = stream_chunk_channel(s.ring.quads, ch, lane >> 4, k), v = 16 * n + (lane & 15);
                        if (ci < 0 || ci >= C) continue;
                        const ColSeg* sg = nullptr;
                        int co = 0;
                        if (v < pad4(nb)) { if (v < nb) { sg = &sb; co = v; } }
                        else if (v - pad4(nb) < na) { sg = &sa; co = v - pad4(nb); }
                        if (!sg) continue;
                        const int cols = (int)h->tensors[sg->w].shape.back();
                        float w = tens(sg->w)[(size_t)(cbase + ci) * cols + sg->col0 + co];
                        if (sg->dw1 >= 0) w = tens(sg->dw1)[cbase + ci] * w;      // folded 1x1 depthwise half, as finalize_op
                        blob[base + ((size_t)(ch * 2 + n) * 64 + lane) * 4 + k] = w;
                    }
        lds += s.chunks * 2 * 64 * 16;
        cbase += C;
    }
    for (int i = 0; i < L; ++i) {                      // conv i: CNN(i+2) for i < L-1, B2 for i == L-1
        const bool is_b2 = i == L - 1;
        const Op& src = is_b2 ? op.fused[L + 2] : op.fused[2 + i];
        const int cin = is_b2 ? nb : h->sched[i];
        StreamConv& cv = a.conv[i];
        cv.in = is_b2 ? a.b1 : fr[i];
        cv.lag = is_b2 ? 2 * L + 1 : 2 * (i + 1);
        cv.to_global = is_b2 ? 1 : 0;
        if (is_b2) { cv.out = StreamRing{-1, 0, pad4(nb) / 4, 0}; }
        else cv.out = fr[i + 1];
        cv.dww = lds;
        const size_t base = blob.size();
        const int quads = pad4(cin) / 4;
        blob.resize(base + (size_t)9 * quads * 4, 0.0f);
        const std::vector<float>& dw = tens(src.dw_w);          // [3, 3, cin, 1]
        for (int k = 0; k < 9; ++k)
            for (int ci = 0; ci < cin; ++ci) blob[base + (size_t)k * quads * 4 + ci] = dw[(size_t)k * cin + ci];
        lds += 9 * quads * 16;
    }
    auto bias_alpha = [&](const Op& o, const ColSeg& sg, int dst, size_t base) {
        for (int co = 0; co < sg.cout; ++co) {
            blob[base + dst + co] = sg.b >= 0 ? tens(sg.b)[sg.col0 + co] : 0.0f;
            blob[base + 32 + dst + co] = (sg.alpha >= 0 ? tens(sg.alpha)[sg.col0 + co] : o.const_alpha) - 1.0f;     // stream_prelu wants alpha - 1
        }
    };
    // --- pointwise filters [chunk][tile][lane] float4, bias, slope of the streamed convs ---
    for (int i = 0; i < L; ++i) {
        const bool is_b2 = i == L - 1;
        const Op& src = is_b2 ? op.fused[L + 2] : op.fused[2 + i];
        const ColSeg& sg = src.segs[0];
        const int cin = is_b2 ? nb : h->sched[i], cout = sg.cout;
        const int chunks = (cin + 15) / 16, tiles = (cout + 15) / 16;
        StreamConv& cv = a.conv[i];
        cv.wp = lds;
        size_t base = blob.size();
        blob.resize(base + (size_t)chunks * tiles * 64 * 4, 0.0f);
        const std::vector<float>& pw = tens(sg.w);              // [1, 1, cin, cout]
        for (int ch = 0; ch < chunks; ++ch)
            for (int n = 0; n < tiles; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int st = 0; st < 4; ++st) {
                        const int ci = stream_chunk_channel(pad4(cin) / 4, ch, lane >> 4, st), co = 16 * n + (lane & 15);
                        if (ci >= 0 && ci < cin && co < cout) blob[base + ((size_t)(ch * tiles + n) * 64 + lane) * 4 + st] = pw[(size_t)ci * cout + co];
                    }
        lds += chunks * tiles * 64 * 16;
        cv.ba = lds;
        base = blob.size();
        blob.resize(base + 64, 0.0f);
        bias_alpha(src, sg, 0, base);
        lds += 256;
    }
    {
        a.nin_ba = lds;
        const size_t base = blob.size();
        blob.resize(base + 64, 0.0f);
        bias_alpha(nin, sb, 0, base);
        bias_alpha(nin, sa, pad4(nb), base);
        lds += 256;
    }
    a.ldsw_src = 0;
    a.ldsw_bytes = lds - a.ring_bytes;
    if ((size_t)a.ldsw_bytes != blob.size() * sizeof(float)) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat_stream LDS image size");
    {
        // wave -> role: wave w runs on SIMD w & 3, and on a SIMD the MFMAs and the VALU instructions of all its waves execute
        // one after the other (tools/mfma_valu_overlap.hip), so a SIMD's time per row is the SUM of its roles' estimated cycles
        // (32 per MFMA + 4.5 per other VALU instruction).  Exhaustive search for the assignment with the smallest maximum:
        // the L A1 || B1 roles are interchangeable, the other L + 1 roles are tried on every SIMD (4^(L+1) <= 65536).
        auto ksteps = [](int quads) { const int ch = (quads + 3) / 4, ql = quads - 4 * (ch - 1); return 4 * (ch - 1) + (ql >= 3 ? 4 : ql); };
        std::vector<int> cost(1 + L), code(1 + L);
        cost[0] = (int)(4.5 * 180); code[0] = 0;                  // CNN1
        int nin_mfma = 0;
        for (int i = 0; i < L; ++i) {
            const int chunks = (a.conv[i].in.quads + 3) / 4, tiles = (a.conv[i].out.quads + 3) / 4;
            cost[1 + i] = 32 * 3 * tiles * ksteps(a.conv[i].in.quads) + (int)(4.5 * (54 * chunks + 36 * tiles + 100));
            code[1 + i] = 1 + i;
            nin_mfma += 6 * ksteps(fr[i].quads);
        }
        const int nin_cost = 32 * nin_mfma / L + (int)(4.5 * 85);
        const int waves = 2 * L + 1;
        int cap[4];
        for (int sd = 0; sd < 4; ++sd) cap[sd] = (waves - sd + 3) / 4;      // waves sd, sd + 4, ... below `waves`
        long best_key = -1;
        std::vector<int> best_sd(1 + L, 0);
        int best_nin[4] = {0, 0, 0, 0};
        const int combos = 1 << (2 * (L + 1));
        for (int m = 0; m < combos; ++m) {
            int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
            for (int r = 0; r <= L; ++r) { const int sd = (m >> (2 * r)) & 3; load[sd] += cost[r]; used[sd] += 1; }
            if (used[0] > cap[0] || used[1] > cap[1] || used[2] > cap[2] || used[3] > cap[3]) continue;
            // the L interchangeable roles: always onto the least loaded SIMD with a free wave
            int nin[4] = {0, 0, 0, 0};
            bool ok = true;
            for (int k = 0; k < L && ok; ++k) {
                int pick = -1;
                for (int sd = 0; sd < 4; ++sd)
                    if (used[sd] + nin[sd] < cap[sd] && (pick < 0 || load[sd] < load[pick])) pick = sd;
                if (pick < 0) { ok = false; break; }
                nin[pick] += 1;
                load[pick] += nin_cost;
            }
            if (!ok) continue;
            const long mx = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
            long sq = 0;
            for (int sd = 0; sd < 4; ++sd) sq += (long)(load[sd] / 16) * (load[sd] / 16);
            const long key = mx * 1000000 + sq / 16;
            if (best_key < 0 || key < best_key) {
                best_key = key;
                for (int r = 0; r <= L; ++r) best_sd[r] = (m >> (2 * r)) & 3;
                for (int sd = 0; sd < 4; ++sd) best_nin[sd] = nin[sd];
            }
        }
        if (best_key < 0) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat_stream role placement");
        int used[4] = {0, 0, 0, 0};
        for (int w = 0; w < 16; ++w) a.role[w] = 0;
        for (int r = 0; r <= L; ++r) { const int sd = best_sd[r]; a.role[sd + 4 * used[sd]] = (int8_t)code[r]; used[sd] += 1; }
        int slot = 0;
        for (int sd = 0; sd < 4; ++sd)
            for (int k = 0; k < best_nin[sd]; ++k) { a.role[sd + 4 * used[sd]] = (int8_t)(16 + slot); used[sd] += 1; slot += 1; }
    }
    if (lds > 160 * 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat_stream needs %d bytes of LDS", lds);

    // --- CNN1 (global, read once into registers): depthwise[9] (+3 pad), pointwise[32], bias[32], slope[32] ---
    {
        const Op& dw1 = op.fused[0];
        const Op& c1 = op.fused[1];
        a.first_w = (int)blob.size();
        blob.resize(blob.size() + 12 + 96, 0.0f);
        for (int k = 0; k < 9; ++k) blob[a.first_w + k] = tens(dw1.dw_w)[k];
        const ColSeg& sg = c1.segs[0];
        for (int co = 0; co < sg.cout; ++co) blob[a.first_w + 12 + co] = tens(sg.w)[co];      // [1, 1, 1, C1]
        bias_alpha(c1, sg, 0, (size_t)a.first_w + 44);
    }
    return upload(h, blob.data(), blob.size() * sizeof(float), (void**)&op.d_w);
}
}  // namespace

// ---- dense per-layer feature buffers ------------------------------------------------------------------
// build_graph lets every feature layer store into its slice of ONE [n, H, W, sum pad4(C_i)] tensor, which makes tf.concat
// free -- but a narrow slice of a wide NHWC record is a partial, misaligned cache-line access per pixel, for the layer that
// writes it and for the layer that reads it (measured on the c-DCSCN nets: two structurally opposite kernels took exactly the
// same time).  When every consumer of the whole concat is a conv_nin launch (A1 || B1, or the non-NIN "C"
// layer), this pass gives each feature layer its own dense [n, H, W, pad4(C_i)] buffer and hands the consumers the list
// of buffers: conv_nin walks them through a per-quad source table (conv_nin.hpp: MULTI).  The virtual channel order is
// unchanged, so chan_map and the packed filters stay as they are.
void densify_features(dcscn_ctx* h) {
    if (!h->dense_features || h->concat_buf < 0) return;
    const int cat = h->concat_buf;
    const int cat_stride = h->bufs[cat].stride;
    std::vector<int> consumers;
    for (size_t i = 0; i < h->ops.size(); ++i) {
        const Op& op = h->ops[i];
        if (op.in_buf != cat) continue;
        bool slice = false;
        for (const auto& sl : h->concat_slices) slice = slice || (op.in_off == sl.first && op.cin_phys == pad4(sl.second));
        if (slice && !(op.in_off == 0 && op.cin_phys == cat_stride)) continue;                 // a feature layer reading its predecessor
        if (op.in_off != 0 || op.cin_phys != cat_stride || !nin_eligible(h, op) || (size_t)((op.cin_phys + 15) / 16) * 64 > 16 * 1024) return;
        consumers.push_back((int)i);
    }
    if (consumers.empty() || h->concat_slices.size() < 2) return;
    std::vector<int> nb;
    // (row strides padded to 64 / 128 bytes were measured: noise on the wide nets, 3-18 % slower on the narrow ones)
    for (const auto& sl : h->concat_slices) nb.push_back(new_buf(h, pad4(sl.second), 1));
    for (Op& op : h->ops) {
        for (size_t k = 0; k < h->concat_slices.size(); ++k) {
            const int off = h->concat_slices[k].first, w4 = pad4(h->concat_slices[k].second);
            for (int o = 0; o < 2; ++o)
                if (op.out_buf[o] == cat && op.out_off[o] == off) { op.out_buf[o] = nb[k]; op.out_off[o] = 0; }
            if (op.in_buf == cat && op.in_off == off && op.cin_phys == w4 && !(off == 0 && w4 == cat_stride)) { op.in_buf = nb[k]; op.in_off = 0; }
        }
    }
    for (int ci : consumers) {
        Op& op = h->ops[ci];
        for (size_t k = 0; k < nb.size(); ++k) op.multi.push_back({nb[k], pad4(h->concat_slices[k].second)});
    }
    bool used = false;
    for (const Op& o : h->ops) used = used || (o.multi.empty() && o.in_buf == cat) || o.out_buf[0] == cat || o.out_buf[1] == cat;
    if (!used) h->bufs[cat].stride = 0;                         // the concat tensor no longer exists
}

int dcscn_finalize(dcscn_handle h) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (h->finalized) return DCSCN_OK;
    for (const TensorSpec& t : h->tensors)
        if (!t.set) return fail(h, DCSCN_ERR_MISSING_TENSOR, "variable '%s' was never set", t.name.c_str());
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->fold_tail) fold_linear_tail(h);      // silently keeps the layer-by-layer graph where it does not apply
    fuse_tail_stream(h);
    fuse_feat_stream(h);
    densify_features(h);
    for (Op& op : h->ops) {
        int rc = finalize_op(h, op);
        if (rc) return rc;
        if (!op.multi.empty()) {
            op.h_srctab.assign((size_t)4 * op.n_chunks, NinSrcQuad{0, 0, 0});
            rc = upload(h, op.h_srctab.data(), op.h_srctab.size() * sizeof(NinSrcQuad), (void**)&op.d_srctab);
            if (rc) return rc;
        }
    }
    h->prof_ms.assign(h->ops.size(), 0.0);
    h->finalized = true;
    return DCSCN_OK;
}

int dcscn_num_layers(dcscn_handle h) { return h ? (int)h->layers.size() : -DCSCN_ERR_INVALID_ARG; }

int dcscn_layer_info_get(dcscn_handle h, int index, dcscn_layer_info* out) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!out || index < 0 || index >= (int)h->layers.size()) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_layer_info_get: bad argument");
    *out = h->layers[index];
    return DCSCN_OK;
}

int dcscn_num_ops(dcscn_handle h) { return h ? (int)h->ops.size() : -DCSCN_ERR_INVALID_ARG; }

int dcscn_op_info_get(dcscn_handle h, int index, dcscn_op_info* out) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!out || index < 0 || index >= (int)h->ops.size()) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_op_info_get: bad argument");
    const Op& op = h->ops[index];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", op.name.c_str());
    snprintf(out->kernel, sizeof out->kernel, "%s", op.kind == OP_CONV ? (op.shape.wino ? "conv_wino2" : op.shape.nin ? "conv_nin" : "conv_igemm") : op.kind == OP_CIN1 ? "conv_cin1" : op.kind == OP_COUT1 ? "conv_cout1" : op.kind == OP_STREAM ? "feat_stream" : op.kind == OP_TAIL ? "tail_stream" : "depthwise");
    out->kernel_size = op.ks;
    out->in_channels = op.cin;
    out->out_channels = op.cout;
    out->resolution = op.res;
    if (op.kind == OP_CONV && h->finalized) {
        out->mt = op.shape.mt; out->nt = op.shape.nt; out->kc = op.shape.kc; out->n_tiles = op.n_tiles;
    }
    out->macs_per_lr_pixel = op.macs;
    out->bytes_per_lr_pixel = op.bytes;
    out->executed_macs_per_lr_pixel = op.macs;
    if (op.kind == OP_CONV && h->finalized) {
        const int64_t r2 = (int64_t)op.res * op.res;
        const int64_t k_exec = (int64_t)op.n_chunks * op.shape.kc;             // padded input channels
        if (op.shape.nin) {
            const int64_t tiles = (int64_t)op.n_tiles * (op.shape.nt - 1) + op.n_full;
            out->executed_macs_per_lr_pixel = r2 * k_exec * tiles * 16;
        } else if (op.shape.wino) {
            const int64_t tiles = (int64_t)op.n_tiles * (op.shape.nt - 1) + op.n_full;
            out->executed_macs_per_lr_pixel = r2 * 4 * k_exec * tiles * 16;    // 16 products per 2x2 outputs
        } else {
            out->executed_macs_per_lr_pixel = r2 * op.ks * op.ks * k_exec * (int64_t)op.n_tiles * op.shape.nt * 16;
        }
    }
    return DCSCN_OK;
}

int dcscn_set_option(dcscn_handle h, const char* key, int64_t value) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!key) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_set_option: null key");
    if (!strcmp(key, "sub_batch_pixels")) {
        if (value < 1) return fail(h, DCSCN_ERR_INVALID_ARG, "sub_batch_pixels must be >= 1");
        h->sub_batch_pixels = value;
        return DCSCN_OK;
    }
    if (!strcmp(key, "workspace_budget_bytes")) {
        if (value < 1) return fail(h, DCSCN_ERR_INVALID_ARG, "workspace_budget_bytes must be >= 1");
        h->workspace_budget = value;
        h->budget_user_set = true;
        return DCSCN_OK;
    }
    if (!strcmp(key, "spatial_tiling")) {
        h->spatial_tiling = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "fold_linear_tail")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the fold_linear_tail option must be set before dcscn_finalize");
        h->fold_tail = value != 0;
        h->fold_force = value == 2;
        return DCSCN_OK;
    }
    if (!strcmp(key, "stream_tail")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the stream_tail option must be set before dcscn_finalize");
        h->stream_tail = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "stream_features")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the stream_features option must be set before dcscn_finalize");
        h->stream_features = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "dense_features")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the dense_features option must be set before dcscn_finalize");
        h->dense_features = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "nin_gemm")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the nin_gemm option must be set before dcscn_finalize");
        h->nin = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "winograd")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the winograd option must be set before dcscn_finalize");
        h->winograd = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "profile")) {
        h->profile = value != 0;
        return DCSCN_OK;
    }
    return fail(h, DCSCN_ERR_INVALID_ARG, "unknown option '%s'", key);
}

int dcscn_forward_device(dcscn_handle h, const float* x, const float* x2, float* y, int n, int height, int width, void* stream) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    return run_forward(h, x, x2, y, n, height, width, stream ? (hipStream_t)stream : h->stream);
}

static int ensure_io(dcscn_ctx* h, size_t lr_floats, size_t hr_floats) {
    if (lr_floats > h->io_x_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->io_x) HIP_TRY(h, hipFree(h->io_x));
        h->io_x = nullptr; h->io_x_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->io_x, lr_floats * sizeof(float)));
        h->io_x_cap = lr_floats;
    }
    if (hr_floats > h->io_y_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->io_x2) HIP_TRY(h, hipFree(h->io_x2));
        if (h->io_y) HIP_TRY(h, hipFree(h->io_y));
        h->io_x2 = h->io_y = nullptr; h->io_y_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->io_x2, hr_floats * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->io_y, hr_floats * sizeof(float)));
        h->io_y_cap = hr_floats;
    }
    return DCSCN_OK;
}

// Host-buffer forward in up to 4 chunks of images: the upload of chunk i+1 and the download of chunk i-1 run while chunk i
// computes (blocking hipMemcpy on pageable user memory runs at PCIe speed and overlaps kernels of the handle's stream;
// hipMemcpyAsync would stage pageable buffers at ~3 GB/s on this stack).  x2 == nullptr: x2 = bicubic(x) on the device.
static int forward_host_chunked(dcscn_ctx* h, const float* x, const float* x2, float* y, int n, int height, int width) {
    const int s = h->cfg.scale;
    const size_t lr1 = (size_t)height * width, hr1 = lr1 * s * s;
    int rc = ensure_io(h, lr1 * n, hr1 * n);
    if (rc) return rc;
    const int chunks = (hr1 * n * sizeof(float) >= ((size_t)8 << 20) && n >= 8) ? 4 : 1;
    while ((int)h->host_ev.size() < chunks) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->host_ev.push_back(e);
    }
    const bool trace = getenv("DCSCN_TRACE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int begin[5];
    for (int i = 0; i <= chunks; ++i) begin[i] = (int)((int64_t)n * i / chunks);
    // the first chunk must be the largest: it sizes the workspace carve the later ones reuse
    auto span = [&](int i, int* b, int* cnt) { *b = begin[chunks - 1 - i] ; *cnt = begin[chunks - i] - begin[chunks - 1 - i]; };
    auto download = [&](int i) -> int {
        int b, cnt;
        span(i, &b, &cnt);
        HIP_TRY(h, hipEventSynchronize(h->host_ev[i]));
        HIP_TRY(h, hipMemcpy(y + (size_t)b * hr1, h->io_y + (size_t)b * hr1, (size_t)cnt * hr1 * sizeof(float), hipMemcpyDeviceToHost));
        return DCSCN_OK;
    };
    for (int i = 0; i < chunks; ++i) {
        int b, cnt;
        span(i, &b, &cnt);
        if (cnt > 0) {
            HIP_TRY(h, hipMemcpy(h->io_x + (size_t)b * lr1, x + (size_t)b * lr1, (size_t)cnt * lr1 * sizeof(float), hipMemcpyHostToDevice));
            if (x2) HIP_TRY(h, hipMemcpy(h->io_x2 + (size_t)b * hr1, x2 + (size_t)b * hr1, (size_t)cnt * hr1 * sizeof(float), hipMemcpyHostToDevice));
            else rc = resize_device(h, h->io_x + (size_t)b * lr1, h->io_x2 + (size_t)b * hr1, cnt, height, width, height * s, width * s, h->stream);   // DCSCN.py:552-554
            if (!rc) rc = run_forward(h, h->io_x + (size_t)b * lr1, h->io_x2 + (size_t)b * hr1, h->io_y + (size_t)b * hr1, cnt, height, width, h->stream);
            if (rc) return rc;
        }
        HIP_TRY(h, hipEventRecord(h->host_ev[i], h->stream));
        if (i > 0 && (rc = download(i - 1))) return rc;
    }
    if ((rc = download(chunks - 1))) return rc;
    if (trace) fprintf(stderr, "dcscn_forward: %d chunk(s), %.2f ms\n", chunks, now() - t0);
    return DCSCN_OK;
}

int dcscn_forward(dcscn_handle h, const float* x, const float* x2, float* y, int n, int height, int width) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_forward before dcscn_finalize");
    if (n < 0 || height <= 0 || width <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape n=%d h=%d w=%d", n, height, width);
    if (n == 0) return DCSCN_OK;
    if (!x || !x2 || !y) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return forward_host_chunked(h, x, x2, y, n, height, width);
}

int dcscn_resample_table(int in_size, int out_size, int* ksize, int* bounds, double* weights, int capacity) {
    if (in_size <= 0 || out_size <= 0 || !ksize) return DCSCN_ERR_INVALID_ARG;
    std::vector<int> b;
    std::vector<double> k;
    *ksize = resample_coeffs(in_size, out_size, &b, &k);
    if (!bounds && !weights) return DCSCN_OK;                       // size query
    if (!bounds || !weights || capacity < (int)k.size()) return DCSCN_ERR_INVALID_ARG;
    std::copy(b.begin(), b.end(), bounds);
    std::copy(k.begin(), k.end(), weights);
    return DCSCN_OK;
}

int dcscn_resize_bicubic(dcscn_handle h, const float* in, float* out, int n, int height, int width, int out_height, int out_width) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (n < 0 || height <= 0 || width <= 0 || out_height <= 0 || out_width <= 0)
        return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape n=%d %dx%d -> %dx%d", n, height, width, out_height, out_width);
    if (n == 0) return DCSCN_OK;
    if (!in || !out) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t ni = (size_t)n * height * width, no = (size_t)n * out_height * out_width;
    int rc = grow(h, &h->rs_in, &h->rs_in_cap, ni, h->stream);
    if (!rc) rc = grow(h, &h->rs_out, &h->rs_out_cap, no, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(h->rs_in, in, ni * sizeof(float), hipMemcpyHostToDevice));
    rc = resize_device(h, h->rs_in, h->rs_out, n, height, width, out_height, out_width, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->rs_out, no * sizeof(float), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int dcscn_resize_bicubic_device(dcscn_handle h, const float* in, float* out, int n, int height, int width, int out_height,
                                int out_width, void* stream) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (n < 0 || height <= 0 || width <= 0 || out_height <= 0 || out_width <= 0)
        return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape n=%d %dx%d -> %dx%d", n, height, width, out_height, out_width);
    if (n == 0) return DCSCN_OK;
    if (!in || !out) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return resize_device(h, in, out, n, height, width, out_height, out_width, stream ? (hipStream_t)stream : h->stream);
}

int dcscn_forward_lr(dcscn_handle h, const float* x, float* y, int n, int height, int width) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_forward_lr before dcscn_finalize");
    if (n < 0 || height <= 0 || width <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape n=%d h=%d w=%d", n, height, width);
    if (n == 0) return DCSCN_OK;
    if (!x || !y) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return forward_host_chunked(h, x, nullptr, y, n, height, width);
}

// do()'s self-ensemble for the image pair already in io_x / io_x2; leaves the float64 mean in ens_out (enqueued, not synchronised)
static int ensemble_on_device(dcscn_ctx* h, int height, int width, int n) {
    const int s = h->cfg.scale;
    const size_t lr = (size_t)height * width, hr = lr * s * s;
    const int na = std::min(n, 4), nb = n - na;               // types 0-3 keep [h, w]; 4-7 are [w, h]
    int rc = grow(h, &h->ens_x, &h->ens_x_cap, n * lr, h->stream);
    if (!rc) rc = grow(h, &h->ens_x2, &h->ens_x2_cap, n * hr, h->stream);
    if (!rc) rc = grow(h, &h->ens_y, &h->ens_y_cap, n * hr, h->stream);
    if (!rc) rc = grow(h, &h->ens_out, &h->ens_out_cap, 2 * hr, h->stream);     // doubles
    if (rc) return rc;
    // util.flip(image, i) for i < n (DCSCN.py:562-564), on the device
    HIP_TRY(h, ensemble_gather_launch(h->io_x, h->ens_x, height, width, n, h->stream));
    HIP_TRY(h, ensemble_gather_launch(h->io_x2, h->ens_x2, height * s, width * s, n, h->stream));
    // two batches: the reference runs n forwards of batch 1 (DCSCN.py:565-569)
    rc = run_forward(h, h->ens_x, h->ens_x2, h->ens_y, na, height, width, h->stream);
    if (!rc && nb > 0)
        rc = run_forward(h, h->ens_x + (size_t)na * lr, h->ens_x2 + (size_t)na * hr, h->ens_y + (size_t)na * hr, nb, width, height, h->stream);
    if (rc) return rc;
    // output = zeros(float64); output += flip(y_i, invert=True) for i ascending; output /= n  (DCSCN.py:560-573)
    HIP_TRY(h, ensemble_reduce_launch(h->ens_y, reinterpret_cast<double*>(h->ens_out), height * s, width * s, n, h->stream));
    return DCSCN_OK;
}

int dcscn_forward_ensemble(dcscn_handle h, const float* x, const float* x2, double* y, int height, int width, int n_ensemble) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_forward_ensemble before dcscn_finalize");
    if (!x || !x2 || !y) return fail(h, DCSCN_ERR_INVALID_ARG, "null image pointer");
    if (n_ensemble < 1 || n_ensemble > 8) return fail(h, DCSCN_ERR_INVALID_ARG, "n_ensemble %d outside [1, 8]", n_ensemble);
    if (height <= 0 || width <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape h=%d w=%d", height, width);
    HIP_TRY(h, hipSetDevice(h->device));
    const int s = h->cfg.scale;
    const size_t lr = (size_t)height * width, hr = lr * s * s;
    int rc = ensure_io(h, lr, hr);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(h->io_x, x, lr * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->io_x2, x2, hr * sizeof(float), hipMemcpyHostToDevice));
    rc = ensemble_on_device(h, height, width, n_ensemble);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(y, h->ens_out, hr * sizeof(double), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

// ---- colour conversions and the RGB pipelines of evaluate.py / sr.py (color.hip) ---------------------------------

static int color_args(dcscn_handle h, const void* a, const void* b, int64_t n, const char* what) {
    if (!a || !b) return fail(h, DCSCN_ERR_INVALID_ARG, "%s: null pointer", what);
    if (n < 0) return fail(h, DCSCN_ERR_INVALID_ARG, "%s: negative pixel count", what);
    return DCSCN_OK;
}

int dcscn_convert_rgb_to_y(dcscn_handle h, const uint8_t* rgb, double* y, int64_t n_pixels) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    int rc = color_args(h, rgb, y, n_pixels, "dcscn_convert_rgb_to_y");
    if (rc || n_pixels == 0) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = grow(h, &h->col_rgb, &h->col_rgb_cap, (size_t)(3 * n_pixels + 3) / 4, h->stream);
    if (!rc) rc = grow(h, &h->col_d, &h->col_d_cap, (size_t)2 * n_pixels, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(h->col_rgb, rgb, (size_t)3 * n_pixels, hipMemcpyHostToDevice));
    HIP_TRY(h, rgb_to_y_launch(reinterpret_cast<const uint8_t*>(h->col_rgb), reinterpret_cast<double*>(h->col_d), nullptr, n_pixels, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(y, h->col_d, (size_t)n_pixels * sizeof(double), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int dcscn_convert_rgb_to_ycbcr(dcscn_handle h, const uint8_t* rgb, double* ycbcr, int64_t n_pixels) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    int rc = color_args(h, rgb, ycbcr, n_pixels, "dcscn_convert_rgb_to_ycbcr");
    if (rc || n_pixels == 0) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = grow(h, &h->col_rgb, &h->col_rgb_cap, (size_t)(3 * n_pixels + 3) / 4, h->stream);
    if (!rc) rc = grow(h, &h->col_d, &h->col_d_cap, (size_t)6 * n_pixels, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(h->col_rgb, rgb, (size_t)3 * n_pixels, hipMemcpyHostToDevice));
    HIP_TRY(h, rgb_to_ycbcr_launch(reinterpret_cast<const uint8_t*>(h->col_rgb), reinterpret_cast<double*>(h->col_d), n_pixels, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(ycbcr, h->col_d, (size_t)3 * n_pixels * sizeof(double), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int dcscn_convert_y_and_cbcr_to_rgb(dcscn_handle h, const double* y, const double* cbcr, double* rgb, int64_t n_pixels) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    int rc = color_args(h, y, cbcr, n_pixels, "dcscn_convert_y_and_cbcr_to_rgb");
    if (!rc && !rgb) rc = fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_convert_y_and_cbcr_to_rgb: null pointer");
    if (rc || n_pixels == 0) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    rc = grow(h, &h->col_d, &h->col_d_cap, (size_t)6 * n_pixels, h->stream);            // y | cbcr
    if (!rc) rc = grow(h, &h->col_d2, &h->col_d2_cap, (size_t)6 * n_pixels, h->stream);
    if (rc) return rc;
    double* dy = reinterpret_cast<double*>(h->col_d);
    double* dc = dy + n_pixels;
    HIP_TRY(h, hipMemcpy(dy, y, (size_t)n_pixels * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dc, cbcr, (size_t)2 * n_pixels * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, y_cbcr_to_rgb_launch(dy, nullptr, dc, nullptr, reinterpret_cast<double*>(h->col_d2), n_pixels, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(rgb, h->col_d2, (size_t)3 * n_pixels * sizeof(double), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

// Y image already on the device in col_y32 [H, W] (float32 of the float64 luma) -> LR, x2, y on the device.
// Leaves the result as float32 in io_y (n_ensemble == 1) or as float64 in ens_out (n_ensemble > 1); enqueued only.
static int sr_from_lr_on_device(dcscn_ctx* h, int lh, int lw, int n_ensemble) {
    const int s = h->cfg.scale;
    int rc = resize_device(h, h->io_x, h->io_x2, 1, lh, lw, lh * s, lw * s, h->stream);                 // DCSCN.py:552-554 / 683
    if (rc) return rc;
    if (n_ensemble > 1) return ensemble_on_device(h, lh, lw, n_ensemble);
    return run_forward(h, h->io_x, h->io_x2, h->io_y, 1, lh, lw, h->stream);
}

static int download_sr(dcscn_ctx* h, size_t hr, int n_ensemble, double* y) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (n_ensemble > 1) {
        HIP_TRY(h, hipMemcpy(y, h->ens_out, hr * sizeof(double), hipMemcpyDeviceToHost));
    } else {                                   // sess.run returns float32 (DCSCN.py:575-578): widened exactly
        std::vector<float> tmp(hr);
        HIP_TRY(h, hipMemcpy(tmp.data(), h->io_y, hr * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < hr; ++i) y[i] = (double)tmp[i];
    }
    return DCSCN_OK;
}

int dcscn_evaluate_rgb(dcscn_handle h, const uint8_t* rgb, int height, int width, int n_ensemble, double* true_y, float* lr, double* y) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_evaluate_rgb before dcscn_finalize");
    if (!rgb || !y) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_evaluate_rgb: null pointer");
    if (n_ensemble < 1 || n_ensemble > 8) return fail(h, DCSCN_ERR_INVALID_ARG, "n_ensemble %d outside [1, 8]", n_ensemble);
    const int s = h->cfg.scale;
    if (height <= 0 || width <= 0 || height % s || width % s)
        return fail(h, DCSCN_ERR_INVALID_ARG, "image %dx%d is not aligned to the scale %d (set_image_alignment, utilty.py:196-208)", height, width, s);
    HIP_TRY(h, hipSetDevice(h->device));
    const int lh = height / s, lw = width / s;
    const size_t hr = (size_t)height * width, lrn = (size_t)lh * lw;
    int rc = ensure_io(h, lrn, hr);
    if (!rc) rc = grow(h, &h->col_rgb, &h->col_rgb_cap, (3 * hr + 3) / 4, h->stream);
    if (!rc) rc = grow(h, &h->col_d, &h->col_d_cap, 2 * hr, h->stream);
    if (!rc) rc = grow(h, &h->col_y32, &h->col_y32_cap, hr, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(h->col_rgb, rgb, 3 * hr, hipMemcpyHostToDevice));
    // convert_rgb_to_y in float64 (utilty.py:146-147); the LR image is Pillow's BICUBIC on the mode-'F' (float32) copy of it
    HIP_TRY(h, rgb_to_y_launch(reinterpret_cast<const uint8_t*>(h->col_rgb), reinterpret_cast<double*>(h->col_d), h->col_y32, (long long)hr, h->stream));
    rc = resize_device(h, h->col_y32, h->io_x, 1, height, width, lh, lw, h->stream);                    // loader.py:64-65
    if (!rc) rc = sr_from_lr_on_device(h, lh, lw, n_ensemble);
    if (!rc) rc = download_sr(h, hr, n_ensemble, y);
    if (rc) return rc;
    if (true_y) HIP_TRY(h, hipMemcpy(true_y, h->col_d, hr * sizeof(double), hipMemcpyDeviceToHost));
    if (lr) HIP_TRY(h, hipMemcpy(lr, h->io_x, lrn * sizeof(float), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int dcscn_sr_rgb(dcscn_handle h, const uint8_t* rgb, const uint8_t* rgb_upscaled, int height, int width, int n_ensemble, double* y, double* rgb_out) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!h->finalized) return fail(h, DCSCN_ERR_STATE, "dcscn_sr_rgb before dcscn_finalize");
    if (!rgb || !rgb_upscaled || !rgb_out) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_sr_rgb: null pointer");
    if (n_ensemble < 1 || n_ensemble > 8) return fail(h, DCSCN_ERR_INVALID_ARG, "n_ensemble %d outside [1, 8]", n_ensemble);
    if (height <= 0 || width <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "bad shape h=%d w=%d", height, width);
    HIP_TRY(h, hipSetDevice(h->device));
    const int s = h->cfg.scale;
    const size_t lrn = (size_t)height * width, hr = lrn * s * s;
    int rc = ensure_io(h, lrn, hr);
    if (!rc) rc = grow(h, &h->col_rgb, &h->col_rgb_cap, (3 * hr + 3) / 4 + (3 * lrn + 3) / 4 + 4, h->stream);
    if (!rc) rc = grow(h, &h->col_d2, &h->col_d2_cap, 6 * hr, h->stream);
    if (rc) return rc;
    uint8_t* d_up = reinterpret_cast<uint8_t*>(h->col_rgb);
    uint8_t* d_lr = d_up + ((3 * hr + 15) & ~(size_t)15);
    HIP_TRY(h, hipMemcpy(d_lr, rgb, 3 * lrn, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_up, rgb_upscaled, 3 * hr, hipMemcpyHostToDevice));
    // input_y_image = convert_rgb_to_y(org_image); do(input_y_image): x = float32(Y) (DCSCN.py:597-601)
    HIP_TRY(h, rgb_to_y_launch(d_lr, nullptr, h->io_x, (long long)lrn, h->stream));
    rc = sr_from_lr_on_device(h, height, width, n_ensemble);
    if (rc) return rc;
    // convert_y_and_cbcr_to_rgb(output_y, convert_rgb_to_ycbcr(bicubic RGB)[:, :, 1:3]) (DCSCN.py:603-605)
    HIP_TRY(h, y_cbcr_to_rgb_launch(n_ensemble > 1 ? reinterpret_cast<const double*>(h->ens_out) : nullptr, n_ensemble > 1 ? nullptr : h->io_y,
                                    nullptr, d_up, reinterpret_cast<double*>(h->col_d2), (long long)hr, h->stream));
    if (y) rc = download_sr(h, hr, n_ensemble, y);
    else HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (rc) return rc;
    HIP_TRY(h, hipMemcpy(rgb_out, h->col_d2, 3 * hr * sizeof(double), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int dcscn_get_profile(dcscn_handle h, double* ms, int capacity) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!ms || capacity < 0) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_get_profile: bad argument");
    const int nops = (int)h->ops.size();
    std::vector<double> acc(nops, 0.0);
    const int forwards = h->ev_forwards;
    if (h->ev_used > 0) {
        HIP_TRY(h, hipDeviceSynchronize());
        const size_t launches = h->ev_used / 2;
        for (size_t l = 0; l < launches; ++l) {
            float t = 0.0f;
            HIP_TRY(h, hipEventElapsedTime(&t, h->ev[2 * l], h->ev[2 * l + 1]));
            acc[l % nops] += t;
        }
    }
    h->ev_used = 0;
    h->ev_forwards = 0;
    if (forwards > 1)
        for (double& v : acc) v /= forwards;
    for (int i = 0; i < std::min(capacity, nops); ++i) ms[i] = acc[i];
    return DCSCN_OK;
}

int64_t dcscn_workspace_bytes(dcscn_handle h) { return h ? (int64_t)h->arena_bytes : -1; }

const char* dcscn_last_error(dcscn_handle h) { return h ? h->error.c_str() : g_global_error.c_str(); }

int dcscn_get_stream(dcscn_handle h, void** stream) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!stream) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_get_stream: null pointer");
    *stream = (void*)h->stream;
    return DCSCN_OK;
}

int dcscn_synchronize(dcscn_handle h) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->has_last) HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return DCSCN_OK;
}

int dcscn_destroy(dcscn_handle h) {
    if (!h) return DCSCN_OK;
    (void)hipSetDevice(h->device);
    if (h->has_last) (void)hipStreamSynchronize(h->last_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->done_ev) (void)hipEventDestroy(h->done_ev);
    for (hipEvent_t e : h->host_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (void* p : h->device_allocs) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    for (float* p : {h->tile_x, h->tile_x2, h->tile_y, h->rs_tmp, h->rs_in, h->rs_out, h->ens_x, h->ens_x2, h->ens_y, h->ens_out, h->col_rgb, h->col_d,
                     h->col_d2, h->col_y32})
        if (p) (void)hipFree(p);
    if (h->io_x) (void)hipFree(h->io_x);
    if (h->io_x2) (void)hipFree(h->io_x2);
    if (h->io_y) (void)hipFree(h->io_y);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return DCSCN_OK;
}

}  // extern "C"
