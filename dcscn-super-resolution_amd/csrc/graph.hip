// The launch plan: SuperResolution.build_graph (DCSCN.py:222-325) restated as a list of graph layers and a list of kernel
// launches over a small set of workspace tensors, and the graph rewrites applied to it before the filters are packed.
//
//   CONCAT  [n, H, W, sum(pad4(filters_i))]  every feature layer stores straight into its channel slice, so tf.concat
//                                            (DCSCN.py:259) costs nothing (densify_features: one dense buffer per layer instead)
//   T1      B1 output;  T2 = Concat2 = [B2 | A1] (DCSCN.py:281), or the "C" layer's output
//   UPk     depth_to_space outputs (the shuffle happens in the producing conv's store)
//   Rk      extra reconstruction layers;  DW  scratch of the depthwise half of separable convs
//
// All slices start on a 4-channel boundary and are padded to 4 channels; the consumer's repacked filter has zero rows for
// padding channels.
#include "plan.h"

#pragma clang fp contract(off)

namespace dcscn_impl {

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
    int concat_stride = 0, total = 0, concat_data = 0;     // concat_data: channels that hold data (each layer padded to 4): the bytes accounting
    for (int i = 0; i < c.layers; ++i) {
        if (h->sched[i] <= 0) return fail(h, DCSCN_ERR_INVALID_ARG, "feature layer %d has %d filters", i + 1, h->sched[i]);
        slice_off[i] = concat_stride;
        concat_stride += (h->sched[i] + 7) & ~7;               // slices start on 8-channel boundaries (r05): a channel octet of the K axis of A1 || B1
                                                               // belongs to ONE layer, with or without densify_features, float32 or P16 (p16.hpp)
        total += h->sched[i];
        concat_data += pad4(h->sched[i]);
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
            f.bytes = 4 * (int64_t)concat_data + 4 * (pad4(na) + pad4(nb));
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

// 3x3 layers the Winograd kernel does not take because they have ONE 16-channel tile or a destination without the 16-byte store form
// (a pixel shuffler to fewer than 4 channels per sub-pixel: the x3 stage of the c-DCSCN nets, 32 -> 9) still go to conv3_h under split16:
// one tile of the direct form on the f16 pipe against conv_igemm's f32 rate (0.41 -> 0.13 ms on that layer); conv_igemm stays behind
// them as the float32 kernel (split16 = 0, flagged images)
bool h16_direct_eligible(const dcscn_ctx* h, const Op& op) {
    return h->winograd && !wino_eligible(h, op) && op.kind == OP_CONV && op.ks == 3 && op.dwk == 0 && op.cin_phys >= 24 && op.segs.size() == 1 &&
           op.tconv_s == 0 && op.fold_s == 0 && op.in_stride_override <= 0;
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
    // With split16 the composite of an x2 stage (16 channels = ONE tile on conv5_h, two workgroups per CU) wins everywhere, the
    // c-DCSCN nets included: 0.25 ms against 0.43 (x2), 0.90 against 1.55 for the second stage of x4; at x3 (36 channels, three
    // tiles, one workgroup per CU) the work rule still decides (0.79 folded against 0.61 for c-DCSCN x3).
    const bool one_tile_h16 = h->split16 && (h->split16_mask & 1) && u.ps == 2;
    if (!h->fold_force && !one_tile_h16 && 25 * pad16(4 * u.ps * u.ps) >= 9 * u.ps * u.ps * u.ps_c) return false;
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

// ---- the WHOLE linear tail as one conv (x3, x4; r06) ------------------------------------------------------
// build_pixel_shuffler_layer is called with activator = None for EVERY stage (tf_graph.py:238-249, DCSCN.py:300-311: no activator argument), so at x4
// the chain  Up-PS conv + bias -> depth_to_space(2) -> Up-PS2 conv + bias -> depth_to_space(2) -> R-CNN1  is one affine map, as the one-stage x3
// chain is, dense or depthwise separable (a separable conv is the dense conv  w[t][ci][co] = dw[t][ci] pw[ci][co]).  HR pixel (S y + a, S x + b)
// is a 5x5 conv of the LR neighbourhood of (y, x) with a kernel per sub-pixel phase (a, b): S^2 <= 16 phases = ONE 16-channel tile on
// conv5_h.  The zero padding of the intermediate maps (which the reference applies to the SHUFFLED maps, not to the LR one) only acts on the
// first / last row and column of the image: there the kernel is another one per (vy, vx) in {interior, first, last, both}^2, and instead of
// computing every variant for every pixel (fold_linear_tail: 4 variants per phase as extra channels -- 36 / 64 channels at x3 / x4, more with
// two stages) the border ring is a launch of its own (conv5_h.hpp: fold_border), 8 % of a 48 x 48 patch.  pack.hip: pack_foldx composes the
// kernels in float64.  Replaces [Up-PS, folded Up-PS2 + R-CNN1] (x4), [Up-PS, R-CNN1] (x3), tail_stream (separable x4); those launches stay
// in `fused` as the float32 plan of a flagged image and the split16 = 0 path.  Option "fold_whole_tail" (0: the r05 plans).
bool fold_whole_tail(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    if (!h->fold_tail || !h->fold_whole || !h->split16 || !(h->split16_mask & 1)) return false;
    if (!c.pixel_shuffler || c.cnn_size != 3 || c.reconstruct_layers > 1 || (c.scale != 3 && c.scale != 4)) return false;
    size_t i0 = h->ops.size();
    for (size_t i = 0; i < h->ops.size(); ++i)
        if (h->ops[i].name.rfind("Up-PS", 0) == 0) { i0 = i; break; }
    if (i0 >= h->ops.size()) return false;
    const Op first = h->ops[i0].kind == OP_TAIL ? h->ops[i0].fused[0] : h->ops[i0];
    if (first.kind != OP_CONV || first.in_buf < 0 || first.res != 1 || first.act != ACT_NONE || first.tconv_s != 0 || first.segs.size() != 1 ||
        first.in_stride_override > 0 || !first.multi.empty()) return false;
    int64_t macs = 0;
    for (size_t i = i0; i < h->ops.size(); ++i) {
        const Op& o = h->ops[i];
        if ((o.kind != OP_CONV && o.kind != OP_COUT1 && o.kind != OP_TAIL) || o.act != ACT_NONE) return false;
        for (size_t k = 0; k < i0; ++k)                        // nothing in front of the tail reads what it writes
            if (o.out_buf[0] >= 0 && h->ops[k].in_buf == o.out_buf[0]) return false;
        macs += o.macs;
    }
    const Op& last = h->ops.back();
    if (last.out_buf[0] != EXT_Y || !last.residual) return false;
    Op f;
    f.kind = OP_FOLDX;
    f.name = "Up-PS.." + std::string(c.reconstruct_layers > 1 ? "R-CNN" : "R-CNN1") + " (folded)";
    f.ks = 5;
    f.cin = first.cin;
    f.cout = c.scale * c.scale;
    f.res = 1;
    f.act = ACT_NONE;
    f.in_buf = first.in_buf; f.in_off = first.in_off; f.cin_phys = first.cin_phys;
    f.chan_map = first.chan_map;
    f.fold_s = c.scale;
    f.ps = c.scale; f.ps_c = 1;
    f.out_buf[0] = f.out_buf[1] = EXT_Y;
    f.out_width[0] = 1;
    f.residual = true;
    f.vec4 = false;
    f.halo = 2;
    f.macs = macs;
    f.bytes = 4 * (int64_t)first.cin_phys + 8 * (int64_t)c.scale * c.scale;
    f.fused.assign(h->ops.begin() + i0, h->ops.end());
    h->ops.erase(h->ops.begin() + i0, h->ops.end());
    h->ops.push_back(f);
    return true;
}

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


// ---- dense per-layer feature buffers ------------------------------------------------------------------
// build_graph lets every feature layer store into its slice of ONE [n, H, W, sum pad4(C_i)] tensor, which makes tf.concat
// free -- but a narrow slice of a wide NHWC record is a partial, misaligned cache-line access per pixel, for the layer that
// writes it and for the layer that reads it (measured on the c-DCSCN nets: two structurally opposite kernels took exactly the
// same time).  When every consumer of the whole concat is a conv_nin launch (A1 || B1, or the non-NIN "C"
// layer), this pass gives each feature layer its own dense [n, H, W, pad4(C_i)] buffer and hands the consumers the list
// of buffers: conv_nin walks them through a per-quad source table (conv_nin.hpp: MULTI).  The virtual channel order is
// unchanged, so chan_map and the packed filters stay as they are.  (The concat's slices start on 8-channel boundaries -- build_graph --
// so the table holds a padding quad behind a layer whose quad count is odd: a channel octet of the K axis belongs to one layer, which is
// what lets the same filter image serve the float32 buffers and their P16 form: a 16-byte unit of p16.hpp holds 8 channels of ONE tensor.)
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

// ---- row-streamed feature extractor of the non-separable narrow nets (feat3_stream.hpp) --------------------------------------
// CNN1 .. CNNL of a net whose feature layers are plain 3x3 convs of at most 32 channels (the c-DCSCN checkpoints: 32 .. 8) become ONE
// launch; runs behind densify_features (every layer stores into its own dense tensor, the 1x1 GEMM A1 || B1 reads them as before).
// The layers' own launches stay in Op::fused: the float32 plan of a flagged image and split16 = 0 run those.
void fuse_feat3_stream(dcscn_ctx* h) {
    const dcscn_config& c = h->cfg;
    const int L = c.layers;
    if (!h->stream_dense || c.depthwise_separable || c.cnn_size != 3 || L < 2 || L > kS3MaxL || (int)h->ops.size() < L) return;
    int waves = 1;                                                // CNN1 + one per conv
    size_t lds = 0;
    for (int i = 0; i < L; ++i) {
        if (h->sched[i] > 32) return;
        if (i > 0) waves += 1;
        if (i + 1 < L) lds += (size_t)4 * kStreamRowPx * (2 * ((h->sched[i] + 7) / 8) + 1) * 16;
    }
    if (waves > kS3MaxWaves || lds > 158 * 1024) return;
    const Op& c1 = h->ops[0];
    if (c1.kind != OP_CIN1 || c1.ks != 3 || c1.in_buf != EXT_X || c1.act != ACT_ALPHA || c1.out_off[0] != 0 || c1.out_buf[0] < 0 || c1.segs.size() != 1) return;
    for (int i = 1; i < L; ++i) {
        const Op& o = h->ops[i];
        if (o.kind != OP_CONV || o.ks != 3 || o.dwk != 0 || o.ps != 1 || o.residual || o.act != ACT_ALPHA || o.segs.size() != 1 || o.tconv_s > 0 || o.fold_s > 0 ||
            o.res != 1 || o.cin != h->sched[i - 1] || o.cout != h->sched[i] || o.in_buf != h->ops[i - 1].out_buf[0] || o.in_off != 0 || o.out_off[0] != 0 ||
            o.out_buf[0] < 0 || o.split < (1 << 29) || !o.multi.empty())
            return;
        for (size_t k = 0; k < o.chan_map.size(); ++k)
            if (o.chan_map[k] != (int)k) return;
        // every layer's tensor must be its own (densify_features): nobody else may write it
        for (int k = 0; k < i; ++k)
            if (h->ops[k].out_buf[0] == o.out_buf[0]) return;
    }
    // r06: A1 || B1 and B2 inside the launch (feat3_stream.hpp: s3_nin_role / s3_trio_role; option "stream_nin") -- no feature map goes to HBM, Concat2
    // [B2 | A1] is the launch's only output.  Instantiated for the c-DCSCN shape: L = 7, B1 = one octet, A1 || B1 = 32 channels, the last two
    // feature layers read two octets and write one tile; anything else keeps the r05 plan (the layers' rows to global memory, conv_nin_h behind).
    bool nin_on = false;
    if (h->stream_nin && c.use_nin && L == 7 && c.nin_filters2 == 8 && c.nin_filters == 24 && (int)h->ops.size() >= L + 2) {
        const Op& nin = h->ops[L];
        const Op& b2 = h->ops[L + 1];
        auto octs = [&](int i) { return (h->sched[i] + 7) / 8; };
        bool ok = nin.kind == OP_CONV && nin.ks == 1 && nin.dwk == 0 && nin.segs.size() == 2 && nin.act == ACT_ALPHA && nin.ps == 1 && !nin.residual &&
                  nin.segs[0].dw1 < 0 && nin.segs[1].dw1 < 0 && (int)nin.multi.size() == L && nin.out_buf[0] >= 0 && nin.out_buf[1] >= 0 &&
                  nin.out_off[0] == 0 && nin.out_off[1] == 8 && nin.cin == [&] { int k = 0; for (int i = 0; i < L; ++i) k += h->sched[i]; return k; }();
        for (int i = 0; ok && i < L; ++i) ok = nin.multi[i].first == h->ops[i].out_buf[0];
        ok = ok && b2.kind == OP_CONV && b2.ks == 3 && b2.dwk == 0 && b2.cin == 8 && b2.cout == 8 && b2.act == ACT_ALPHA && b2.ps == 1 && !b2.residual &&
             b2.segs.size() == 1 && b2.in_buf == nin.out_buf[0] && b2.in_off == 0 && b2.out_buf[0] == nin.out_buf[1] && b2.out_off[0] == 0 &&
             b2.split >= (1 << 29) && b2.tconv_s == 0 && b2.fold_s == 0 && b2.res == 1 && h->bufs[nin.out_buf[1]].stride == 32;
        // the pair roles are instantiated for: conv[L - 4] reads three octets, conv[L - 3] and conv[L - 2] two; one output tile each
        ok = ok && octs(L - 4) == 3 && octs(L - 3) == 2 && octs(L - 2) == 2 && h->sched[L - 3] <= 16 && h->sched[L - 2] <= 16 && h->sched[L - 1] <= 16;
        const size_t lds2 = lds + (size_t)4 * kStreamRowPx * (2 * octs(L - 1) + 1) * 16 + (size_t)4 * kStreamRowPx * 3 * 16;
        nin_on = ok && lds2 <= 158 * 1024;
    }
    Op f;
    f.kind = OP_STREAM3;
    f.name = "CNN1.." + h->ops[nin_on ? L + 1 : L - 1].name + " (streamed)";
    f.ks = 3;
    f.cin = 1;
    f.cout = nin_on ? c.nin_filters + c.nin_filters2 : h->sched[L - 1];
    f.res = 1;
    f.act = ACT_ALPHA;
    f.in_buf = EXT_X;
    f.out_buf[0] = f.out_buf[1] = nin_on ? h->ops[L].out_buf[1] : h->ops[L - 1].out_buf[0];
    f.out_width[0] = nin_on ? 32 : pad4(h->sched[L - 1]);
    f.halo = nin_on ? L + 1 : L;
    f.bytes = 4;
    const int n_rep = nin_on ? L + 2 : L;
    for (int i = 0; i < n_rep; ++i) {
        f.macs += h->ops[i].macs;
        if (!nin_on) {
            f.bytes += 4 * (int64_t)pad4(h->sched[i]);
            f.extra_out.push_back(h->ops[i].out_buf[0]);
        }
        f.fused.push_back(h->ops[i]);
    }
    if (nin_on) {
        f.bytes += 4 * 32;
        f.extra_out.push_back(f.out_buf[0]);
        f.stream3.nin.on = 1;                                    // (pack_feat3_stream fills the rest)
    }
    h->ops.erase(h->ops.begin(), h->ops.begin() + n_rep);
    h->ops.insert(h->ops.begin(), f);
}

// ---- P16 tensors (p16.hpp) ----------------------------------------------------------------------------
// A workspace tensor is kept in the pre-split form when EVERY launch that writes it can store (hi | lo) units -- conv_cin1, conv3_h,
// conv3_h8, conv_nin_h with a plain NHWC destination on a 16-channel boundary, conv3_h through a pixel shuffler to >= 16 channels -- and EVERY launch that reads it is a split16 kernel
// reading the whole tensor from channel 0 in its natural channel order (conv3_h / conv3_h8 / conv5_h; conv_nin_h when ALL its sources
// qualify).  Fixed point over the launch list; runs after finalize_op (it needs to know which launches have a split16 variant).
void plan_p16(dcscn_ctx* h) {
    const size_t nbuf = h->bufs.size();
    std::vector<char> ok(nbuf, 0), written(nbuf, 0), read(nbuf, 0);
    for (size_t b = 0; b < nbuf; ++b) ok[b] = h->bufs[b].stride > 0;
    const bool mixed = false;
    auto inputs = [&](const Op& op) {
        std::vector<int> v;
        if (!op.multi.empty()) for (const auto& m : op.multi) v.push_back(m.first);
        else if (op.in_buf >= 0) v.push_back(op.in_buf);
        return v;
    };
    auto can_read = [&](const Op& op) {
        if ((op.kind != OP_CONV && op.kind != OP_FOLDX) || !op.h16.on || op.dwk != 0 || op.in_stride_override > 0) return false;
        if (!op.multi.empty()) return op.shape.nin != 0;          // (the pad-8 virtual K axis of densify_features)
        if (op.in_buf < 0 || op.in_off != 0 || op.cin_phys != h->bufs[op.in_buf].stride) return false;
        for (size_t i = 0; i < op.chan_map.size(); ++i)
            if (op.chan_map[i] != (int)i) return false;
        return true;
    };
    auto can_write = [&](const Op& op, int k) {
        if (op.kind == OP_STREAM3) return op.h16.on;             // (stores P16 units or float32, per tensor)
        if (op.kind == OP_CIN1) return k == 0 && op.out_off[0] == 0 && op.ks <= 3;   // (conv_cin1's octet-per-thread store path holds 2 x taps filter quads)
        if (op.kind != OP_CONV || !op.h16.on || op.fold_s > 0 || op.residual || op.dwk != 0) return false;
        // a pixel shuffler whose sub-pixels take whole 16-channel tiles (conv3_h's P16 epilogue with depth_to_space addressing): ONE destination
        if (op.ps != 1 && (op.shape.nin || op.ps_c % 16 != 0 || k != 0 || op.split < (1 << 29))) return false;
        if (op.out_off[k] % 16 != 0) return false;
        return k == 0 || op.split % 16 == 0;
    };
    for (bool changed = !mixed; changed;) {
        changed = false;
        for (const Op& op : h->ops) {
            const std::vector<int> in = inputs(op);
            bool all = can_read(op);
            for (int b : in) all = all && ok[b];
            if (!all)
                for (int b : in)
                    if (ok[b]) { ok[b] = 0; changed = true; }
            for (int k = 0; k < 2; ++k) {
                const int b = op.out_buf[k];
                if (b < 0 || (k == 1 && op.split >= (1 << 29))) continue;
                if (ok[b] && !can_write(op, k)) { ok[b] = 0; changed = true; }
            }
            for (int b : op.extra_out)
                if (ok[b] && !can_write(op, 0)) { ok[b] = 0; changed = true; }
        }
    }
    for (const Op& op : h->ops) {
        for (int b : inputs(op)) read[b] = 1;
        for (int k = 0; k < 2; ++k)
            if (op.out_buf[k] >= 0 && !(k == 1 && op.split >= (1 << 29))) written[op.out_buf[k]] = 1;
        for (int b : op.extra_out) written[b] = 1;
    }
    h->any_p16 = false;
    h->p16_max_res = 1;
    for (size_t b = 0; b < nbuf; ++b) {
        WsBuf& wb = h->bufs[b];
        wb.p16_ok = !mixed && ok[b] && written[b] && read[b];
        wb.octs = (wb.stride + 7) / 8;
        if (wb.p16_ok) { h->any_p16 = true; h->p16_max_res = std::max(h->p16_max_res, wb.res); }
    }
    for (Op& op : h->ops) {
        const std::vector<int> in = inputs(op);
        bool all = !in.empty() && can_read(op);
        for (int b : in) all = all && h->bufs[b].p16_ok;
        op.h16.in16_ok = all;
    }
    // the float32 plan of a flagged image (exec.hip: run_forward): every launch downstream of a split16 launch or of a P16 tensor
    std::vector<char> dirty(nbuf, 0);
    for (Op& op : h->ops) {
        bool r = (op.kind == OP_CONV || op.kind == OP_STREAM || op.kind == OP_TAIL || op.kind == OP_STREAM3 || op.kind == OP_FOLDX) && op.h16.on;
        for (int b : inputs(op)) r = r || dirty[b];
        for (int k = 0; k < 2; ++k)
            if (op.out_buf[k] >= 0 && !(k == 1 && op.split >= (1 << 29))) r = r || h->bufs[op.out_buf[k]].p16_ok;
        op.h16.rerun = r;
        for (int b : op.extra_out) r = r || h->bufs[b].p16_ok;
        if (r) {
            for (int k = 0; k < 2; ++k)
                if (op.out_buf[k] >= 0) dirty[op.out_buf[k]] = 1;
            for (int b : op.extra_out) dirty[b] = 1;
        }
    }
}

}  // namespace dcscn_impl
