// Weight repack: every launch's filters in the exact LDS image of the kernel that consumes them (dcscn_finalize).
#include "plan.h"
#include "split16_pack.hpp"

#pragma clang fp contract(off)

namespace dcscn_impl {

// ---- weight repack -----------------------------------------------------------------------------

int upload(dcscn_ctx* h, const void* host, size_t bytes, void** dev) {
    HIP_TRY(h, hipMalloc(dev, bytes));
    h->device_allocs.push_back(*dev);
    HIP_TRY(h, hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return DCSCN_OK;
}


// conv3_h's image of a 3x3 layer (one column segment): direct form on the f16 pipe with its own channel groups (up to kC3hMaxNT tiles),
// untransformed filters as (hi, lo) fragments scaled by 2^e, bias and slopes in the padded group layout
static int pack_conv3_h16(dcscn_ctx* h, Op& op, const TensorSpec& tw, int wcols, int tiles16) {
    const ColSeg& sg = op.segs[0];
    const int cin = (int)op.chan_map.size();
    Op::Split16& s16 = op.h16;
    s16.n_tiles = (tiles16 + kC3hMaxNT - 1) / kC3hMaxNT;
    s16.nt = (tiles16 + s16.n_tiles - 1) / s16.n_tiles;
    s16.n_full = tiles16 - s16.n_tiles * (s16.nt - 1);
    s16.n_chunks = (op.cin_phys + kC3hKC - 1) / kC3hKC;
    const int nt16 = s16.nt, ctot16 = s16.n_tiles * nt16 * 16;
    auto padded16 = [&](int cc) {
        const int t = cc / 16;
        const int wide = s16.n_full * nt16;
        const int g = t < wide ? t / nt16 : s16.n_full + (t - wide) / (nt16 - 1);
        const int tg = t < wide ? t % nt16 : (t - wide) % (nt16 - 1);
        return (g * nt16 + tg) * 16 + cc % 16;
    };
    std::vector<float> dense((size_t)9 * op.cin_phys * ctot16, 0.0f), b16(ctot16, 0.0f), a16(ctot16, 0.0f);
    for (int t = 0; t < 9; ++t)
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < sg.cout; ++co)
                dense[((size_t)t * op.cin_phys + op.chan_map[ci]) * ctot16 + padded16(sg.dst + co)] = tw.data[((size_t)t * cin + ci) * wcols + sg.col0 + co];
    for (int co = 0; co < sg.cout; ++co) {
        const int pc = padded16(sg.dst + co);
        if (sg.b >= 0) b16[pc] = h->tensors[sg.b].data[sg.col0 + co];
        a16[pc] = sg.alpha >= 0 ? h->tensors[sg.alpha].data[sg.col0 + co] : op.const_alpha;
    }
    const int e = split16_scale_exp(dense.data(), dense.size());
    s16.inv_scale = std::ldexp(1.0f, -e);
    s16.tail_octs = c3h_tail_octs(op.cin_phys);
    const std::vector<uint16_t> img = pack_conv16(dense, 9, op.cin_phys, ctot16, s16.n_tiles, nt16, s16.n_chunks, e, s16.tail_octs);
    int rc = upload(h, img.data(), img.size() * sizeof(uint16_t), &s16.d_w);
    if (!rc) rc = upload(h, b16.data(), b16.size() * sizeof(float), (void**)&s16.d_bias);
    if (!rc) rc = upload(h, a16.data(), a16.size() * sizeof(float), (void**)&s16.d_alpha);
    s16.on = rc == DCSCN_OK;
    return rc;
}

// ---- fold_whole_tail (graph.hip): the composite 5x5 kernels of the whole tail, one set per border-position class -------------------------
namespace {
struct FoldStage {                  // one pixel-shuffler stage as a dense 3x3 conv: w [9][cin][s * s * c], b [s * s * c]
    int s = 1, cin = 0, c = 0;
    std::vector<double> w, b;
};

// the dense equivalent [9][cin][cout] of the 3x3 conv variable `var` (tf_graph.py build_conv / build_depthwise_separable_conv:
// tf.nn.separable_conv2d with channel_multiplier 1 is the dense conv w[t][ci][co] = depthwise[t][ci] * pointwise[ci][co])
bool foldx_dense(const dcscn_ctx* h, const std::string& var, int cin, int cout, bool bias, std::vector<double>& w, std::vector<double>& b) {
    auto find = [&](const std::string& n) -> const TensorSpec* {
        auto it = h->tensor_index.find(n);
        return it == h->tensor_index.end() ? nullptr : &h->tensors[it->second];
    };
    w.assign((size_t)9 * cin * cout, 0.0);
    b.assign(cout, 0.0);
    if (const TensorSpec* t = find(var + "/conv_W")) {
        if (t->data.size() != w.size()) return false;
        for (size_t i = 0; i < w.size(); ++i) w[i] = t->data[i];
    } else {
        const TensorSpec* dw = find(var + "/depthwise_W");
        const TensorSpec* pw = find(var + "/pointwise_W");
        if (!dw || !pw || dw->data.size() != (size_t)9 * cin || pw->data.size() != (size_t)cin * cout) return false;
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cout; ++co) w[((size_t)t * cin + ci) * cout + co] = (double)dw->data[(size_t)t * cin + ci] * (double)pw->data[(size_t)ci * cout + co];
    }
    if (bias) {
        const TensorSpec* bt = find(var + "/conv_B");
        if (!bt || bt->data.size() != (size_t)cout) return false;
        for (int co = 0; co < cout; ++co) b[co] = bt->data[co];
    }
    return true;
}

// position classes of a pixel along one axis: 0 interior, 1 first row, 2 last row, 3 both (a one-pixel axis) -- as (size of a virtual image, the
// pixel's coordinate in it).  The padding of the intermediate maps reaches one LR pixel deep, the composite's support two: 5 / 3 / 3 / 1 pixels.
const int kFoldHv[4] = {5, 3, 3, 1}, kFoldYv[4] = {2, 0, 2, 0};

// Which (last-conv tap, stage taps ...) chains of HR row S y + a survive the zero padding of the intermediate maps, with the LR offset each
// ends at: two position classes with the same list have the same kernel (the LR map's own padding multiplies taps by zero: not part of it)
std::vector<int> foldx_axis_sig(const std::vector<FoldStage>& up, int S, int a, int variant) {
    const int Hv = kFoldHv[variant], y = kFoldYv[variant];
    std::vector<std::pair<int, int>> cur, nxt;       // (position, path code)
    for (int e = -1; e <= 1; ++e) {
        const int Y = S * y + a + e;
        if (Y >= 0 && Y < S * Hv) cur.push_back({Y, e + 1});
    }
    int res = S;
    for (int k = (int)up.size() - 1; k >= 0; --k) {
        res /= up[k].s;
        nxt.clear();
        for (const auto& p : cur)
            for (int f = -1; f <= 1; ++f) {
                const int q = p.first / up[k].s + f;
                if (k > 0 && (q < 0 || q >= res * Hv)) continue;
                nxt.push_back({q, p.second * 3 + f + 1});
            }
        cur.swap(nxt);
    }
    std::vector<int> sig;
    for (const auto& p : cur) { sig.push_back(p.second); sig.push_back(p.first - y); }
    return sig;
}

// d y(S y + a, S x + b) / d X(y + dy, x + dx, ci) for a pixel of position class (vy, vx), and the constant term: the adjoint of the chain, from the
// output pixel back through the last conv, and per stage depth_to_space and the 3x3 conv, dropping every tap that leaves an intermediate map
void foldx_kernel(const std::vector<FoldStage>& up, const std::vector<double>& r, int S, int a, int b, int vy, int vx, std::vector<double>& kw, double& kb) {
    const int Hv = kFoldHv[vy], y = kFoldYv[vy], Wv = kFoldHv[vx], x = kFoldYv[vx];
    const int n = (int)up.size(), c_last = up[n - 1].c;
    typedef std::map<std::pair<int, int>, std::vector<double>> Adj;
    Adj g;
    for (int ey = -1; ey <= 1; ++ey)
        for (int ex = -1; ex <= 1; ++ex) {
            const int Y = S * y + a + ey, X = S * x + b + ex;
            if (Y < 0 || Y >= S * Hv || X < 0 || X >= S * Wv) continue;          // the last conv's SAME padding of the HR map
            std::vector<double>& v = g[{Y, X}];
            v.assign(c_last, 0.0);
            for (int c = 0; c < c_last; ++c) v[c] = r[(size_t)((ey + 1) * 3 + (ex + 1)) * c_last + c];
        }
    kb = 0.0;
    int res = S;
    for (int k = n - 1; k >= 0; --k) {
        const FoldStage& st = up[k];
        const int s = st.s, co = s * s * st.c;
        res /= s;                                    // resolution of the stage's input map
        Adj gp;
        for (const auto& kv : g) {
            const int Y = kv.first.first, X = kv.first.second;
            const int py = Y / s, px = X / s;
            const int ch0 = ((Y - py * s) * s + (X - px * s)) * st.c;            // depth_to_space: conv channel of (sub-pixel, c)
            const std::vector<double>& gv = kv.second;
            for (int c = 0; c < st.c; ++c) kb += gv[c] * st.b[ch0 + c];
            for (int fy = -1; fy <= 1; ++fy)
                for (int fx = -1; fx <= 1; ++fx) {
                    const int qy = py + fy, qx = px + fx;
                    if (k > 0 && (qy < 0 || qy >= res * Hv || qx < 0 || qx >= res * Wv)) continue;     // SAME padding of an intermediate map
                    std::vector<double>& d = gp[{qy, qx}];
                    if (d.empty()) d.assign(st.cin, 0.0);
                    const double* wt = &st.w[(size_t)((fy + 1) * 3 + (fx + 1)) * st.cin * co];
                    for (int ci = 0; ci < st.cin; ++ci) {
                        const double* wr = wt + (size_t)ci * co + ch0;
                        double sum = 0.0;
                        for (int c = 0; c < st.c; ++c) sum += gv[c] * wr[c];
                        d[ci] += sum;
                    }
                }
        }
        g.swap(gp);
    }
    const int cin0 = up[0].cin;
    kw.assign((size_t)25 * cin0, 0.0);
    for (const auto& kv : g) {
        const int dy = kv.first.first - y, dx = kv.first.second - x;
        if (dy < -2 || dy > 2 || dx < -2 || dx > 2) continue;                    // (cannot happen: the chain's support is 5 x 5)
        for (int ci = 0; ci < cin0; ++ci) kw[(size_t)((dy + 2) * 5 + (dx + 2)) * cin0 + ci] = kv.second[ci];
    }
}
}  // namespace

int pack_foldx(dcscn_ctx* h, Op& op) {
    for (Op& sub : op.fused) {                       // the launches it replaces: the float32 plan of a flagged image, and split16 = 0
        const int rc = finalize_op(h, sub);
        if (rc) return rc;
    }
    const dcscn_config& c = h->cfg;
    const int S = op.fold_s, cin = (int)op.chan_map.size();
    const int ps_out = c.pixel_shuffler_filters != 0 ? c.pixel_shuffler_filters : cin;
    std::vector<FoldStage> up;
    if (S == 4) {
        up.resize(2);
        up[0].s = 2; up[0].cin = cin; up[0].c = cin;
        up[1].s = 2; up[1].cin = cin; up[1].c = ps_out;
    } else {
        up.resize(1);
        up[0].s = S; up[0].cin = cin; up[0].c = ps_out;
    }
    const char* names[2] = {"Up-PS/Up-PS_CNN", "Up-PS2/Up-PS2_CNN"};
    for (size_t k = 0; k < up.size(); ++k)
        if (!foldx_dense(h, names[k], up[k].cin, up[k].s * up[k].s * up[k].c, true, up[k].w, up[k].b))
            return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: fold_whole_tail does not find the variables of %s", names[k]);
    std::vector<double> r, r_b;
    if (!foldx_dense(h, "R-CNN1", ps_out, 1, false, r, r_b)) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: fold_whole_tail does not find the variables of R-CNN1");
    // position classes along an axis that share a kernel for phase a (the first class with the same surviving chains)
    int eff[4][4];
    for (int a = 0; a < S; ++a) {
        std::vector<int> sig[4];
        for (int v = 0; v < 4; ++v) {
            sig[v] = foldx_axis_sig(up, S, a, v);
            eff[a][v] = v;
            for (int u = 0; u < v; ++u)
                if (sig[u] == sig[v]) { eff[a][v] = u; break; }
        }
    }
    std::map<std::vector<int>, std::pair<std::vector<double>, double>> memo;
    const int ncols = 16;
    std::vector<std::vector<float>> dense(kFoldVariants, std::vector<float>((size_t)25 * op.cin_phys * ncols, 0.0f));
    std::vector<float> b16((size_t)kFoldVariants * ncols, 0.0f);
    for (int vy = 0; vy < 4; ++vy)
        for (int vx = 0; vx < 4; ++vx)
            for (int a = 0; a < S; ++a)
                for (int b = 0; b < S; ++b) {
                    const std::vector<int> key = {a, eff[a][vy], b, eff[b][vx]};
                    auto it = memo.find(key);
                    if (it == memo.end()) {
                        std::pair<std::vector<double>, double> kk;
                        foldx_kernel(up, r, S, a, b, eff[a][vy], eff[b][vx], kk.first, kk.second);
                        it = memo.emplace(key, std::move(kk)).first;
                    }
                    const int v = vy * 4 + vx, p = a * S + b;
                    const std::vector<double>& kw = it->second.first;
                    for (int t = 0; t < 25; ++t)
                        for (int ci = 0; ci < cin; ++ci) dense[v][((size_t)t * op.cin_phys + op.chan_map[ci]) * ncols + p] = (float)kw[(size_t)t * cin + ci];
                    b16[(size_t)v * ncols + p] = (float)it->second.second;
                }
    Op::Split16& s16 = op.h16;
    s16.nt = 1; s16.n_tiles = 1; s16.n_full = 1;
    s16.n_chunks = (op.cin_phys + kC3hKC - 1) / kC3hKC;
    int e = 0;
    {
        float m = 0.0f;
        for (const auto& d : dense)
            for (float w : d) m = std::fmax(m, std::fabs(w));
        e = split16_scale_exp(&m, 1);
    }
    s16.inv_scale = std::ldexp(1.0f, -e);
    std::vector<uint16_t> img;
    for (int v = 0; v < kFoldVariants; ++v) {
        const std::vector<uint16_t> one = pack_conv16(dense[v], 25, op.cin_phys, ncols, 1, 1, s16.n_chunks, e);
        img.insert(img.end(), one.begin(), one.end());
    }
    int rc = upload(h, img.data(), img.size() * sizeof(uint16_t), &s16.d_w);
    if (!rc) rc = upload(h, b16.data(), b16.size() * sizeof(float), (void**)&s16.d_bias);
    s16.on = rc == DCSCN_OK;
    return rc;
}

int finalize_op(dcscn_ctx* h, Op& op) {
    if (op.kind == OP_FOLDX) return pack_foldx(h, op);
    if (op.kind == OP_STREAM3) return pack_feat3_stream(h, op);
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
        std::vector<float> dense((size_t)op.cin_phys * op.ctot, 0.0f);          // [physical input channel][padded output channel], for the split16 image
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
                    const float wv = sg.dw1 >= 0 ? dscale * wrow[co] : wrow[co];
                    pack[((size_t)grp * op.n_chunks + chunk) * chunk_floats + (size_t)row * ns + jn] = wv;
                    dense[(size_t)kp * op.ctot + pc] = wv;
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
        if (!rcn) {
            // conv_nin_h: the same channel groups, 32-channel chunks, f16 (hi, lo) fragments of the filters scaled by 2^e
            Op::Split16& s16 = op.h16;
            s16.nt = nt; s16.n_tiles = op.n_tiles; s16.n_full = op.n_full;
            s16.n_chunks = (op.cin_phys + kNinHKC - 1) / kNinHKC;
            const int e = split16_scale_exp(dense.data(), dense.size());
            s16.inv_scale = std::ldexp(1.0f, -e);
            const std::vector<uint16_t> img = pack_conv16(dense, 1, op.cin_phys, op.ctot, op.n_tiles, nt, s16.n_chunks, e);
            rcn = upload(h, img.data(), img.size() * sizeof(uint16_t), &s16.d_w);
            s16.on = rcn == DCSCN_OK;
        }
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
        if (!rcw) rcw = pack_conv3_h16(h, op, tw, wcols, tiles16);
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
    if (!rc && op.fold_s > 0 && op.ks == 5 && op.shape.mt == 4 && op.segs.size() == 1 && op.dwk == 0) {
        // conv5_h: the folded tail on the f16 pipe (one channel group of ceil(4 s^2 / 16) tiles; conv_igemm<5, 4, ...> with the
        // same 16x16 pixel tiles stays behind it as the f32 fallback of flagged tiles)
        Op::Split16& s16 = op.h16;
        const ColSeg& sg = op.segs[0];
        const int cin = (int)op.chan_map.size();
        const int nt16 = (sg.dst + sg.cout + 15) / 16;
        if (nt16 == 1 || nt16 == 3 || nt16 == 4) {
            s16.nt = nt16; s16.n_tiles = 1; s16.n_full = 1;
            s16.n_chunks = (op.cin_phys + kC3hKC - 1) / kC3hKC;
            const int ctot16 = nt16 * 16;
            std::vector<float> dense((size_t)25 * op.cin_phys * ctot16, 0.0f), b16(ctot16, 0.0f);
            for (int t = 0; t < 25; ++t)
                for (int ci = 0; ci < cin; ++ci)
                    for (int co = 0; co < sg.cout; ++co)
                        dense[((size_t)t * op.cin_phys + op.chan_map[ci]) * ctot16 + sg.dst + co] = derived.data[((size_t)t * cin + ci) * sg.cout + co];
            for (int co = 0; co < sg.cout; ++co) b16[sg.dst + co] = derived_bias[co];
            const int e = split16_scale_exp(dense.data(), dense.size());
            s16.inv_scale = std::ldexp(1.0f, -e);
            const std::vector<uint16_t> img = pack_conv16(dense, 25, op.cin_phys, ctot16, 1, nt16, s16.n_chunks, e);
            rc = upload(h, img.data(), img.size() * sizeof(uint16_t), &s16.d_w);
            if (!rc) rc = upload(h, b16.data(), b16.size() * sizeof(float), (void**)&s16.d_bias);
            s16.on = rc == DCSCN_OK;
        }
    }
    // one-tile / scalar-store 3x3 layers: conv3_h in front of the conv_igemm kernel packed above (graph.hip: h16_direct_eligible)
    if (!rc && h16_direct_eligible(h, op)) rc = pack_conv3_h16(h, op, h->tensors[op.segs[0].w], (int)h->tensors[op.segs[0].w].shape.back(), tiles16);
    return rc;
}

// f16 image of one pointwise GEMM of the streamed kernels (feat_stream.hpp: stream_dw_pw, F16), written over its float32 image's slots
// (same size): `quads` input channel quads in 16-channel chunks, `tiles` output tiles; per chunk PAIR and tile two 1 KB fragments --
// lane (i = lane & 15: output column 16 n + i, q = lane >> 4) holds k = 8 q + t: t < 4 channel 16 (2p) + 4 q + t, else 16 (2p + 1) + 4 q + t - 4 --
// hi at slot (2p * tiles + 2n), lo right behind it; an odd last chunk keeps one slot per tile: [hi t 0-3 | lo t 0-3].  get(ci, co) = the
// weight (0 past the real channels); the weights are multiplied by 2^e.
template <typename Get>
static void stream_f16_image(float* dst, int quads, int tiles, const Get& get, int e) {
    const int chunks = (quads + 3) / 4;
    uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
    auto chan = [&](int ch, int q, int t) { return q < std::min(4, quads - 4 * ch) ? 16 * ch + 4 * q + t : -1; };
    for (int n = 0; n < tiles; ++n)
        for (int lane = 0; lane < 64; ++lane) {
            const int q = lane >> 4, co = 16 * n + (lane & 15);
            for (int p = 0; 2 * p + 1 < chunks; ++p) {
                uint16_t* hi = d16 + ((size_t)(2 * p * tiles + 2 * n) * 64 + lane) * 8;
                uint16_t* lo = hi + 64 * 8;
                for (int t = 0; t < 8; ++t) {
                    const int ci = chan(2 * p + (t >> 2), q, t & 3);
                    split16_host(std::ldexp(ci >= 0 ? get(ci, co) : 0.0f, e), &hi[t], &lo[t]);
                }
            }
            if (chunks & 1) {
                const int ch = chunks - 1;
                uint16_t* s = d16 + ((size_t)(ch * tiles + n) * 64 + lane) * 8;
                for (int t = 0; t < 4; ++t) {
                    const int ci = chan(ch, q, t);
                    split16_host(std::ldexp(ci >= 0 ? get(ci, co) : 0.0f, e), &s[t], &s[4 + t]);
                }
            }
        }
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
    size_t a_wp_base = 0, a_bias_base = 0, b_wp_base = 0, b_bias_base = 0;     // regions the f16 image rewrites
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
        a_wp_base = base;
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
        a_bias_base = base;
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
        b_wp_base = base;
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
        b_bias_base = base;
        const ColSeg& sg = u2.segs[0];
        for (int co = 0; co < 4; ++co) blob[base + co] = sg.b >= 0 ? tens(sg.b)[co] : 0.0f;
    }
    a.ldsw_bytes = lds - a.ring_bytes;
    if (lds > 160 * 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: tail_stream needs %d bytes of LDS", lds);
    for (int k = 0; k < 9; ++k) a.c_w[k] = tens(rc.dw_w)[k];         // [3, 3, 1, 1]
    a.c_scale = tens(rc.segs[0].w)[0];                               // [1, 1, 1, 1]
    int urc = upload(h, blob.data(), blob.size() * sizeof(float), (void**)&op.d_w);
    if (urc) return urc;
    // the F16 kernel's image (pack_feat_stream does the same for its GEMMs)
    std::vector<float> blob16 = blob;
    {
        const std::vector<float>& pw = tens(u1.segs[0].w);
        const int tiles = C > 16 ? 2 : 1;
        const int e = split16_scale_exp(pw.data(), pw.size());
        auto get = [&](int ci, int co) {
            const int n = co / 16, ph = n / tiles, cc = 16 * (n % tiles) + co % 16;
            return ci < cin && ph < 4 && cc < C ? pw[(size_t)ci * 4 * C + ph * C + cc] : 0.0f;
        };
        stream_f16_image(&blob16[a_wp_base], cin / 4, 8, get, e);
        for (int k = 0; k < 8 * 16; ++k) blob16[a_bias_base + k] = std::ldexp(blob[a_bias_base + k], e);
        a.a_inv = std::ldexp(1.0f, -e);
    }
    {
        const std::vector<float>& pw = tens(u2.segs[0].w);
        const int e = split16_scale_exp(pw.data(), pw.size());
        auto get = [&](int ci, int co) { return ci < C && co < 4 ? pw[(size_t)ci * 4 + co] : 0.0f; };
        stream_f16_image(&blob16[b_wp_base], C / 4, 1, get, e);
        for (int k = 0; k < 4; ++k) blob16[b_bias_base + k] = std::ldexp(blob[b_bias_base + k], e);
        a.b_inv = std::ldexp(1.0f, -e);
    }
    urc = upload(h, blob16.data(), blob16.size() * sizeof(float), &op.h16.d_w);
    op.h16.on = urc == DCSCN_OK;
    return urc;
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
    std::vector<size_t> nin_base(L), wp_base(L), ba_base(L);        // blob offsets (floats) of the regions the f16 image rewrites
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
        nin_base[i] = base;
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
        wp_base[i] = base;
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
        ba_base[i] = base;
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
    int rc = upload(h, blob.data(), blob.size() * sizeof(float), (void**)&op.d_w);
    if (rc) return rc;
    // --- the F16 kernel's image: the same blob with the pointwise and A1 || B1 filters as f16 (hi, lo) fragments scaled by 2^e per GEMM,
    // the convs' biases times 2^e (they are the first MFMA's C operand); 2^-e goes to the kernel arguments ---
    std::vector<float> blob16 = blob;
    for (int i = 0; i < L; ++i) {
        const bool is_b2 = i == L - 1;
        const Op& src = is_b2 ? op.fused[L + 2] : op.fused[2 + i];
        const ColSeg& sg = src.segs[0];
        const int cin = is_b2 ? nb : h->sched[i], cout = sg.cout;
        const std::vector<float>& pw = tens(sg.w);
        const int e = split16_scale_exp(pw.data(), (size_t)cin * cout);
        auto get = [&](int ci, int co) { return ci < cin && co < cout ? pw[(size_t)ci * cout + co] : 0.0f; };
        stream_f16_image(&blob16[wp_base[i]], pad4(cin) / 4, (cout + 15) / 16, get, e);
        for (int k = 0; k < 32; ++k) blob16[ba_base[i] + k] = std::ldexp(blob[ba_base[i] + k], e);
        a.conv[i].inv = std::ldexp(1.0f, -e);
    }
    {
        auto ninw = [&](int layer_base, int ci, int v) -> float {
            const ColSeg* sg = nullptr;
            int co = 0;
            if (v < pad4(nb)) { if (v < nb) { sg = &sb; co = v; } }
            else if (v - pad4(nb) < na) { sg = &sa; co = v - pad4(nb); }
            if (!sg) return 0.0f;
            const int cols = (int)h->tensors[sg->w].shape.back();
            float w = tens(sg->w)[(size_t)(layer_base + ci) * cols + sg->col0 + co];
            if (sg->dw1 >= 0) w = tens(sg->dw1)[layer_base + ci] * w;
            return w;
        };
        std::vector<float> all;
        int cb = 0;
        for (int i = 0; i < L; ++i) {
            for (int ci = 0; ci < h->sched[i]; ++ci)
                for (int v = 0; v < 32; ++v) all.push_back(ninw(cb, ci, v));
            cb += h->sched[i];
        }
        const int e = split16_scale_exp(all.data(), all.size());
        cb = 0;
        for (int i = 0; i < L; ++i) {
            const int C = h->sched[i];
            auto get = [&](int ci, int v) { return ci < C ? ninw(cb, ci, v) : 0.0f; };
            stream_f16_image(&blob16[nin_base[i]], a.nin[i].ring.quads, 2, get, e);
            cb += C;
        }
        a.nin_inv = std::ldexp(1.0f, -e);
    }
    rc = upload(h, blob16.data(), blob16.size() * sizeof(float), &op.h16.d_w);
    op.h16.on = rc == DCSCN_OK;
    return rc;
}

// ---- feat3_stream (feat3_stream.hpp): rings, role table and filter fragments of the fused CNN1 .. CNNL launch ---------------------
int pack_feat3_stream(dcscn_ctx* h, Op& op) {
    const bool nin_on = op.stream3.nin.on != 0;                 // fuse_feat3_stream: B1+A1 and B2 are part of the launch (op.fused[L], op.fused[L + 1])
    const int L = (int)op.fused.size() - (nin_on ? 2 : 0);
    for (Op& sub : op.fused) {                      // the layers' own launches: the float32 plan of a flagged image, and split16 = 0
        const int rc = finalize_op(h, sub);
        if (rc) return rc;
    }
    Stream3Args& a = op.stream3;
    a = Stream3Args{};
    a.L = L;
    a.nin.on = nin_on ? 1 : 0;
    a.total_lag = nin_on ? 2 * L + 1 : 2 * (L - 1);             // (B2 computes stream row t - (2 L + 1): two behind the row A1 || B1 finishes at step g + 2 L - 1)
    auto tens = [&](int id) -> const std::vector<float>& { return h->tensors[id].data; };
    int lds = 0;
    std::vector<S3Ring> ring(L);
    for (int i = 0; i < L; ++i) {
        const int octs = (h->sched[i] + 7) / 8;
        const bool has = i + 1 < L || nin_on;                    // the last layer's rows are read by the A1 || B1 waves only
        ring[i] = S3Ring{lds, has ? (2 * octs + 1) * 16 : 0, octs};
        if (has) lds += 4 * kStreamRowPx * ring[i].px;           // four slots (feat3_stream.hpp)
    }
    if (nin_on) {
        a.nin.b1 = S3Ring{lds, 3 * 16, 1};
        lds += 4 * kStreamRowPx * a.nin.b1.px;
    }
    a.ring_bytes = lds;
    if (lds > 160 * 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat3_stream needs %d bytes of LDS", lds);
    a.first_out = ring[0];
    std::vector<float> blob;
    {   // CNN1: filter [9][32] (tap major), bias [32], slope - 1 [32]
        const Op& c1 = op.fused[0];
        const ColSeg& sg = c1.segs[0];
        const int C = h->sched[0];
        a.first_w = 0;
        blob.resize(288 + 64, 0.0f);
        const std::vector<float>& w = tens(sg.w);              // [3, 3, 1, C]
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < C; ++co) blob[(size_t)t * 32 + co] = w[(size_t)t * C + sg.col0 + co];
        for (int co = 0; co < C; ++co) {
            blob[288 + co] = sg.b >= 0 ? tens(sg.b)[sg.col0 + co] : 0.0f;
            blob[320 + co] = (sg.alpha >= 0 ? tens(sg.alpha)[sg.col0 + co] : c1.const_alpha) - 1.0f;
        }
    }
    // a 3x3 conv of the stream: conv[ci] computes `cout` channels from the `cin` channels of ring `in`; fragments [step][tile][hi | lo][64 lanes][8 halfs]
    auto pack_conv = [&](int ci, const Op& o, int cin, int cout, const S3Ring& in, const S3Ring& out, int lag) {
        const ColSeg& sg = o.segs[0];
        const int octs = in.octs, steps = (9 * octs + 3) / 4, tiles = (cout + 15) / 16;
        S3Conv& cv = a.conv[ci];
        cv.in = in;
        cv.out = out;
        cv.lag = lag;
        cv.tiles = tiles;
        const std::vector<float>& w = tens(sg.w);              // [3, 3, cin, cout_total]
        const int wcols = (int)h->tensors[sg.w].shape.back();
        std::vector<float> all((size_t)9 * cin * cout);
        for (int t = 0; t < 9; ++t)
            for (int c0 = 0; c0 < cin; ++c0)
                for (int co = 0; co < cout; ++co) all[((size_t)t * cin + c0) * cout + co] = w[((size_t)t * cin + c0) * wcols + sg.col0 + co];
        const int e = split16_scale_exp(all.data(), all.size());
        cv.inv = std::ldexp(1.0f, -e);
        cv.w_off = (int)blob.size();
        blob.resize(blob.size() + (size_t)steps * tiles * 2 * 64 * 4, 0.0f);
        uint16_t* d16 = reinterpret_cast<uint16_t*>(&blob[cv.w_off]);
        for (int s = 0; s < steps; ++s)
            for (int n = 0; n < tiles; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, co = 16 * n + (lane & 15);
                    const int p = 4 * s + q, tap = p / octs, oct = p - tap * octs;
                    uint16_t* hi = d16 + ((size_t)((s * tiles + n) * 2 + 0) * 64 + lane) * 8;
                    uint16_t* lo = d16 + ((size_t)((s * tiles + n) * 2 + 1) * 64 + lane) * 8;
                    for (int t8 = 0; t8 < 8; ++t8) {
                        const int c0 = 8 * oct + t8;
                        const float wv = tap < 9 && c0 < cin && co < cout ? all[((size_t)tap * cin + c0) * cout + co] : 0.0f;
                        split16_host(std::ldexp(wv, e), &hi[t8], &lo[t8]);
                    }
                }
        cv.ba_off = (int)blob.size();
        blob.resize(blob.size() + 64, 0.0f);
        for (int co = 0; co < cout; ++co) {
            blob[cv.ba_off + co] = std::ldexp(sg.b >= 0 ? tens(sg.b)[sg.col0 + co] : 0.0f, e);
            blob[cv.ba_off + 32 + co] = (sg.alpha >= 0 ? tens(sg.alpha)[sg.col0 + co] : o.const_alpha) - 1.0f;
        }
        return 9 * steps * tiles;                               // MFMAs per row
    };
    int waves = 0;
    std::vector<int> cost;                          // MFMAs per row of each wave
    a.role_conv[waves] = -1; a.role_tile[waves] = 0; cost.push_back(0); ++waves;
    int pair_cost[2] = {0, 0};
    for (int i = 1; i < L; ++i) {
        const int mf = pack_conv(i - 1, op.fused[i], h->sched[i - 1], h->sched[i], ring[i - 1], ring[i], 2 * i);
        if (nin_on && i >= L - 2) { pair_cost[0] += mf; continue; }             // conv[L - 3], conv[L - 2] share a wave
        if (nin_on && i == L - 3) { pair_cost[1] += mf; continue; }             // conv[L - 4] shares one with B2
        if (waves >= kS3MaxWaves) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat3_stream needs more than %d waves", kS3MaxWaves);
        a.role_conv[waves] = (int8_t)(i - 1); a.role_tile[waves] = 0; cost.push_back(mf); ++waves;
    }
    if (nin_on) {
        const dcscn_config& c = h->cfg;
        const int nb = c.nin_filters2, na = c.nin_filters;      // 8, 24 (fuse_feat3_stream)
        const Op& nin = op.fused[L];
        const Op& b2 = op.fused[L + 1];
        pair_cost[1] += pack_conv(L - 1, b2, nb, nb, a.nin.b1, S3Ring{0, 0, 1}, 2 * L + 1);
        if (waves + 4 > kS3MaxWaves) return fail(h, DCSCN_ERR_UNSUPPORTED, "internal: feat3_stream with A1 || B1 needs more than %d waves", kS3MaxWaves);
        for (int k = 0; k < 2; ++k) { a.role_conv[waves] = (int8_t)(kS3RolePair + k); cost.push_back(pair_cost[k] + 60); ++waves; }   // (+ a second conv's fixed work)
        // A1 || B1: conv channel c < nb = B1 channel c, else A1 channel c - nb; the K axis layer by layer, one K = 32 fragment per (layer, tile)
        const ColSeg& sb = nin.segs[0];
        const ColSeg& sa = nin.segs[1];
        const std::vector<float>& wb = tens(sb.w);             // [1, 1, K, nb_total]
        const std::vector<float>& wa = tens(sa.w);
        const int bcols = (int)h->tensors[sb.w].shape.back(), acols = (int)h->tensors[sa.w].shape.back();
        int ktot = 0;
        for (int i = 0; i < L; ++i) ktot += h->sched[i];
        std::vector<float> all((size_t)ktot * 32, 0.0f);
        for (int k = 0; k < ktot; ++k) {
            for (int co = 0; co < nb; ++co) all[(size_t)k * 32 + co] = wb[(size_t)k * bcols + sb.col0 + co];
            for (int co = 0; co < na; ++co) all[(size_t)k * 32 + nb + co] = wa[(size_t)k * acols + sa.col0 + co];
        }
        const int e = split16_scale_exp(all.data(), all.size());
        a.nin.inv = std::ldexp(1.0f, -e);
        a.nin.w_off = (int)blob.size();
        blob.resize(blob.size() + (size_t)L * 2 * 2 * 64 * 4, 0.0f);
        uint16_t* d16 = reinterpret_cast<uint16_t*>(&blob[a.nin.w_off]);
        int k0 = 0;
        for (int i = 0; i < L; ++i) {
            for (int n = 0; n < 2; ++n)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, co = 16 * n + (lane & 15);
                    uint16_t* hi = d16 + ((size_t)((i * 2 + n) * 2 + 0) * 64 + lane) * 8;
                    uint16_t* lo = d16 + ((size_t)((i * 2 + n) * 2 + 1) * 64 + lane) * 8;
                    for (int t8 = 0; t8 < 8; ++t8) {
                        const int c0 = 8 * q + t8;                 // (a lane group past the layer's last octet re-reads that octet: zero rows here)
                        const float wv = c0 < h->sched[i] && 8 * q < 8 * ring[i].octs ? all[(size_t)(k0 + c0) * 32 + co] : 0.0f;
                        split16_host(std::ldexp(wv, e), &hi[t8], &lo[t8]);
                    }
                }
            k0 += h->sched[i];
        }
        a.nin.ba_off = (int)blob.size();
        blob.resize(blob.size() + 64, 0.0f);
        for (int co = 0; co < nb; ++co) {
            blob[a.nin.ba_off + co] = std::ldexp(sb.b >= 0 ? tens(sb.b)[sb.col0 + co] : 0.0f, e);
            blob[a.nin.ba_off + 32 + co] = (sb.alpha >= 0 ? tens(sb.alpha)[sb.col0 + co] : nin.const_alpha) - 1.0f;
        }
        for (int co = 0; co < na; ++co) {
            blob[a.nin.ba_off + nb + co] = std::ldexp(sa.b >= 0 ? tens(sa.b)[sa.col0 + co] : 0.0f, e);
            blob[a.nin.ba_off + 32 + nb + co] = (sa.alpha >= 0 ? tens(sa.alpha)[sa.col0 + co] : nin.const_alpha) - 1.0f;
        }
        for (int n = 0; n < 2; ++n) { a.role_conv[waves] = (int8_t)(kS3RoleNin + n); cost.push_back(9 * L); ++waves; }
    }
    a.n_waves = waves;
    {   // wave w runs on SIMD w & 3: deal the roles, heaviest first, onto the least loaded SIMD that has a wave slot left
        std::vector<int> order(waves);
        for (int i = 0; i < waves; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
        int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
        int8_t rc[kS3MaxWaves], rt[kS3MaxWaves];
        for (int w = 0; w < kS3MaxWaves; ++w) { rc[w] = -1; rt[w] = 0; }
        for (int idx : order) {
            int pick = -1;
            for (int sd = 0; sd < 4; ++sd) {
                const int cap = (waves - sd + 3) / 4;
                if (used[sd] < cap && (pick < 0 || load[sd] < load[pick])) pick = sd;
            }
            const int w = pick + 4 * used[pick];
            rc[w] = a.role_conv[idx]; rt[w] = a.role_tile[idx];
            used[pick] += 1;
            load[pick] += cost[idx] + 20;
        }
        for (int w = 0; w < kS3MaxWaves; ++w) { a.role_conv[w] = rc[w]; a.role_tile[w] = rt[w]; }
    }
    const int rc = upload(h, blob.data(), blob.size() * sizeof(float), (void**)&op.d_w);
    op.h16.on = rc == DCSCN_OK;
    return rc;
}

}  // namespace dcscn_impl
