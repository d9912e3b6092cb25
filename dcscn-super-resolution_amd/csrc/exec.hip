// Execution: workspace carving, the launches of one pass, sub-batching, spatial tiling of over-sized images, bicubic resize.
#include "plan.h"

#pragma clang fp contract(off)

namespace dcscn_impl {

// ---- workspace ---------------------------------------------------------------------------------

// (Re)carves the arena for passes of nb images of H x W.  `stream` is the stream the coming forward runs on: the clear
// of the new carve is enqueued there, behind an event wait on the previous forward (which may have run on another
// stream and may still be in flight) -- nothing is cleared or re-carved underneath live kernels.
int ensure_workspace(dcscn_ctx* h, int nb, int H, int W, hipStream_t stream) {
    const bool want16 = p16_active(h);
    if (h->arena && nb <= h->lay_n && H == h->lay_h && W == h->lay_w && want16 == h->p16_now) return DCSCN_OK;
    std::vector<size_t> offsets(h->bufs.size()), sizes(h->bufs.size());
    size_t total = 0;
    for (size_t i = 0; i < h->bufs.size(); ++i) {
        const WsBuf& b = h->bufs[i];
        offsets[i] = total;
        const long long npix = (long long)nb * H * b.res * W * b.res;
        size_t bytes = (size_t)npix * b.stride * sizeof(float);
        // a P16 tensor (p16.hpp) is never smaller than its float32 form: the float32 plan of a flagged image reuses the bytes
        if (want16 && b.p16_ok && b.stride > 0) bytes = std::max(bytes, (size_t)p16_tensor_bytes(npix, b.octs));
        sizes[i] = bytes;
        total += (bytes + 255) & ~(size_t)255;
    }
    // redo flags of a pass (split16.hpp): [0] = some image was flagged, [1 + image]; cleared per pass
    const size_t redo_ints = 1 + (size_t)nb;
    h->redo_off = total;
    h->redo_ints = redo_ints;
    total += (redo_ints * sizeof(int32_t) + 255) & ~(size_t)255;
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
    for (size_t i = 0; i < h->bufs.size(); ++i) {
        WsBuf& b = h->bufs[i];
        b.offset = offsets[i];
        b.bytes = sizes[i];
        b.p16 = want16 && b.p16_ok && b.stride > 0;
        b.plane = p16_plane_bytes((long long)nb * H * b.res * W * b.res);
    }
    h->p16_now = want16;
    // padding channels that no kernel writes (depth_to_space outputs with C % 4 != 0) must hold
    // finite values: clear the bytes of the new carve
    HIP_TRY(h, hipMemsetAsync(h->arena, 0, total, stream));
    h->carve_gen += 1;
    h->lay_n = nb;
    h->lay_h = H;
    h->lay_w = W;
    return DCSCN_OK;
}



static P16Desc p16_desc(const dcscn_ctx* h, int id) {
    const WsBuf& b = h->bufs[id];
    return P16Desc{static_cast<char*>(h->arena) + b.offset, b.plane, b.octs, 0};
}

bool op_on_split16(const dcscn_ctx* h, const Op& op) {
    if (op.kind == OP_STREAM || op.kind == OP_TAIL || op.kind == OP_STREAM3 || op.kind == OP_FOLDX) return h->split16 && op.h16.on && (h->split16_mask & 1);      // the F16 instantiation of the streamed kernels
    return op.kind == OP_CONV && h->split16 && op.h16.on && (h->split16_mask & (op.shape.nin ? 2 : 1));
}

// conv3_h8 takes a 3x3 launch with two channel groups whose tensors are all P16 or all float32 (c3e_eligible without the pointers)
bool op_takes_h8(const dcscn_ctx* h, const Op& op) {
    if (!op_on_split16(h, op) || !h->conv3_h8 || op.shape.nin || op.fold_s > 0) return false;
    ConvArgs a{};
    a.n_full = op.h16.n_full; a.ps = op.ps; a.res = op.residual ? reinterpret_cast<const float*>(1) : nullptr; a.act = op.act; a.n_chunks = op.h16.n_chunks;
    if (!c3e_eligible(op.h16.nt, a, op.h16.n_tiles)) return false;
    const bool on16 = p16_active(h);                              // (what the next carve will hold; == h->p16_now inside a forward)
    const bool in16 = on16 && op.h16.in16_ok;
    bool all16 = true, any16 = false;
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && op.split >= (1 << 29)) continue;
        const bool o = on16 && op.out_buf[k] >= 0 && h->bufs[op.out_buf[k]].p16_ok && h->bufs[op.out_buf[k]].stride > 0;
        all16 = all16 && o; any16 = any16 || o;
    }
    return in16 ? all16 : !any16;
}

int launch_op(dcscn_ctx* h, const Op& op, int nb, int H, int W, const float* x, const float* x2, float* y,
              hipStream_t stream, bool redo) {
    const int Hr = H * op.res, Wr = W * op.res;
    int32_t* const redo_flags = reinterpret_cast<int32_t*>(static_cast<char*>(h->arena) + h->redo_off);
    const bool stream16 = !redo && op_on_split16(h, op);          // streamed kernels: the F16 instantiation on its own filter image
    if (op.kind == OP_STREAM3) {
        if (!stream16) {
            // split16 = 0, or the float32 plan of a flagged image: the layers one by one on their own float32 kernels
            for (const Op& sub : op.fused) {
                const int rc = launch_op(h, sub, nb, H, W, x, x2, y, stream, redo);
                if (rc) return rc;
            }
            return DCSCN_OK;
        }
        Stream3Args a = op.stream3;
        a.x = x;
        a.blob = op.d_w;
        a.N = nb; a.H = H; a.W = W;
        a.halo = a.L + (a.nin.on ? 1 : 0);                // receptive-field radius of the fused chain (B2 behind A1 || B1 adds one)
        if (W <= kStreamPX) { a.n_strips = 1; a.useful_w = W; }
        else { a.useful_w = kStreamPX - 2 * a.halo; a.n_strips = (W + a.useful_w - 1) / a.useful_w; }
        const int64_t cols = (int64_t)nb * a.n_strips;
        const int want = (int)std::max<int64_t>(1, (512 + cols - 1) / cols);
        a.useful_h = std::max(32, (H + want - 1) / want);
        a.n_blocks = (H + a.useful_h - 1) / a.useful_h;
        a.rows_c = a.n_blocks == 1 ? H : a.useful_h + 2 * a.halo;
        a.n_jobs = (int)(cols * a.n_blocks);
        a.jobs_per_wg = (a.n_jobs + h->n_cus - 1) / h->n_cus;
        const int grid = (a.n_jobs + a.jobs_per_wg - 1) / a.jobs_per_wg;
        auto out_desc = [&](int id, int lo, int hi) {
            S3Out o{};
            o.ptr = buf_ptr(h, id);
            o.stride = h->bufs[id].stride;
            o.width = h->bufs[id].stride;
            o.p16 = h->bufs[id].p16 ? p16_desc(h, id) : P16Desc{nullptr, 0, 0, 0};
            o.lo = lo; o.hi = hi < 0 ? o.width : hi;
            return o;
        };
        if (a.nin.on) {
            // no layer's rows leave the CU; Concat2 = [B2 (conv[L - 1]: channels 0 .. 7) | A1 (the A1 || B1 waves: channels 8 .. 31)]
            for (int i = 0; i < kS3MaxL; ++i) a.out[i] = S3Out{};
            a.out[a.L] = out_desc(op.extra_out[0], 0, 8);
            a.out2 = out_desc(op.extra_out[0], 8, 32);
        } else
        for (size_t i = 0; i < op.extra_out.size(); ++i) a.out[i] = out_desc(op.extra_out[i], 0, -1);
        a.redo = redo_flags;
        HIP_TRY(h, stream3_launch(a, grid, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_FOLDX) {
        if (!stream16) {
            // split16 = 0, or the float32 plan of a flagged image: the launches the fold replaces (graph.hip: fold_whole_tail)
            for (const Op& sub : op.fused) {
                const int rc = launch_op(h, sub, nb, H, W, x, x2, y, stream, redo);
                if (rc) return rc;
            }
            return DCSCN_OK;
        }
        ConvArgs b{};
        b.in = buf_ptr(h, op.in_buf);
        b.in_stride = h->bufs[op.in_buf].stride;
        b.in_off = op.in_off;
        b.cin_phys = op.cin_phys;
        if (h->p16_now && op.h16.in16_ok) {
            b.in16 = p16_desc(h, op.in_buf);
            b.in = nullptr;
        }
        b.n_chunks = op.h16.n_chunks;
        b.wpack16 = op.h16.d_w;                                  // [16 variants][chunk][25 taps][hi | lo]: variant 0 = the interior kernels
        b.bias = op.h16.d_bias;                                  // [16 variants][16 phases]
        b.inv_scale = op.h16.inv_scale;
        b.act = ACT_NONE;
        b.N = nb; b.H = H; b.W = W;
        b.tiles_x = (W + 15) / 16;
        b.tiles_y = (H + 15) / 16;
        b.out0.ptr = y; b.out0.stride = 1; b.out0.width = 1;
        b.out1 = b.out0;
        b.split = 1 << 30;
        b.ps = op.fold_s; b.ps_c = 1;
        b.res = x2; b.res_stride = 1;
        b.fold = 2;
        b.redo = redo_flags;
        HIP_TRY(h, c5h_launch(1, b, stream));
        HIP_TRY(h, c5h_border_launch(b, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_TAIL) {
        TailArgs a = op.tail;
        a.c2 = buf_ptr(h, op.in_buf);
        a.c2_stride = h->bufs[op.in_buf].stride;
        a.x2 = x2;
        a.y = y;
        a.blob = stream16 ? static_cast<const float*>(op.h16.d_w) : op.d_w;
        a.redo = redo_flags; a.redo_check = redo ? 1 : 0;
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
        HIP_TRY(h, tail_launch(a, grid, stream16, stream));
        return DCSCN_OK;
    }
    if (op.kind == OP_STREAM) {
        StreamArgs a = op.stream;
        a.x = x;
        a.out = buf_ptr(h, op.out_buf[0]);
        a.out_stride = h->bufs[op.out_buf[0]].stride;
        a.blob = stream16 ? static_cast<const float*>(op.h16.d_w) : op.d_w;
        a.redo = redo_flags; a.redo_check = redo ? 1 : 0;
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
#ifdef STREAM_DBG                              // timing-probe builds only (tools/stream_dbg.sh): shader clocks of workgroup 0 to a file
        static long long* dbg = nullptr;
        if (getenv("DCSCN_STREAM_DBG")) {
            if (!dbg) HIP_TRY(h, hipMalloc((void**)&dbg, 16 * 64 * 4 * sizeof(long long)));
            HIP_TRY(h, hipMemsetAsync(dbg, 0, 16 * 64 * 4 * sizeof(long long), stream));
            a.dbg = dbg;
        }
#endif
        HIP_TRY(h, stream_launch(a, grid, stream16, stream));
#ifdef STREAM_DBG
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
#endif
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
        a.redo = redo_flags; a.redo_check = redo ? 1 : 0;
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
        a.redo = redo_flags; a.redo_check = redo ? 1 : 0;
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
        if (!redo && h->bufs[op.out_buf[0]].p16) a.out.p16 = p16_desc(h, op.out_buf[0]);
        a.redo = redo_flags; a.redo_check = redo ? 1 : 0;
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
    a.redo = redo_flags;
    if (!redo && op_on_split16(h, op)) {
        // the contraction on the f16 matrix pipe; an image with a value beyond the f16 range (a non-finite accumulator, a P16 output
        // that does not fit) raises its redo flag and is recomputed by the float32 plan behind the pass (run_forward)
        ConvArgs b = a;
        b.wpack16 = op.h16.d_w;
        b.inv_scale = op.h16.inv_scale;
        b.n_chunks = op.h16.n_chunks;
        b.n_full = op.h16.n_full;
        if (h->p16_now && op.h16.in16_ok) {                   // P16 sources (p16.hpp)
            b.in16 = p16_desc(h, op.multi.empty() ? op.in_buf : op.multi[0].first);
            b.in = nullptr;
            if (op.shape.nin) b.srctab = op.h16.d_tab16;
        }
        for (int i = 0; i < 2; ++i) {
            OutDesc& o = i == 0 ? b.out0 : b.out1;
            if (op.out_buf[i] >= 0 && h->bufs[op.out_buf[i]].p16) o.p16 = p16_desc(h, op.out_buf[i]);
        }
        if (op.shape.nin) HIP_TRY(h, nin_h_launch(op.h16.nt, b, op.h16.n_tiles, stream));
        else if (op.fold_s > 0) {
            b.bias = op.h16.d_bias;
            HIP_TRY(h, c5h_launch(op.h16.nt, b, stream));
        } else {
            b.bias = op.h16.d_bias;
            b.tail_octs = op.h16.tail_octs;
            b.alpha = op.h16.d_alpha;
            b.tiles_y = (Hr + 15) / 16;                           // (16 x 16 pixel tiles whatever the float32 kernel behind the layer uses)
            if (op_takes_h8(h, op)) HIP_TRY(h, c3e_launch(op.h16.nt, b, op.h16.n_tiles, h->n_cus, stream));
            else HIP_TRY(h, c3h_launch(op.h16.nt, b, op.h16.n_tiles, stream));
        }
        return DCSCN_OK;
    }
    a.redo_check = redo ? 1 : 0;
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
        // the old buffers are freed: the previous forward (possibly on another stream) may still read them
        if (h->has_last) HIP_TRY(h, hipStreamSynchronize(h->last_stream));
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
    // a P16 tensor (p16.hpp) is carved at max(float32 bytes, P16 bytes): its last chunk's record is padded to whole 32-byte octet pairs, so
    // a tensor whose stride is 4 mod 8 is larger than its float32 form (the budget is the hard knob, count what is carved; the
    // per-plane zero records and 256-byte roundings are the slack of 1 / 64 below)
    const bool want16 = p16_active(h);
    for (const WsBuf& b : h->bufs) {
        int64_t per_pixel = (int64_t)b.stride * (int64_t)sizeof(float);
        if (want16 && b.p16_ok && b.stride > 0) per_pixel = std::max<int64_t>(per_pixel, 128 * (b.octs / 4) + 32 * (b.octs % 4));
        ws_per_lr_pixel += (int64_t)b.res * b.res * per_pixel;
    }
    ws_per_lr_pixel += (ws_per_lr_pixel + 63) / 64;
    int64_t pass_pixels = std::min<int64_t>(h->sub_batch_pixels, h->workspace_budget / std::max<int64_t>(ws_per_lr_pixel, 1));
    // sub_batch_pixels is a soft knob (a pass holds at least one image); the workspace budget is the hard one
    int64_t budget_pixels = h->workspace_budget / std::max<int64_t>(ws_per_lr_pixel, 1);
    if (want16) {
        // record offsets inside a P16 plane are 32-bit (kernels.h: kP16MaxPixels): bounds the pixels of a pass at the tensors' resolution
        const int64_t cap = kP16MaxPixels / ((int64_t)h->p16_max_res * h->p16_max_res);
        pass_pixels = std::min(pass_pixels, cap);
        budget_pixels = std::min(budget_pixels, cap);
    }
    // two forwards of one handle share the arena and the tile staging buffers: a forward on another stream than the previous
    // one waits for it -- before anything of this forward is enqueued, the gathers of run_tiled included
    if (h->has_last && h->last_stream != stream) HIP_TRY(h, hipStreamWaitEvent(stream, h->done_ev, 0));
    if (per_image > budget_pixels && h->spatial_tiling) return run_tiled(h, x, x2, y, n, H, W, budget_pixels, stream);
    int nb = (int)std::max<int64_t>(1, std::min<int64_t>(n, pass_pixels / per_image));
    int rc = ensure_workspace(h, nb, H, W, stream);
    while (rc == DCSCN_ERR_NOMEM && nb > 1) {            // less free memory than the budget assumed: smaller passes
        nb = (nb + 1) / 2;
        rc = ensure_workspace(h, nb, H, W, stream);
    }
    if (rc) return rc;
    if (h->tables_gen != h->carve_gen) {
        // the multi-source tables hold arena addresses: refill them behind the re-carve, on the launch stream
        std::vector<Op*> with_tables;                   // the launches of the plan and the ones inside a streamed launch (its float32 plan)
        for (Op& op : h->ops) {
            with_tables.push_back(&op);
            for (Op& sub : op.fused) with_tables.push_back(&sub);
        }
        for (Op* opp : with_tables) {
            Op& op = *opp;
            if (op.multi.empty() || !op.d_srctab) continue;
            // float32 form: one entry per channel quad of the pad-8 virtual K axis (densify_features); padding quads point at readable
            // memory with stride 0 (conv_nin_h fetches every quad; their filter rows are zero)
            const unsigned long long pad_ptr = (unsigned long long)(uintptr_t)buf_ptr(h, op.multi[0].first);
            size_t q = 0;
            for (const auto& sg : op.multi) {
                const char* base = reinterpret_cast<const char*>(buf_ptr(h, sg.first));
                const unsigned stride = (unsigned)(h->bufs[sg.first].stride * sizeof(float));
                for (int c4 = 0; c4 < ((sg.second + 7) & ~7) / 4 && q < op.h_srctab.size(); ++c4, ++q)
                    op.h_srctab[q] = c4 < sg.second / 4 ? NinSrcQuad{(unsigned long long)(uintptr_t)(base + 16 * c4), stride, 1u} : NinSrcQuad{pad_ptr, 0, 0};
            }
            for (; q < op.h_srctab.size(); ++q) op.h_srctab[q] = NinSrcQuad{pad_ptr, 0, 0};
            HIP_TRY(h, hipMemcpyAsync(op.d_srctab, op.h_srctab.data(), op.h_srctab.size() * sizeof(NinSrcQuad), hipMemcpyHostToDevice, stream));
        }
        // P16 form of the 1x1 GEMMs' K axis (conv_nin_h.hpp, SRC = 2): one entry per channel OCTET = {the octet's hi unit in the record of
        // pixel 0, record bytes}; entries past the last octet read a zero record with stride 0
        for (Op& op : h->ops) {
            if (!op.h16.d_tab16 || !h->p16_now || !op.h16.in16_ok) continue;
            std::vector<std::pair<int, int>> srcs = op.multi;
            if (srcs.empty()) srcs.push_back({op.in_buf, op.cin_phys});
            const unsigned long long zero_rec = (unsigned long long)(uintptr_t)buf_ptr(h, srcs[0].first);
            size_t q = 0;
            for (const auto& sg : srcs) {
                const WsBuf& wb = h->bufs[sg.first];
                const char* base = reinterpret_cast<const char*>(buf_ptr(h, sg.first));
                for (int o = 0; o < wb.octs && q < op.h16.h_tab16.size(); ++o, ++q)
                    op.h16.h_tab16[q] = NinSrcQuad{(unsigned long long)(uintptr_t)(base + (long long)(o >> 2) * wb.plane + 128 + (o & 3) * 32),
                                                   (unsigned)p16_rec_bytes(wb.octs, o >> 2), 1u};
            }
            for (; q < op.h16.h_tab16.size(); ++q) op.h16.h_tab16[q] = NinSrcQuad{zero_rec, 0, 0};
            HIP_TRY(h, hipMemcpyAsync(op.h16.d_tab16, op.h16.h_tab16.data(), op.h16.h_tab16.size() * sizeof(NinSrcQuad), hipMemcpyHostToDevice, stream));
        }
        // the zero records of the P16 planes: re-cleared at the start of every pass (the float32 plan of a flagged image writes float32
        // tensors over the same bytes)
        h->h_zrec.clear();
        for (const WsBuf& wb : h->bufs)
            if (wb.p16)
                for (int c = 0; c < (wb.octs + 3) / 4; ++c) h->h_zrec.push_back((unsigned long long)(uintptr_t)(static_cast<char*>(h->arena) + wb.offset + (long long)c * wb.plane));
        if (!h->h_zrec.empty()) {
            if (h->h_zrec.size() > h->zrec_cap) {
                HIP_TRY(h, hipStreamSynchronize(stream));
                if (h->d_zrec) HIP_TRY(h, hipFree(h->d_zrec));
                h->d_zrec = nullptr;
                h->zrec_cap = 0;
                HIP_TRY(h, hipMalloc((void**)&h->d_zrec, h->h_zrec.size() * 2 * sizeof(unsigned long long)));
                h->zrec_cap = h->h_zrec.size() * 2;
            }
            HIP_TRY(h, hipMemcpyAsync(h->d_zrec, h->h_zrec.data(), h->h_zrec.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
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
        const size_t need = ev_base + (size_t)batches * (nops + 1) * 2;      // one pair per launch, one per pass around its float32 plan
        while (h->ev.size() < need) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
        h->ev_forwards += 1;
    }
    // (pairs are handed out in launch order; ev_op remembers which launch a pair belongs to: nops = the float32 plan of a pass)
    auto ev_pair = [&](int op_index) -> size_t {
        const size_t at = h->ev_used;
        h->ev_used += 2;
        h->ev_op.push_back(op_index);
        return at;
    };
    // graph replay (option "graph_replay"): the launch sequence below depends only on the arguments and on the arena's carve, so a
    // forward whose arguments repeat is captured the second time it is seen and replayed afterwards
    dcscn_ctx::GraphKey gkey;
    gkey.x = x; gkey.x2 = x2; gkey.y = y; gkey.stream = stream; gkey.n = n; gkey.H = H; gkey.W = W;
    gkey.split16 = h->split16 ? h->split16_mask : 0;
    gkey.nb = nb;                                        // the pass size (sub_batch_pixels / budget) shapes the launch sequence too
    gkey.h8 = h->conv3_h8 ? 1 : 0;
    gkey.p16 = h->p16_now ? 1 : 0;
    gkey.carve = (unsigned long long)h->carve_gen;
    const bool graphs = h->graph_replay && !h->profile && !h->debug_digest && !h->debug_poison;   // (debug launches allocate / are not part of the key)
    bool capturing = false;
    if (graphs && h->graph_exec && gkey == h->graph_key) {
        HIP_TRY(h, hipGraphLaunch(h->graph_exec, stream));
        HIP_TRY(h, hipEventRecord(h->done_ev, stream));
        h->last_stream = stream;
        h->has_last = true;
        return DCSCN_OK;
    }
    if (graphs && gkey == h->graph_seen) {
        // a caller that is capturing this stream itself (or any other reason capture cannot start): plain launches, replay off for this key
        if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
        else (void)hipGetLastError();
    }
    h->graph_seen = gkey;
    auto abandon_capture = [&]() {
        if (!capturing) return;
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(stream, &g);
        if (g) (void)hipGraphDestroy(g);
        capturing = false;
    };
    if (h->debug_digest) {
        if (nops + 1 > 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "debug_digest: more than 1023 launches");
        if (!h->d_digest) HIP_TRY(h, hipMalloc((void**)&h->d_digest, 1024 * sizeof(unsigned long long)));
        HIP_TRY(h, hipMemsetAsync(h->d_digest, 0, 1024 * sizeof(unsigned long long), stream));
    }
    for (int b = 0; b < batches; ++b) {
        const int b0 = b * nb;
        const int cnt = std::min(nb, n - b0);
        const float* xb = x + (size_t)b0 * H * W;
        const float* x2b = x2 + (size_t)b0 * H * s * W * s;
        float* yb = y + (size_t)b0 * H * s * W * s;
        bool any_h16 = false;
        for (const Op& op : h->ops) any_h16 = any_h16 || op_on_split16(h, op);
        if (any_h16) {                                   // redo flags and the P16 planes' zero records
            const hipError_t me = pass_begin_launch(reinterpret_cast<int32_t*>(static_cast<char*>(h->arena) + h->redo_off), (int)h->redo_ints,
                                                    h->p16_now ? h->d_zrec : nullptr, h->p16_now ? (int)h->h_zrec.size() : 0, stream);
            if (me != hipSuccess) {
                abandon_capture();
                HIP_TRY(h, me);
            }
        }
        for (int i = 0; i < nops; ++i) {
            size_t evi = 0;
            if (h->profile) { evi = ev_pair(i); HIP_TRY(h, hipEventRecord(h->ev[evi], stream)); }
            if (h->debug_poison) HIP_TRY(h, debug_poison_launch(h->debug_poison, stream));
            rc = launch_op(h, h->ops[i], cnt, H, W, xb, x2b, yb, stream);
            if (rc) {
                abandon_capture();
                return rc;
            }
            if (h->profile) HIP_TRY(h, hipEventRecord(h->ev[evi + 1], stream));
            if (h->debug_digest) {                        // checksum of the buffer(s) this launch writes
                const Op& op = h->ops[i];
                for (int k = 0; k < 2; ++k) {
                    const int idx = op.out_buf[k];
                    if (k == 1 && idx == op.out_buf[0]) break;
                    if (idx >= 0) {
                        const WsBuf& wb = h->bufs[idx];
                        HIP_TRY(h, debug_digest_launch(static_cast<const char*>(h->arena) + wb.offset, wb.p16 ? wb.bytes / 4 : (size_t)cnt * H * wb.res * W * wb.res * wb.stride, h->d_digest + i, stream));
                    } else if (idx == EXT_Y) {
                        HIP_TRY(h, debug_digest_launch(yb, (size_t)cnt * H * s * W * s, h->d_digest + i, stream));
                    }
                }
            }
        }
        // the float32 plan: every launch again on its float32 kernel, gated by the pass's redo flags -- computes the images a split16 launch
        // flagged (a value beyond the f16 range) from the first layer on, as float32 tensors in the same workspace; exits at once otherwise
        if (any_h16) {
            size_t evi = 0;
            if (h->profile) { evi = ev_pair(nops); HIP_TRY(h, hipEventRecord(h->ev[evi], stream)); }
            for (int i = 0; i < nops; ++i) {
                if (!h->ops[i].h16.rerun) continue;           // upstream of every split16 launch: its float32 outputs stand
                rc = launch_op(h, h->ops[i], cnt, H, W, xb, x2b, yb, stream, true);
                if (rc) {
                    abandon_capture();
                    return rc;
                }
            }
            if (h->profile) HIP_TRY(h, hipEventRecord(h->ev[evi + 1], stream));
        }
        if (h->debug_digest) HIP_TRY(h, debug_digest_launch(yb, (size_t)cnt * H * s * W * s, h->d_digest + nops, stream));
    }
    if (capturing) {
        hipGraph_t g = nullptr;
        hipError_t ce = hipStreamEndCapture(stream, &g);
        if (h->graph_exec) {
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        if (ce == hipSuccess) ce = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (ce == hipSuccess) ce = hipGraphLaunch(h->graph_exec, stream);
        if (ce != hipSuccess) {
            // nothing captured has run: forget the graph, forget the key, and issue this forward as plain launches
            (void)hipGetLastError();
            if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
            h->graph_seen = dcscn_ctx::GraphKey{};
            const bool was = h->graph_replay;
            h->graph_replay = false;
            const int rr = run_forward(h, x, x2, y, n, H, W, stream);
            h->graph_replay = was;
            return rr;
        }
        h->graph_key = gkey;
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

}  // namespace dcscn_impl
