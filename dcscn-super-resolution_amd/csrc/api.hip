// C ABI of include/dcscn.h: the extern "C" surface over the plan (plan.h: graph.hip / pack.hip / exec.hip).
#include "plan.h"

#pragma clang fp contract(off)

namespace dcscn_impl {

thread_local std::string g_global_error;

void set_global_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_global_error = buf;
}

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

}  // namespace dcscn_impl

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
        h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
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
    fuse_feat3_stream(h);
    fold_whole_tail(h);                         // x3 / x4: whatever tail the rewrites above left becomes the float32 plan of ONE folded launch
    for (Op& op : h->ops) {
        int rc = finalize_op(h, op);
        if (rc) return rc;
    }
    plan_p16(h);
    for (Op& op : h->ops) {
        int rc = DCSCN_OK;
        if (op.kind == OP_CONV && op.shape.nin && op.h16.on && op.h16.in16_ok) {
            // conv_nin_h with P16 sources: 4 octet entries per 32-channel chunk
            op.h16.h_tab16.assign((size_t)4 * op.h16.n_chunks, NinSrcQuad{0, 0, 0});
            rc = upload(h, op.h16.h_tab16.data(), op.h16.h_tab16.size() * sizeof(NinSrcQuad), (void**)&op.h16.d_tab16);
            if (rc) return rc;
        }
        auto alloc_srctab = [&](Op& o) {
            if (o.multi.empty()) return (int)DCSCN_OK;
            // 4 quads per 16-channel chunk of conv_nin; conv_nin_h walks the same table 8 quads per 32-channel chunk
            o.h_srctab.assign(std::max<size_t>((size_t)4 * o.n_chunks, (size_t)8 * ((o.cin_phys + kNinHKC - 1) / kNinHKC)), NinSrcQuad{0, 0, 0});
            return upload(h, o.h_srctab.data(), o.h_srctab.size() * sizeof(NinSrcQuad), (void**)&o.d_srctab);
        };
        rc = alloc_srctab(op);
        if (rc) return rc;
        for (Op& sub : op.fused) {                      // (the 1x1 GEMM inside a streamed launch: its own float32 launch reads the layers' tensors)
            rc = alloc_srctab(sub);
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
    const bool s16 = op_on_split16(h, op);
    const bool h8 = op_takes_h8(h, op);                          // (the predicate launch_op itself uses)
    snprintf(out->kernel, sizeof out->kernel, "%s", op.kind == OP_CONV ? (s16 ? (op.shape.nin ? "conv_nin_h" : op.fold_s > 0 ? "conv5_h" : h8 ? "conv3_h8" : "conv3_h") : op.shape.wino ? "conv_wino2" : op.shape.nin ? "conv_nin" : "conv_igemm") : op.kind == OP_CIN1 ? "conv_cin1" : op.kind == OP_COUT1 ? "conv_cout1" : op.kind == OP_STREAM ? "feat_stream" : op.kind == OP_TAIL ? "tail_stream" : op.kind == OP_STREAM3 ? (s16 ? "feat3_stream" : "layer by layer") : op.kind == OP_FOLDX ? (s16 ? "conv5_h" : "layer by layer") : "depthwise");
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
    if (op.kind == OP_FOLDX && h->finalized && s16) {
        out->nt = 1; out->kc = 32; out->n_tiles = 1;
        out->executed_macs_per_lr_pixel = 3 * 25 * (int64_t)op.h16.n_chunks * 32 * 16;      // (the border ring's launch repeats 8 % of it on a 48 x 48 patch)
    }
    if (op.kind == OP_CONV && h->finalized) {
        const int64_t r2 = (int64_t)op.res * op.res;
        const int64_t k_exec = (int64_t)op.n_chunks * op.shape.kc;             // padded input channels
        if (s16) {
            // f16 multiply-accumulates issued: 3 products, input channels padded to 32, output channels to 16
            const int64_t tiles = (int64_t)op.h16.n_tiles * (op.h16.nt - 1) + op.h16.n_full;
            out->nt = op.h16.nt; out->kc = 32; out->n_tiles = op.h16.n_tiles;
            // MFMA steps of K = 32: 9 taps per chunk, 3 / 5 / 7 in a packed last chunk (conv3_h.hpp)
            const int64_t ksteps = op.shape.nin ? op.h16.n_chunks : op.fold_s > 0 ? 25 * (int64_t)op.h16.n_chunks
                                                : 9 * (int64_t)(op.h16.n_chunks - (op.h16.tail_octs ? 1 : 0)) + (op.h16.tail_octs ? c3h_tail_steps(op.h16.tail_octs) : 0);
            out->executed_macs_per_lr_pixel = r2 * 3 * ksteps * 32 * tiles * 16;
        } else if (op.shape.nin) {
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
    if (!strcmp(key, "fold_whole_tail")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the fold_whole_tail option must be set before dcscn_finalize");
        h->fold_whole = value != 0;
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
    if (!strcmp(key, "stream_nin")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the stream_nin option must be set before dcscn_finalize");
        h->stream_nin = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "stream_dense")) {
        if (h->finalized) return fail(h, DCSCN_ERR_STATE, "the stream_dense option must be set before dcscn_finalize");
        h->stream_dense = value != 0;
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
    if (!strcmp(key, "split16")) {                  // any time: the f16 images are always built, the option picks the launch
        h->split16 = value != 0;
        h->split16_mask = value == 2 ? 1 : value == 3 ? 2 : 3;
        return DCSCN_OK;
    }
    if (!strcmp(key, "p16")) {                      // any time: the next forward re-carves the workspace (1 = pre-split tensors between split16 launches)
        h->p16 = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "conv3_h8")) {                 // any time: the two kernels take the same filter image
        h->conv3_h8 = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "debug_digest")) {            // debug aid: a checksum of the whole workspace behind every launch (dcscn_debug_digests)
        if ((int)h->ops.size() + 1 > 1024) return fail(h, DCSCN_ERR_UNSUPPORTED, "debug_digest: more than 1023 launches");
        h->debug_digest = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "debug_poison")) {            // debug aid (tools/determinism_check.py): no kernel may depend on what LDS / registers held before it
        h->debug_poison = (int)value;
        return DCSCN_OK;
    }
    if (!strcmp(key, "profile")) {
        h->profile = value != 0;
        return DCSCN_OK;
    }
    if (!strcmp(key, "graph_replay")) {
        h->graph_replay = value != 0;
        if (!h->graph_replay && h->graph_exec) {
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        h->graph_seen = dcscn_ctx::GraphKey{};
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
    // ordered like a forward (exec.hip: run_forward): behind the previous forward / resize of this handle when that ran on
    // another stream (the scratch row buffer rs_tmp is shared, and grow() frees it after synchronising `st` only -- which by
    // then waits for that event too), and recorded so that the next call and dcscn_synchronize see it
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (h->has_last && h->last_stream != st) HIP_TRY(h, hipStreamWaitEvent(st, h->done_ev, 0));
    const int rc = resize_device(h, in, out, n, height, width, out_height, out_width, st);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->done_ev, st));
    h->last_stream = st;
    h->has_last = true;
    return DCSCN_OK;
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
    std::vector<double> acc(nops + 1, 0.0);                 // [nops]: the float32 plan behind the passes (gated launches that normally exit at once)
    const int forwards = h->ev_forwards;
    if (h->ev_used > 0) {
        HIP_TRY(h, hipDeviceSynchronize());
        const size_t launches = h->ev_used / 2;
        for (size_t l = 0; l < launches && l < h->ev_op.size(); ++l) {
            float t = 0.0f;
            HIP_TRY(h, hipEventElapsedTime(&t, h->ev[2 * l], h->ev[2 * l + 1]));
            acc[std::min(std::max(h->ev_op[l], 0), nops)] += t;
        }
    }
    h->ev_used = 0;
    h->ev_forwards = 0;
    h->ev_op.clear();
    if (forwards > 1)
        for (double& v : acc) v /= forwards;
    for (int i = 0; i < std::min(capacity, nops + 1); ++i) ms[i] = acc[i];
    return DCSCN_OK;
}

int dcscn_debug_digests(dcscn_handle h, uint64_t* out, int capacity) {
    if (!h) return DCSCN_ERR_INVALID_ARG;
    if (!out || capacity < 0) return fail(h, DCSCN_ERR_INVALID_ARG, "dcscn_debug_digests: bad argument");
    if (!h->d_digest) return fail(h, DCSCN_ERR_STATE, "dcscn_debug_digests: no forward has run with the debug_digest option on");
    HIP_TRY(h, hipDeviceSynchronize());
    const int n = std::min(capacity, (int)h->ops.size() + 1);
    HIP_TRY(h, hipMemcpy(out, h->d_digest, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return DCSCN_OK;
}

int64_t dcscn_workspace_bytes(dcscn_handle h) { return h ? (int64_t)h->arena_bytes : -1; }

int dcscn_num_presplit_tensors(dcscn_handle h) {
    if (!h) return -DCSCN_ERR_INVALID_ARG;
    if (!h->finalized || !p16_active(h)) return 0;
    int n = 0;
    for (const WsBuf& b : h->bufs) n += b.p16_ok && b.stride > 0;
    return n;
}

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
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->done_ev) (void)hipEventDestroy(h->done_ev);
    for (hipEvent_t e : h->host_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (void* p : h->device_allocs) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    if (h->d_zrec) (void)hipFree(h->d_zrec);
    if (h->d_digest) (void)hipFree(h->d_digest);
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
