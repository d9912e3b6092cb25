/*
 * dcscn.h -- C ABI of the MI355X-native DCSCN forward pass (libdcscn_hip.so).
 *
 * The reference (jiny2001/dcscn-super-resolution) has no FFI of its own: its forward pass is a
 * TensorFlow session call.  This header is the drop-in seam for that call; every entry point cites
 * the reference interface it stands in for (file:line into the reference tree).
 *
 *   reference                                                      this library
 *   ------------------------------------------------------------   ---------------------------
 *   SuperResolution.__init__ + build_graph  (DCSCN.py:29-106,222)   dcscn_create
 *   tf.train.Saver.restore by variable name (tf_graph.py:263-280)   dcscn_set_tensor (+ tensor_info)
 *   init_all_variables / first sess.run     (tf_graph.py:73-75)     dcscn_finalize
 *   sess.run(self.y_, {x, x2, dropout:1.0, is_training:0})          dcscn_forward / dcscn_forward_device
 *                                           (DCSCN.py:565-569,575-578)
 *   self-ensemble loop of do()              (DCSCN.py:559-573)      dcscn_forward_ensemble
 *   logging "Complexity" / layer list       (DCSCN.py:331-332)      dcscn_layer_info
 *   sess.close()                                                    dcscn_destroy
 *
 * Conventions: plain C types only; all image tensors are dense NHWC float32 with C == 1
 * (x: [n, h, w, 1], x2 / y: [n, s*h, s*w, 1], s = scale).  Weights are passed exactly as the
 * checkpoint stores them (conv filters HWIO) under the checkpoint's variable names.  Every call
 * returns a dcscn_status; nothing aborts the process.  A handle owns one device, one HIP stream and
 * its workspace, and must not be used from two threads at once.  There is no CPU fallback: on a
 * machine without a usable HIP device dcscn_create fails with DCSCN_ERR_HIP.
 */
#ifndef DCSCN_H_
#define DCSCN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCSCN_ABI_VERSION 1
#define DCSCN_MAX_NAME 128

typedef struct dcscn_ctx* dcscn_handle;

typedef enum dcscn_status {
    DCSCN_OK = 0,
    DCSCN_ERR_INVALID_ARG = 1,    /* null pointer, bad size, bad enum */
    DCSCN_ERR_UNSUPPORTED = 2,    /* a flag combination the HIP path does not implement */
    DCSCN_ERR_MISSING_TENSOR = 3, /* finalize(): a variable the graph needs was never set
                                     (reference: Saver.restore raises NotFoundError) */
    DCSCN_ERR_SHAPE = 4,          /* set_tensor(): unknown name or shape mismatch */
    DCSCN_ERR_HIP = 5,            /* HIP runtime error / no device */
    DCSCN_ERR_STATE = 6,          /* call order violated (e.g. forward before finalize) */
    DCSCN_ERR_NOMEM = 7
} dcscn_status;

/* helper/tf_graph.py:77-102 build_activator */
typedef enum dcscn_activator {
    DCSCN_ACT_NONE = 0,
    DCSCN_ACT_PRELU = 1,
    DCSCN_ACT_RELU = 2,
    DCSCN_ACT_LEAKY_RELU = 3,
    DCSCN_ACT_SIGMOID = 4,
    DCSCN_ACT_TANH = 5,
    DCSCN_ACT_SELU = 6
} dcscn_activator;

/* Model flags of helper/args.py:17-36 as consumed by DCSCN.py:33-48 and build_graph. */
typedef struct dcscn_config {
    int32_t struct_size;            /* = sizeof(dcscn_config), ABI guard */
    int32_t scale;                  /* 2, 3 or 4 */
    int32_t layers;                 /* feature-extraction layers */
    int32_t filters;
    int32_t min_filters;
    double  filters_decay_gamma;
    int32_t cnn_size;               /* 3 (1 is accepted) */
    int32_t use_nin;
    int32_t nin_filters;            /* A1 */
    int32_t nin_filters2;           /* B1, B2 */
    int32_t reconstruct_layers;     /* max(flag, 1) applied inside, DCSCN.py:42 */
    int32_t reconstruct_filters;
    int32_t activator;              /* dcscn_activator */
    int32_t pixel_shuffler;         /* 1: pixel shuffler (tf_graph.py:238-249); 0: transposed conv "Up-TCNN"
                                       (tf_graph.py:219-236), run as its equivalent 3x3 conv + depth_to_space */
    int32_t pixel_shuffler_filters; /* 0 = same as input channels */
    int32_t depthwise_separable;
    int32_t channels;               /* must be 1 */
    int32_t legacy_no_c;            /* 1: graph of the shipped dcscn_L2_* checkpoints (use_nin=0 and
                                       no 1x1 "C" layer between H_concat and the upsampler) */
    int32_t batch_norm;             /* must be 0 */
    int32_t reserved[8];
} dcscn_config;

/* One conv layer of the graph, in build order (CNN1.., A1, B1, B2 | C, Up-PS.., R-CNN..). */
typedef struct dcscn_layer_info {
    char    name[DCSCN_MAX_NAME];   /* checkpoint scope, e.g. "CNN3", "Up-PS/Up-PS_CNN" */
    int32_t kernel_size;
    int32_t in_channels;
    int32_t out_channels;
    int32_t depthwise_separable;
    int32_t has_bias;
    int32_t activator;
    int32_t resolution;             /* pixels per LR pixel along one axis where the conv runs */
    int64_t macs_per_lr_pixel;      /* multiply-accumulates per LR pixel */
} dcscn_layer_info;

/* Library / ABI version (DCSCN_ABI_VERSION). */
int dcscn_abi_version(void);

/* Message of the last failing call on this thread that had no usable handle (e.g. dcscn_create). */
const char* dcscn_last_global_error(void);

/* Number of visible HIP devices, or a negative dcscn_status. */
int dcscn_device_count(void);

/* Feature-extraction filter schedule of DCSCN.py:232,240-244; writes `layers` ints. */
int dcscn_filter_schedule(int layers, int filters, int min_filters, double gamma, int32_t* out);

/* Build the graph for `cfg` on HIP device `device` (SuperResolution.__init__ + build_graph). */
int dcscn_create(const dcscn_config* cfg, int device, dcscn_handle* out);

/* Variables the graph expects, in checkpoint naming (tf_graph.py:117-216). shape has 4 slots. */
int dcscn_num_tensors(dcscn_handle h);
int dcscn_tensor_info(dcscn_handle h, int index, char* name, int name_capacity, int64_t* shape, int* rank);

/* Copy one variable in (Saver.restore); `shape`/`rank` must match the graph. Host memory. */
int dcscn_set_tensor(dcscn_handle h, const char* name, const float* data, const int64_t* shape, int rank);

/* Repack weights for the MFMA kernels and upload them.  All variables must have been set. */
int dcscn_finalize(dcscn_handle h);

/* Conv layers of the graph (for logging / FLOP accounting). */
int dcscn_num_layers(dcscn_handle h);
int dcscn_layer_info_get(dcscn_handle h, int index, dcscn_layer_info* out);

/* Kernel launches one forward pass is made of, in stream order (valid after dcscn_finalize).
 * A launch covers one graph layer, or two when they are fused (A1 and B1 share one GEMM). */
typedef struct dcscn_op_info {
    char    name[DCSCN_MAX_NAME];   /* e.g. "CNN2", "B1+A1", "CNN3/depthwise" */
    char    kernel[32];             /* "conv_igemm", "conv_wino", "conv_cin1", "conv_cout1", "depthwise" */
    int32_t kernel_size;
    int32_t in_channels;            /* logical */
    int32_t out_channels;           /* logical */
    int32_t resolution;
    int32_t mt, nt, kc, n_tiles;    /* conv_igemm variant (0 for the other kernels) */
    int64_t macs_per_lr_pixel;      /* algorithmic multiply-accumulates per LR pixel */
    int64_t bytes_per_lr_pixel;     /* algorithmic activation bytes read + written per LR pixel */
    int64_t executed_macs_per_lr_pixel; /* multiply-accumulates the kernel really issues: channel padding
                                       included, 16/36 of the direct form for the Winograd kernel */
} dcscn_op_info;
int dcscn_num_ops(dcscn_handle h);
int dcscn_op_info_get(dcscn_handle h, int index, dcscn_op_info* out);

/* Tunables: "sub_batch_pixels" (LR pixels processed per pass through the layer chain, default 4 Mi),
 * "workspace_budget_bytes" (caps the pass size so the activation workspace stays below it, default
 * 48 GiB; an image whose workspace alone exceeds it is cut into equally shaped windows overlapping by twice
 * the network's receptive-field radius, run as a batch and stitched -- "spatial_tiling" 0 disables that and
 * lets the allocation fail instead), "profile" (1: time every launch with HIP events, read back with dcscn_get_profile),
 * "winograd" (default 1; before dcscn_finalize only: 0 keeps every 3x3 conv on the direct
 * implicit-GEMM kernel instead of the Winograd F(2x2,3x3) kernel), "nin_gemm" (default 1; before dcscn_finalize only: 0 sends
 * the wide 1x1 convs -- A1 || B1 over the skip-concat -- to the generic implicit-GEMM kernel instead of conv_nin_h / conv_nin),
 * "fold_linear_tail" (default 1; before dcscn_finalize only): the last pixel-shuffler conv, depth_to_space and the last
 * reconstruction conv -- all linear, no activator between them, DCSCN.py:293-323 -- run as ONE 5x5 conv of the
 * low-resolution map with per-phase / per-border kernels composed in float64 from the checkpoint tensors; the same
 * function, f32 results differ from the layer-by-layer graph by re-association only (parity-tested against the float64
 * oracle at the same 1e-4 bar).  0 = the escape hatch: execute the reference's layers one by one.  Ignored where the
 * graph has no such tail: separable convs, transposed-conv upsampler, reconstruct_layers > 1, cnn_size != 3 -- and,
 * with the default value 1, where the composite would be MORE work than the layers (pixel shufflers to fewer than 12
 * channels at x2: the c-DCSCN nets); 2 folds there too.  The work rule is evaluated ONCE, in dcscn_finalize, for the "split16"
 * value in force then (a one-tile composite is cheap on conv5_h and folds; on the f32 kernel it would not): a handle whose
 * split16 option is flipped afterwards keeps the plan it was finalized with.
 * "fold_whole_tail" (default 1; before dcscn_finalize only; needs fold_linear_tail != 0 and split16 = 1 at finalize): at x3 and x4 EVERY
 * pixel-shuffler stage is built without an activator (tf_graph.py:238-249 as called from DCSCN.py:300-311), so the whole tail -- Up-PS,
 * depth_to_space, (Up-PS2, depth_to_space,) the last reconstruction conv, dense or depthwise separable -- is one affine map of the LR tensor:
 * a 5x5 conv to scale^2 <= 16 sub-pixel phases (ONE channel tile) whose kernel differs only on the first / last row and column of the
 * image, where the reference zero-pads the SHUFFLED maps.  One launch computes the interior with the interior kernels, a second one the
 * border ring with the kernels of its 15 (row, column) position classes, all composed in float64 (csrc/pack.hip: pack_foldx); the launches it
 * replaces -- the shuffler conv(s), the r05 fold of the last stage, tail_stream -- remain the float32 plan of a flagged image and the
 * split16 = 0 path.  Same function, same parity bars.  0 = the r05 plans.
 * "dense_features" (default 1; before dcscn_finalize only): every feature layer stores into its own dense NHWC buffer and
 * the 1x1 layer(s) that consume tf.concat (DCSCN.py:234) walk the list of buffers, instead of all layers sharing one
 * [n, h, w, sum C_i] tensor -- same bits, full cache lines.  0 = one concat tensor.  Ignored where a consumer of the
 * concat is not a 1x1 GEMM launch (cnn_size > 1 reconstruction without NIN).
 * "stream_features" / "stream_tail" (default 1; before dcscn_finalize only): the separable narrow nets (depthwise_separable,
 * <= 7 feature layers of <= 32 filters, NIN of <= 32 channels) run CNN1 .. CNNL, A1 || B1 and B2 as ONE row-streamed launch
 * with every intermediate tensor in LDS, and -- x4 models with a 32-channel pixel shuffler -- Up-PS, Up-PS2, the last
 * reconstruction conv and the residual add as a second one.  0 = the layer-by-layer launches (same function, f32 results
 * differ by accumulation order only).
 * "stream_dense" (default 1; before dcscn_finalize only): the NON-separable narrow nets (plain 3x3 feature layers of <= 32 filters --
 * the c-DCSCN checkpoints) run CNN1 .. CNNL as one row-streamed launch on the f16 matrix pipe (csrc/feat3_stream.hpp: three-row rings of
 * pre-split units in LDS, filter fragments in registers, every layer's rows written once for A1 || B1); with split16 = 0, and for a
 * flagged image, the layers run one by one.  0 = always layer by layer.
 * "stream_nin" (default 1; before dcscn_finalize only): in that launch also A1 || B1 -- accumulated in registers as the layers' rows appear in
 * their LDS rings -- and B2, where the net has the c-DCSCN shape (7 feature layers, nin_filters 24, nin_filters2 8): no feature map is
 * written to device memory, Concat2 [B2 | A1] is the launch's only output (DCSCN.py:258-291).  0 = every layer's rows go to device memory
 * and the 1x1 GEMM reads them back (the r05 plan; same function, f32 results differ by accumulation order only).
 * "split16" (default 1; any time): the 3x3 convs the Winograd kernel would take and the wide 1x1 convs run their contraction
 * on the f16 matrix pipe at f32 accuracy -- every f32 operand as an f16 (hi, lo) pair, three products per MAC, f32
 * accumulation; measured error at the f32 kernels' level (profiles/r03_f16x3_numerics.txt).  Weights are scaled
 * by a power of two per layer before the split; activations are split unscaled, so below |x| ~ 2^-3 their `lo` piece is an f16
 * subnormal and the pair carries an ABSOLUTE error floor of ~3e-8 per element instead of a relative 2^-22 -- immaterial for the
 * network's data (tested: inputs in [0, 1] as --max_value=1 feeds them meet the same 5e-6 bar on the bare branch,
 * test_small_magnitude_inputs_on_split16), but not "f32 accuracy for any input".  Per layer the split16 kernels are about twice
 * as far from the float64 value as the f32 Winograd kernel (max error 2.9e-4 against 1.4e-4 on outputs of magnitude 260, rms 1.8e-5
 * against 1.6e-5: profiles/r04_h16_conv3_harness.txt); end to end both families meet the same bars.  An activation beyond the f16
 * range (|x| >= 65520) cannot be carried as a pair: the launch that meets it (a non-finite accumulator) or produces it (a P16 output,
 * see "p16") raises the IMAGE's redo flag, and behind every pass the float32 launches of all layers run once more, gated by those
 * flags: a flagged image is recomputed from the first layer on by the float32 kernels (bit-identical to a split16 = 0 run of
 * that image), every other image of the batch keeps its split16 bits -- results never depend on what else is in the batch.  With no
 * flag set (always, for image data) the gated launches exit at once.
 * 0 = the pure f32 kernels (conv_wino2 / conv_nin).
 * "p16" (default 1; any time, the next forward re-carves the workspace): tensors that only split16 launches write and read (the
 * feature maps, B1, Concat2) are kept PRE-SPLIT in the workspace -- per pixel and 32-channel chunk one aligned 128-byte record of
 * f16 (hi | lo) units instead of float32 values, same 4 bytes per value (csrc/p16.hpp) -- so the split happens once, in the producer's
 * epilogue, and the consumers stage their input tiles by LDS-DMA.  Same split, same products in the same order: results are
 * bit-identical to p16 = 0.  Needs split16 = 1 (both kernel families); otherwise the float32 tensors of r04 are used.
 * "conv3_h8" (default 1; any time): 3x3 layers whose output channels form two channel groups (7 .. 12 tiles of 16) run on
 * conv3_h8 -- one persistent 8-wave workgroup per CU that stages a pixel tile's input once for both groups -- instead of two
 * conv3_h workgroups per tile; same filter image, bit-identical results.
 * "graph_replay" (default 0; any time): a dcscn_forward_device call whose arguments repeat (same x / x2 / y pointers, shape and
 * stream) is captured into a hipGraph the second time it is seen and replayed from then on: one graph launch instead of the
 * pass's ~30 kernel launches (the launch gaps are 0.4 % of a 1024-patch pass of the L12 model, 3 % for the narrow nets).  The
 * data in the buffers may change between calls, the pointers may not; any other call falls back to plain launches. */
int dcscn_set_option(dcscn_handle h, const char* key, int64_t value);

/* Forward pass on host buffers: H2D, kernels, D2H, synchronous. */
int dcscn_forward(dcscn_handle h, const float* x, const float* x2, float* y, int n, int height, int width);

/* Bicubic resize of `n` single-channel float images [height, width] -> [out_height, out_width], bit-compatible
 * with Pillow's Image.resize(BICUBIC) on mode-"F" images (a = -0.5 cubic, antialiased when shrinking, float64
 * accumulation, horizontal pass first): replaces util.resize_image_by_pil (helper/utilty.py:211-239) for the
 * single-channel images of the path.  Host buffers, synchronous; the _device form takes device pointers and a
 * hipStream_t (NULL = the handle's stream) and does not synchronise.  Usable before dcscn_finalize. */
int dcscn_resize_bicubic(dcscn_handle h, const float* in, float* out, int n, int height, int width,
                         int out_height, int out_width);
/* The per-axis tables the resize uses (Pillow's precompute_coeffs for BICUBIC over the whole axis): for output
 * index i, `bounds[2i]` = first input index, `bounds[2i+1]` = taps, `weights[i * ksize + t]` = normalised float64
 * weight of tap t.  Host only, needs no device.  Call with bounds = weights = NULL to get `ksize`; `capacity` is
 * the number of doubles `weights` can hold (>= out_size * ksize). */
int dcscn_resample_table(int in_size, int out_size, int* ksize, int* bounds, double* weights, int capacity);
int dcscn_resize_bicubic_device(dcscn_handle h, const float* in, float* out, int n, int height, int width,
                                int out_height, int out_width, void* stream);

/* do(input_image, bicubic_input_image=None) (DCSCN.py:547-554): x2 is the bicubic upscale of x, computed on the
 * device with dcscn_resize_bicubic; otherwise as dcscn_forward. */
int dcscn_forward_lr(dcscn_handle h, const float* x, float* y, int n, int height, int width);

/* Forward pass on device buffers, enqueued on `stream` (a hipStream_t, NULL = the handle's own
 * stream, see dcscn_get_stream) without synchronising.  Workspace growth (first call / larger shape) does
 * synchronise.  Consecutive calls on DIFFERENT streams are ordered by the library (they share the workspace): the
 * later call waits, on the device, for the earlier one.  The reference analogue is a second sess.run on the same
 * session (DCSCN.py:565-569): it simply runs after the first. */
int dcscn_forward_device(dcscn_handle h, const float* x, const float* x2, float* y,
                         int n, int height, int width, void* stream);

/* The handle's own hipStream_t (created non-blocking: it does NOT synchronise with the legacy default stream), so a
 * caller that passes stream = NULL above can order its own work against it (hipStreamWaitEvent / hipStreamSynchronize). */
int dcscn_get_stream(dcscn_handle h, void** stream);

/* Blocks until every forward / resize enqueued through this handle has finished, whatever stream it ran on
 * (sess.run is synchronous; this is the explicit form for the _device entry points). */
int dcscn_synchronize(dcscn_handle h);

/* do() with self_ensemble = n_ensemble in [1, 8] for ONE image (DCSCN.py:559-573): the flipped /
 * rotated copies (utilty.py:595-617) run as two batches on the device; the mean over copies is
 * accumulated in float64 in the reference's order.  x: [h, w], x2: [s*h, s*w], y: float64 [s*h, s*w]. */
int dcscn_forward_ensemble(dcscn_handle h, const float* x, const float* x2, double* y,
                           int height, int width, int n_ensemble);

/* Colour conversions of helper/utilty.py:142-193 on the device, float64 like the reference's numpy (host buffers,
 * synchronous; usable before dcscn_finalize).  rgb: uint8 [n_pixels, 3] interleaved.
 *   dcscn_convert_rgb_to_y            utilty.py:142-149   y     float64 [n_pixels]     = [65.738 129.057 25.064]/256 . rgb + 16
 *   dcscn_convert_rgb_to_ycbcr        utilty.py:152-165   ycbcr float64 [n_pixels, 3]
 *   dcscn_convert_y_and_cbcr_to_rgb   utilty.py:168-193   y float64 [n_pixels], cbcr float64 [n_pixels, 2] -> rgb float64 [n_pixels, 3] */
int dcscn_convert_rgb_to_y(dcscn_handle h, const uint8_t* rgb, double* y, int64_t n_pixels);
int dcscn_convert_rgb_to_ycbcr(dcscn_handle h, const uint8_t* rgb, double* ycbcr, int64_t n_pixels);
int dcscn_convert_y_and_cbcr_to_rgb(dcscn_handle h, const double* y, const double* cbcr, double* rgb, int64_t n_pixels);

/* do_for_evaluate's image pipeline (DCSCN.py:672-703, loader.py:42-67) with ONE upload: the aligned true image
 * (uint8 RGB [height, width, 3], both multiples of the scale) -> Y in float64 (convert_rgb_to_y) -> LR = Pillow BICUBIC
 * of the mode-'F' Y at 1/scale -> x2 = BICUBIC of LR at scale -> do() with self_ensemble = n_ensemble.
 * Outputs (host): y float64 [height, width] (n_ensemble == 1: the float32 network output, widened), and optionally
 * true_y float64 [height, width] (what the PSNR is measured against) and lr float32 [height/s, width/s]. */
int dcscn_evaluate_rgb(dcscn_handle h, const uint8_t* rgb, int height, int width, int n_ensemble,
                       double* true_y, float* lr, double* y);

/* do_for_file's colour pipeline (sr.py; DCSCN.py:588-614): rgb uint8 [height, width, 3] is the input image,
 * rgb_upscaled uint8 [s*height, s*width, 3] its Pillow-upscaled copy (the reference builds it on the host for the
 * "_bicubic" output anyway; Pillow's uint8 RGB resize is integer arithmetic and stays there).  Device: Y of the input
 * -> x, x2 = BICUBIC(x), do() -> y; CbCr of rgb_upscaled; rgb_out = convert_y_and_cbcr_to_rgb(y, cbcr), float64
 * [s*height, s*width, 3].  y (optional) receives the super-resolved luma, float64. */
int dcscn_sr_rgb(dcscn_handle h, const uint8_t* rgb, const uint8_t* rgb_upscaled, int height, int width, int n_ensemble,
                 double* y, double* rgb_out);

/* Per-launch (dcscn_op_info order) device milliseconds, summed over sub-batches and averaged over
 * the forwards run with the profile option on since the previous call (which this call resets);
 * `ms` receives min(capacity, num_ops + 1) entries; entry num_ops is the float32 plan behind the split16 passes (the gated float32
 * launches, see "split16": microseconds unless an image left the f16 range).  Synchronises the device. */
int dcscn_get_profile(dcscn_handle h, double* ms, int capacity);

/* Debug aid (no reference counterpart): with dcscn_set_option("debug_digest", 1) every launch of a forward is followed by a
 * position-weighted checksum of the workspace tensor(s) it writes, and the pass ends with one of its output; `out` receives
 * min(capacity, num_ops + 1) values of the LAST pass.  Two runs of the same input must agree entry by entry -- the first entry
 * that differs names the launch whose result changed (tools/determinism_check.py).  "debug_poison" (1 LDS, 2 vector
 * registers, 3 both) fills those with NaN patterns in front of every launch: no kernel may depend on what they held. */
int dcscn_debug_digests(dcscn_handle h, uint64_t* out, int capacity);

/* Bytes of device workspace currently held. */
int64_t dcscn_workspace_bytes(dcscn_handle h);
/* How many workspace tensors the next forward keeps pre-split (option "p16"; 0 when the option, split16 or the graph rules it out).
 * The reference has no counterpart (sess.run hides its buffers); diagnostic for tests and benchmarks.  After dcscn_finalize. */
int dcscn_num_presplit_tensors(dcscn_handle h);

const char* dcscn_last_error(dcscn_handle h);
int dcscn_destroy(dcscn_handle h);

#ifdef __cplusplus
}
#endif
#endif /* DCSCN_H_ */
