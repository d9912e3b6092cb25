"""ctypes binding of libdcscn_hip.so (include/dcscn.h): the device side of ``sess.run(self.y_, ...)``.

There is deliberately no fallback: if the shared library is missing or no gfx950 device is usable,
constructing an :class:`Engine` raises :class:`EngineError`.
"""

import ctypes
import os

import numpy as np

from . import build as _build

ABI_VERSION = 1
MAX_NAME = 128

ACTIVATORS = {None: 0, "": 0, "none": 0, "prelu": 1, "relu": 2, "leaky_relu": 3, "sigmoid": 4, "tanh": 5, "selu": 6}

STATUS_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "UNSUPPORTED", 3: "MISSING_TENSOR", 4: "SHAPE", 5: "HIP", 6: "STATE",
                7: "NOMEM"}

# every symbol include/dcscn.h declares (tests check the library exports exactly these)
EXPORTED_SYMBOLS = (
    "dcscn_abi_version", "dcscn_last_global_error", "dcscn_device_count", "dcscn_filter_schedule", "dcscn_create",
    "dcscn_num_tensors", "dcscn_tensor_info", "dcscn_set_tensor", "dcscn_finalize", "dcscn_num_layers",
    "dcscn_layer_info_get", "dcscn_num_ops", "dcscn_op_info_get", "dcscn_set_option", "dcscn_forward",
    "dcscn_forward_device", "dcscn_forward_ensemble", "dcscn_get_profile", "dcscn_debug_digests", "dcscn_workspace_bytes", "dcscn_num_presplit_tensors",
    "dcscn_last_error", "dcscn_destroy", "dcscn_resize_bicubic", "dcscn_resize_bicubic_device", "dcscn_forward_lr",
    "dcscn_resample_table", "dcscn_get_stream", "dcscn_synchronize", "dcscn_convert_rgb_to_y", "dcscn_convert_rgb_to_ycbcr",
    "dcscn_convert_y_and_cbcr_to_rgb", "dcscn_evaluate_rgb", "dcscn_sr_rgb",
)


class EngineError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("dcscn: %s (%s)" % (message, STATUS_NAMES.get(status, status)))
        self.status = status
        self.message = message


class Config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("scale", ctypes.c_int32),
        ("layers", ctypes.c_int32),
        ("filters", ctypes.c_int32),
        ("min_filters", ctypes.c_int32),
        ("filters_decay_gamma", ctypes.c_double),
        ("cnn_size", ctypes.c_int32),
        ("use_nin", ctypes.c_int32),
        ("nin_filters", ctypes.c_int32),
        ("nin_filters2", ctypes.c_int32),
        ("reconstruct_layers", ctypes.c_int32),
        ("reconstruct_filters", ctypes.c_int32),
        ("activator", ctypes.c_int32),
        ("pixel_shuffler", ctypes.c_int32),
        ("pixel_shuffler_filters", ctypes.c_int32),
        ("depthwise_separable", ctypes.c_int32),
        ("channels", ctypes.c_int32),
        ("legacy_no_c", ctypes.c_int32),
        ("batch_norm", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 8),
    ]


class LayerInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * MAX_NAME),
        ("kernel_size", ctypes.c_int32),
        ("in_channels", ctypes.c_int32),
        ("out_channels", ctypes.c_int32),
        ("depthwise_separable", ctypes.c_int32),
        ("has_bias", ctypes.c_int32),
        ("activator", ctypes.c_int32),
        ("resolution", ctypes.c_int32),
        ("macs_per_lr_pixel", ctypes.c_int64),
    ]


class OpInfo(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * MAX_NAME),
        ("kernel", ctypes.c_char * 32),
        ("kernel_size", ctypes.c_int32),
        ("in_channels", ctypes.c_int32),
        ("out_channels", ctypes.c_int32),
        ("resolution", ctypes.c_int32),
        ("mt", ctypes.c_int32),
        ("nt", ctypes.c_int32),
        ("kc", ctypes.c_int32),
        ("n_tiles", ctypes.c_int32),
        ("macs_per_lr_pixel", ctypes.c_int64),
        ("bytes_per_lr_pixel", ctypes.c_int64),
        ("executed_macs_per_lr_pixel", ctypes.c_int64),
    ]


_lib = None


def library_path():
    return _build.LIB_PATH


def load_library():
    """dlopen libdcscn_hip.so (built in-tree by build.py / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise EngineError(5, "HIP extension %s is missing; run `python __graft_entry__.py` (or "
                             "dcscn-super-resolution_amd/build.py) to compile it with hipcc" % path)
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise EngineError(5, "cannot load %s: %s" % (path, exc))
    c = ctypes
    fp, dp, vp = c.POINTER(c.c_float), c.POINTER(c.c_double), c.c_void_p
    lib.dcscn_abi_version.restype = c.c_int
    lib.dcscn_last_global_error.restype = c.c_char_p
    lib.dcscn_device_count.restype = c.c_int
    lib.dcscn_filter_schedule.argtypes = [c.c_int, c.c_int, c.c_int, c.c_double, c.POINTER(c.c_int32)]
    lib.dcscn_create.argtypes = [c.POINTER(Config), c.c_int, c.POINTER(vp)]
    lib.dcscn_num_tensors.argtypes = [vp]
    lib.dcscn_tensor_info.argtypes = [vp, c.c_int, c.c_char_p, c.c_int, c.POINTER(c.c_int64), c.POINTER(c.c_int)]
    lib.dcscn_set_tensor.argtypes = [vp, c.c_char_p, fp, c.POINTER(c.c_int64), c.c_int]
    lib.dcscn_finalize.argtypes = [vp]
    lib.dcscn_num_layers.argtypes = [vp]
    lib.dcscn_layer_info_get.argtypes = [vp, c.c_int, c.POINTER(LayerInfo)]
    lib.dcscn_num_ops.argtypes = [vp]
    lib.dcscn_op_info_get.argtypes = [vp, c.c_int, c.POINTER(OpInfo)]
    lib.dcscn_set_option.argtypes = [vp, c.c_char_p, c.c_int64]
    lib.dcscn_forward.argtypes = [vp, fp, fp, fp, c.c_int, c.c_int, c.c_int]
    lib.dcscn_forward_device.argtypes = [vp, vp, vp, vp, c.c_int, c.c_int, c.c_int, vp]
    lib.dcscn_forward_ensemble.argtypes = [vp, fp, fp, dp, c.c_int, c.c_int, c.c_int]
    lib.dcscn_resize_bicubic.argtypes = [vp, fp, fp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int]
    lib.dcscn_resize_bicubic_device.argtypes = [vp, vp, vp, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, vp]
    lib.dcscn_forward_lr.argtypes = [vp, fp, fp, c.c_int, c.c_int, c.c_int]
    lib.dcscn_resample_table.argtypes = [c.c_int, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int), dp, c.c_int]
    lib.dcscn_get_profile.argtypes = [vp, dp, c.c_int]
    lib.dcscn_debug_digests.argtypes = [vp, c.POINTER(c.c_uint64), c.c_int]
    lib.dcscn_workspace_bytes.argtypes = [vp]
    lib.dcscn_num_presplit_tensors.argtypes = [vp]
    lib.dcscn_get_stream.argtypes = [vp, c.POINTER(vp)]
    u8p = c.POINTER(c.c_uint8)
    lib.dcscn_convert_rgb_to_y.argtypes = [vp, u8p, dp, c.c_int64]
    lib.dcscn_convert_rgb_to_ycbcr.argtypes = [vp, u8p, dp, c.c_int64]
    lib.dcscn_convert_y_and_cbcr_to_rgb.argtypes = [vp, dp, dp, dp, c.c_int64]
    lib.dcscn_evaluate_rgb.argtypes = [vp, u8p, c.c_int, c.c_int, c.c_int, dp, fp, dp]
    lib.dcscn_sr_rgb.argtypes = [vp, u8p, u8p, c.c_int, c.c_int, c.c_int, dp, dp]
    lib.dcscn_synchronize.argtypes = [vp]
    lib.dcscn_workspace_bytes.restype = c.c_int64
    lib.dcscn_last_error.argtypes = [vp]
    lib.dcscn_last_error.restype = c.c_char_p
    lib.dcscn_destroy.argtypes = [vp]
    if lib.dcscn_abi_version() != ABI_VERSION:
        raise EngineError(5, "ABI mismatch: library %d, binding %d" % (lib.dcscn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def filter_schedule(layers, filters, min_filters, gamma):
    lib = load_library()
    out = (ctypes.c_int32 * layers)()
    rc = lib.dcscn_filter_schedule(layers, filters, min_filters, gamma, out)
    if rc:
        raise EngineError(rc, lib.dcscn_last_global_error().decode())
    return list(out)


def device_count():
    return load_library().dcscn_device_count()


def make_config(cfg):
    """Translate a flags-style dict / object (names of helper/args.py) into the C struct."""
    get = cfg.get if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
    c = Config()
    c.struct_size = ctypes.sizeof(Config)
    c.scale = int(get("scale", 2))
    c.layers = int(get("layers", 12))
    c.filters = int(get("filters", 196))
    c.min_filters = int(get("min_filters", 48))
    c.filters_decay_gamma = float(get("filters_decay_gamma", 1.5))
    c.cnn_size = int(get("cnn_size", 3))
    c.use_nin = int(bool(get("use_nin", True)))
    c.nin_filters = int(get("nin_filters", 64))
    c.nin_filters2 = int(get("nin_filters2", 32))
    c.reconstruct_layers = int(get("reconstruct_layers", 1))
    c.reconstruct_filters = int(get("reconstruct_filters", 32))
    act = get("activator", "prelu")
    if act not in ACTIVATORS:
        raise NameError("Not implemented activator:%s" % act)      # tf_graph.py:98
    c.activator = ACTIVATORS[act]
    c.pixel_shuffler = int(bool(get("pixel_shuffler", True)))
    c.pixel_shuffler_filters = int(get("pixel_shuffler_filters", 0))
    c.depthwise_separable = int(bool(get("depthwise_separable", False)))
    c.channels = int(get("channels", 1))
    c.legacy_no_c = int(bool(get("legacy_no_c", False)))
    c.batch_norm = int(bool(get("batch_norm", False)))
    return c


def resample_table(in_size, out_size):
    """(bounds [out, 2] int32, weights [out, ksize] float64) of the Pillow-compatible bicubic resize along one
    axis (dcscn_resample_table); host code of the library, needs no GPU."""
    lib = load_library()
    ks = ctypes.c_int(0)
    rc = lib.dcscn_resample_table(int(in_size), int(out_size), ctypes.byref(ks), None, None, 0)
    if rc:
        raise EngineError(rc, "dcscn_resample_table(%d, %d)" % (in_size, out_size))
    bounds = np.zeros((int(out_size), 2), np.int32)
    weights = np.zeros((int(out_size), ks.value), np.float64)
    rc = lib.dcscn_resample_table(int(in_size), int(out_size), ctypes.byref(ks),
                                  bounds.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                  weights.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), weights.size)
    if rc:
        raise EngineError(rc, "dcscn_resample_table(%d, %d)" % (in_size, out_size))
    return bounds, weights


class Engine:
    """One DCSCN graph resident on one MI355X (``dcscn_handle``)."""

    def __init__(self, cfg, device=0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.scale = int(cfg.get("scale", 2) if isinstance(cfg, dict) else getattr(cfg, "scale", 2))
        c = make_config(cfg)
        rc = self._lib.dcscn_create(ctypes.byref(c), int(device), ctypes.byref(self._h))
        if rc:
            self._h = ctypes.c_void_p()
            raise EngineError(rc, self._lib.dcscn_last_global_error().decode())
        self.device = int(device)
        self.finalized = False

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.dcscn_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc:
            raise EngineError(rc, self._lib.dcscn_last_error(self._h).decode())

    # -- graph description -------------------------------------------------------------------
    def tensor_specs(self):
        """[(checkpoint variable name, shape)] the graph expects."""
        out = []
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int64 * 4)()
        rank = ctypes.c_int()
        for i in range(self._lib.dcscn_num_tensors(self._h)):
            self._check(self._lib.dcscn_tensor_info(self._h, i, name, 256, shape, ctypes.byref(rank)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(rank.value))))
        return out

    def layers(self):
        out = []
        info = LayerInfo()
        for i in range(self._lib.dcscn_num_layers(self._h)):
            self._check(self._lib.dcscn_layer_info_get(self._h, i, ctypes.byref(info)))
            out.append({f: (getattr(info, f).decode() if f == "name" else getattr(info, f)) for f, _ in info._fields_})
        return out

    def ops(self):
        out = []
        info = OpInfo()
        for i in range(self._lib.dcscn_num_ops(self._h)):
            self._check(self._lib.dcscn_op_info_get(self._h, i, ctypes.byref(info)))
            out.append({f: (getattr(info, f).decode() if f in ("name", "kernel") else getattr(info, f))
                        for f, _ in info._fields_})
        return out

    # -- weights -----------------------------------------------------------------------------
    def set_tensor(self, name, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
        self._check(self._lib.dcscn_set_tensor(self._h, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                               shape, a.ndim))

    def load_weights(self, tensors, winograd=None, fold_tail=None, split16=None):
        """Feed every variable the graph needs from ``{name: ndarray}`` and finalize.
        ``split16=False`` (or DCSCN_SPLIT16=0) keeps every contraction on the f32 kernels; the library default runs the wide
        3x3 and 1x1 convs on the f16 matrix pipe at f32 accuracy (f16 hi/lo split, 3 products, f32 accumulate -- "split16"
        in include/dcscn.h; the option can also be flipped after finalize with ``set_option("split16", 0 / 1)``).
        ``winograd=False`` keeps every 3x3 conv on the direct implicit-GEMM kernel (default: library choice).
        ``fold_tail=False`` executes the reference's layers one by one; the library default runs the linear tail (last
        pixel-shuffler conv, depth_to_space, last reconstruction conv) as one 5x5 conv where that is less work;
        ``fold_tail=True`` (or the environment variable DCSCN_FOLD_TAIL=1; =0 for False) folds wherever the graph has
        such a tail -- see "fold_linear_tail" in include/dcscn.h."""
        if winograd is not None:
            self.set_option("winograd", 1 if winograd else 0)
        if split16 is None and os.environ.get("DCSCN_SPLIT16") in ("0", "1"):
            split16 = os.environ["DCSCN_SPLIT16"] == "1"
        if split16 is not None:
            self.set_option("split16", 1 if split16 else 0)
        if fold_tail is None and os.environ.get("DCSCN_FOLD_TAIL") in ("0", "1"):
            fold_tail = os.environ["DCSCN_FOLD_TAIL"] == "1"
        if fold_tail is not None:
            self.set_option("fold_linear_tail", 2 if fold_tail else 0)       # an explicit request folds wherever it is possible
        for name, _ in self.tensor_specs():
            if name not in tensors:
                raise EngineError(3, "variable '%s' is missing from the checkpoint" % name)
            self.set_tensor(name, tensors[name])
        self.finalize()

    def finalize(self):
        self._check(self._lib.dcscn_finalize(self._h))
        self.finalized = True

    def set_option(self, key, value):
        self._check(self._lib.dcscn_set_option(self._h, key.encode(), int(value)))

    # -- forward -----------------------------------------------------------------------------
    @staticmethod
    def _out_buffer(out, shape):
        if out is None:
            return np.empty(shape, dtype=np.float32)
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.size != int(np.prod(shape)):
            raise EngineError(1, "out must be a C-contiguous float32 array of %s" % (shape,))
        return out.reshape(shape)

    def forward(self, x, x2, out=None):
        """x: [n, h, w, 1] (or [n, h, w]) float32 host array, x2: [n, s*h, s*w, 1]; returns y like x2 (``out``: reuse a
        result buffer -- a fresh 40 MB numpy array costs 2-3 ms of page faults under the download)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        x2 = np.ascontiguousarray(x2, dtype=np.float32)
        if x.ndim == 4:
            if x.shape[3] != 1:
                raise EngineError(1, "x must have one channel, got shape %s" % (x.shape,))
        elif x.ndim != 3:
            raise EngineError(1, "x must be [n, h, w, 1], got shape %s" % (x.shape,))
        n, h, w = x.shape[:3]
        s = self.scale
        if x2.size != n * h * s * w * s:
            raise EngineError(1, "x2 has %d elements, expected %d x %d x %d" % (x2.size, n, h * s, w * s))
        y = self._out_buffer(out, (n, h * s, w * s, 1))
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self._lib.dcscn_forward(self._h, x.ctypes.data_as(fp), x2.ctypes.data_as(fp),
                                            y.ctypes.data_as(fp), n, h, w))
        return y

    def forward_lr(self, x, out=None):
        """``do(input_image, bicubic_input_image=None)`` (DCSCN.py:547-554): x [n, h, w, 1] (or [n, h, w]); the bicubic
        x2 is computed on the device, bit-compatible with Pillow.  Returns y [n, s*h, s*w, 1]."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim == 4:
            x = x[..., 0]
        n, h, w = x.shape
        s = self.scale
        y = self._out_buffer(out, (n, h * s, w * s, 1))
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self._lib.dcscn_forward_lr(self._h, np.ascontiguousarray(x).ctypes.data_as(fp), y.ctypes.data_as(fp), n, h, w))
        return y

    def resize_bicubic(self, images, out_height, out_width):
        """Pillow-compatible (mode 'F', BICUBIC) resize of [n, h, w] float images on the device -> [n, oh, ow]."""
        a = np.ascontiguousarray(images, dtype=np.float32)
        single = a.ndim == 2
        if single:
            a = a[None]
        n, h, w = a.shape
        out = np.empty((n, int(out_height), int(out_width)), np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self._lib.dcscn_resize_bicubic(self._h, a.ctypes.data_as(fp), out.ctypes.data_as(fp), n, h, w,
                                                   int(out_height), int(out_width)))
        return out[0] if single else out

    def stream(self):
        """The handle's own hipStream_t as an int (non-blocking: not ordered against the legacy default stream)."""
        out = ctypes.c_void_p()
        self._check(self._lib.dcscn_get_stream(self._h, ctypes.byref(out)))
        return int(out.value or 0)

    def synchronize(self):
        """Wait for everything enqueued through this handle (whatever stream it ran on)."""
        self._check(self._lib.dcscn_synchronize(self._h))

    def forward_device(self, x_ptr, x2_ptr, y_ptr, n, h, w, stream=None):
        """Enqueue on device pointers (ints, e.g. ``torch.Tensor.data_ptr()``); does not synchronise.

        ``stream``: a hipStream_t as an int; None / 0 means the handle's OWN stream (``self.stream()``), which is not
        ordered against the caller's default stream -- order with ``synchronize()`` or events, or pass a real stream."""
        self._check(self._lib.dcscn_forward_device(self._h, ctypes.c_void_p(x_ptr), ctypes.c_void_p(x2_ptr),
                                                   ctypes.c_void_p(y_ptr), n, h, w,
                                                   ctypes.c_void_p(stream) if stream else None))

    # ---- colour path (helper/utilty.py:142-193 and the RGB pipelines of evaluate.py / sr.py on the device) ----
    @staticmethod
    def _rgb8(image):
        a = np.ascontiguousarray(image)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise EngineError(1, "expected a uint8 RGB image [h, w, 3]")
        return a

    def convert_rgb_to_y(self, image):
        a = self._rgb8(image)
        out = np.empty(a.shape[:2] + (1,), np.float64)
        self._check(self._lib.dcscn_convert_rgb_to_y(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.shape[0] * a.shape[1]))
        return out

    def convert_rgb_to_ycbcr(self, image):
        a = self._rgb8(image)
        out = np.empty(a.shape, np.float64)
        self._check(self._lib.dcscn_convert_rgb_to_ycbcr(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.shape[0] * a.shape[1]))
        return out

    def convert_y_and_cbcr_to_rgb(self, y_image, cbcr_image):
        dp = ctypes.POINTER(ctypes.c_double)
        y = np.ascontiguousarray(np.asarray(y_image, np.float64).reshape(y_image.shape[0], y_image.shape[1], -1)[:, :, 0])
        cbcr = np.ascontiguousarray(np.asarray(cbcr_image, np.float64)[:, :, 0:2])
        out = np.empty(y.shape + (3,), np.float64)
        self._check(self._lib.dcscn_convert_y_and_cbcr_to_rgb(self._h, y.ctypes.data_as(dp), cbcr.ctypes.data_as(dp), out.ctypes.data_as(dp), y.size))
        return out

    def evaluate_rgb(self, true_image, n_ensemble=1, want_inputs=False):
        """do_for_evaluate's pipeline for an aligned uint8 RGB image: returns (true_y float64 [H, W, 1], y [H, W, 1]
        -- float32 for n_ensemble <= 1 like sess.run, float64 otherwise like do()'s mean) and, with want_inputs, the LR input."""
        a = self._rgb8(true_image)
        hh, ww = a.shape[:2]
        s = self.scale
        true_y = np.empty((hh, ww, 1), np.float64)
        y = np.empty((hh, ww, 1), np.float64)
        lr = np.empty((hh // s, ww // s, 1), np.float32) if want_inputs else None
        self._check(self._lib.dcscn_evaluate_rgb(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), hh, ww, max(1, int(n_ensemble)),
                                                 true_y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                 lr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if want_inputs else None,
                                                 y.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        if n_ensemble <= 1:
            y = y.astype(np.float32)
        return (true_y, y, lr) if want_inputs else (true_y, y)

    def sr_rgb(self, image, upscaled_image, n_ensemble=1):
        """do_for_file's colour branch: (super-resolved Y [s*h, s*w, 1], RGB float64 [s*h, s*w, 3])."""
        a, b = self._rgb8(image), self._rgb8(upscaled_image)
        h, w = a.shape[:2]
        s = self.scale
        if b.shape[:2] != (h * s, w * s):
            raise EngineError(1, "the upscaled image must be %d times the input" % s)
        y = np.empty((h * s, w * s, 1), np.float64)
        rgb = np.empty((h * s, w * s, 3), np.float64)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        dp = ctypes.POINTER(ctypes.c_double)
        self._check(self._lib.dcscn_sr_rgb(self._h, a.ctypes.data_as(u8), b.ctypes.data_as(u8), h, w, max(1, int(n_ensemble)),
                                           y.ctypes.data_as(dp), rgb.ctypes.data_as(dp)))
        if n_ensemble <= 1:
            y = y.astype(np.float32)
        return y, rgb

    def forward_ensemble(self, x, x2, n_ensemble):
        """One image [h, w(,1)] + bicubic [s*h, s*w(,1)] -> float64 [s*h, s*w, 1] (do(), DCSCN.py:559-573)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        x2 = np.ascontiguousarray(x2, dtype=np.float32)
        h, w = x.shape[:2]
        s = self.scale
        if x.size != h * w or x2.size != h * s * w * s:
            raise EngineError(1, "forward_ensemble expects single-channel images")
        y = np.empty((h * s, w * s, 1), dtype=np.float64)
        fp = ctypes.POINTER(ctypes.c_float)
        self._check(self._lib.dcscn_forward_ensemble(self._h, x.ctypes.data_as(fp), x2.ctypes.data_as(fp),
                                                     y.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), h, w,
                                                     int(n_ensemble)))
        return y

    def profile(self, with_float32_plan=False):
        """Per-launch milliseconds of the last forward (needs set_option('profile', 1)).  with_float32_plan: one more entry, the gated
        float32 launches behind every split16 pass (they exit at once unless an image left the f16 range; include/dcscn.h "split16")."""
        n = self._lib.dcscn_num_ops(self._h)
        ms = (ctypes.c_double * (n + 1))()
        self._check(self._lib.dcscn_get_profile(self._h, ms, n + 1))
        return list(ms) if with_float32_plan else list(ms)[:n]

    def debug_digests(self):
        """Workspace checksum behind every launch of the last pass + one of its output (needs set_option('debug_digest', 1))."""
        n = self._lib.dcscn_num_ops(self._h) + 1
        out = (ctypes.c_uint64 * n)()
        self._check(self._lib.dcscn_debug_digests(self._h, out, n))
        return [int(v) for v in out]

    def workspace_bytes(self):
        return int(self._lib.dcscn_workspace_bytes(self._h))

    def num_presplit_tensors(self):
        """Workspace tensors the next forward keeps pre-split (option "p16", csrc/p16.hpp)."""
        return int(self._lib.dcscn_num_presplit_tensors(self._h))
