"""CPU ORACLE for the DCSCN forward pass -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The shipped path (``dcscn-super-resolution_amd``) never does: it runs the HIP kernels or fails.

What it is: a plain numpy restatement (float64 by default) of the graph the reference builds with
TensorFlow in ``DCSCN.py:222-325`` (``build_graph``) from the layer helpers in
``helper/tf_graph.py:77-249``, and of the inference driver ``DCSCN.py:547-586`` (``do`` + self
ensemble, flips from ``helper/utilty.py:595-617``).  The arithmetic of the reference lives in
TensorFlow (``tensorflow>=2.0.0``, un-pinned, ``Pipfile:9``), which is not installable here; the
TF ops are restated from their published semantics:

  tf.nn.conv2d  stride 1, padding SAME, NHWC input, HWIO filter  (tf_graph.py:105)
  tf.nn.separable_conv2d = depthwise k x k SAME (channel multiplier 1) then 1x1    (tf_graph.py:161)
  tf.depth_to_space(block b): out[n, h*b+i, w*b+j, c] = in[n, h, w, (i*b+j)*C + c] (tf_graph.py:248)
  PReLU as written: relu(x) + alpha * (x - |x|) * 0.5                              (tf_graph.py:94)
  tf.nn.conv2d_transpose stride s, SAME, filter [k, k, out, in]                    (tf_graph.py:227)
  tf.concat axis 3, tf.add                                                         (DCSCN.py:259,281,325)

PARITY PINNING: the reference ships no tests and no golden vectors, and TF
cannot run here, so bit-level parity with the reference binary is UNPINNED.  What pins this oracle
is the reference's published PSNR table (README.md:55-65, 2 decimals) reproduced with the shipped
checkpoints on the shipped Set5 images (tests/test_oracle_golden.py).
"""

import math

import numpy as np

# --------------------------------------------------------------------------------------------
# configuration (mirrors the model flags of helper/args.py:17-36)
# --------------------------------------------------------------------------------------------

DEFAULT_CONFIG = dict(
    scale=2, layers=12, filters=196, min_filters=48, filters_decay_gamma=1.5,
    use_nin=True, nin_filters=64, nin_filters2=32, cnn_size=3,
    reconstruct_layers=1, reconstruct_filters=32, activator="prelu",
    pixel_shuffler=True, pixel_shuffler_filters=0, depthwise_separable=False,
    channels=1, legacy_no_c=False,
)


def make_config(**overrides):
    cfg = dict(DEFAULT_CONFIG)
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError("unknown config keys: %s" % sorted(unknown))
    cfg.update(overrides)
    cfg["min_filters"] = min(cfg["filters"], cfg["min_filters"])        # DCSCN.py:36
    cfg["reconstruct_layers"] = max(cfg["reconstruct_layers"], 1)       # DCSCN.py:42
    return cfg


def filter_schedule(layers, filters, min_filters, gamma):
    """Feature-extraction filter counts, DCSCN.py:232,240-244."""
    out = []
    n = filters
    for i in range(layers):
        if min_filters != 0 and i > 0:
            x1 = i / float(layers - 1)
            y1 = pow(x1, 1.0 / gamma)
            n = int((filters - min_filters) * (1 - y1) + min_filters)
        out.append(n)
    return out


# --------------------------------------------------------------------------------------------
# topology: a flat list of ops over named tensors, in the order build_graph creates them
# --------------------------------------------------------------------------------------------

def build_topology(cfg):
    """Return the op list of ``build_graph`` (DCSCN.py:222-325).

    Each conv op: dict(op="conv", name, var (variable scope prefix in the checkpoint), src, dst,
    k, cin, cout, bias, act, ds).  Other ops: concat / depth_to_space / add.
    """
    ops = []
    ds = bool(cfg["depthwise_separable"])
    act = cfg["activator"]
    k = cfg["cnn_size"]

    def conv(name, src, ksize, cin, cout, bias, activator, separable, var=None):
        ops.append(dict(op="conv", name=name, var=var or name, src=src, dst=name, k=ksize, cin=cin,
                        cout=cout, bias=bias, act=activator, ds=separable))
        return name

    # feature extraction (DCSCN.py:240-256)
    sched = filter_schedule(cfg["layers"], cfg["filters"], cfg["min_filters"], cfg["filters_decay_gamma"])
    src, cin = "x", cfg["channels"]
    feats = []
    for i, cout in enumerate(sched):
        src = conv("CNN%d" % (i + 1), src, k, cin, cout, True, act, ds)
        feats.append(src)
        cin = cout
    total = sum(sched)
    ops.append(dict(op="concat", srcs=list(feats), dst="H_concat"))                 # DCSCN.py:258-259

    # reconstruction (DCSCN.py:262-291)
    if cfg["use_nin"]:
        conv("A1", "H_concat", 1, total, cfg["nin_filters"], True, act, ds)
        conv("B1", "H_concat", 1, total, cfg["nin_filters2"], True, act, ds)
        conv("B2", "B1", 3, cfg["nin_filters2"], cfg["nin_filters2"], True, act, ds)
        ops.append(dict(op="concat", srcs=["B2", "A1"], dst="Concat2"))             # DCSCN.py:281 (B2 first)
        src, cin = "Concat2", cfg["nin_filters"] + cfg["nin_filters2"]
    elif cfg["legacy_no_c"]:
        # graph of the shipped dcscn_L2_* checkpoints: H_concat feeds the upsampler directly
        src, cin = "H_concat", total
    else:
        src = conv("C", "H_concat", 1, total, cfg["filters"], True, act, ds)
        cin = cfg["filters"]

    # upsampling (DCSCN.py:293-311, tf_graph.py:219-249)
    if cfg["pixel_shuffler"]:
        cout = cfg["pixel_shuffler_filters"] if cfg["pixel_shuffler_filters"] != 0 else cin
        stages = [("Up-PS", 2, cin), ("Up-PS2", 2, cout)] if cfg["scale"] == 4 else [("Up-PS", cfg["scale"], cout)]
        for name, s, c_out in stages:
            conv(name + "_CNN", src, k, cin, s * s * c_out, True, None, ds, var=name + "/" + name + "_CNN")
            ops.append(dict(op="depth_to_space", src=name + "_CNN", dst=name, block=s))
            src, cin = name, c_out
    else:
        # build_transposed_conv("Up-TCNN", H[-1], scale, channels): one stage for any scale, no bias, no
        # activator, filter [k, k, C, C] with k = 2 s - s % 2 (utilty.py:377-390)
        ops.append(dict(op="conv_transpose", name="Up-TCNN", var="Up-TCNN", src=src, dst="Up-TCNN",
                        scale=cfg["scale"], channels=cin))
        src = "Up-TCNN"

    # reconstruction convs at HR (DCSCN.py:313-323); extra layers always use build_conv (never DS)
    rl = cfg["reconstruct_layers"]
    for i in range(rl - 1):
        src = conv("R-CNN%d" % (i + 1), src, k, cin, cfg["reconstruct_filters"], True, act, False)
        cin = cfg["reconstruct_filters"]
    src = conv("R-CNN%d" % rl, src, k, cin, 1, False, None, ds)
    ops.append(dict(op="add", srcs=[src, "x2"], dst="y_"))                          # DCSCN.py:325
    return ops


def variable_shapes(cfg):
    """``{checkpoint variable name: shape}`` the topology consumes (names as in tf_graph.py:117-216)."""
    shapes = {}
    for op in build_topology(cfg):
        if op["op"] == "conv_transpose":
            ks = 2 * op["scale"] - op["scale"] % 2
            shapes[op["var"] + "/Tconv_W"] = (ks, ks, op["channels"], op["channels"])
            continue
        if op["op"] != "conv":
            continue
        v, k, cin, cout = op["var"], op["k"], op["cin"], op["cout"]
        if op["ds"]:
            shapes[v + "/depthwise_W"] = (k, k, cin, 1)
            shapes[v + "/pointwise_W"] = (1, 1, cin, cout)
        else:
            shapes[v + "/conv_W"] = (k, k, cin, cout)
        if op["bias"]:
            shapes[v + "/conv_B"] = (cout,)
        if op["act"] == "prelu":
            shapes[v + "/prelu/" + op["name"] + "_prelu"] = (cout,)
    return shapes


def macs_per_lr_pixel(cfg):
    """Multiply-accumulates per LR pixel; HR-side layers count scale^2."""
    res, total = 1, 0
    for op in build_topology(cfg):
        if op["op"] == "depth_to_space":
            res *= op["block"] ** 2
        elif op["op"] == "conv_transpose":
            ks = 2 * op["scale"] - op["scale"] % 2
            total += res * ks * ks * op["channels"] ** 2        # every input pixel meets every filter tap
            res *= op["scale"] ** 2
        elif op["op"] == "conv":
            k, cin, cout = op["k"], op["cin"], op["cout"]
            total += res * ((k * k * cin + cin * cout) if op["ds"] else k * k * cin * cout)
    return total


# --------------------------------------------------------------------------------------------
# synthetic weights -- seeded, shared by the oracle and the HIP path in tests
# --------------------------------------------------------------------------------------------

def synthetic_weights(cfg, seed=0):
    """He truncated-normal conv weights (utilty.py:360-363), bias ~ N(0, 0.1), alpha ~ U(0.05, 0.3).

    The last reconstruction conv is scaled by 0.01 so that outputs stay near [0, 255].
    """
    rng = np.random.default_rng(seed)
    weights = {}
    shapes = variable_shapes(cfg)
    last = "R-CNN%d" % cfg["reconstruct_layers"]
    for name in sorted(shapes):
        shape = shapes[name]
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "Tconv_W":
            # bilinear initialiser of the reference (utilty.py:366-390) plus seeded noise so that every
            # (out, in) channel pair and every tap is exercised
            size = shape[0]
            factor = (size + 1) // 2
            center = factor - 1 if size % 2 == 1 else factor - 0.5
            og = np.ogrid[:size, :size]
            bil = (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)
            w = 0.05 * rng.standard_normal(shape)
            for i in range(shape[2]):
                w[:, :, i, i] += bil
            weights[name] = w.astype(np.float32)
        elif leaf in ("conv_W", "depthwise_W", "pointwise_W"):
            fan_in = shape[0] * shape[1] * shape[2]
            std = math.sqrt(2.0 / fan_in)
            w = rng.standard_normal(shape)
            bad = np.abs(w) > 2.0                      # truncated normal: resample beyond 2 sigma
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2.0
            w *= std
            if name.startswith(last + "/"):
                w *= 0.01 if leaf != "depthwise_W" else 1.0
            weights[name] = w.astype(np.float32)
        elif leaf == "conv_B":
            weights[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            weights[name] = rng.uniform(0.05, 0.3, shape).astype(np.float32)
    return weights


# --------------------------------------------------------------------------------------------
# layer semantics
# --------------------------------------------------------------------------------------------

def conv2d_same(x, w):
    """tf.nn.conv2d, stride 1, SAME, NHWC x HWIO (tf_graph.py:105). Odd kernel sizes only."""
    kh, kw, cin, cout = w.shape
    assert x.shape[3] == cin, (x.shape, w.shape)
    assert kh % 2 == 1 and kw % 2 == 1
    n, h, wd, _ = x.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin), dtype=x.dtype)
    xp[:, ph:ph + h, pw:pw + wd, :] = x
    out = np.zeros((n, h, wd, cout), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            out += xp[:, dy:dy + h, dx:dx + wd, :] @ w[dy, dx]
    return out


def depthwise_conv2d_same(x, w):
    """tf.nn.depthwise_conv2d with channel multiplier 1 (first half of tf.nn.separable_conv2d)."""
    kh, kw, cin, mult = w.shape
    assert mult == 1 and x.shape[3] == cin
    n, h, wd, _ = x.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((n, h + 2 * ph, wd + 2 * pw, cin), dtype=x.dtype)
    xp[:, ph:ph + h, pw:pw + wd, :] = x
    out = np.zeros_like(x)
    for dy in range(kh):
        for dx in range(kw):
            out += xp[:, dy:dy + h, dx:dx + wd, :] * w[dy, dx, :, 0]
    return out


def conv2d_transpose_same(x, w, s):
    """tf.nn.conv2d_transpose(x, w, [N, sH, sW, C], strides s, padding SAME) (tf_graph.py:227): filter
    [kh, kw, out_c, in_c]; out[n, h*s + ky - pt, w*s + kx - pl, oc] += x[n, h, w, ic] * w[ky, kx, oc, ic]
    with pt = pl = (k - s) // 2 (the SAME padding of the forward conv this op is the gradient of)."""
    kh, kw, oc, ic = w.shape
    n, h, wd, c = x.shape
    assert c == ic
    pt = (kh - s) // 2
    big = np.zeros((n, (h - 1) * s + kh, (wd - 1) * s + kw, oc), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            big[:, ky:ky + (h - 1) * s + 1:s, kx:kx + (wd - 1) * s + 1:s, :] += x @ w[ky, kx].T
    return np.ascontiguousarray(big[:, pt:pt + h * s, pt:pt + wd * s, :])


def activate(x, kind, alpha=None):
    """build_activator, tf_graph.py:77-102."""
    if kind is None or kind == "":
        return x
    if kind == "prelu":
        return np.maximum(x, 0) + alpha * (x - np.abs(x)) * 0.5
    if kind == "relu":
        return np.maximum(x, 0)
    if kind == "leaky_relu":
        return np.maximum(x, 0.1 * x)
    if kind == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    if kind == "tanh":
        return np.tanh(x)
    if kind == "selu":
        scale, a = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717
        return scale * np.where(x > 0, x, a * (np.exp(np.minimum(x, 0)) - 1.0))
    raise NameError("Not implemented activator:%s" % kind)


def depth_to_space(x, block):
    n, h, w, c = x.shape
    cout = c // (block * block)
    assert cout * block * block == c
    x = x.reshape(n, h, w, block, block, cout)
    x = x.transpose(0, 1, 3, 2, 4, 5)
    return np.ascontiguousarray(x.reshape(n, h * block, w * block, cout))


def forward(cfg, weights, x, x2, dtype=np.float64, return_intermediates=False):
    """``sess.run(self.y_, {x, x2, dropout: 1.0, is_training: 0})`` (DCSCN.py:565-569, 575-578).

    x: [N, H, W, channels], x2: [N, sH, sW, 1]; returns y_: [N, sH, sW, 1] in ``dtype``.
    Dropout at keep-rate 1.0 is the identity (tf_graph.py:129-130) and is omitted.
    """
    t = {"x": np.asarray(x, dtype=dtype), "x2": np.asarray(x2, dtype=dtype)}
    for op in build_topology(cfg):
        kind = op["op"]
        if kind == "conv":
            v = op["var"]
            src = t[op["src"]]
            if op["ds"]:
                h = depthwise_conv2d_same(src, weights[v + "/depthwise_W"].astype(dtype))
                h = conv2d_same(h, weights[v + "/pointwise_W"].astype(dtype))
            else:
                h = conv2d_same(src, weights[v + "/conv_W"].astype(dtype))
            if op["bias"]:
                h = h + weights[v + "/conv_B"].astype(dtype)
            alpha = None
            if op["act"] == "prelu":
                alpha = weights[v + "/prelu/" + op["name"] + "_prelu"].astype(dtype)
            t[op["dst"]] = activate(h, op["act"], alpha)
        elif kind == "concat":
            t[op["dst"]] = np.concatenate([t[s] for s in op["srcs"]], axis=3)
        elif kind == "depth_to_space":
            t[op["dst"]] = depth_to_space(t[op["src"]], op["block"])
        elif kind == "conv_transpose":
            t[op["dst"]] = conv2d_transpose_same(t[op["src"]], weights[op["var"] + "/Tconv_W"].astype(dtype), op["scale"])
        elif kind == "add":
            t[op["dst"]] = t[op["srcs"][0]] + t[op["srcs"][1]]
        else:
            raise ValueError(kind)
    return (t["y_"], t) if return_intermediates else t["y_"]


# --------------------------------------------------------------------------------------------
# inference driver: do() + self ensemble (DCSCN.py:547-586, utilty.py:595-617)
# --------------------------------------------------------------------------------------------

def flip(image, flip_type, invert=False):
    """helper/utilty.py:595-617, on [H, W, C] arrays."""
    if flip_type == 0:
        return image
    if flip_type == 1:
        return np.flipud(image)
    if flip_type == 2:
        return np.fliplr(image)
    if flip_type == 3:
        return np.flipud(np.fliplr(image))
    if flip_type == 4:
        return np.rot90(image, 1 if not invert else -1)
    if flip_type == 5:
        return np.rot90(image, -1 if not invert else 1)
    if flip_type == 6:
        return np.flipud(np.rot90(image)) if not invert else np.rot90(np.flipud(image), -1)
    if flip_type == 7:
        return np.flipud(np.rot90(image, -1)) if not invert else np.rot90(np.flipud(image), 1)
    raise ValueError("flip_type must be in [0, 7]")


def do(cfg, weights, input_image, bicubic_image, self_ensemble=1, max_value=255.0, dtype=np.float64):
    """``SuperResolution.do`` for one [h, w, 1] image with its [sh, sw, 1] bicubic upscale.

    Each forward runs at batch 1; the ensemble mean accumulates in float64 exactly as
    ``np.zeros`` + ``+=`` + ``/=`` do in DCSCN.py:560-573.  The forward output is rounded to
    float32 first when ``dtype`` is float32 (what ``sess.run`` returns).
    """
    input_image = np.asarray(input_image)
    bicubic_image = np.asarray(bicubic_image)
    if max_value != 255.0:
        input_image = np.multiply(input_image, max_value / 255.0)
        bicubic_image = np.multiply(bicubic_image, max_value / 255.0)
    if self_ensemble > 1:
        s = cfg["scale"]
        h, w = input_image.shape[:2]
        output = np.zeros([s * h, s * w, 1])
        for i in range(self_ensemble):
            img = flip(input_image, i)
            bic = flip(bicubic_image, i)
            y = forward(cfg, weights, img[None], bic[None], dtype=dtype)
            output += flip(y[0], i, invert=True)
        output /= self_ensemble
    else:
        output = forward(cfg, weights, input_image[None], bicubic_image[None], dtype=dtype)[0]
    if max_value != 255.0:
        output = np.multiply(output, 255.0 / max_value)
    return output


# --------------------------------------------------------------------------------------------
# evaluation recipe (DCSCN.py:672-703, loader.py:42-67, utilty.py:142-149,196-239,501-536)
# --------------------------------------------------------------------------------------------

def rgb_to_y(image):
    """utilty.py:142-149 (float64, no rounding)."""
    if image.ndim <= 2 or image.shape[2] == 1:
        return image
    xform = np.array([[65.738 / 256.0, 129.057 / 256.0, 25.064 / 256.0]])
    return image.dot(xform.T) + 16.0


def align(image, alignment):
    """utilty.py:196-208."""
    h = (image.shape[0] // alignment) * alignment
    w = (image.shape[1] // alignment) * alignment
    image = image[:h, :w, :]
    if image.shape[2] >= 4:
        image = image[:, :, 0:3]
    return image


def pil_bicubic(image, scale):
    """utilty.py:211-239 for single-channel float images: PIL mode 'F' BICUBIC."""
    from PIL import Image
    h, w = image.shape[:2]
    nw, nh = int(w * scale), int(h * scale)
    im = Image.fromarray(np.asarray(image).reshape(h, w))
    im = im.resize([nw, nh], resample=Image.BICUBIC)
    return np.asarray(im).reshape(nh, nw, 1)


def psnr_y(true_y, out_y, border):
    """utilty.py:501-533: rint, clip to [0, 255], float32, shave ``border`` px, 10 log10(255^2 / mse)."""
    a = np.clip(np.rint(true_y), 0, 255).astype(np.float32)
    b = np.clip(np.rint(out_y), 0, 255).astype(np.float32)
    if border > 0:
        a = a[border:-border, border:-border]
        b = b[border:-border, border:-border]
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10.0 * math.log10(255.0 ** 2 / mse)


def evaluate_image(cfg, weights, rgb_or_gray_u8, self_ensemble=1, dtype=np.float64):
    """``do_for_evaluate`` (DCSCN.py:672-703) on a loaded uint8 [H, W, C] image: returns
    (psnr, lr_y, bicubic_y, output_y)."""
    s = cfg["scale"]
    true_image = align(np.atleast_3d(rgb_or_gray_u8), s)
    true_y = rgb_to_y(true_image) if true_image.shape[2] == 3 else true_image
    lr = pil_bicubic(true_y, 1.0 / s)
    bic = pil_bicubic(lr, s)
    out = do(cfg, weights, lr, bic, self_ensemble=self_ensemble, dtype=dtype)
    return psnr_y(true_y, out, s), lr, bic, out
