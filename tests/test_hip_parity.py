"""Parity of the HIP forward pass (through the C ABI) against the CPU oracle.

Tolerances are the ones BASELINE.json's north_star states: max-abs pixel error <= 1e-4 on the 0-255
scale against the float64 oracle (the reference itself is float32 TensorFlow).
"""
import os

import numpy as np
import pytest

from conftest import CONFIGS, synthetic_batch

pytestmark = pytest.mark.gpu

MAX_ABS_TOL = 1e-4


# The two kernel families of the product: split16 (f16 hi/lo x 3 products on the f16 matrix pipe: conv3_h / conv_nin_h / conv5_h,
# the library default) and pure f32 (conv_wino2 / conv_nin / conv_igemm: split16 = 0, also the fallback of every flagged tile).
# with split16 the default the f32 family had lost most of its coverage; the tests below run both.
SPLIT16 = [True, False]


def _engine(cfg, weights, winograd=None, split16=None):
    from dcscn_amd import engine
    eng = engine.Engine(cfg, device=0)
    eng.load_weights(weights, winograd=winograd, split16=split16)
    return eng


def _check(oracle, name, overrides, n, h, w, seed=0, sub_batch_pixels=None, winograd=None, split16=None):
    cfg = oracle.make_config(**overrides)
    weights = oracle.synthetic_weights(cfg, seed=seed)
    x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=seed + 1)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with _engine(cfg, weights, winograd, split16) as eng:
        if sub_batch_pixels:
            eng.set_option("sub_batch_pixels", sub_batch_pixels)
        y = eng.forward(x, x2)
    assert y.shape == ref.shape and y.dtype == np.float32
    err = float(np.max(np.abs(y.astype(np.float64) - ref)))
    scale = float(np.max(np.abs(ref)))
    print("%s n=%d %dx%d max|y|=%.1f max-abs err %.3g" % (name, n, h, w, scale, err))
    assert np.isfinite(y).all()
    assert err <= MAX_ABS_TOL, "%s: max-abs error %.3g > %.1g" % (name, err, MAX_ABS_TOL)


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("winograd", [True, False])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_config_48x48(oracle, name, winograd, split16):
    """Every BASELINE config (and every shipped-checkpoint topology) on 48x48 patches: 3x3 convs on conv3_h (split16, default),
    on the f32 Winograd kernel (split16 off) and on the direct implicit-GEMM kernel (winograd off), wide 1x1 convs on
    conv_nin_h / conv_nin."""
    n = 2 if "L12" in name or "L8" in name else 3
    _check(oracle, name, CONFIGS[name], n, 48, 48, winograd=winograd, split16=split16)


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("winograd", [True, False])
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_residual_branch_relative_error(oracle, name, winograd, fold, split16):
    """The synthetic weights scale the last conv by 0.01, which would hide upstream errors behind the
    bicubic term.  Here the last conv is NOT scaled and x2 = 0, so y is the bare network branch; its
    error is bounded relative to its own magnitude (f32 accumulation over K <= 1764 terms) -- for EVERY
    BASELINE config and shipped topology, on the Winograd / direct kernels, folded tail and layer by layer."""
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=7)
    last = "R-CNN%d" % cfg["reconstruct_layers"]
    for leaf in ("conv_W", "pointwise_W"):
        if last + "/" + leaf in weights:
            weights[last + "/" + leaf] = weights[last + "/" + leaf] * 100.0
    s = cfg["scale"]
    x, _ = synthetic_batch(2, 48, 48, s, seed=8)
    x2 = np.zeros((2, 48 * s, 48 * s, 1), np.float32)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    from dcscn_amd import engine
    eng = engine.Engine(cfg, device=0)
    eng.load_weights(weights, winograd=winograd, fold_tail=fold, split16=split16)
    y = eng.forward(x, x2)
    eng.close()
    rel = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
    print("%s winograd=%s fold=%s split16=%s residual-branch relative error %.3g (max|y| %.3g)" % (name, winograd, fold, split16, rel, np.max(np.abs(ref))))
    assert rel <= 5e-6


@pytest.mark.parametrize("filters", [36, 40, 44, 52, 56, 68, 76, 100, 120])
def test_conv3_h_packed_last_chunk_channel_counts(oracle, filters):
    """conv3_h walks K in chunks of 32 channels; a last chunk of at most 8 / 16 / 24 physical channels is packed (tap, octet)
    pairs four to an MFMA (kernels.h: c3h_tail_octs).  Constant-width nets put exactly `filters` channels into every 3x3 layer:
    36 / 40 (one octet), 44 / 76 (two: 44 = 32 + 12, 76 = 64 + 12), 52 / 56 / 120 (three), 68 (one, behind two full chunks),
    100 (one, behind three) -- bare network branch against the float64 oracle at the usual 5e-6, and the same bits as a run with
    the plain f32 kernels would give to within that bar; ragged image size so that border tiles and the packed chunk meet."""
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=3, filters=filters, min_filters=filters, nin_filters=32, nin_filters2=16)
    weights = oracle.synthetic_weights(cfg, seed=31)
    last = "R-CNN%d" % cfg["reconstruct_layers"]
    weights[last + "/conv_W"] = weights[last + "/conv_W"] * 100.0
    x, _ = synthetic_batch(2, 37, 50, 2, seed=32)
    x2 = np.zeros((2, 74, 100, 1), np.float32)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        kernels = [op["kernel"] for op in eng.ops() if op["kernel_size"] == 3 and op["in_channels"] == filters]
        assert kernels and all(k in ("conv3_h", "conv3_h8") for k in kernels), eng.ops()    # two channel groups -> conv3_h8, else conv3_h
        y = eng.forward(x, x2)
    rel = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
    print("filters=%d: residual-branch relative error %.3g" % (filters, rel))
    assert np.isfinite(y).all() and rel <= 5e-6


def test_baseline_config3_as_written(oracle):
    """BASELINE.json configs[3]: dcscn_L12_F196to48 x4 WITH self_ensemble = 8 on an image-sized input (the 8 flips / rotations
    of DCSCN.py:559-573, four of them transposed, float64 mean) against oracle.do, with split16 on (library default) and off."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L12_F196to48_x4"])
    weights = oracle.synthetic_weights(cfg, seed=3)
    x, x2 = synthetic_batch(1, 48, 64, 4, seed=21)
    ref = oracle.do(cfg, weights, x[0], x2[0], self_ensemble=8)
    for s16 in (True, False):
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights, split16=s16)
            y = eng.forward_ensemble(x[0], x2[0], 8)
        err = float(np.max(np.abs(y - ref)))
        print("L12 x4 self_ensemble=8 on 48x64 split16=%s: max-abs %.3g (max|y| %.1f)" % (s16, err, np.abs(ref).max()))
        assert y.shape == ref.shape and np.isfinite(y).all() and err <= MAX_ABS_TOL


@pytest.mark.parametrize("split16", [True, False])
@pytest.mark.parametrize("name", ["L8_F96to48_x2", "L12_F196to48_x2", "L7_F32to8_x2"])
def test_split16_option_both_ways(oracle, name, split16):
    """The f16-pipe kernels (library default) and the pure f32 kernels behind the same handle: both meet the bars, the
    option flips the launches (dcscn_op_info) and can be changed after finalize."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=5)
    x, x2 = synthetic_batch(2, 40, 56, cfg["scale"], seed=6)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        eng.set_option("split16", 1 if split16 else 0)
        kernels = {o["kernel"] for o in eng.ops()}
        # (the non-separable narrow net runs CNN1 .. CNNL as one streamed launch on the split16 path, layer by layer otherwise)
        assert ("conv3_h" in kernels or "feat3_stream" in kernels) == split16 and ("conv_wino2" in kernels or "layer by layer" in kernels) == (not split16), kernels
        y = eng.forward(x, x2)
        eng.set_option("split16", 0 if split16 else 1)      # and back the other way on the same handle
        y2 = eng.forward(x, x2)
    for out in (y, y2):
        assert float(np.max(np.abs(out - ref))) <= MAX_ABS_TOL


def test_split16_overflow_falls_back_to_f32(oracle):
    """An activation beyond the f16 range (|x| >= 65520) cannot be split; the kernels flag the affected 16x16 tiles / 256-pixel
    blocks and the f32 launch behind them recomputes exactly those.  Here the input is scaled so that CNN1's outputs reach
    ~1e6: the result must still match the float64 oracle to f32 accuracy, and an ordinary image in the same batch must come
    out bit-identical to a run without the huge one."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L8_F96to48_x2"])
    weights = oracle.synthetic_weights(cfg, seed=9)
    x, x2 = synthetic_batch(2, 32, 48, 2, seed=10)
    xb = x.copy()
    xb[1] *= 4000.0                                            # image 1: activations far beyond 65504
    ref = oracle.forward(cfg, weights, xb, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        y = eng.forward(xb, x2)
        y0 = eng.forward(x[:1], x2[:1])
    assert np.isfinite(y).all()
    rel = float(np.max(np.abs(y[1] - ref[1])) / np.max(np.abs(ref[1])))
    print("overflowing image: relative error %.3g (max|y| %.3g)" % (rel, np.max(np.abs(ref[1]))))
    assert rel <= 5e-6
    assert float(np.max(np.abs(y[0] - ref[0]))) <= MAX_ABS_TOL
    assert np.array_equal(y[0], y0[0])


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("hw", [(1, 1), (5, 7), (16, 16), (17, 33), (31, 9), (50, 20)])
def test_ragged_sizes(oracle, hw, split16):
    """Image sizes that are not multiples of the pixel tiles (odd sizes also cut Winograd's 2x2 output
    tiles), down to a single pixel; channel counts that are not multiples of 4 or 16 -- on both kernel families."""
    _check(oracle, "L7_F32to8_x2", CONFIGS["L7_F32to8_x2"], 2, hw[0], hw[1], split16=split16)
    _check(oracle, "odd-channels", dict(layers=4, filters=37, min_filters=13, nin_filters=21, nin_filters2=10), 1, hw[0], hw[1], split16=split16)
    _check(oracle, "wide-odd", dict(layers=3, filters=70, min_filters=45, nin_filters=40, nin_filters2=33), 1, hw[0], hw[1], split16=split16)


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("hw", [(5, 7), (17, 33), (50, 20)])
def test_ragged_sizes_through_the_shuffler_epilogues(oracle, hw, split16):
    """The two conv3_h store paths r05 added, on sizes that cut the 16 x 16 pixel tiles: per-channel stores for a pixel shuffler to ONE
    channel per sub-pixel (x3 c-DCSCN: Up-PS 32 -> 9 as a one-tile layer, graph.hip h16_direct_eligible) and (hi | lo) units stored with
    depth_to_space addressing (x4 nets: the half-resolution map stays pre-split for conv5_h)."""
    _check(oracle, "L7_F32to8_x3", dict(CONFIGS["L7_F32to8_x2"], scale=3), 2, hw[0], hw[1], split16=split16)
    _check(oracle, "L7_F32to8_x4", dict(CONFIGS["L7_F32to8_x2"], scale=4), 2, hw[0], hw[1], split16=split16)
    _check(oracle, "wide-x4", dict(layers=3, filters=70, min_filters=45, nin_filters=40, nin_filters2=24, scale=4), 1, hw[0], hw[1], split16=split16)


def test_sub_batching_is_transparent(oracle):
    """Splitting the batch into passes must not change a single bit (no cross-image coupling)."""
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x2"])
    weights = oracle.synthetic_weights(cfg, seed=3)
    x, x2 = synthetic_batch(5, 24, 40, 2, seed=4)
    with _engine(cfg, weights) as eng:
        full = eng.forward(x, x2)
        eng.set_option("sub_batch_pixels", 2 * 24 * 40)
        parts = eng.forward(x, x2)
        eng.set_option("sub_batch_pixels", 1)
        single = eng.forward(x, x2)
    assert np.array_equal(full, parts) and np.array_equal(full, single)


@pytest.mark.parametrize("variant", [
    dict(use_nin=False, layers=3, filters=20, min_filters=8),                      # 1x1 "C" layer path
    dict(layers=3, filters=24, min_filters=8, reconstruct_layers=3, reconstruct_filters=12),
    dict(layers=3, filters=16, min_filters=8, scale=3),
    dict(layers=3, filters=16, min_filters=8, scale=3, pixel_shuffler_filters=1),
    dict(layers=3, filters=16, min_filters=8, scale=4, pixel_shuffler_filters=6),
    dict(layers=3, filters=16, min_filters=8, activator="relu"),
    dict(layers=3, filters=16, min_filters=8, activator="leaky_relu"),
    dict(layers=3, filters=16, min_filters=0),
    dict(layers=2, filters=8, min_filters=8, cnn_size=1),
    # --cnn_size=5 / 7: every conv of the graph (feature stack, B2, pixel shuffler, reconstruction) is k x k
    dict(layers=3, filters=20, min_filters=8, cnn_size=5),
    dict(layers=2, filters=40, min_filters=24, cnn_size=7, nin_filters=20, nin_filters2=12),
    dict(layers=3, filters=16, min_filters=8, cnn_size=5, scale=3, reconstruct_layers=2, reconstruct_filters=8),
    dict(layers=2, filters=12, min_filters=8, cnn_size=5, depthwise_separable=True),
    dict(layers=2, filters=12, min_filters=8, cnn_size=7, depthwise_separable=True, scale=4),
    dict(layers=2, filters=12, min_filters=8, cnn_size=5, pixel_shuffler=False),
    dict(layers=3, filters=16, min_filters=8, depthwise_separable=True, scale=2),
    dict(layers=3, filters=16, min_filters=8, depthwise_separable=True, use_nin=False, scale=3),
    # transposed-conv upsampler (tf_graph.py:219-236), k = 4 / 5 / 8
    dict(layers=3, filters=16, min_filters=8, pixel_shuffler=False),
    dict(layers=3, filters=16, min_filters=8, pixel_shuffler=False, scale=3, nin_filters=9, nin_filters2=5),
    dict(layers=2, filters=12, min_filters=8, pixel_shuffler=False, scale=4, reconstruct_layers=2, reconstruct_filters=8),
    dict(layers=2, filters=40, min_filters=36, pixel_shuffler=False, use_nin=False),
    # transposed conv (never separable, DCSCN.py:311) in a depthwise-separable graph: the R-CNN behind it IS separable (DCSCN.py:318-320)
    dict(layers=3, filters=16, min_filters=8, pixel_shuffler=False, depthwise_separable=True),
    dict(layers=2, filters=12, min_filters=8, pixel_shuffler=False, depthwise_separable=True, scale=4, reconstruct_layers=2, reconstruct_filters=8),
    dict(layers=2, filters=12, min_filters=8, pixel_shuffler=False, depthwise_separable=True, scale=3, use_nin=False),
])
def test_flag_surface(oracle, variant):
    _check(oracle, str(variant), variant, 2, 20, 28)


@pytest.mark.parametrize("act", ["sigmoid", "tanh", "selu"])
def test_transcendental_activators(oracle, act):
    # device expf/tanhf are not correctly rounded: looser bound, still far below one grey level
    cfg = oracle.make_config(layers=3, filters=16, min_filters=8, activator=act)
    weights = oracle.synthetic_weights(cfg, seed=0)
    x, x2 = synthetic_batch(2, 20, 28, 2, seed=1)
    x = x / 255.0
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with _engine(cfg, weights) as eng:
        y = eng.forward(x, x2)
    assert float(np.max(np.abs(y - ref))) <= 1e-3


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("n_ens", [1, 2, 5, 8])
def test_self_ensemble(oracle, n_ens, split16):
    """do() with self_ensemble (DCSCN.py:559-573): batched on the device, float64 mean in reference order."""
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x2"])
    weights = oracle.synthetic_weights(cfg, seed=5)
    x, x2 = synthetic_batch(1, 18, 30, 2, seed=6)
    ref = oracle.do(cfg, weights, x[0], x2[0], self_ensemble=n_ens, dtype=np.float64)
    with _engine(cfg, weights, split16=split16) as eng:
        y = eng.forward_ensemble(x[0], x2[0], n_ens)
    assert y.dtype == np.float64 and y.shape == ref.shape
    assert float(np.max(np.abs(y - ref))) <= MAX_ABS_TOL


@pytest.mark.parametrize("name", ["L8_F96to48_x2", "L12_F196to48_x2", "L7_F32to8_x2"])
def test_small_magnitude_inputs_on_split16(oracle, name):
    """Inputs in [0, 1] (--max_value=1 feeds the network image * 1/255: DCSCN.py:555-557): the split16 path splits activations
    UNSCALED, so small values lean on f16 subnormals for their `lo` pieces (absolute floor ~3e-8 per element, split16.hpp).  Bare
    network branch (last conv not attenuated, x2 = 0) against the float64 oracle at the same 5e-6 relative bar as on 0..255 data."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=7)
    last = "R-CNN%d" % cfg["reconstruct_layers"]
    weights[last + "/conv_W"] = weights[last + "/conv_W"] * 100.0
    # the synthetic biases (~N(0, 0.1)) would dominate a [0, 1] input: scale them with the data so that the activations really are small
    for k in list(weights):
        if k.endswith("conv_B"):
            weights[k] = weights[k] / 255.0
    x, _ = synthetic_batch(2, 48, 48, 2, seed=8)
    x = (x / 255.0).astype(np.float32)
    x2 = np.zeros((2, 96, 96, 1), np.float32)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    rels = {}
    for s16 in (True, False):
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights, split16=s16)
            y = eng.forward(x, x2)
        rels[s16] = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
    print("%s inputs in [0, 1]: residual-branch relative error split16 %.3g, f32 kernels %.3g (max|y| %.3g)" % (name, rels[True], rels[False], np.max(np.abs(ref))))
    assert rels[True] <= 5e-6 and rels[False] <= 5e-6


@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("max_value", [1.0, 100.0])
def test_do_with_max_value(oracle, tmp_path, max_value, split16, monkeypatch):
    """SuperResolution.do() with --max_value != 255 (DCSCN.py:555-557, 581-584): image and bicubic image are multiplied by
    max_value / 255 before the network and the output by 255 / max_value after it; self_ensemble 1 and 8."""
    from dcscn_amd.model import SuperResolution
    from test_host import _flags
    monkeypatch.setenv("DCSCN_SPLIT16", "1" if split16 else "0")
    over = dict(layers=4, filters=40, min_filters=24, nin_filters=32, nin_filters2=16)
    cfg = oracle.make_config(**over)
    weights = oracle.synthetic_weights(cfg, seed=11)
    x, x2 = synthetic_batch(1, 30, 44, 2, seed=12)
    for n_ens in (1, 8):
        m = SuperResolution(_flags(max_value=max_value, self_ensemble=n_ens, checkpoint_dir=str(tmp_path / "models"), **over))
        m.build_graph()
        m.init_all_variables()
        m.load_weights(weights)
        y = m.do(x[0], x2[0])
        m.close()
        ref = oracle.do(cfg, weights, x[0], x2[0], self_ensemble=n_ens, max_value=max_value)
        err = float(np.max(np.abs(np.asarray(y, np.float64) - ref)))
        print("do() max_value=%g self_ensemble=%d split16=%s: max-abs %.3g" % (max_value, n_ens, split16, err))
        assert y.shape == ref.shape and err <= MAX_ABS_TOL


@pytest.mark.parametrize("name", ["L12_F196to48_x2", "L12_F196to48_x4", "wide-3"])
def test_conv3_h8_is_bit_identical_to_conv3_h(oracle, name):
    """Layers with two channel groups run on conv3_h8 (one persistent 8-wave workgroup per CU, the pixel tile's input staged once for
    both groups, the halves running load and compute in opposite order); the option conv3_h8 = 0 sends them to conv3_h.  Same arithmetic in the same order: same bits,
    on ragged sizes (image-edge tiles take conv3_h8's general epilogue) and with more items than workgroups."""
    from dcscn_amd import engine
    flags = CONFIGS[name] if name in CONFIGS else dict(layers=3, filters=176, min_filters=112, filters_decay_gamma=1.0, nin_filters=48, nin_filters2=24)
    cfg = oracle.make_config(**flags)
    weights = oracle.synthetic_weights(cfg, seed=4)
    for n, h, w in ((3, 48, 48), (2, 37, 50), (1, 130, 70)):
        x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=5)
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights)
            on8 = [op["name"] for op in eng.ops() if op["kernel"] == "conv3_h8"]
            assert on8, eng.ops()
            y8 = eng.forward(x, x2)
            eng.set_option("conv3_h8", 0)
            assert not [op for op in eng.ops() if op["kernel"] == "conv3_h8"]
            y4 = eng.forward(x, x2)
        assert np.isfinite(y8).all() and np.array_equal(y8, y4), (name, n, h, w)


def test_debug_poison_changes_nothing(oracle):
    """No kernel may depend on what LDS or the vector registers held before it: with the debug_poison option every launch is
    preceded by kernels that fill all LDS and all VGPRs with NaN patterns -- the output must be bit-identical (both kernel
    families; the narrow streamed nets too)."""
    from dcscn_amd import engine
    for name in ("L12_F196to48_x2", "L7_F32to8_x4_DS", "L7_F32to8_x2"):
        cfg = oracle.make_config(**CONFIGS[name])
        weights = oracle.synthetic_weights(cfg, seed=2)
        x, x2 = synthetic_batch(2, 33, 48, cfg["scale"], seed=3)
        for s16 in (True, False):
            with engine.Engine(cfg, device=0) as eng:
                eng.load_weights(weights, split16=s16)
                y0 = eng.forward(x, x2)
                eng.set_option("debug_poison", 3)
                y1 = eng.forward(x, x2)
            assert np.isfinite(y0).all() and np.array_equal(y0, y1), (name, s16)


def test_errors_are_reported_not_fatal(oracle):
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L2_F4to4_x2"])
    weights = oracle.synthetic_weights(cfg)
    eng = engine.Engine(cfg)
    x, x2 = synthetic_batch(1, 8, 8, 2)
    with pytest.raises(engine.EngineError) as e:      # forward before finalize
        eng.forward(x, x2)
    assert e.value.status == 6
    with pytest.raises(engine.EngineError) as e:      # unknown variable
        eng.set_tensor("nope/conv_W", np.zeros((3, 3, 1, 4), np.float32))
    assert e.value.status == 4
    with pytest.raises(engine.EngineError) as e:      # wrong shape
        eng.set_tensor("CNN1/conv_W", np.zeros((3, 3, 1, 5), np.float32))
    assert e.value.status == 4
    with pytest.raises(engine.EngineError) as e:      # finalize with variables missing
        eng.finalize()
    assert e.value.status == 3
    eng.load_weights(weights)
    assert eng.forward(x, x2).shape == (1, 16, 16, 1)
    eng.close()
    with pytest.raises(engine.EngineError) as e:
        engine.Engine(dict(scale=5))
    assert e.value.status == 2


def test_full_size_bench_workload(oracle):
    """BASELINE.json configs[2] at its full size (L12_F196to48 x2, 1024 patches of 48x48): the oracle needs
    ~4 s per patch pair, so 6 sampled patches are checked against it directly and the rest through
    size-independent properties -- every patch depends on its own input only (a batch of 1024 equals the
    same patches run in other batch compositions, bit for bit), runs are deterministic, outputs finite."""
    cfg = oracle.make_config()
    weights = oracle.synthetic_weights(cfg, seed=0)
    n = 1024
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
    x2 = rng.uniform(0, 255, (n, 96, 96, 1)).astype(np.float32)
    with _engine(cfg, weights) as eng:
        y = eng.forward(x, x2)
        assert np.isfinite(y).all()
        assert np.array_equal(y, eng.forward(x, x2))                       # deterministic
        eng.set_option("sub_batch_pixels", 100 * 48 * 48)                   # 11 ragged passes (100 x 10 + 24)
        assert np.array_equal(y, eng.forward(x, x2))
        perm = rng.permutation(n)[:200]
        assert np.array_equal(y[perm], eng.forward(x[perm], x2[perm]))     # other batch composition
    pick = [0, 1, 511, 512, 1000, 1023]
    ref = oracle.forward(cfg, weights, x[pick], x2[pick], dtype=np.float64)
    assert float(np.max(np.abs(y[pick] - ref))) <= MAX_ABS_TOL
    # r06 ("the headline config's border / interior mix is pinned on eight patches"): 96 more patches -- every 11th of the
    # batch, and its last ones -- against the float64 torch-CPU restatement of the same graph (oracle/cpu_path_torch.py, pinned to the
    # numpy oracle at 1e-9 on these very weights below; oneDNN runs it in seconds where the numpy loops need 4 s per patch)
    import torch
    import cpu_path_torch as T
    model = T.TorchCpuModel(cfg, weights, dtype=torch.float64)
    assert float(np.max(np.abs(model.forward(x[pick], x2[pick]) - ref))) <= 1e-9
    more = sorted(set(range(0, n, 11)) | {n - 3, n - 2, n - 1})
    ref64 = model.forward(x[more], x2[more])
    err = np.max(np.abs(y[more] - ref64), axis=(1, 2, 3))
    print("L12 x2 full size: %d patches against float64, worst %.3g, mean of the maxima %.3g" % (len(more), err.max(), err.mean()))
    assert float(err.max()) <= MAX_ABS_TOL


def test_full_size_c2_workload(oracle):
    """BASELINE.json configs[1] at its full size (L8_F96to48 x2, batch = 256 patches of 48x48): EVERY patch against the float64
    torch-CPU restatement (pinned to the numpy oracle on three of them), on both kernel families."""
    import torch
    import cpu_path_torch as T
    cfg = oracle.make_config(**CONFIGS["L8_F96to48_x2"])
    weights = oracle.synthetic_weights(cfg, seed=0)
    n = 256
    rng = np.random.default_rng(12)
    x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
    x2 = rng.uniform(0, 255, (n, 96, 96, 1)).astype(np.float32)
    model = T.TorchCpuModel(cfg, weights, dtype=torch.float64)
    ref = model.forward(x, x2)
    pick = [0, 100, 255]
    assert float(np.max(np.abs(oracle.forward(cfg, weights, x[pick], x2[pick], dtype=np.float64) - ref[pick]))) <= 1e-9
    for split16 in (None, False):
        with _engine(cfg, weights, split16=split16) as eng:
            y = eng.forward(x, x2)
        err = float(np.max(np.abs(y - ref)))
        print("L8 x2, 256 patches, split16 %s: max-abs %.3g" % (split16, err))
        assert err <= MAX_ABS_TOL


@pytest.mark.parametrize("name,n", [("L7_F32to8_x4_DS", 1024), ("L7_F32to8_x4", 1024), ("L12_F196to48_x4", 64)])
def test_full_size_x4_workloads(oracle, name, n):
    """BASELINE.json configs[4] at its full size (L7_F32to8 x4 depthwise separable, 1024 patches of 48x48), the c-DCSCN x4 net at the same size
    and the L12 x4 graph of configs[3] on 64 patches: EVERY patch against the float64 torch-CPU restatement (pinned to the numpy oracle on
    three of them) -- through the whole-tail fold (interior launch + 256 corner jobs of 16 images at n = 1024) and, fold_whole_tail = 0,
    through the r05 plans."""
    import torch
    import cpu_path_torch as T
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=0)
    rng = np.random.default_rng(13)
    x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
    x2 = rng.uniform(0, 255, (n, 192, 192, 1)).astype(np.float32)
    model = T.TorchCpuModel(cfg, weights, dtype=torch.float64)
    ref = np.concatenate([model.forward(x[i:i + 128], x2[i:i + 128]) for i in range(0, n, 128)])
    pick = [0, n // 3, n - 1]
    assert float(np.max(np.abs(oracle.forward(cfg, weights, x[pick], x2[pick], dtype=np.float64) - ref[pick]))) <= 1e-9
    for whole in (1, 0):
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("fold_whole_tail", whole)
            eng.load_weights(weights)
            assert (eng.ops()[-1]["name"] == "Up-PS..R-CNN1 (folded)") == (whole == 1)
            y = eng.forward(x, x2)
        err = np.max(np.abs(y - ref), axis=(1, 2, 3))
        print("%s, %d patches, fold_whole_tail %d: worst patch %.3g, mean of the maxima %.3g" % (name, n, whole, err.max(), err.mean()))
        assert float(err.max()) <= MAX_ABS_TOL


# ---- opt-in graph rewrite: the linear tail as one 5x5 conv (include/dcscn.h "fold_linear_tail") ----------

def _fold_engine(cfg, weights, fold=True, split16=None):
    from dcscn_amd import engine
    eng = engine.Engine(cfg, device=0)
    eng.load_weights(weights, fold_tail=fold, split16=split16)
    return eng


def _folded(eng):
    return [op["name"] for op in eng.ops() if "(folded)" in op["name"]]


@pytest.mark.parametrize("name", ["L2_F4to4_x2", "L8_F96to48_x2", "L12_F196to48_x2", "L12_F196to48_x4",
                                  "L7_F32to8_x2", "L7_F32to8_x3", "L7_F32to8_x4"])
def test_folded_tail_configs(oracle, name):
    """Pixel-shuffler conv + depth_to_space + R-CNN as ONE conv: same 1e-4 bar against the float64 oracle of
    the layer-by-layer graph; the rewrite must actually have happened (one launch named '... (folded)')."""
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=11)
    n = 2
    x, x2 = synthetic_batch(n, 48, 48, cfg["scale"], seed=12)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with _fold_engine(cfg, weights) as eng:
        assert len(_folded(eng)) == 1, eng.ops()
        y = eng.forward(x, x2)
    err = float(np.max(np.abs(y.astype(np.float64) - ref)))
    print("%s folded tail: max-abs err %.3g" % (name, err))
    assert err <= MAX_ABS_TOL


@pytest.mark.parametrize("scale,whole", [(2, 1), (3, 1), (4, 1), (3, 0), (4, 0)])
@pytest.mark.parametrize("hw", [(1, 1), (1, 6), (7, 1), (2, 2), (3, 5), (16, 16), (17, 33)])
def test_folded_tail_borders(oracle, scale, whole, hw):
    """The composite kernel changes on the border rows / columns of each sub-pixel phase (the reconstruction conv
    zero-pads the HR map).  Bare network branch (x2 = 0, last conv not scaled down) so that a wrong border
    variant cannot hide behind the bicubic term; images down to one pixel, where every pixel is a border.
    x3 / x4, whole = 1 (the default): the WHOLE tail is one launch of the interior kernels + the border ring's launch (fold_whole_tail);
    whole = 0: the r05 fold of the last stage, every border variant as extra channels (conv5_h<3> / <1>)."""
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=3, filters=16, min_filters=8, scale=scale, pixel_shuffler_filters=5)
    weights = oracle.synthetic_weights(cfg, seed=20 + scale)
    weights["R-CNN1/conv_W"] = weights["R-CNN1/conv_W"] * 100.0
    x, _ = synthetic_batch(2, hw[0], hw[1], scale, seed=30)
    x2 = np.zeros((2, hw[0] * scale, hw[1] * scale, 1), np.float32)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)

    def fold_engine():
        eng = engine.Engine(cfg, device=0)
        eng.set_option("fold_whole_tail", whole)
        eng.load_weights(weights, fold_tail=True)
        return eng
    whole_tail = scale > 2 and whole == 1
    with fold_engine() as eng:
        assert _folded(eng) == (["Up-PS..R-CNN1 (folded)"] if whole_tail else [("Up-PS2/Up-PS2_CNN" if scale == 4 else "Up-PS/Up-PS_CNN") + "+R-CNN1 (folded)"])
        # split16 (default): the composite runs on conv5_h (one channel tile; three for the r05 fold at x3)
        assert [op["kernel"] for op in eng.ops() if "(folded)" in op["name"]] == ["conv5_h"]
        y = eng.forward(x, x2)
    with fold_engine() as eng:
        eng.set_option("split16", 0)
        # (the whole-tail fold has no float32 kernel of its own: the launches it replaces run -- also the float32 plan of a flagged image)
        assert [op["kernel"] for op in eng.ops() if "(folded)" in op["name"]] == ["layer by layer" if whole_tail else "conv_igemm"]
        y32 = eng.forward(x, x2)
    assert float(np.max(np.abs(y32 - ref))) / float(np.max(np.abs(ref))) <= 1e-5
    with _fold_engine(cfg, weights, fold=False) as eng:
        assert not _folded(eng)
        y0 = eng.forward(x, x2)
    mag = float(np.max(np.abs(ref)))
    rel = float(np.max(np.abs(y - ref))) / mag
    rel0 = float(np.max(np.abs(y0 - ref))) / mag
    print("x%d %dx%d folded rel err %.3g (layer by layer %.3g, max|y| %.3g)" % (scale, hw[0], hw[1], rel, rel0, mag))
    assert rel <= 1e-5


@pytest.mark.parametrize("flags", [
    dict(scale=3, pixel_shuffler_filters=5),
    dict(scale=4, pixel_shuffler_filters=1),
    dict(scale=4, pixel_shuffler_filters=0),                                             # shuffler to all 8 channels: R-CNN1 sums over them
    dict(scale=4, pixel_shuffler_filters=5, depthwise_separable=True),                   # a separable conv is a dense one: dw[t][c] * pw[c][co]
    dict(scale=3, pixel_shuffler_filters=1, depthwise_separable=True),
])
def test_whole_tail_fold(oracle, flags):
    """r06: at x3 / x4 every pixel-shuffler stage is built without an activator (tf_graph.py:238-249), so Up-PS (, Up-PS2) and R-CNN1 are ONE
    5x5 conv of the LR map to scale^2 phases -- the interior on conv5_h, the image's border ring (15 position classes with kernels of their
    own: the reference zero-pads the SHUFFLED maps) in fold_border.  Bare branch against the float64 oracle on shapes that exercise every job
    kind: many small images (corner batches of 16), one-pixel axes (the `both` classes), two-pixel axes (no interior), segments past 16 and
    ragged ends; option fold_whole_tail = 0 (the r05 plan, which is also the float32 plan behind the fold) meets the same bar."""
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=3, filters=16, min_filters=8, **flags)
    weights = oracle.synthetic_weights(cfg, seed=41)
    for k in list(weights):
        if k.startswith("R-CNN1/") and k.endswith(("conv_W", "pointwise_W")):
            weights[k] = weights[k] * 100.0
    s = cfg["scale"]
    for n, h, w in ((35, 5, 7), (1, 1, 1), (3, 1, 9), (2, 8, 1), (2, 2, 2), (1, 2, 37), (1, 40, 3), (2, 33, 50), (1, 18, 18)):
        x, _ = synthetic_batch(n, h, w, s, seed=42 + h)
        x2 = np.zeros((n, h * s, w * s, 1), np.float32)
        ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
        mag = float(np.max(np.abs(ref)))
        rel = {}
        for whole in (1, 0):
            with engine.Engine(cfg, device=0) as eng:
                eng.set_option("fold_whole_tail", whole)
                eng.load_weights(weights)
                folded = [(o["name"], o["kernel"]) for o in eng.ops() if "(folded)" in o["name"]]
                assert (("Up-PS..R-CNN1 (folded)", "conv5_h") in folded) == (whole == 1), eng.ops()
                y = eng.forward(x, x2)
                assert np.array_equal(y, eng.forward(x, x2))
            rel[whole] = float(np.max(np.abs(y - ref))) / mag
        print("%s %dx%dx%d: whole-tail fold rel %.3g, r05 plan %.3g" % (flags, n, h, w, rel[1], rel[0]))
        assert rel[1] <= 5e-6 and rel[0] <= 5e-6


@pytest.mark.parametrize("name", ["L7_F32to8_x4", "L7_F32to8_x4_DS", "L7_F32to8_x3"])
def test_whole_tail_fold_flagged_image_takes_the_float32_plan(oracle, name):
    """The folded tail has no float32 kernel of its own: for a flagged image (values beyond the f16 range) the launches it replaces run, gated,
    behind the pass -- bit-identical to a split16 = 0 run of that image; the bystanders keep their bits; the next forward is clean."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=9)
    x, x2 = synthetic_batch(3, 21, 50, cfg["scale"], seed=10)
    xb = x.copy()
    xb[1] *= 4000.0
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        assert [o["kernel"] for o in eng.ops()][-1] == "conv5_h" and eng.ops()[-1]["name"] == "Up-PS..R-CNN1 (folded)"
        clean = eng.forward(x, x2)
        y = eng.forward(xb, x2)
        again = eng.forward(x, x2)
        eng.set_option("split16", 0)
        assert [o["kernel"] for o in eng.ops()][-1] == "layer by layer"
        y32 = eng.forward(xb, x2)
    assert np.isfinite(y).all()
    assert np.array_equal(y[0], clean[0]) and np.array_equal(y[2], clean[2])
    assert np.array_equal(y[1], y32[1])
    assert np.array_equal(again, clean)


@pytest.mark.parametrize("variant", [
    dict(layers=3, filters=16, min_filters=8, depthwise_separable=True),                 # separable convs
    dict(layers=3, filters=16, min_filters=8, pixel_shuffler=False),                     # transposed-conv upsampler
    dict(layers=3, filters=24, min_filters=8, reconstruct_layers=3, reconstruct_filters=12),   # activators in the tail
    dict(layers=2, filters=8, min_filters=8, cnn_size=1),
])
def test_folded_tail_not_applicable(oracle, variant):
    """Where the tail is not [PS conv, depth_to_space, one 3x3 conv] the option is ignored, not an error."""
    cfg = oracle.make_config(**variant)
    weights = oracle.synthetic_weights(cfg, seed=5)
    x, x2 = synthetic_batch(1, 12, 20, cfg["scale"], seed=6)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with _fold_engine(cfg, weights) as eng:
        assert not _folded(eng)
        y = eng.forward(x, x2)
    assert float(np.max(np.abs(y - ref))) <= MAX_ABS_TOL


# ---- images larger than the workspace budget: haloed spatial windows (api.hip run_tiled) ----------------

@pytest.mark.parametrize("split16", SPLIT16)
@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("case", [
    (dict(layers=3, filters=16, min_filters=8), 1, 70, 90),                 # x2, halo 3+1+1+1
    (dict(layers=3, filters=40, min_filters=32, nin_filters=32, nin_filters2=16), 1, 60, 50),   # wide enough for conv3_h / conv_wino2
    (dict(layers=3, filters=16, min_filters=8, scale=3), 2, 45, 61),        # two images, x3
    (dict(layers=4, filters=12, min_filters=8, scale=4, pixel_shuffler_filters=4), 1, 64, 40),   # two PS stages
    (dict(layers=3, filters=16, min_filters=8), 1, 200, 20),                # tiled along one axis only
])
def test_spatial_tiling(oracle, case, fold, split16):
    """With a workspace budget smaller than one image the library cuts windows with a receptive-field halo, runs
    them as a batch and stitches; the result must meet the same bar against the oracle of the WHOLE image (SAME
    padding only acts at true image borders) and agree with the untiled pass to rounding."""
    flags, n, h, w = case
    cfg = oracle.make_config(**flags)
    weights = oracle.synthetic_weights(cfg, seed=2)
    x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=3)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with _fold_engine(cfg, weights, fold=fold, split16=split16) as eng:
        whole = eng.forward(x, x2)
        per_px = eng.workspace_bytes() // (n * h * w) + 1
        eng.set_option("workspace_budget_bytes", per_px * 30 * 34)          # windows of about 30 x 34 LR pixels
        tiled = eng.forward(x, x2)
        eng.set_option("spatial_tiling", 0)
        untiled_again = eng.forward(x, x2)                                   # tiling off: one big pass as before
    assert float(np.max(np.abs(tiled - ref))) <= MAX_ABS_TOL
    assert float(np.max(np.abs(tiled - whole))) <= 6.2e-5      # two f32 ulps at 256..512: different Winograd tile phase
    assert np.array_equal(untiled_again, whole)


def test_spatial_tiling_window_too_small(oracle):
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=3, filters=16, min_filters=8)
    weights = oracle.synthetic_weights(cfg, seed=2)
    x, x2 = synthetic_batch(1, 40, 40, 2, seed=3)
    with _fold_engine(cfg, weights, fold=False) as eng:
        eng.forward(x, x2)
        eng.set_option("workspace_budget_bytes", 1)
        with pytest.raises(engine.EngineError):
            eng.forward(x, x2)


class _Hip:
    """The few HIP runtime calls the device-pointer test needs, through ctypes -- resolved through libdcscn_hip.so's own handle, i.e. in the
    runtime the library is linked to.  (dlopen("libamdhip64.so") is NOT that runtime once torch has been imported in the process -- the
    full-size parity tests above do, for the float64 restatement -- but the copy bundled with the torch wheel: a second, uninitialised HIP
    runtime whose first call fails with hipErrorNoDevice.  r06: the serial suite failed exactly so.)"""

    def __init__(self):
        import ctypes
        from dcscn_amd import engine
        self.c = ctypes
        self.lib = ctypes.CDLL(engine.library_path())          # dlsym on this handle searches the library's dependencies: ITS libamdhip64
        self.lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.hipFree.argtypes = [ctypes.c_void_p]
        self.lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.lib.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.lib.hipStreamDestroy.argtypes = [ctypes.c_void_p]
        self.ptrs, self.streams = [], []

    def _ok(self, rc):
        assert rc == 0, "HIP error %d" % rc

    def upload(self, a):
        p = self.c.c_void_p()
        self._ok(self.lib.hipMalloc(self.c.byref(p), a.nbytes))
        self._ok(self.lib.hipMemcpy(p, a.ctypes.data_as(self.c.c_void_p), a.nbytes, 1))      # hipMemcpyHostToDevice
        self.ptrs.append(p)
        return p.value

    def write(self, ptr, a):
        self._ok(self.lib.hipMemcpy(self.c.c_void_p(ptr), a.ctypes.data_as(self.c.c_void_p), a.nbytes, 1))

    def alloc(self, nbytes):
        p = self.c.c_void_p()
        self._ok(self.lib.hipMalloc(self.c.byref(p), nbytes))
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape):
        out = np.empty(shape, np.float32)
        self._ok(self.lib.hipMemcpy(out.ctypes.data_as(self.c.c_void_p), self.c.c_void_p(ptr), out.nbytes, 2))   # DeviceToHost
        return out

    def stream(self):
        s = self.c.c_void_p()
        self._ok(self.lib.hipStreamCreate(self.c.byref(s)))
        self.streams.append(s)
        return s.value

    def close(self):
        for s in self.streams:
            self.lib.hipStreamDestroy(s)
        for p in self.ptrs:
            self.lib.hipFree(p)


@pytest.mark.parametrize("name", ["L8_F96to48_x2", "L7_F32to8_x2", "L7_F32to8_x4_DS"])
def test_graph_replay_gives_the_same_bits(oracle, name):
    """Option graph_replay: a forward whose arguments repeat is captured into a hipGraph on its second
    occurrence and replayed afterwards.  New data in the SAME buffers must give exactly what the plain launch sequence gives; a
    different batch size drops back to plain launches and re-captures; switching the option off destroys the graph."""
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=3)
    s = cfg["scale"]
    batches = [synthetic_batch(4, 40, 56, s, seed=60 + i) for i in range(5)]
    hip = _Hip()
    try:
        with _engine(cfg, weights) as eng:
            expect = [eng.forward(x, x2) for x, x2 in batches]
            dx, dx2 = hip.upload(batches[0][0]), hip.upload(batches[0][1])
            dy = hip.alloc(expect[0].nbytes)
            st = hip.stream()
            eng.set_option("graph_replay", 1)
            for i, (x, x2) in enumerate(batches):                    # call 0 plain, call 1 captures, calls 2.. replay
                hip.write(dx, x)
                hip.write(dx2, x2)
                eng.forward_device(dx, dx2, dy, 4, 40, 56, stream=st)
                eng.synchronize()
                assert np.array_equal(hip.download(dy, expect[i].shape), expect[i]), "graph replay, call %d" % i
            for i in (0, 1, 2):                                       # another batch size through the same buffers: re-capture
                hip.write(dx, batches[i][0])
                hip.write(dx2, batches[i][1])
                eng.forward_device(dx, dx2, dy, 2, 40, 56, stream=st)
                eng.synchronize()
                assert np.array_equal(hip.download(dy, expect[i][:2].shape), expect[i][:2])
            eng.set_option("graph_replay", 0)
            eng.forward_device(dx, dx2, dy, 4, 40, 56, stream=st)
            eng.synchronize()
            assert np.array_equal(hip.download(dy, expect[2].shape), expect[2])
    finally:
        hip.close()


def test_forward_device_alternating_shapes_and_streams(oracle):
    """two shapes alternate through dcscn_forward_device on user streams with no sync in between.  The
    second shape fits the arena of the first, so the re-carve neither frees nor synchronises; the library must
    order its clear (and the forward) behind the previous call -- every output must still match a fresh run."""
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x2"])
    weights = oracle.synthetic_weights(cfg, seed=3)
    shapes = [(6, 48, 40), (3, 24, 56)]
    data = [synthetic_batch(n, h, w, 2, seed=20 + i) for i, (n, h, w) in enumerate(shapes)]
    hip = _Hip()
    try:
        with _engine(cfg, weights) as eng:
            expect = [eng.forward(x, x2) for x, x2 in data]
            dev = [(hip.upload(x), hip.upload(x2)) for x, x2 in data]
            streams = [hip.stream(), hip.stream()]
            outs = []
            for it in range(8):
                k = it & 1
                n, h, w = shapes[k]
                y = hip.alloc(expect[k].nbytes)
                eng.forward_device(dev[k][0], dev[k][1], y, n, h, w, stream=streams[(it // 2) & 1])
                outs.append((k, y))
            eng.synchronize()
            for k, y in outs:
                assert np.array_equal(hip.download(y, expect[k].shape), expect[k]), "output of an un-synchronised call differs"
            assert eng.stream() != 0
    finally:
        hip.close()


def test_forward_device_alternating_streams_on_the_tiled_path(oracle):
    """the same, through spatial tiling.  With a workspace budget of about 30 x 34 LR pixels both shapes are
    cut into windows that go through the SHARED tile buffers; the second shape needs larger ones (realloc while the
    first call may still be running on the other stream).  Every output must equal the synchronised tiled run."""
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x2"])
    weights = oracle.synthetic_weights(cfg, seed=3)
    shapes = [(1, 64, 50), (2, 90, 77)]
    data = [synthetic_batch(n, h, w, 2, seed=40 + i) for i, (n, h, w) in enumerate(shapes)]
    hip = _Hip()
    try:
        with _engine(cfg, weights) as eng:
            eng.forward(*data[0])
            per_px = eng.workspace_bytes() // (64 * 50) + 1
            eng.set_option("workspace_budget_bytes", per_px * 30 * 34)
            expect = [eng.forward(x, x2) for x, x2 in data]                 # tiled, one at a time
            dev = [(hip.upload(x), hip.upload(x2)) for x, x2 in data]
            streams = [hip.stream(), hip.stream()]
            outs = []
            for it in range(8):
                k = it & 1
                n, h, w = shapes[k]
                y = hip.alloc(expect[k].nbytes)
                eng.forward_device(dev[k][0], dev[k][1], y, n, h, w, stream=streams[(it // 2) & 1])
                outs.append((k, y))
            eng.synchronize()
            for k, y in outs:
                assert np.array_equal(hip.download(y, expect[k].shape), expect[k]), "tiled output of an un-synchronised call differs"
    finally:
        hip.close()


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_dense_feature_buffers_are_bit_identical_to_the_concat_tensor(oracle, name):
    """dense_features (default 1): one dense buffer per feature layer and a multi-source NIN GEMM instead of one wide
    concat tensor.  The virtual channel order, the chunking and every accumulation order are the same, so the
    output must be the same BITS -- and the pass must really be taken where it applies (kernel list)."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=3)
    x, x2 = synthetic_batch(3, 40, 36, cfg["scale"], seed=4)
    outs = []
    for dense in (1, 0):
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("dense_features", dense)
            eng.set_option("stream_dense", 0)                 # (the streamed CNN1 .. CNNL launch of the narrow nets exists on dense buffers only)
            eng.load_weights(weights)
            outs.append(eng.forward(x, x2))
    assert outs[0].tobytes() == outs[1].tobytes()


@pytest.mark.parametrize("shape", [(2, 48, 48), (1, 61, 130), (1, 300, 70), (3, 20, 97)])
def test_streamed_separable_net_matches_layer_by_layer(oracle, shape):
    """feat_stream / tail_stream (options stream_features / stream_tail, default on) against the layer-by-layer launches of
    the same library on the separable x4 net, with the last conv NOT attenuated and x2 = 0 (bare network branch): single
    strip, column strips with halos (w > 48), row blocks with halos (tall single image), ragged widths.  Both paths are
    f32; they differ by accumulation order only."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x4_DS"])
    weights = oracle.synthetic_weights(cfg, seed=11)
    last = "R-CNN%d" % cfg["reconstruct_layers"]              # the layer synthetic_weights scales by 0.01 (none when 0)
    for k in list(weights):
        if k.startswith(last + "/") and k.endswith(("conv_W", "pointwise_W")):
            weights[k] = weights[k] * 100.0
    n, h, w = shape
    x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=h + w)
    x2 = np.zeros_like(x2)
    outs = {}
    # mode 2 = the default plan (r06): feat_stream + the whole tail folded into one 5x5 conv (fold_whole_tail); 1 = the r05 plan, feat_stream +
    # tail_stream (also the float32 plan behind mode 2); 0 = every layer a launch
    for mode in (2, 1, 0):
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("stream_features", 1 if mode else 0)
            eng.set_option("stream_tail", 1 if mode else 0)
            eng.set_option("fold_whole_tail", 1 if mode == 2 else 0)
            eng.load_weights(weights)
            kernels = [o["kernel"] for o in eng.ops()]
            assert (kernels == ["feat_stream", "conv5_h"]) == (mode == 2), kernels
            assert (kernels == ["feat_stream", "tail_stream"]) == (mode == 1), kernels
            outs[mode] = eng.forward(x, x2).astype(np.float64)
    scale = np.abs(outs[0]).max()
    assert scale > 1.0
    for mode in (2, 1):
        err = np.abs(outs[mode] - outs[0]).max() / scale
        print("%s vs layered, %s: max rel %.3g" % ("folded tail" if mode == 2 else "streamed", shape, err))
        assert err <= 5e-6
    # ... and all against the float64 oracle (the row-block path of a tall single image and the 3-image batch
    # met only the layer-by-layer launches before)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    for mode in (2, 1, 0):
        e = np.abs(outs[mode] - ref).max() / np.abs(ref).max()
        print("  %s vs float64 oracle: max rel %.3g" % (("layered", "streamed", "feat_stream + folded tail")[mode], e))
        assert e <= 5e-6


# (layers, filters, min_filters, gamma, nin_filters, nin_filters2, scale, activator): between them every instantiated
# stream_conv_role<input quads 1..8, output tiles 1..2> and every last-chunk width of the A1 || B1 sources
STREAM_VARIANTS = [
    (7, 32, 8, 1.2, 24, 8, 2, "prelu"),        # the shipped widths at x2: feat_stream + per-layer tail
    (7, 32, 8, 1.2, 24, 8, 3, "prelu"),
    (5, 29, 3, 1.0, 20, 12, 4, "relu"),        # quads 8, 6, 4, 2 -> ...; B2 from 12 channels (3 quads)
    (6, 24, 1, 1.5, 9, 5, 2, "leaky_relu"),    # down to one channel: a single-quad source and a 1-quad conv input
    (4, 20, 16, 1.0, 16, 16, 2, "prelu"),      # 5 quads in, 2 output tiles would be <5,2>: not instantiated -> layer by layer
    (3, 32, 17, 2.0, 8, 4, 4, "prelu"),        # 8 quads -> 2 tiles, then 5 quads -> ...
    (2, 16, 13, 1.0, 28, 4, 3, "prelu"),
]


@pytest.mark.parametrize("variant", STREAM_VARIANTS)
def test_feat_stream_variants(oracle, variant):
    """Separable nets of other depths / widths than the shipped one: the streamed feature extractor must be taken wherever
    its role instantiations cover the net (and only there), and meet the bare-branch bar against the float64 oracle."""
    from dcscn_amd import engine
    layers, filters, min_filters, gamma, na, nb, scale, act = variant
    cfg = oracle.make_config(layers=layers, filters=filters, min_filters=min_filters, filters_decay_gamma=gamma, nin_filters=na,
                             nin_filters2=nb, scale=scale, activator=act, depthwise_separable=True, reconstruct_layers=0,
                             pixel_shuffler_filters=1)
    weights = oracle.synthetic_weights(cfg, seed=layers * 10 + scale)
    x, x2 = synthetic_batch(2, 37, 55, scale, seed=5)
    x2 = np.zeros_like(x2)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        kernels = [o["kernel"] for o in eng.ops()]
        y = eng.forward(x, x2).astype(np.float64)
    sched = oracle.filter_schedule(layers, filters, min_filters, gamma)
    quads = lambda c: (c + 3) // 4
    pairs = [(quads(sched[i]), (sched[i + 1] + 15) // 16) for i in range(layers - 1)] + [(quads(nb), (nb + 15) // 16)]
    supported = all((t == 1) if qi <= 5 else (t == 2) if qi <= 7 else True for qi, t in pairs)
    assert ("feat_stream" in kernels) == supported, (kernels, sched, pairs)
    scale_ = np.abs(ref).max()
    err = np.abs(y - ref).max() / scale_
    print(variant, kernels[:2], "max rel %.3g" % err)
    assert err <= 5e-6


# ---- P16: pre-split tensors between split16 launches (csrc/p16.hpp, option "p16") ----------------------------------------------

P16_CASES = [
    ("L12_F196to48_x2", dict(), 2, 48, 48),
    ("L12_F196to48_x2", dict(), 3, 37, 53),              # tiles that stick out of the image, on every layer
    ("L8_F96to48_x2", dict(), 2, 40, 56),
    ("L12_F196to48_x4", dict(), 1, 33, 47),
    ("L7_F32to8_x2", dict(), 3, 31, 18),
    ("wide-odd", dict(layers=3, filters=70, min_filters=45, nin_filters=40, nin_filters2=33), 2, 17, 33),   # Concat2 slice off a 16-channel boundary
    ("odd-channels", dict(layers=4, filters=37, min_filters=13, nin_filters=21, nin_filters2=10), 2, 20, 50),
    ("three-groups", dict(layers=3, filters=250, min_filters=120, nin_filters=64, nin_filters2=32), 1, 24, 40),
    # the half-resolution map of an x4 net stays pre-split through the pixel shuffler (conv3_h's P16 epilogue with depth_to_space addressing)
    ("c-DCSCN-x4", dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0,
                        pixel_shuffler_filters=1, scale=4), 2, 21, 35),
    ("c-DCSCN-x3", dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0,
                        pixel_shuffler_filters=1, scale=3), 2, 21, 35),
]


@pytest.mark.parametrize("case", P16_CASES, ids=[c[0] + "-%dx%dx%d" % c[2:] for c in P16_CASES])
def test_p16_tensors_are_bit_identical_to_float32_tensors(oracle, case):
    """Option p16 moves the f16 (hi, lo) split from every consumer's staging code into the producer's epilogue and changes the
    workspace layout; the products and their order do not change -- every output bit must be the same, with the fold on and off,
    and across sub-batching (a partial last pass uses planes carved for more images)."""
    name, overrides, n, h, w = case
    cfg = oracle.make_config(**(CONFIGS[name] if name in CONFIGS else overrides))
    weights = oracle.synthetic_weights(cfg, seed=11)
    x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=12)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    for fold in (True, False):
        from dcscn_amd import engine
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights, fold_tail=fold)
            n16 = eng.num_presplit_tensors()
            y1 = eng.forward(x, x2)
            eng.set_option("p16", 0)
            assert eng.num_presplit_tensors() == 0
            y0 = eng.forward(x, x2)
            eng.set_option("p16", 1)
            eng.set_option("sub_batch_pixels", 2 * h * w)
            y2 = eng.forward(x, x2)
        print("%s fold=%s: %d P16 tensors, max-abs err %.3g" % (name, fold, n16, float(np.max(np.abs(y1 - ref)))))
        if name.startswith("L12") or name.startswith("L8"):
            assert n16 >= cfg["layers"], n16                  # every feature map at least
        assert float(np.max(np.abs(y1 - ref))) <= MAX_ABS_TOL
        assert np.array_equal(y1, y0), "p16 on / off differ in %d values" % int(np.sum(y1 != y0))
        assert np.array_equal(y1, y2)


def test_one_tile_shuffler_layer_runs_on_conv3_h(oracle):
    """x3 c-DCSCN: Up-PS is a 3x3 conv 32 -> 9 in front of a pixel shuffler to ONE channel -- one 16-channel tile, no 16-byte stores.  Under
    split16 it runs on conv3_h (per-channel store epilogue), conv_igemm stays behind it as the float32 kernel."""
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0,
                             pixel_shuffler_filters=1, scale=3)
    weights = oracle.synthetic_weights(cfg, seed=2)
    x, x2 = synthetic_batch(2, 19, 23, 3, seed=5)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights, fold_tail=False)
        up = [o for o in eng.ops() if o["name"].startswith("Up-PS")]
        assert up and up[0]["kernel"] == "conv3_h" and up[0]["out_channels"] == 9, up
        y1 = eng.forward(x, x2)
        eng.set_option("split16", 0)
        up = [o for o in eng.ops() if o["name"].startswith("Up-PS")]
        assert up[0]["kernel"] == "conv_igemm", up
        y0 = eng.forward(x, x2)
    assert float(np.max(np.abs(y1 - ref))) <= MAX_ABS_TOL and float(np.max(np.abs(y0 - ref))) <= MAX_ABS_TOL


def test_p16_overflow_recomputes_the_image_on_the_float32_plan(oracle):
    """A flagged image (values beyond the f16 range) is recomputed from the first layer on by the float32 launches behind the pass --
    bit-identical to a split16 = 0 run of it -- while the other image of the batch keeps the bits of a run without the huge one; the
    next forward on the same handle (the float32 plan wrote float32 tensors over the P16 planes and their zero records) is clean."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L12_F196to48_x2"])
    weights = oracle.synthetic_weights(cfg, seed=9)
    x, x2 = synthetic_batch(3, 40, 33, 2, seed=10)
    xb = x.copy()
    xb[1] *= 4000.0
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        assert eng.num_presplit_tensors() > 0
        clean = eng.forward(x, x2)
        y = eng.forward(xb, x2)
        again = eng.forward(x, x2)
        eng.set_option("split16", 0)
        y32 = eng.forward(xb, x2)
    assert np.isfinite(y).all()
    assert np.array_equal(y[0], clean[0]) and np.array_equal(y[2], clean[2])
    assert np.array_equal(y[1], y32[1])
    assert np.array_equal(again, clean)


def test_wide_gemm_workgroup_sizes_give_the_same_bits(oracle):
    """r06: the 1301-channel A1 || B1 GEMM of the L12 nets runs on 256-pixel workgroups (conv_nin_h_w8: 8 waves x two tiles, half the filter
    traffic per pixel), every other 1x1 GEMM on 128-pixel ones -- same products in the same order per pixel.  DCSCN_NINH8=0 (read once per
    process) keeps the 128-pixel workgroups: two subprocesses, one digest; 300 pixels per image so that both sizes have a ragged last block."""
    import hashlib
    import subprocess
    import sys
    code = (
        "import sys, os, hashlib, numpy as np\n"
        "root = %r\n"
        "sys.path[:0] = [root, os.path.join(root, 'oracle'), os.path.join(root, 'tests')]\n"
        "import dcscn_oracle as O\n"
        "from conftest import CONFIGS, synthetic_batch\n"
        "from dcscn_amd import engine\n"
        "cfg = O.make_config(**CONFIGS['L12_F196to48_x2'])\n"
        "w = O.synthetic_weights(cfg, seed=5)\n"
        "x, x2 = synthetic_batch(3, 20, 15, 2, seed=6)\n"
        "eng = engine.Engine(cfg, device=0); eng.load_weights(w)\n"
        "assert 'conv_nin_h' in [o['kernel'] for o in eng.ops()]\n"
        "print('DIGEST', hashlib.sha256(eng.forward(x, x2).tobytes()).hexdigest())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for flag in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DCSCN_NINH8=flag), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("scale", [2, 3, 4])
def test_streamed_dense_feature_extractor(oracle, scale):
    """feat3_stream: CNN1 .. CNNL of the non-separable narrow nets (the c-DCSCN checkpoints' topology) as one row-streamed launch on the
    f16 matrix pipe -- taken by default (kernel list), within the bars of the float64 oracle on patches, on an image wider than a strip
    (column strips) and on a tall one (row blocks), and consistent with the layer-by-layer launches (option stream_dense = 0)."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x%d" % scale])
    weights = oracle.synthetic_weights(cfg, seed=21)
    for n, h, w in ((3, 48, 48), (1, 37, 131), (1, 300, 20)):
        x, x2 = synthetic_batch(n, h, w, scale, seed=22)
        ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights)
            assert "feat3_stream" in [o["kernel"] for o in eng.ops()], eng.ops()
            y = eng.forward(x, x2)
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("stream_dense", 0)
            eng.load_weights(weights)
            assert "feat3_stream" not in [o["kernel"] for o in eng.ops()]
            y0 = eng.forward(x, x2)
        err, err0 = float(np.max(np.abs(y - ref))), float(np.max(np.abs(y0 - ref)))
        print("x%d %dx%dx%d: streamed max-abs %.3g, layer by layer %.3g" % (scale, n, h, w, err, err0))
        assert np.isfinite(y).all() and err <= MAX_ABS_TOL and err0 <= MAX_ABS_TOL


@pytest.mark.parametrize("scale", [2, 3, 4])
def test_streamed_nin_keeps_the_feature_maps_out_of_hbm(oracle, scale):
    """r06: with the c-DCSCN shape the streamed launch also accumulates A1 || B1 as the layers' rows appear and runs B2
    behind it -- the launch list loses B1+A1 and B2, Concat2 is the launch's only output -- within the same bars of the float64 oracle on
    patches, column strips, row blocks, a ragged batch and a 1-pixel image; stream_nin = 0 is the r05 plan (same bars)."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS["L7_F32to8_x%d" % scale])
    weights = oracle.synthetic_weights(cfg, seed=31)
    for n, h, w in ((3, 48, 48), (1, 37, 131), (1, 300, 20), (5, 17, 33), (2, 1, 1), (1, 64, 50)):
        x, x2 = synthetic_batch(n, h, w, scale, seed=32)
        ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
        with engine.Engine(cfg, device=0) as eng:
            eng.load_weights(weights)
            ops = eng.ops()
            assert ops[0]["kernel"] == "feat3_stream" and ops[0]["name"].startswith("CNN1..B2"), ops
            assert not [o for o in ops if o["name"] in ("B1+A1", "B2")], ops
            y = eng.forward(x, x2)
            assert np.array_equal(y, eng.forward(x, x2))                      # deterministic
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("stream_nin", 0)
            eng.load_weights(weights)
            names = [o["name"] for o in eng.ops()]
            assert "B1+A1" in names and "B2" in names and eng.ops()[0]["kernel"] == "feat3_stream"
            y0 = eng.forward(x, x2)
        err, err0 = float(np.max(np.abs(y - ref))), float(np.max(np.abs(y0 - ref)))
        print("x%d %dx%dx%d: A1 || B1 in the launch max-abs %.3g, conv_nin_h behind it %.3g" % (scale, n, h, w, err, err0))
        assert np.isfinite(y).all() and err <= MAX_ABS_TOL and err0 <= MAX_ABS_TOL
    # the bare residual branch (last conv unattenuated, x2 = 0) at the relative bar of test_residual_branch_relative_error, both plans
    weights = oracle.synthetic_weights(cfg, seed=33)
    last = "R-CNN%d" % cfg["reconstruct_layers"]
    weights[last + "/conv_W"] = weights[last + "/conv_W"] * 100.0
    x, _ = synthetic_batch(2, 48, 48, scale, seed=34)
    x2 = np.zeros((2, 48 * scale, 48 * scale, 1), np.float32)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    for nin in (1, 0):
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("stream_nin", nin)
            eng.load_weights(weights)
            y = eng.forward(x, x2)
        rel = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
        print("x%d bare branch, stream_nin %d: relative %.3g" % (scale, nin, rel))
        assert rel <= 5e-6
