"""Host-side pieces that need no GPU: checkpoint reader, flag surface, imaging helpers, model naming,
the C ABI's symbol table and its loud failure without a device."""
import json
import os
import re
import sys

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, ROOT


# ---- TF checkpoint reader -------------------------------------------------------------------------
def test_checkpoint_reader_reads_shipped_bundle():
    from dcscn_amd import ckpt
    prefix = os.path.join(GOLDEN, "models", "dcscn_L2_F4to4_PS_R1F4.ckpt")
    names = dict(ckpt.list_variables(prefix))
    assert names["CNN1/conv_W"] == (3, 3, 1, 4) and names["Up-PS/Up-PS_CNN/conv_W"] == (3, 3, 8, 32)
    assert any(ckpt.is_optimizer_slot(n) for n in names)            # Adam slots are in the file ...
    tensors = ckpt.load_checkpoint(prefix)
    assert not any(ckpt.is_optimizer_slot(n) for n in tensors)      # ... and dropped by default
    golden = np.load(os.path.join(GOLDEN, "weights_L2_x2.npz"))
    assert set(golden.files) == set(tensors)
    for k in golden.files:
        assert np.array_equal(golden[k], tensors[k]) and tensors[k].dtype == np.float32


def test_checkpoint_reader_rejects_garbage(tmp_path):
    from dcscn_amd import ckpt
    bad = tmp_path / "x.ckpt.index"
    bad.write_bytes(b"\x00" * 100)
    with pytest.raises(ckpt.CheckpointError):
        ckpt.read_index(str(bad))
    short = tmp_path / "y.ckpt.index"
    short.write_bytes(b"abc")
    with pytest.raises(ckpt.CheckpointError):
        ckpt.read_index(str(short))
    # index present, data shard missing (the reference's L8 / L12 checkpoints ship like this)
    src = os.path.join(GOLDEN, "models", "dcscn_L2_F4to4_PS_R1F4.ckpt.index")
    lone = tmp_path / "z.ckpt.index"
    lone.write_bytes(open(src, "rb").read())
    assert not ckpt.has_data(str(tmp_path / "z.ckpt"))
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_checkpoint(str(tmp_path / "z.ckpt"))


# ---- flags ----------------------------------------------------------------------------------------
def _fresh_flags():
    from dcscn_amd import flags
    fv = flags.FlagValues()
    flags.DEFINE_integer("scale", 2, "", flag_values=fv)
    flags.DEFINE_float("gamma", 1.5, "", flag_values=fv)
    flags.DEFINE_boolean("use_nin", True, "", flag_values=fv)
    flags.DEFINE_boolean("ds", False, "", flag_values=fv)
    flags.DEFINE_string("file", "image.jpg", "", flag_values=fv)
    return flags, fv


def test_flag_syntax():
    flags, fv = _fresh_flags()
    assert (fv.scale, fv.gamma, fv.use_nin, fv.ds, fv.file) == (2, 1.5, True, False, "image.jpg")
    rest = fv(["prog", "--scale=4", "--gamma", "1.2", "--nouse_nin", "--ds", "-file=a.png", "extra"])
    assert rest == ["prog", "extra"]
    assert (fv.scale, fv.gamma, fv.use_nin, fv.ds, fv.file) == (4, 1.2, False, True, "a.png")
    fv.reset()
    fv(["prog", "--use_nin=false", "--ds=True", "--", "--scale=9"])
    assert (fv.use_nin, fv.ds, fv.scale) == (False, True, 2)
    with pytest.raises(flags.FlagError):
        fv(["prog", "--unknown=1"])
    with pytest.raises(flags.FlagError):
        fv(["prog", "--scale=abc"])
    with pytest.raises(flags.FlagError):
        fv(["prog", "--scale"])
    fv.scale = 3
    assert fv.scale == 3 and fv.flag_values_dict()["scale"] == 3


def test_reference_flag_surface():
    """Every flag of the reference's helper/args.py:17-98 with its default."""
    from helper import args
    expect = dict(scale=2, layers=12, filters=196, min_filters=48, filters_decay_gamma=1.5, use_nin=True,
                  nin_filters=64, nin_filters2=32, cnn_size=3, reconstruct_layers=1, reconstruct_filters=32,
                  dropout_rate=0.8, activator="prelu", pixel_shuffler=True, pixel_shuffler_filters=0,
                  self_ensemble=8, batch_norm=False, depthwise_separable=False, bicubic_init=True, clipping_norm=5.0,
                  initializer="he", weight_dev=0.01, l2_decay=0.0001, optimizer="adam", beta1=0.9, beta2=0.999,
                  epsilon=1e-8, momentum=0.9, batch_num=20, batch_image_size=48, stride_size=0,
                  training_images=24000, use_l1_loss=False, initial_lr=0.002, lr_decay=0.5, lr_decay_epoch=9,
                  end_lr=2e-5, dataset="bsd200", test_dataset="set5", tests=1, do_benchmark=False, max_value=255.0,
                  channels=1, psnr_calc_border_size=-1, build_batch=False, checkpoint_dir="models",
                  graph_dir="graphs", data_dir="data", batch_dir="batch_data", output_dir="output",
                  tf_log_dir="tf_log", log_filename="log.txt", model_name="", load_model_name="",
                  initialize_tf_log=True, enable_log=True, save_weights=True, save_images=False, save_images_num=20,
                  save_meta_data=False, gpu_device_id=0, frozenInference=False,
                  frozen_graph_path="./model_to_freeze/frozen_model_optimized.pb")
    defaults = {n: args.FLAGS._flags[n].default for n in args.FLAGS}
    for name, value in expect.items():
        assert name in defaults, name
        assert defaults[name] == value and type(defaults[name]) is type(value), name


# ---- model naming (selects the checkpoint file) -----------------------------------------------------
class _F(dict):
    __getattr__ = dict.__getitem__


def _flags(**over):
    from helper import args
    d = {n: args.FLAGS._flags[n].default for n in args.FLAGS}
    d.update(over)
    d["log_filename"] = ""
    return _F(d)


def test_model_names_select_the_shipped_checkpoints(tmp_path):
    from dcscn_amd.model import SuperResolution
    ck = str(tmp_path / "models")
    L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8,
              reconstruct_layers=0, self_ensemble=1, pixel_shuffler_filters=1, checkpoint_dir=ck)
    cases = [
        (dict(checkpoint_dir=ck), "dcscn_L12_F196to48_NIN_A64_PS_R1F32"),
        (dict(scale=3, checkpoint_dir=ck), "dcscn_L12_F196to48_Sc3_NIN_A64_PS_R1F32"),
        (dict(layers=8, filters=96, scale=4, checkpoint_dir=ck), "dcscn_L8_F96to48_Sc4_NIN_A64_PS_R1F32"),
        (L7, "dcscn_L7_F32to8_G1.20_NIN_A24_B8_PS_R1F32"),
        (dict(L7, scale=4, depthwise_separable=True), "dcscn_L7_F32to8_G1.20_Sc4_NIN_A24_B8_PS_DS_R1F32"),
        (dict(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4, checkpoint_dir=ck),
         "dcscn_L2_F4to4_PS_R1F4"),
    ]
    for over, name in cases:
        assert SuperResolution(_flags(**over)).name == name
    assert SuperResolution(_flags(checkpoint_dir=ck), model_name="abc").name == "dcscn_abc"


# ---- imaging ----------------------------------------------------------------------------------------
def test_imaging_helpers():
    from dcscn_amd import imaging as util
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (10, 14, 3)).astype(np.uint8)
    for t in range(8):
        assert np.array_equal(util.flip(util.flip(img, t), t, invert=True), img)
    assert util.set_image_alignment(img, 4).shape == (8, 12, 3)
    y = util.convert_rgb_to_y(img)
    ycc = util.convert_rgb_to_ycbcr(img)
    assert y.dtype == np.float64 and np.allclose(y[:, :, 0], ycc[:, :, 0])
    back = util.convert_ycbcr_to_rgb(ycc)
    assert np.max(np.abs(back - img)) < 0.01
    lr = util.resize_image_by_pil(y, 0.5)
    assert lr.shape == (5, 7, 1) and lr.dtype == np.float32
    assert util.resize_image_by_pil(img, 2).shape == (20, 28, 3)
    a = rng.uniform(0, 255, (40, 40, 1))
    psnr, ssim = util.compute_psnr_and_ssim(a, a, border_size=2)
    assert psnr == float("inf") and abs(ssim - 1.0) < 1e-12
    b = a + rng.normal(0, 5, a.shape)
    psnr, ssim = util.compute_psnr_and_ssim(a, b, border_size=2)
    assert 30 < psnr < 40 and 0.5 < ssim < 1.0
    assert util.compute_psnr_and_ssim(a, b[:-1]) is None


def test_ssim_vectorised_equals_per_column_loop():
    """The reference's SSIM quirk (columns as channels, utilty.py:529-535): the vectorised evaluation
    must equal the literal per-column loop."""
    from dcscn_amd import imaging as util
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 255, (37, 23)).round()
    b = np.clip(a + rng.normal(0, 9, a.shape), 0, 255).round()
    fast = util._ssim_last_axis_channels(a, b, 255, 1.5, 0.01, 0.03)
    loop = float(np.mean([util._ssim_gaussian(a[:, c], b[:, c], 255, 1.5, 0.01, 0.03) for c in range(a.shape[1])]))
    assert abs(fast - loop) < 1e-14
    with pytest.raises(ValueError):
        util._ssim_last_axis_channels(a[:10], b[:10], 255, 1.5, 0.01, 0.03)


def test_ssim_against_a_first_principles_restatement():
    """scikit-image is not installable here, so SSIM is pinned to its PUBLISHED algorithm instead (Wang et al. 2004 as
    implemented by skimage.metrics.structural_similarity with win_size=11, gaussian_weights=True, sigma=1.5,
    use_sample_covariance=True, K1=.01, K2=.03, data_range=255 -- the call at utilty.py:533-535), restated here with
    nothing but numpy: an explicit normalised Gaussian of radius int(3.5*1.5+0.5)=5, scipy's 'reflect' (half-sample
    symmetric) boundary written out by hand, sample-covariance factor NP/(NP-1) with NP = 11 (1-D windows: the
    multichannel=True quirk makes every image COLUMN a 1-D signal), 5 border samples cropped from the mean."""
    from dcscn_amd import imaging as util
    rng = np.random.default_rng(11)
    a = rng.uniform(0, 255, (45, 9)).round()
    b = np.clip(a + rng.normal(0, 12, a.shape), 0, 255).round()

    sigma, radius = 1.5, 5
    g = np.exp(-0.5 * (np.arange(-radius, radius + 1) / sigma) ** 2)
    g /= g.sum()

    def filt(x):                                  # 1-D correlation with reflect padding: (d c b a | a b c d | d c b a)
        n = len(x)
        out = np.empty(n)
        for i in range(n):
            acc = 0.0
            for k in range(-radius, radius + 1):
                j = i + k
                while j < 0 or j >= n:
                    j = -j - 1 if j < 0 else 2 * n - 1 - j
                acc += g[k + radius] * x[j]
            out[i] = acc
        return out

    c1, c2, cov_norm = (0.01 * 255) ** 2, (0.03 * 255) ** 2, 11.0 / 10.0
    per_column = []
    for col in range(a.shape[1]):
        x, y = a[:, col].astype(np.float64), b[:, col].astype(np.float64)
        ux, uy = filt(x), filt(y)
        vx = cov_norm * (filt(x * x) - ux * ux)
        vy = cov_norm * (filt(y * y) - uy * uy)
        vxy = cov_norm * (filt(x * y) - ux * uy)
        s_map = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        per_column.append(s_map[radius:-radius].mean())
    want = float(np.mean(per_column))
    got = util._ssim_last_axis_channels(a, b, 255, 1.5, 0.01, 0.03)
    assert abs(got - want) < 1e-12, (got, want)
    # and through the public entry point (rint / clip / shave as utilty.py:501-527; border 2 here)
    psnr, ssim = util.compute_psnr_and_ssim(a[:, :, None], b[:, :, None], border_size=0)
    assert abs(ssim - want) < 1e-12


def test_bicubic_goldens_through_the_host_glue(oracle):
    """evaluate_bicubic's recipe with the package's own helpers reproduces the oracle's numbers."""
    from dcscn_amd import imaging as util
    from dcscn_amd.model import build_input_image
    with open(os.path.join(GOLDEN, "goldens.json")) as f:
        g = json.load(f)
    for s in (2, 3, 4):
        for i in (1, 4):
            img = util.set_image_alignment(util.load_image(os.path.join(GOLDEN, "set5", g["files"][i]),
                                                           print_console=False), s)
            lr = build_input_image(img, channels=1, scale=s, alignment=s, convert_ycbcr=True)
            bic = util.resize_image_by_pil(lr, s)
            psnr, _ = util.compute_psnr_and_ssim(util.convert_rgb_to_y(img), bic, border_size=s)
            assert abs(psnr - g["bicubic"]["x%d" % s][i]) < 1e-9


def test_save_and_load_roundtrip(tmp_path):
    from dcscn_amd import imaging as util
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (6, 9, 3)).astype(np.uint8)
    grey = rng.uniform(0, 255, (6, 9, 1))
    util.save_image(str(tmp_path / "a" / "rgb.png"), rgb, print_console=False)
    util.save_image(str(tmp_path / "grey.png"), grey, print_console=False)
    assert np.array_equal(util.load_image(str(tmp_path / "a" / "rgb.png"), print_console=False), rgb)
    assert np.array_equal(util.load_image(str(tmp_path / "grey.png"), print_console=False)[:, :, 0],
                          grey[:, :, 0].astype(np.uint8))
    with pytest.raises(util.LoadError):
        util.load_image(str(tmp_path / "missing.png"))


# ---- C ABI ------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from dcscn_amd import build, engine
    build.build()                                        # no-op when up to date; hipcc cross-compiles on CPU
    header = open(os.path.join(ROOT, "include", "dcscn.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dcscn_[a-z_]+)\s*\(", header))
    assert declared == set(engine.EXPORTED_SYMBOLS)
    lib = engine.load_library()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.dcscn_abi_version() == engine.ABI_VERSION
    import ctypes
    assert ctypes.sizeof(engine.Config) == 120    # dcscn_config: 5 x i32, pad, f64, 13 x i32, 8 reserved, tail pad
    # host-side helper that needs no device
    assert engine.filter_schedule(12, 196, 48, 1.5) == [196, 166, 148, 133, 120, 108, 97, 86, 76, 66, 57, 48]
    assert engine.filter_schedule(7, 32, 8, 1.2) == [32, 26, 22, 18, 14, 11, 8]


def test_filter_schedule_matches_python_for_many_flags(oracle):
    from dcscn_amd import engine
    for layers in (1, 2, 3, 7, 12, 20):
        for filters, minf in ((196, 48), (96, 48), (32, 8), (64, 64), (10, 0), (8, 32)):
            for gamma in (1.0, 1.2, 1.5, 2.0, 0.7):
                if layers == 1 and minf != 0:
                    continue      # the reference divides by (layers - 1)
                want = oracle.filter_schedule(layers, filters, min(filters, minf), gamma)
                assert engine.filter_schedule(layers, filters, minf, gamma) == want


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from dcscn_amd import engine
    with pytest.raises(engine.EngineError) as e:
        engine.Engine(dict())
    assert e.value.status == 5
    src = open(os.path.join(ROOT, "dcscn-super-resolution_amd", "model.py")).read() + \
        open(os.path.join(ROOT, "dcscn-super-resolution_amd", "engine.py")).read()
    assert "oracle" not in src


def test_batch_norm_is_refused_not_ignored(tmp_path):
    """--batch_norm=true (tf_graph.py:112-113, 175) is in no shipped model and not implemented; the library must say so
    (DCSCN_ERR_UNSUPPORTED from dcscn_create, before it even looks for a device) instead of silently running the graph without it --
    through the C ABI and through the reference-shaped model class (whose name still gets the reference's _BN suffix, DCSCN.py:131)."""
    from dcscn_amd import engine
    from dcscn_amd.model import SuperResolution
    with pytest.raises(engine.EngineError) as e:
        engine.Engine(dict(batch_norm=True))
    assert e.value.status == 2 and engine.STATUS_NAMES[2] == "UNSUPPORTED" and "batch_norm" in str(e.value)
    m = SuperResolution(_flags(batch_norm=True, checkpoint_dir=str(tmp_path / "models")))
    assert m.name.endswith("_BN_R1F32")
    with pytest.raises(engine.EngineError) as e:
        m.build_graph()
    assert e.value.status == 2
    # the other flags the survey marks unsupported are refused the same way, not approximated
    for bad in (dict(channels=3), dict(cnn_size=4), dict(scale=5)):
        with pytest.raises(engine.EngineError) as e:
            engine.Engine(bad)
        assert e.value.status == 2, bad


def test_resample_tables_reproduce_pillow():
    """The library's per-axis bicubic tables (host code, no GPU), applied in float64 in tap order with one rounding
    per pass, must give Pillow's mode-'F' BICUBIC resize bit for bit -- the same tables drive the device kernels
    (tests/test_resize_hip.py checks those against Pillow on the GPU)."""
    from PIL import Image
    from dcscn_amd import engine

    def one_axis(img, out):                       # resize axis 1 of [rows, n] float32
        bounds, w = engine.resample_table(img.shape[1], out)
        res = np.zeros((img.shape[0], out), np.float32)
        for i in range(out):
            x0, n = bounds[i]
            ss = np.zeros(img.shape[0])
            for t in range(n):
                ss = ss + img[:, x0 + t].astype(np.float64) * w[i, t]
            res[:, i] = ss.astype(np.float32)
        return res

    rng = np.random.default_rng(0)
    for (h, w), (oh, ow) in [((12, 14), (36, 42)), ((31, 17), (62, 34)), ((40, 36), (10, 9)), ((9, 20), (27, 5))]:
        img = rng.uniform(0, 255, (h, w)).astype(np.float32)
        ref = np.asarray(Image.fromarray(img).resize([ow, oh], resample=Image.BICUBIC))
        got = one_axis(one_axis(img, ow).T.copy(), oh).T          # horizontal pass first, as Pillow
        assert np.array_equal(got, ref)


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r01_bench_n1.json is the line `python bench.py` printed on the MI355X box: the keys the driver and the
    judge read must all be there, with the metric / workload BASELINE.json names."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r01_bench_n1.json")) as f:
        d = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("LR Mpixels/sec at 48x48 patches, L12_F196to48 x2")
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - 1024 * 48 * 48 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3


def test_committed_r06_bench_line_carries_the_other_configs_and_the_parity_half():
    """profiles/r06_bench_n1.json (the line of the r06 tree on an MI355X box): besides the contract fields, BASELINE's other
    single-GPU configs with their roofline objects ("also C2, C5"), the metric's "PSNR delta vs ref on Set5" computed
    in the run, and counters replayed only from a PMC file taken on the same kernels."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r06_bench_n1.json")) as f:
        d = json.load(f)
    assert d["metric"].startswith("LR Mpixels/sec at 48x48 patches, L12_F196to48 x2") and d["n_gpus"] == 1
    assert abs(d["value"] - 1024 * 48 * 48 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["useful_frac"] <= r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic_detail"]["replayed"] is True
    for key, bound, px in (("C2", "mfma", 256 * 48 * 48), ("C5", "hbm", 1024 * 48 * 48)):
        c = d["other_configs"][key]
        assert "error" not in c and c["steps"] >= 20 and c["unit"] == "LR Mpix/s"
        assert abs(c["value"] - px / (c["ms_per_step"] * 1e-3) / 1e6) / c["value"] < 1e-3
        rr = c["roofline"]
        assert rr["bound"] == bound and 0 < rr["frac"] <= 1.0 and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
        assert c["kernel_ms_per_step"] <= c["ms_per_step"]
    fl = d["other_configs"]["C5"]["roofline"]["floor_ms"]
    assert fl["valu_plus_mfma"] > fl["hbm"] > 0                 # the narrow net's floor is its arithmetic, not its compulsory I/O
    par = d["parity"]
    assert 0 <= par["set5_psnr_delta_db"] <= par["bars"]["set5_psnr_delta_db"] == 1e-3
    assert 0 <= par["max_abs"] <= par["bars"]["max_abs"] == 1e-4


def test_stale_pmc_files_are_refused():
    """bench.py replays HBM / MFMA counters from the newest committed PMC summary only when the kernels that did real work under
    the profiler are the kernels of the run ("change a kernel, forget tools/rocprof_bench.sh, and the line quotes
    stale counters against fresh times")."""
    import bench
    run = {"conv_cin1", "conv3_h8", "conv3_h", "conv_nin_h", "conv5_h"}
    path, why = bench._pmc_file("conv3_h", run)
    assert path is not None and why is None and os.path.basename(path).startswith("r0")
    path, why = bench._pmc_file("conv3_h", (run - {"conv3_h"}) | {"conv3_hc"})
    assert path is None and why.startswith("stale: ")
    traffic, ns = bench.pmc_replay(15.5, 2.9, 2.3e10, (run - {"conv3_h"}) | {"conv3_hc"})
    assert traffic is None and ns["traffic"] is None and ns["reason"].startswith("stale: ")
    assert bench._kernel_family("void dcscn::conv3_h8<6, 5, 0, 6, true>(dcscn::ConvArgs)") == "conv3_h8"
    assert bench._kernel_family("conv_nin_h<6, 2, 3, 2>") == "conv_nin_h"
    assert len(bench.csrc_digest()) == 64


def test_library_has_no_packed_f32_instructions(tmp_path):
    """v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 return wrong low halves (lanes 48-63) when ANOTHER process's MFMA work shares the
    SIMD (tools/xproc_triage.hip: victim cin1p vs cin1s beside aggressor mfma) -- the r03 cross-process
    corruption.  build.py compiles every source with -target-feature -packed-fp32-ops; this checks the shipped code objects."""
    import glob
    import shutil
    import subprocess
    from conftest import ROOT
    lib = os.path.join(ROOT, "dcscn-super-resolution_amd", "libdcscn_hip.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    # (every image this suite runs in carries ROCm's llvm-objdump: a missing tool is a failure, not a skip -- ADVICE r04 -- so that a
    # toolchain change which drops the feature flag cannot pass unseen)
    assert os.path.isfile(objdump), "llvm-objdump not found: the packed-f32 check cannot run"
    if not os.path.isfile(lib):
        pytest.skip("library not built")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950*"))
    assert objs, os.listdir(tmp_path)
    packed = mfma = 0
    for o in objs:
        dis = subprocess.run([objdump, "-d", o], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        packed += sum(1 for ln in dis.splitlines() if "v_pk_fma_f32" in ln or "v_pk_mul_f32" in ln or "v_pk_add_f32" in ln)
        mfma += dis.count("v_mfma_f32_16x16x32_f16")
    assert mfma > 1000, "disassembly did not find the kernels"
    assert packed == 0, "%d packed-f32 VALU instructions in libdcscn_hip.so" % packed


def test_every_option_of_the_library_is_documented():
    """Each key dcscn_set_option accepts (csrc/api.hip) is described in include/dcscn.h and listed in INTEGRATION.md."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = [d for d in os.listdir(root) if d.endswith("_amd") and os.path.isdir(os.path.join(root, d))][0]
    api = open(os.path.join(root, pkg, "csrc", "api.hip")).read()
    keys = sorted(set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', api)))
    assert len(keys) >= 10
    header = open(os.path.join(root, "include", "dcscn.h")).read()
    integration = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [k for k in keys if '"%s"' % k not in header or "`%s`" % k not in integration]
    assert not missing, "undocumented options: %s" % missing
