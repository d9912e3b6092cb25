"""Device colour conversions and the RGB pipelines of evaluate.py / sr.py (csrc/color.hip, dcscn_evaluate_rgb, dcscn_sr_rgb)
against the host restatement of helper/utilty.py:142-193 (imaging.py, numpy float64) and against the host path of the
model (Pillow + numpy + dcscn_forward): the device pipeline must give the same arrays."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_golden_hip import _goldens, _model

pytestmark = pytest.mark.gpu


def _images():
    from dcscn_amd import imaging as util
    g = _goldens()
    out = [util.load_image(os.path.join(GOLDEN, "set5", f), print_console=False) for f in g["files"][:3]]
    rng = np.random.default_rng(5)
    out.append(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8))
    out.append(np.stack(list(np.meshgrid(np.arange(256), np.arange(256))) + [np.full((256, 256), 77)], -1).astype(np.uint8))
    return [im for im in out if im.ndim == 3 and im.shape[2] == 3]


def test_colour_conversions_match_numpy_bit_for_bit(tmp_path):
    from dcscn_amd import imaging as util
    g, m = _model(tmp_path, "L7_x2")
    eng = m._ready_engine()
    for im in _images():
        y = eng.convert_rgb_to_y(im)
        assert y.dtype == np.float64 and np.array_equal(y, util.convert_rgb_to_y(im))
        ycc = eng.convert_rgb_to_ycbcr(im)
        assert np.array_equal(ycc, util.convert_rgb_to_ycbcr(im))
        rng = np.random.default_rng(1)
        ynew = y + rng.normal(0, 3, y.shape)
        rgb = eng.convert_y_and_cbcr_to_rgb(ynew, ycc[:, :, 1:3])
        assert np.array_equal(rgb, util.convert_y_and_cbcr_to_rgb(ynew, ycc[:, :, 1:3]))
    m.close()


@pytest.mark.parametrize("key,ens", [("L7_x2", 1), ("L7_x3", 1), ("L7_x4", 1), ("L7_x2", 8), ("L7_x4", 3)])
def test_evaluate_rgb_equals_the_host_pipeline(tmp_path, key, ens):
    """dcscn_evaluate_rgb (one upload) == convert_rgb_to_y + two Pillow resizes + do() on the host path."""
    from dcscn_amd import imaging as util
    g, m = _model(tmp_path, key, self_ensemble=ens)
    eng = m._ready_engine()
    for f in g["files"][:3]:
        path = os.path.join(GOLDEN, "set5", f)
        true_image, true_y, input_image, bicubic = m._evaluation_inputs(path)
        want = m.do(input_image, bicubic)
        ty, y, lr = eng.evaluate_rgb(true_image, ens, want_inputs=True)
        assert np.array_equal(ty, true_y)
        assert np.array_equal(lr, np.asarray(input_image, np.float32))
        assert y.dtype == want.dtype and np.array_equal(y, want)
        psnr_dev = m.do_for_evaluate(path)[0]                     # takes the device pipeline for RGB files
        psnr_host = util.compute_psnr_and_ssim(true_y, want, border_size=m.psnr_calc_border_size)[0]
        assert psnr_dev == psnr_host
    m.close()


def test_sr_rgb_equals_do_for_file_colour_branch(tmp_path):
    from dcscn_amd import imaging as util
    g, m = _model(tmp_path, "L7_x2")
    eng = m._ready_engine()
    for im in _images()[:3]:
        im = np.ascontiguousarray(im[:96, :80])
        scaled = util.resize_image_by_pil(im, m.scale, m.resampling_method)
        y_want = m.do(util.convert_rgb_to_y(im))
        rgb_want = util.convert_y_and_cbcr_to_rgb(y_want, util.convert_rgb_to_ycbcr(scaled)[:, :, 1:3])
        y, rgb = eng.sr_rgb(im, scaled, 1)
        assert np.array_equal(y, y_want) and np.array_equal(rgb, rgb_want)
    m.close()


def test_evaluate_rgb_rejects_unaligned_images(tmp_path):
    g, m = _model(tmp_path, "L7_x2")
    with pytest.raises(Exception) as e:
        m._ready_engine().evaluate_rgb(np.zeros((31, 40, 3), np.uint8))
    assert type(e.value).__name__ == "EngineError" and e.value.status == 1
    m.close()


def test_do_for_file_device_colour_path_writes_the_same_images(tmp_path):
    """sr.py's work (do_for_file, DCSCN.py:588-614) through dcscn_sr_rgb: same output files, same pixels as the host colour path."""
    from PIL import Image
    g, m = _model(tmp_path, "L7_x2")
    src = os.path.join(GOLDEN, "set5", g["files"][2])
    crop = tmp_path / "in.png"
    Image.open(src).convert("RGB").crop((0, 0, 64, 48)).save(crop)
    m.do_for_file(str(crop), str(tmp_path / "dev"))
    assert m._device_colour_path(np.asarray(Image.open(crop)))
    # host path: force it by pretending the resampling method differs only in name
    m._device_colour_path = lambda image: False
    m.do_for_file(str(crop), str(tmp_path / "host"))
    names = sorted(os.listdir(tmp_path / "dev" / m.name))
    assert names == sorted(os.listdir(tmp_path / "host" / m.name)) and "in_result.png" in names and "in_result_y.png" in names
    for nme in names:
        a = np.asarray(Image.open(tmp_path / "dev" / m.name / nme))
        b = np.asarray(Image.open(tmp_path / "host" / m.name / nme))
        assert np.array_equal(a, b), nme
    m.close()
