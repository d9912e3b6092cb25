"""dcscn_resize_bicubic / dcscn_forward_lr against Pillow itself (the reference's util.resize_image_by_pil,
helper/utilty.py:211-239, goes through Image.resize(BICUBIC) on mode-'F' images).  The bar is BIT equality:
same coefficient tables, same float64 summation order, one rounding to float32 per pass."""
import numpy as np
import pytest
from PIL import Image

from conftest import CONFIGS

pytestmark = pytest.mark.gpu


def _pil(img, oh, ow):
    return np.asarray(Image.fromarray(np.ascontiguousarray(img, dtype=np.float32)).resize([ow, oh], resample=Image.BICUBIC))


@pytest.fixture(scope="module")
def eng(oracle):
    from dcscn_amd import engine
    cfg = oracle.make_config(layers=2, filters=8, min_filters=8)
    e = engine.Engine(cfg, device=0)
    yield e
    e.close()


@pytest.mark.parametrize("shape,out", [
    ((48, 48), (96, 96)), ((48, 48), (144, 144)), ((48, 48), (192, 192)),          # the x2 / x3 / x4 residual inputs
    ((17, 33), (34, 66)), ((1, 1), (2, 2)), ((1, 7), (4, 28)), ((5, 1), (15, 3)),
    ((96, 96), (48, 48)), ((99, 120), (33, 40)), ((128, 64), (32, 16)),            # antialiased shrinking (LR building)
    ((37, 53), (37, 106)), ((37, 53), (74, 53)), ((37, 53), (37, 53)),             # one pass only / identity
    ((50, 70), (61, 23)),                                                          # up in one axis, down in the other
])
def test_resize_is_bit_identical_to_pillow(eng, shape, out):
    rng = np.random.default_rng(shape[0] * 1000 + out[1])
    imgs = rng.uniform(0, 255, (3,) + shape).astype(np.float32)
    got = eng.resize_bicubic(imgs, out[0], out[1])
    for i in range(3):
        ref = _pil(imgs[i], out[0], out[1])
        assert got[i].shape == ref.shape
        assert np.array_equal(got[i], ref), "max diff %g" % float(np.max(np.abs(got[i] - ref)))


def test_resize_extreme_values(eng):
    img = np.array([[0, 255, 0, 255], [255, 0, 255, 0], [1e-30, -5.5, 300.25, 16.0]], np.float32)
    assert np.array_equal(eng.resize_bicubic(img, 6, 8), _pil(img, 6, 8))
    assert np.array_equal(eng.resize_bicubic(img, 9, 12), _pil(img, 9, 12))


@pytest.mark.parametrize("name", ["L7_F32to8_x2", "L7_F32to8_x3", "L7_F32to8_x4"])
def test_forward_lr_equals_forward_with_pillow_bicubic(oracle, name):
    """do(input, bicubic=None): the device-side bicubic must give the very same y as feeding Pillow's."""
    from dcscn_amd import engine
    cfg = oracle.make_config(**CONFIGS[name])
    weights = oracle.synthetic_weights(cfg, seed=1)
    s = cfg["scale"]
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 255, (2, 21, 30, 1)).astype(np.float32)
    x2 = np.stack([_pil(x[i, :, :, 0], 21 * s, 30 * s) for i in range(2)])[..., None]
    with engine.Engine(cfg, device=0) as e:
        e.load_weights(weights)
        assert np.array_equal(e.forward_lr(x), e.forward(x, x2))
