"""Seeded random walk over the flag surface of helper/args.py x image sizes x engine options (Winograd on/off, folded
tail on/off, forced spatial tiling): every draw must meet the 1e-4 bar against the float64 oracle.  Complements the
hand-picked cases of test_hip_parity.py; channel counts are chosen so that 3k+1-tile layers (tail launches), 4-tile
layers, partial channel tiles and multi-group Winograd layers all occur."""
import numpy as np
import pytest

from conftest import synthetic_batch

pytestmark = pytest.mark.gpu


def _draw(rng):
    scale = int(rng.choice([2, 2, 3, 4]))
    ds = bool(rng.random() < 0.2)
    flags = dict(
        scale=scale,
        layers=int(rng.integers(1, 6)),
        filters=int(rng.choice([4, 9, 24, 37, 52, 57, 66, 97, 112, 148])),
        min_filters=int(rng.choice([1, 4, 8, 20, 48])),
        filters_decay_gamma=float(rng.choice([1.0, 1.2, 1.5, 2.0])),
        cnn_size=int(rng.choice([3, 3, 3, 3, 1, 5, 7])),
        use_nin=bool(rng.random() < 0.7),
        nin_filters=int(rng.choice([4, 9, 24, 64])),
        nin_filters2=int(rng.choice([3, 8, 32])),
        reconstruct_layers=int(rng.choice([0, 1, 1, 2, 3])),
        reconstruct_filters=int(rng.choice([4, 12, 32])),
        activator=str(rng.choice(["prelu", "prelu", "relu", "leaky_relu"])),
        pixel_shuffler=bool(rng.random() < 0.85),
        pixel_shuffler_filters=int(rng.choice([0, 0, 1, 5, 16])),
        depthwise_separable=ds,
    )
    flags["min_filters"] = min(flags["min_filters"], flags["filters"])
    if flags["cnn_size"] == 7:
        flags["filters"] = min(flags["filters"], 66)
    h, w = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    n = int(rng.integers(1, 4))
    opts = dict(winograd=bool(rng.random() < 0.7), fold=bool(rng.random() < 0.5), tile=bool(rng.random() < 0.25))
    return flags, n, h, w, opts


def _run_draw(oracle, seed, split16):
    from dcscn_amd import engine
    rng = np.random.default_rng(1000 + seed)
    flags, n, h, w, opts = _draw(rng)
    cfg = oracle.make_config(**flags)
    weights = oracle.synthetic_weights(cfg, seed=seed)
    x, x2 = synthetic_batch(n, h, w, cfg["scale"], seed=seed + 1)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights, winograd=opts["winograd"], fold_tail=opts["fold"], split16=split16)
        y = eng.forward(x, x2)
        if opts["tile"] and h * w >= 600:
            per_px = eng.workspace_bytes() // (n * h * w) + 1
            eng.set_option("workspace_budget_bytes", per_px * (h * w // 2))
            try:
                yt = eng.forward(x, x2)
            except engine.EngineError:
                yt = None                           # windows smaller than the halo: a reported error, not a crash
            if yt is not None:
                assert float(np.max(np.abs(yt - ref))) <= 1e-4, (flags, n, h, w, opts)
    err = float(np.max(np.abs(y - ref)))
    assert np.isfinite(y).all() and err <= 1e-4, (err, flags, n, h, w, opts)


@pytest.mark.parametrize("seed", range(200))
def test_random_flag_surface(oracle, seed):
    _run_draw(oracle, seed, None)                   # library default: split16 on


@pytest.mark.parametrize("seed", range(60))
def test_random_flag_surface_f32_kernels(oracle, seed):
    """The first 60 draws again with split16 off: conv_wino2 / conv_nin / conv_igemm meet every drawn flag combination too."""
    _run_draw(oracle, seed, False)


@pytest.mark.parametrize("seed", range(16))
def test_random_self_ensemble(oracle, seed):
    """Device-side flip gather + inverse-flip float64 mean (ensemble.hip) against the oracle's do() for random
    non-square shapes, scales and ensemble sizes; each flip type must also be exact on its own."""
    from dcscn_amd import engine
    rng = np.random.default_rng(2000 + seed)
    scale = int(rng.choice([2, 3, 4]))
    cfg = oracle.make_config(layers=2, filters=12, min_filters=8, scale=scale, nin_filters=6, nin_filters2=5)
    weights = oracle.synthetic_weights(cfg, seed=seed)
    h, w = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    n_ens = int(rng.integers(1, 9))
    x, x2 = synthetic_batch(1, h, w, scale, seed=seed + 3)
    ref = oracle.do(cfg, weights, x[0], x2[0], self_ensemble=n_ens, dtype=np.float64)
    with engine.Engine(cfg, device=0) as eng:
        eng.load_weights(weights)
        y = eng.forward_ensemble(x[0], x2[0], n_ens)
    assert y.dtype == np.float64 and y.shape == ref.shape
    assert float(np.max(np.abs(y - ref))) <= 1e-4, (h, w, scale, n_ens)
