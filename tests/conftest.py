import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The model configurations of BASELINE.json (flag names of helper/args.py).
CONFIGS = {
    # C1: shipped toy checkpoint, legacy graph without the 1x1 "C" layer
    "L2_F4to4_x2": dict(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4, legacy_no_c=True),
    # C2
    "L8_F96to48_x2": dict(layers=8, filters=96),
    # C3 (defaults of helper/args.py)
    "L12_F196to48_x2": dict(),
    # C4
    "L12_F196to48_x4": dict(scale=4),
    # shipped c-DCSCN checkpoints (README.md:86)
    "L7_F32to8_x2": dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8,
                         reconstruct_layers=0, pixel_shuffler_filters=1),
    "L7_F32to8_x3": dict(scale=3, layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24,
                         nin_filters2=8, reconstruct_layers=0, pixel_shuffler_filters=1),
    "L7_F32to8_x4": dict(scale=4, layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24,
                         nin_filters2=8, reconstruct_layers=0, pixel_shuffler_filters=1),
    # C5
    "L7_F32to8_x4_DS": dict(scale=4, layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24,
                            nin_filters2=8, reconstruct_layers=0, pixel_shuffler_filters=1, depthwise_separable=True),
}


@pytest.fixture(scope="session")
def oracle():
    import dcscn_oracle
    return dcscn_oracle


def synthetic_batch(n, h, w, scale, seed=0):
    """x ~ U(0, 255); x2 = PIL bicubic upscale of x per patch."""
    import dcscn_oracle as O
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 255, (n, h, w, 1)).astype(np.float32)
    x2 = np.stack([O.pil_bicubic(x[i], scale) for i in range(n)]).astype(np.float32)
    return x, x2
