#!/usr/bin/env python
"""fold_whole_tail (x3 / x4: the whole tail as one 5x5 conv + a border-ring launch) against the float64 oracle: bare residual branch on ragged
sizes down to one pixel, both plans; then the BASELINE nets.   python tools/foldx_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcscn_oracle as O
from conftest import CONFIGS, synthetic_batch
from dcscn_amd import engine

worst = 0.0
for scale in (3, 4):
    for extra in (dict(pixel_shuffler_filters=5), dict(pixel_shuffler_filters=1), dict(pixel_shuffler_filters=5, depthwise_separable=True)):
        cfg = O.make_config(layers=3, filters=16, min_filters=8, scale=scale, **extra)
        weights = O.synthetic_weights(cfg, seed=20 + scale)
        key = [k for k in weights if k.startswith("R-CNN1/") and k.endswith("_W")]
        for k in key[:1]:
            weights[k] = weights[k] * 100.0
        for hw in ((1, 1), (1, 6), (7, 1), (2, 2), (2, 9), (3, 5), (16, 16), (17, 33), (48, 48), (35, 3)):
            x, _ = synthetic_batch(3, hw[0], hw[1], scale, seed=30)
            x2 = np.zeros((3, hw[0] * scale, hw[1] * scale, 1), np.float32)
            ref = O.forward(cfg, weights, x, x2, dtype=np.float64)
            mag = float(np.max(np.abs(ref)))
            out = []
            for whole in (1, 0):
                with engine.Engine(cfg, device=0) as eng:
                    eng.set_option("fold_whole_tail", whole)
                    eng.load_weights(weights)
                    names = [(o["name"], o["kernel"]) for o in eng.ops()][-2:]
                    y = eng.forward(x, x2)
                out.append(float(np.max(np.abs(y - ref))) / mag)
            worst = max(worst, out[0])
            flag = "" if out[0] <= 1e-5 else "   <-- FAIL"
            print("x%d %s %dx%d: whole-tail fold rel %.3g, r05 plan %.3g  %s%s" % (scale, extra, hw[0], hw[1], out[0], out[1], names, flag), flush=True)
print("worst relative error of the whole-tail fold: %.3g" % worst)
for name in ("L7_F32to8_x3", "L7_F32to8_x4", "L7_F32to8_x4_DS", "L12_F196to48_x4"):
    cfg = O.make_config(**CONFIGS[name])
    weights = O.synthetic_weights(cfg, seed=11)
    x, x2 = synthetic_batch(2, 48, 48, cfg["scale"], seed=12)
    ref = O.forward(cfg, weights, x, x2, dtype=np.float64)
    for whole in (1, 0):
        with engine.Engine(cfg, device=0) as eng:
            eng.set_option("fold_whole_tail", whole)
            eng.load_weights(weights)
            ops = [(o["name"], o["kernel"]) for o in eng.ops()]
            y = eng.forward(x, x2)
        print("%s fold_whole_tail %d: max-abs %.3g  %s" % (name, whole, float(np.max(np.abs(y - ref))), ops[-3:]), flush=True)
