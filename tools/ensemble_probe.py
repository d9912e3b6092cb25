"""Cost split of dcscn_forward_ensemble (C4: L12 x4, self_ensemble 8) on one BSD100-sized image."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as O
from dcscn_amd import engine
cfg = O.make_config(scale=4)
eng = engine.Engine(cfg)
eng.load_weights(O.synthetic_weights(cfg, seed=0))
rng = np.random.default_rng(0)
for (h, w) in [(80, 120), (128, 128), (64, 64)]:
    x = rng.uniform(0, 255, (h, w)).astype(np.float32)
    x2 = rng.uniform(0, 255, (4 * h, 4 * w)).astype(np.float32)
    eng.forward_ensemble(x, x2, 8)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward_ensemble(x, x2, 8)
    te = (time.perf_counter() - t0) / 5
    xb = np.repeat(x[None, :, :, None], 8, 0); x2b = np.repeat(x2[None, :, :, None], 8, 0)
    eng.forward(xb, x2b)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward(xb, x2b)
    tf = (time.perf_counter() - t0) / 5
    print("LR %dx%d x4 ens8: forward_ensemble %.2f ms; plain batch-8 forward (no flips) %.2f ms" % (h, w, te * 1e3, tf * 1e3), flush=True)
