"""Where the host-buffer entry point spends its time (DCSCN_TRACE_HOST=1 makes the library print its own split)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as O
from dcscn_amd import engine
cfg = O.make_config()
eng = engine.Engine(cfg)
eng.load_weights(O.synthetic_weights(cfg, seed=0))
n = 1024
rng = np.random.default_rng(0)
x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
x2 = rng.uniform(0, 255, (n, 96, 96, 1)).astype(np.float32)
for i in range(3):
    t0 = time.perf_counter()
    y = eng.forward(x, x2)
    print("python forward(): %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
