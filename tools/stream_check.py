"""feat_stream (option stream_features) against the layer-by-layer launches of the same library, and against the float64
oracle: max-abs differences for several image shapes (single strip, column strips, row blocks), then timing."""
import os
import sys
import time

import numpy as np
import torch  # before the library: torch must initialise HIP first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402  (registers the dcscn_amd alias)
from dcscn_amd import engine  # noqa: E402
if os.environ.get("DCSCN_LIB"):
    from dcscn_amd import build as _b
    _b.LIB_PATH = os.environ["DCSCN_LIB"]

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as oracle  # noqa: E402

cfg = oracle.make_config(**conftest.CONFIGS["L7_F32to8_x4_DS"])
weights = oracle.synthetic_weights(cfg, seed=5)
torch.zeros(1, device="cuda")
shapes = [(3, 48, 48), (2, 40, 36), (1, 7, 5), (1, 100, 130), (1, 300, 70), (5, 64, 49)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
TIMING_ONLY = bool(os.environ.get("DCSCN_LIB"))
engs = {}
for mode in ((1,) if TIMING_ONLY else (1, 0)):
    e = engine.Engine(cfg, device=0)
    e.set_option("stream_features", mode)
    e.set_option("stream_tail", mode)
    e.load_weights(weights)
    engs[mode] = e
print("ops streamed:", [o["kernel"] for o in engs[1].ops()])
for n, h, w in ([] if TIMING_ONLY else shapes):
    x, x2 = conftest.synthetic_batch(n, h, w, cfg["scale"], seed=n + h)
    y1 = engs[1].forward(x, x2)
    y0 = engs[0].forward(x, x2)
    ref = oracle.forward(cfg, weights, x, x2, dtype=np.float64) if n * h * w < 20000 else None
    d = np.abs(y1.astype(np.float64) - y0)
    msg = "n=%d %dx%d  streamed vs layered max %.3g" % (n, h, w, d.max())
    if ref is not None:
        msg += "   vs oracle: streamed %.3g layered %.3g" % (np.abs(y1 - ref).max(), np.abs(y0 - ref).max())
    if d.max() > 1e-3:
        bad = np.argwhere(d > 1e-3)
        msg += "   BAD at %s ... (%d px)" % (bad[0], len(bad))
    print(msg, flush=True)
s4 = cfg["scale"]
tx = torch.rand((1024, 48, 48, 1), device="cuda") * 255
tx2 = torch.rand((1024, 48 * s4, 48 * s4, 1), device="cuda") * 255
ty = torch.empty_like(tx2)
st = torch.cuda.current_stream().cuda_stream
for mode in ((1,) if TIMING_ONLY else (1, 0)):
    e = engs[mode]
    for _ in range(3):
        e.forward_device(tx.data_ptr(), tx2.data_ptr(), ty.data_ptr(), 1024, 48, 48, st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        e.forward_device(tx.data_ptr(), tx2.data_ptr(), ty.data_ptr(), 1024, 48, 48, st)
    torch.cuda.synchronize()
    print("stream_features=%d: %.3f ms / 1024 patches" % (mode, (time.perf_counter() - t) * 100))
    e.set_option("profile", 1)
    e.forward_device(tx.data_ptr(), tx2.data_ptr(), ty.data_ptr(), 1024, 48, 48, st)
    torch.cuda.synchronize()
    for o, ms in zip(e.ops(), e.profile()):
        print("    %-34s %-12s %.3f ms" % (o["name"], o["kernel"], ms))
