"""Debug: dump the outputs of every split16 launch for several forwards under load and report which layer first differs."""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as O
from dcscn_amd import engine
import torch
dump = os.environ["DCSCN_DUMP_C3H"]
cfg = O.make_config()
weights = O.synthetic_weights(cfg, seed=0)
rng = np.random.default_rng(5)
n = int(os.environ.get('DET_N', '32'))
reps = int(os.environ.get('DET_REPS', '8'))
x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
x2 = rng.uniform(0, 255, (n, 96, 96, 1)).astype(np.float32)
xd, x2d = torch.from_numpy(x).cuda(), torch.from_numpy(x2).cuda()
yd = torch.empty_like(x2d)
stream = torch.cuda.Stream()
torch.cuda.synchronize()
ys = []
with engine.Engine(cfg, device=0) as eng:
    eng.load_weights(weights)
    eng.set_option("split16", int(os.environ.get("DET_MODE", "2")))
    ops = [o for o in eng.ops() if o["kernel"] in ("conv3_h", "conv_nin_h")]
    if os.path.exists(dump):
        os.remove(dump)
    for r in range(reps):
        eng.forward_device(xd.data_ptr(), x2d.data_ptr(), yd.data_ptr(), n, 48, 48, stream.cuda_stream)
        eng.synchronize()
        torch.cuda.synchronize()
        ys.append(hashlib.sha256(yd.cpu().numpy().tobytes()).hexdigest()[:10])
print("y digests:", ys)
if not os.path.exists(dump):
    sys.exit(0)
raw = np.fromfile(dump, np.float32)
strides = []
for o in ops:
    c = o["out_channels"]
    if os.environ.get("DCSCN_DUMP_INPUT"):
        c = o["in_channels"]
        strides.append((c + 3) // 4 * 4)
    else:
        strides.append(96 if o["name"].startswith("B2") else (c + 3) // 4 * 4)
per_run = sum(n * 2304 * s for s in strides)
print("ops", [(o["name"], s) for o, s in zip(ops, strides)], "floats per run", per_run, "file", raw.size, raw.size / per_run)
runs = raw[:per_run * (raw.size // per_run)].reshape(-1, per_run)
for r in range(1, runs.shape[0]):
    off = 0
    msgs = []
    for o, s in zip(ops, strides):
        cnt = n * 2304 * s
        a0, ar = runs[0][off:off + cnt], runs[r][off:off + cnt]
        nd = int((a0 != ar).sum())
        if nd:
            d = np.abs(a0 - ar)
            idx = np.argwhere((a0 != ar).reshape(n, 48, 48, s))
            msgs.append("%s: %d differ (max %.3g) imgs %s rows %d-%d cols %d-%d ch %d-%d" % (
                o["name"], nd, d.max(), sorted(set(idx[:, 0].tolist()))[:4], idx[:, 1].min(), idx[:, 1].max(), idx[:, 2].min(), idx[:, 2].max(), idx[:, 3].min(), idx[:, 3].max()))
        off += cnt
    print("run %d vs 0: %s" % (r, "identical" if not msgs else " | ".join(msgs[:3])))
