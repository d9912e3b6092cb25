"""Debug: F_1 (input) and F_2 (output) of the CNN2 conv3_h launch for many forwards under multi-process churn."""
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
n = 32
reps = int(os.environ["DCSCN_DUMP_COUNT"])
x = rng.uniform(0, 255, (n, 48, 48, 1)).astype(np.float32)
x2 = rng.uniform(0, 255, (n, 96, 96, 1)).astype(np.float32)
xd, x2d = torch.from_numpy(x).cuda(), torch.from_numpy(x2).cuda()
yd = torch.empty_like(x2d)
stream = torch.cuda.Stream()
torch.cuda.synchronize()
with engine.Engine(cfg, device=0) as eng:
    eng.load_weights(weights)
    eng.set_option("split16", int(os.environ.get("DET_MODE", "2")))
    if os.path.exists(dump):
        os.remove(dump)
    for r in range(reps):
        eng.forward_device(xd.data_ptr(), x2d.data_ptr(), yd.data_ptr(), n, 48, 48, stream.cuda_stream)
    eng.synchronize()
raw = np.fromfile(dump, np.float32)
per = n * 2304 * (196 + 168)
raw = raw.reshape(reps, per)
f1 = raw[:, :n * 2304 * 196].reshape(reps, n, 48, 48, 196)
f2 = raw[:, n * 2304 * 196:].reshape(reps, n, 48, 48, 168)
# majority reference: the most common digest
d2 = [hashlib.sha256(f2[r].tobytes()).hexdigest()[:8] for r in range(reps)]
d1 = [hashlib.sha256(f1[r].tobytes()).hexdigest()[:8] for r in range(reps)]
print("F1 digests", d1)
print("F2 digests", d2)
ref = max(set(d2), key=d2.count)
r0 = d2.index(ref)
for r in range(reps):
    if d2[r] == ref:
        continue
    diff = f2[r] != f2[r0]
    idx = np.argwhere(diff)
    print("run %d: F1 %s; F2: %d elements differ, max %.3g" % (r, "same" if d1[r] == d1[r0] else "DIFFERENT", len(idx), np.abs(f2[r] - f2[r0]).max()))
    d1i = np.argwhere(f1[r] != f1[r0])
    if len(d1i):
        print("   F1: %d elements differ, max %.3g; images %s" % (len(d1i), np.abs(f1[r] - f1[r0]).max(), sorted(set(d1i[:, 0].tolist()))[:10]))
        pix = sorted(set((a, b, c) for a, b, c, _ in d1i.tolist()))
        print("   F1 pixels (img,row,col) [%d]: %s" % (len(pix), pix[:14]))
        a, b, c = pix[0]
        chs = d1i[(d1i[:, 0] == a) & (d1i[:, 1] == b) & (d1i[:, 2] == c)][:, 3]
        print("   first pixel: channels %d..%d (%d); ref %s  got %s" % (chs.min(), chs.max(), len(chs), f1[r0][a, b, c, chs[:5]], f1[r][a, b, c, chs[:5]]))
        # does the wrong pixel hold another pixel's data?
        got = f1[r][a, b, c]
        match = np.argwhere(np.all(np.isclose(f1[r0], got, rtol=0, atol=0), axis=3))
        print("   pixels of the reference F1 equal to the wrong record:", match[:4].tolist())
    for img in sorted(set(idx[:, 0].tolist()))[:2]:
        s = idx[idx[:, 0] == img]
        print("   img %d rows %d-%d cols %d-%d ch %d-%d (%d); distinct rows %s cols %s" % (img, s[:, 1].min(), s[:, 1].max(), s[:, 2].min(), s[:, 2].max(),
              s[:, 3].min(), s[:, 3].max(), len(s), sorted(set(s[:, 1].tolist()))[:24], sorted(set(s[:, 2].tolist()))[:24]))
