import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import dcscn_oracle as O
from dcscn_amd import engine
L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8, reconstruct_layers=0, pixel_shuffler_filters=1)
for scale in (3, 4):
  for fold in (1, 2):
    cfg = O.make_config(**dict(L7, scale=scale))
    eng = engine.Engine(cfg)
    eng.set_option("fold_linear_tail", fold)
    eng.load_weights(O.synthetic_weights(cfg, seed=0))
    n = 1024; s = scale
    x = torch.rand((n, 48, 48, 1), device="cuda") * 255
    x2 = torch.rand((n, 48 * s, 48 * s, 1), device="cuda") * 255
    y = torch.empty_like(x2)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, 48, 48, st)
    eng.set_option("profile", 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, 48, 48, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    ms = eng.profile(); ops = eng.ops()
    print("x%d fold=%d: %.3f ms/step; " % (scale, fold, dt * 1e3) + ", ".join("%s %s %.3f" % (o["name"][:22], o["kernel"], m) for o, m in zip(ops, ms)))
