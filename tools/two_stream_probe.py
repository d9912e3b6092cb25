"""Does running two half batches on two streams (tails of one overlapping the other) beat one stream?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import dcscn_oracle as O
from dcscn_amd import engine
cfg = O.make_config()
w = O.synthetic_weights(cfg, seed=0)
def mk():
    e = engine.Engine(cfg); e.load_weights(w); return e
n = 1024
x = torch.rand((n, 48, 48, 1), device="cuda") * 255; x2 = torch.rand((n, 96, 96, 1), device="cuda") * 255; y = torch.empty_like(x2)
e0 = mk()
st = torch.cuda.current_stream().cuda_stream
def one():
    e0.forward_device(x.data_ptr(), x2.data_ptr(), y.data_ptr(), n, 48, 48, st)
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): one()
torch.cuda.synchronize(); print("one stream, 1024 patches: %.2f ms" % ((time.perf_counter() - t0) * 100))
for parts in (2, 4):
    engs = [mk() for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    m = n // parts
    def multi():
        for i, (e, s) in enumerate(zip(engs, streams)):
            o = i * m
            e.forward_device(x[o:o + m].data_ptr(), x2[o:o + m].data_ptr(), y[o:o + m].data_ptr(), m, 48, 48, s.cuda_stream)
    for _ in range(3): multi()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): multi()
    torch.cuda.synchronize(); print("%d streams x %d patches: %.2f ms" % (parts, m, (time.perf_counter() - t0) * 100))
