"""can the HBM-bound launches of the L12 pass (first layer, wide 1x1 GEMM, folded tail: ~3.9 of ~19.4 ms, matrix pipe
idle) run in the shadow of the matrix-bound 3x3 stack of ANOTHER half batch?

  DCSCN_CU_SPLIT=k  (library hook of the experiment commit, since removed): the HBM-bound launches of a handle go to a stream masked to k compute
  units (k / 8 per XCD), the 3x3 launches to a stream masked to the other 256 - k (conv3_h8's persistent grid shrinks to match).
Two handles, 512 patches each, enqueued alternately; reference = one handle, 1024 patches, one stream, no masks.

  python tools/overlap_probe.py [k ...]        (each k runs in a fresh process: the masks are made at dcscn_create)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(k, parts):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import dcscn_oracle as O
    from dcscn_amd import engine
    cfg = O.make_config()
    w = O.synthetic_weights(cfg, seed=0)
    n = 1024
    x = torch.rand((n, 48, 48, 1), device="cuda") * 255
    x2 = torch.rand((n, 96, 96, 1), device="cuda") * 255
    y = torch.empty_like(x2)
    engs = []
    for _ in range(parts):
        e = engine.Engine(cfg)
        e.load_weights(w)
        engs.append(e)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    m = n // parts

    def step():
        for i, (e, s) in enumerate(zip(engs, streams)):
            o = i * m
            e.forward_device(x[o:o + m].data_ptr(), x2[o:o + m].data_ptr(), y[o:o + m].data_ptr(), m, 48, 48, s.cuda_stream)

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 100)
    print("CU split %3d, %d handle(s) x %4d patches: %.2f ms per 1024 patches   (checksum %.6g)" % (k, parts, m, best, float(y.double().sum())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    ks = [int(a) for a in sys.argv[1:]] or [0, 16, 32, 48, 64, 96]
    for k in ks:
        for parts in ((1, 2) if k == 0 else (1, 2, 4)):
            env = dict(os.environ)
            if k:
                env["DCSCN_CU_SPLIT"] = str(k)
            else:
                env.pop("DCSCN_CU_SPLIT", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(k), str(parts)], env=env, check=False)
