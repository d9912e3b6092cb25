#!/usr/bin/env python
"""Where the per-image time of evaluate.py goes for BASELINE configs[3] (L12 x4, self_ensemble 8, Set14): cProfile over
do_for_evaluate with a synthetic L12 x4 checkpoint (GPU box).  python tools/eval_profile_c4.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import dcscn_oracle as O
    from test_host import _flags
    from dcscn_amd.model import SuperResolution
    cfg = O.make_config(scale=4)
    m = SuperResolution(_flags(scale=4, self_ensemble=8, checkpoint_dir="/tmp"))
    m.build_graph()
    m.init_all_variables()
    m.load_weights(O.synthetic_weights(cfg, seed=0))
    d = os.path.join(ROOT, "tests", "golden", "set14")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d))]
    m.do_for_evaluate(files[0])
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.time()
    for f in files:
        m.do_for_evaluate(f)
    dt = time.time() - t0
    pr.disable()
    print("%d images, %.1f ms per image (serial do_for_evaluate loop, under cProfile)" % (len(files), dt / len(files) * 1e3))
    serial = []
    t0 = time.time()
    for f in files:
        serial.append(m.do_for_evaluate(f))
    dt_serial = time.time() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        piped = m.do_for_evaluate_many(files)
        best = min(best, time.time() - t0)
    same = all(a[0] == b[0] and a[1] == b[1] for a, b in zip(serial, piped))
    print("serial loop %.1f ms per image; do_for_evaluate_many (decode / device / metrics pipelined) %.1f ms per image; PSNR and SSIM identical: %s"
          % (dt_serial / len(files) * 1e3, best / len(files) * 1e3, same))
    if "--profile" in sys.argv:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
