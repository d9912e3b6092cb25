#!/usr/bin/env python
"""Engine.evaluate_rgb (the device pipeline of do_for_evaluate) over the Set14 RGB images, L12 x4, self_ensemble 8: wall time per
call against the kernel time of its forwards.  For `rocprofv3 --hip-trace --stats` (which host calls the rest is spent in)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import dcscn_oracle as O
    from dcscn_amd import engine, imaging
    cfg = O.make_config(scale=4)
    eng = engine.Engine(cfg, device=0)
    eng.load_weights(O.synthetic_weights(cfg, seed=0))
    d = os.path.join(ROOT, "tests", "golden", "set14")
    imgs = []
    for f in sorted(os.listdir(d)):
        img = imaging.set_image_alignment(imaging.load_image(os.path.join(d, f), print_console=False), 4)
        if img.ndim == 3 and img.shape[2] == 3:
            imgs.append((f, img))
    for rnd in range(3):
        t0 = time.perf_counter()
        per = []
        for f, img in imgs:
            t1 = time.perf_counter()
            eng.evaluate_rgb(img, 8)
            per.append((time.perf_counter() - t1) * 1e3)
        print("round %d: %.1f ms for %d images: %s" % (rnd, (time.perf_counter() - t0) * 1e3, len(imgs), " ".join("%.1f" % p for p in per)))


if __name__ == "__main__":
    main()
