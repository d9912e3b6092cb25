#!/usr/bin/env python
"""C4-style end-to-end wall clock (BASELINE.json configs[3]): evaluate.py on the committed Set5 / Set14 images with the
L12_F196to48 x4 graph and self_ensemble=8.  The reference does not ship L12 weights, so a seeded synthetic checkpoint is
written with the TF-free checkpoint writer (ckpt.save_checkpoint) and loaded by evaluate.py exactly like a real one.
PSNR values are meaningless (random weights); the point is the wall clock of the whole path: image load, colour
conversion, both bicubic resizes, 8 flipped forwards, float64 mean, PSNR / SSIM, (optional) image saves.

    python tools/eval_wallclock.py            (on the GPU box)
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import dcscn_oracle as O
    from dcscn_amd import ckpt
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "models"))
    for ds in ("set5", "set14"):
        shutil.copytree(os.path.join(ROOT, "tests", "golden", ds), os.path.join(tmp, "data", ds))
    runs = [("dcscn_L12_F196to48_Sc4_NIN_A64_PS_R1F32", dict(scale=4), ["--scale=4"]),
            ("dcscn_L12_F196to48_NIN_A64_PS_R1F32", dict(), ["--scale=2"])]
    for name, flags, cli in runs:
        cfg = O.make_config(**flags)
        ckpt.save_checkpoint(os.path.join(tmp, "models", name + ".ckpt"), O.synthetic_weights(cfg, seed=0))
        for dataset in ("set5", "set14"):
            n = len(os.listdir(os.path.join(tmp, "data", dataset)))
            for ens in (8, 1):
                for save in ("false", "true"):
                    cmd = [sys.executable, os.path.join(ROOT, "evaluate.py"), "--test_dataset=" + dataset, "--self_ensemble=%d" % ens,
                           "--save_results=" + save, "--checkpoint_dir=" + os.path.join(tmp, "models"), "--data_dir=" + os.path.join(tmp, "data"),
                           "--output_dir=" + os.path.join(tmp, "out"), "--log_filename=" + os.path.join(tmp, "log.txt")] + cli
                    t0 = time.time()
                    p = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                    dt = time.time() - t0
                    line = [ln for ln in p.stdout.splitlines() if "Model Average" in ln]
                    print("%-44s %-5s ensemble %d save_results=%-5s  %6.2f s wall (%d images, incl. python start + model build)  %s"
                          % (name, dataset, ens, save, dt, n, line[-1].split("] ", 1)[-1] if line else "FAILED rc=%d\n%s" % (p.returncode, p.stdout[-800:])),
                          flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
