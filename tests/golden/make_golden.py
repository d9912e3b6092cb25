"""Generates the committed fixtures under tests/golden/ from the reference tree (run in the build
container, where /root/reference is mounted; the GPU box never sees that path):

  models/dcscn_L2_F4to4_PS_R1F4.ckpt.*   the shipped toy checkpoint, verbatim (TF bundle reader test,
                                         legacy no-"C" topology)
  weights_<model>.npz                    inference variables of the shipped c-DCSCN checkpoints
  set5/img_00N.png                       the Set5 images shipped in the reference's data/set5
  goldens.json                           float64-oracle PSNR per image (do_for_evaluate recipe,
                                         DCSCN.py:672-703), bicubic PSNR, and a 48x48-crop output vector
  crop_L7_x2.npz                         LR crop, its bicubic, and the float64 oracle output

The oracle that produces the numbers is pinned by the README PSNR table (README.md:57-62):
Set5 x2/x3/x4 = 37.15 / 33.09 / 30.85 for these checkpoints.
"""
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dcscn_oracle as O           # noqa: E402
from dcscn_amd import ckpt          # noqa: E402

REF = "/root/reference"
L7 = dict(layers=7, filters=32, min_filters=8, filters_decay_gamma=1.2, nin_filters=24, nin_filters2=8,
          reconstruct_layers=0, pixel_shuffler_filters=1)
MODELS = {
    "L7_x2": (dict(L7), "dcscn_L7_F32to8_G1.20_NIN_A24_B8_PS_R1F32"),
    "L7_x3": (dict(L7, scale=3), "dcscn_L7_F32to8_G1.20_Sc3_NIN_A24_B8_PS_R1F32"),
    "L7_x4": (dict(L7, scale=4), "dcscn_L7_F32to8_G1.20_Sc4_NIN_A24_B8_PS_R1F32"),
    "L7_x4_DS": (dict(L7, scale=4, depthwise_separable=True), "dcscn_L7_F32to8_G1.20_Sc4_NIN_A24_B8_PS_DS_R1F32"),
    "L2_x2": (dict(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4, legacy_no_c=True),
              "dcscn_L2_F4to4_PS_R1F4"),
}


def main():
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copy(os.path.join(REF, "models", "dcscn_L2_F4to4_PS_R1F4.ckpt" + ext), os.path.join(HERE, "models"))
    files = sorted(os.listdir(os.path.join(REF, "data", "set5")))
    for f in files:
        shutil.copy(os.path.join(REF, "data", "set5", f), os.path.join(HERE, "set5", f))
    images = [np.atleast_3d(np.array(Image.open(os.path.join(HERE, "set5", f)))) for f in files]

    goldens = {"files": files, "models": {}, "bicubic": {}}
    for key, (overrides, name) in MODELS.items():
        cfg = O.make_config(**overrides)
        tensors = ckpt.load_checkpoint(os.path.join(REF, "models", name + ".ckpt"))
        needed = O.variable_shapes(cfg)
        weights = {k: tensors[k] for k in needed}
        assert all(weights[k].shape == v for k, v in needed.items())
        np.savez_compressed(os.path.join(HERE, "weights_%s.npz" % key), **weights)
        psnrs = [O.evaluate_image(cfg, weights, img)[0] for img in images]
        entry = {"checkpoint": name, "flags": overrides, "set5_psnr": psnrs, "set5_mean": float(np.mean(psnrs))}
        if key == "L7_x2":
            entry["set5_psnr_ensemble8"] = [O.evaluate_image(cfg, weights, img, self_ensemble=8)[0] for img in images]
            entry["set5_mean_ensemble8"] = float(np.mean(entry["set5_psnr_ensemble8"]))
            # one 48x48 LR crop with its full float64 output
            _, lr, bic, _ = O.evaluate_image(cfg, weights, images[2])
            lr_c = np.ascontiguousarray(lr[40:88, 30:78])
            bic_c = O.pil_bicubic(lr_c, 2)
            out = O.forward(cfg, weights, lr_c[None], bic_c[None])[0]
            np.savez_compressed(os.path.join(HERE, "crop_L7_x2.npz"), lr=lr_c.astype(np.float32),
                                bicubic=bic_c.astype(np.float32), output=out)
        goldens["models"][key] = entry
        print(key, entry["set5_mean"])
    for s in (2, 3, 4):
        vals = []
        for img in images:
            t = O.align(img, s)
            y = O.rgb_to_y(t)
            vals.append(O.psnr_y(y, O.pil_bicubic(O.pil_bicubic(y, 1.0 / s), s), s))
        goldens["bicubic"]["x%d" % s] = vals
        print("bicubic x%d" % s, np.mean(vals))
    with open(os.path.join(HERE, "goldens.json"), "w") as f:
        json.dump(goldens, f, indent=1)


def set14():
    """Adds the reference's data/set14 images and the float64-oracle PSNR of the shipped c-DCSCN x2 / x3 / x4
    checkpoints on them (README.md:60-62: 32.74 / 29.47 / 27.76 dB) to goldens.json, keeping everything else."""
    os.makedirs(os.path.join(HERE, "set14"), exist_ok=True)
    files = sorted(os.listdir(os.path.join(REF, "data", "set14")))
    for f in files:
        shutil.copy(os.path.join(REF, "data", "set14", f), os.path.join(HERE, "set14", f))
    images = [np.atleast_3d(np.array(Image.open(os.path.join(HERE, "set14", f)))) for f in files]
    with open(os.path.join(HERE, "goldens.json")) as f:
        goldens = json.load(f)
    entry = {"files": files}
    for key in ("L7_x2", "L7_x3", "L7_x4"):
        cfg = O.make_config(**MODELS[key][0])
        weights = dict(np.load(os.path.join(HERE, "weights_%s.npz" % key)))
        psnrs = [O.evaluate_image(cfg, weights, img)[0] for img in images]
        entry[key] = {"psnr": psnrs, "mean": float(np.mean(psnrs))}
        print("set14", key, entry[key]["mean"], flush=True)
    goldens["set14"] = entry
    with open(os.path.join(HERE, "goldens.json"), "w") as f:
        json.dump(goldens, f, indent=1)


def bsd100():
    """Oracle PSNR on the reference's data/bsd100 (README.md:63-65: 31.61 / 28.52 / 27.06 dB).  The 100 images are NOT
    copied (too large for a fixture); only the per-image numbers are recorded, as one more pin of the oracle."""
    d = os.path.join(REF, "data", "bsd100")
    files = sorted(os.listdir(d))
    images = [np.atleast_3d(np.array(Image.open(os.path.join(d, f)))) for f in files]
    with open(os.path.join(HERE, "goldens.json")) as f:
        goldens = json.load(f)
    entry = {"files": files, "note": "images not committed; numbers produced in the build container by make_golden.py --bsd100"}
    for key in ("L7_x2", "L7_x3", "L7_x4"):
        cfg = O.make_config(**MODELS[key][0])
        weights = dict(np.load(os.path.join(HERE, "weights_%s.npz" % key)))
        psnrs = [O.evaluate_image(cfg, weights, img)[0] for img in images]
        entry[key] = {"psnr": psnrs, "mean": float(np.mean(psnrs))}
        print("bsd100", key, entry[key]["mean"], flush=True)
    goldens["bsd100"] = entry
    with open(os.path.join(HERE, "goldens.json"), "w") as f:
        json.dump(goldens, f, indent=1)


if __name__ == "__main__":
    if "--bsd100" in sys.argv:
        bsd100()
    elif "--set14" in sys.argv:
        set14()
    else:
        main()
