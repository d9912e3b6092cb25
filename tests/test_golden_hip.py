"""HIP path on the reference's real weights and real images (committed fixtures): the north_star's
headline parity claim -- PSNR within 1e-3 dB and max-abs pixel error within 1e-4 of the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host import _flags

pytestmark = pytest.mark.gpu


def _goldens():
    with open(os.path.join(GOLDEN, "goldens.json")) as f:
        return json.load(f)


def _model(tmp_path, key, **extra):
    from dcscn_amd.model import SuperResolution
    g = _goldens()
    flags = dict(g["models"][key]["flags"])
    flags.pop("legacy_no_c", None)
    flags.update(checkpoint_dir=str(tmp_path / "models"), self_ensemble=1)
    flags.update(extra)
    m = SuperResolution(_flags(**flags))
    m.build_graph()
    m.init_all_variables()
    m.load_weights(dict(np.load(os.path.join(GOLDEN, "weights_%s.npz" % key))))
    return g, m


@pytest.mark.parametrize("split16", ["1", "0"])
@pytest.mark.parametrize("key", ["L7_x2", "L7_x3", "L7_x4", "L7_x4_DS", "L2_x2"])
def test_set5_psnr_matches_oracle(tmp_path, key, split16, monkeypatch):
    """Real weights, real images, both kernel families (DCSCN_SPLIT16 = 1: f16 hi/lo x 3 products, the default; 0: pure f32)."""
    monkeypatch.setenv("DCSCN_SPLIT16", split16)
    g, m = _model(tmp_path, key)
    psnrs = [m.do_for_evaluate(os.path.join(GOLDEN, "set5", f))[0] for f in g["files"]]
    m.close()
    want = g["models"][key]["set5_psnr"]
    print(key, np.mean(psnrs), g["models"][key]["set5_mean"])
    assert max(abs(a - b) for a, b in zip(psnrs, want)) <= 1e-3
    assert abs(float(np.mean(psnrs)) - g["models"][key]["set5_mean"]) <= 1e-3


@pytest.mark.parametrize("split16", ["1", "0"])
@pytest.mark.parametrize("key", ["L7_x2", "L7_x3", "L7_x4"])
def test_set14_psnr_matches_oracle(tmp_path, key, split16, monkeypatch):
    """Set14 (README.md:60-62: 32.74 / 29.47 / 27.76 dB for these checkpoints), including the grayscale image that
    takes the monochrome branch of do_for_evaluate (DCSCN.py:688-696: uint8 'L'-mode resizes)."""
    monkeypatch.setenv("DCSCN_SPLIT16", split16)
    g, m = _model(tmp_path, key)
    psnrs = [m.do_for_evaluate(os.path.join(GOLDEN, "set14", f))[0] for f in g["set14"]["files"]]
    m.close()
    want = g["set14"][key]["psnr"]
    print(key, np.mean(psnrs), g["set14"][key]["mean"])
    assert max(abs(a - b) for a, b in zip(psnrs, want)) <= 1e-3
    assert abs(float(np.mean(psnrs)) - g["set14"][key]["mean"]) <= 1e-3


@pytest.mark.parametrize("ens", [1, 8])
def test_pipelined_evaluation_gives_identical_values(tmp_path, ens):
    """do_for_evaluate_many (decode of the next files / device / PSNR + SSIM of the previous files on different threads) returns
    exactly the values of the serial do_for_evaluate loop (evaluate.py:89-107), Set14 incl. its grayscale image."""
    g, m = _model(tmp_path, "L7_x2", self_ensemble=ens)
    files = [os.path.join(GOLDEN, "set14", f) for f in g["set14"]["files"]]
    serial = [m.do_for_evaluate(f) for f in files]
    piped = m.do_for_evaluate_many(files)
    m.close()
    assert len(piped) == len(files)
    for (p0, s0), (p1, s1, sec) in zip(serial, piped):
        assert p0 == p1 and s0 == s1 and sec > 0


def test_set5_psnr_self_ensemble_8(tmp_path):
    g, m = _model(tmp_path, "L7_x2", self_ensemble=8)
    psnrs = [m.do_for_evaluate(os.path.join(GOLDEN, "set5", f))[0] for f in g["files"]]
    m.close()
    assert max(abs(a - b) for a, b in zip(psnrs, g["models"]["L7_x2"]["set5_psnr_ensemble8"])) <= 1e-3


def test_crop_vector_max_abs(tmp_path):
    g, m = _model(tmp_path, "L7_x2")
    crop = np.load(os.path.join(GOLDEN, "crop_L7_x2.npz"))
    out = m.do(crop["lr"], crop["bicubic"])
    m.close()
    assert out.dtype == np.float32 and out.shape == crop["output"].shape
    assert float(np.max(np.abs(out - crop["output"]))) <= 1e-4


def test_load_model_reads_the_tf_checkpoint(tmp_path):
    """load_model: TF bundle -> engine, including the legacy no-"C" topology of the shipped L2 file."""
    from dcscn_amd.model import SuperResolution
    g = _goldens()
    m = SuperResolution(_flags(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4,
                               self_ensemble=1, checkpoint_dir=os.path.join(GOLDEN, "models")))
    assert m.name == "dcscn_L2_F4to4_PS_R1F4"
    m.build_graph()
    m.init_all_variables()
    m.load_model()
    assert m.legacy_no_c
    psnr, ssim = m.do_for_evaluate(os.path.join(GOLDEN, "set5", g["files"][1]))
    m.close()
    assert abs(psnr - g["models"]["L2_x2"]["set5_psnr"][1]) <= 1e-3 and 0 < ssim <= 1


def test_evaluate_cli_end_to_end(tmp_path):
    """python evaluate.py on the committed Set5 copy: log line format and PSNR (evaluate.py:106-107)."""
    import re
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    g = _goldens()
    data = tmp_path / "data" / "set5"
    shutil.copytree(os.path.join(GOLDEN, "set5"), data)
    cmd = [sys.executable, os.path.join(ROOT, "evaluate.py"), "--test_dataset=set5", "--layers=2", "--filters=4",
           "--min_filters=4", "--use_nin=false", "--reconstruct_filters=4", "--self_ensemble=1",
           "--checkpoint_dir=" + os.path.join(GOLDEN, "models"), "--data_dir=" + str(tmp_path / "data"),
           "--output_dir=" + str(tmp_path / "out"), "--log_filename=" + str(tmp_path / "log.txt"),
           "--compute_bicubic"]
    p = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    mm = re.search(r"Model Average \[set5\] PSNR:([0-9.]+), SSIM:([0-9.]+), Time \(s\): ([0-9.]+)", p.stdout)
    mb = re.search(r"Bicubic Average \[set5\] PSNR:([0-9.]+), SSIM:([0-9.]+)", p.stdout)
    assert mm and mb, p.stdout
    assert abs(float(mm.group(1)) - g["models"]["L2_x2"]["set5_mean"]) <= 1e-3
    assert abs(float(mb.group(1)) - float(np.mean(g["bicubic"]["x2"]))) <= 1e-3
    # os.listdir order is arbitrary, averages are order independent; result images exist
    assert any(n.endswith("_result.png") for _, _, fs in os.walk(tmp_path / "out") for n in fs)
