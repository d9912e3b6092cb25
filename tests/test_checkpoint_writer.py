"""f4 of SURVEY.md section 8: the TensorFlow-free checkpoint WRITER (tf.train.Saver.save, helper/tf_graph.py:282-296) and the
frozen-graph reader (--frozenInference, DCSCN.py:192-220).

The strongest available pin: re-writing the tensors of a checkpoint the REFERENCE ships reproduces its ``.index`` and
``.data-00000-of-00001`` byte for byte (table layout, prefix compression, restart points, masked CRC32C of blocks and tensors)."""
import filecmp
import os

import numpy as np
import pytest

from conftest import GOLDEN

REF_CKPT = os.path.join(GOLDEN, "models", "dcscn_L2_F4to4_PS_R1F4.ckpt")      # the reference's own file (fixture copy)


def test_rewriting_a_reference_checkpoint_is_byte_identical(tmp_path):
    from dcscn_amd import ckpt
    tensors = ckpt.load_checkpoint(REF_CKPT, include_optimizer_slots=True)
    out = str(tmp_path / "copy.ckpt")
    ckpt.save_checkpoint(out, tensors)
    assert filecmp.cmp(REF_CKPT + ".data-00000-of-00001", out + ".data-00000-of-00001", shallow=False)
    assert filecmp.cmp(REF_CKPT + ".index", out + ".index", shallow=False)
    assert 'model_checkpoint_path: "copy.ckpt"' in (tmp_path / "checkpoint").read_text()


def test_crc32c_known_answers():
    from dcscn_amd import ckpt
    assert ckpt._crc32c(b"123456789") == 0xE3069283          # the CRC-32C check value
    assert ckpt._crc32c(b"\x00" * 32) == 0x8A9136AA          # RFC 3720 B.4
    assert ckpt._crc32c(bytes(range(32))) == 0x46DD794E


@pytest.mark.parametrize("n_vars", [1, 40, 3000])
def test_round_trip_random_tensors(tmp_path, n_vars):
    """Several data blocks (3000 variables > 256 KB of index entries), scalars, long shared prefixes."""
    from dcscn_amd import ckpt
    rng = np.random.default_rng(n_vars)
    tensors = {}
    for i in range(n_vars):
        shape = tuple(int(d) for d in rng.integers(1, 5, size=int(rng.integers(0, 5))))
        tensors["scope_%04d/layer/with/a/long/common/prefix/var_%d" % (i // 7, i)] = rng.standard_normal(shape).astype(np.float32)
    out = str(tmp_path / "m.ckpt")
    ckpt.save_checkpoint(out, tensors)
    back = ckpt.load_checkpoint(out, include_optimizer_slots=True)
    assert sorted(back) == sorted(tensors)
    assert all(back[k].shape == tensors[k].shape and np.array_equal(back[k], tensors[k]) for k in tensors)


def test_frozen_graph_round_trip(tmp_path):
    from dcscn_amd import ckpt, frozen
    tensors = ckpt.load_checkpoint(REF_CKPT)
    pb = str(tmp_path / "frozen_model.pb")
    frozen.write_frozen_graph(pb, tensors)
    nodes = frozen.read_graph_nodes(pb)
    names = [n[0] for n in nodes]
    assert names[:3] == ["x", "x2", "dropout_keep_rate"] and "CNN1/conv_W" in names and "CNN1/conv_W/read" in names
    back = frozen.read_frozen_graph(pb)
    assert sorted(back) == sorted(tensors) and all(np.array_equal(back[k], tensors[k]) for k in tensors)


def test_frozen_reader_rejects_unfrozen_and_garbage(tmp_path):
    from dcscn_amd import ckpt, frozen
    bad = tmp_path / "x.pb"
    bad.write_bytes(b"\x0a\x12\x0a\x03foo\x12\x0bVariableV2")          # one NodeDef: name "foo", op "VariableV2"
    with pytest.raises(ckpt.CheckpointError):
        frozen.read_frozen_graph(str(bad))
    bad.write_bytes(b"not a protobuf at all \xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff")
    with pytest.raises(ckpt.CheckpointError):
        frozen.read_frozen_graph(str(bad))


def test_float_val_and_broadcast_consts(tmp_path):
    """TensorProto with float_val instead of tensor_content (how small / constant tensors are stored)."""
    from dcscn_amd import frozen
    from dcscn_amd.frozen import _ld, _attr, _shape_proto
    import struct

    def const(name, shape, floats, packed):
        fv = _ld(5, b"".join(struct.pack("<f", v) for v in floats)) if packed else b"".join(b"\x2d" + struct.pack("<f", v) for v in floats)
        tensor = b"\x08\x01" + _ld(2, _shape_proto(shape)) + fv
        return _ld(1, _ld(1, name.encode()) + _ld(2, b"Const") + _attr("dtype", b"\x30\x01") + _attr("value", _ld(8, tensor)))

    pb = tmp_path / "c.pb"
    pb.write_bytes(const("a", (2, 2), [1, 2, 3, 4], True) + const("b", (3,), [0.5], False) + const("c", (), [7.0], False))
    t = frozen.read_frozen_graph(str(pb))
    assert np.array_equal(t["a"], np.array([[1, 2], [3, 4]], np.float32))
    assert np.array_equal(t["b"], np.full((3,), 0.5, np.float32)) and t["c"].shape == () and float(t["c"]) == 7.0


@pytest.mark.gpu
def test_save_model_and_frozen_inference_reproduce_the_checkpoint_psnr(tmp_path):
    """model.save_model -> load_model, and evaluate.py --frozenInference on a frozen copy: same PSNR as the original checkpoint."""
    import json
    import re
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    from test_host import _flags
    from dcscn_amd import frozen
    from dcscn_amd.model import SuperResolution
    with open(os.path.join(GOLDEN, "goldens.json")) as f:
        g = json.load(f)
    flags = dict(layers=2, filters=4, min_filters=4, use_nin=False, reconstruct_filters=4, self_ensemble=1)
    m = SuperResolution(_flags(checkpoint_dir=os.path.join(GOLDEN, "models"), **flags))
    m.build_graph()
    m.load_model()
    image = os.path.join(GOLDEN, "set5", g["files"][1])
    want = m.do_for_evaluate(image)[0]
    m.checkpoint_dir = str(tmp_path / "saved")
    m.save_model(trial=2)
    frozen.write_frozen_graph(str(tmp_path / "frozen.pb"), m._weights)
    m.close()
    m2 = SuperResolution(_flags(checkpoint_dir=str(tmp_path / "saved"), **flags))
    m2.build_graph()
    m2.load_model(trial=2)
    assert m2.do_for_evaluate(image)[0] == want
    m2.close()
    data = tmp_path / "data" / "set5"
    shutil.copytree(os.path.join(GOLDEN, "set5"), data)
    cmd = [sys.executable, os.path.join(ROOT, "evaluate.py"), "--test_dataset=set5", "--layers=2", "--filters=4", "--min_filters=4",
           "--use_nin=false", "--reconstruct_filters=4", "--self_ensemble=1", "--frozenInference", "--frozen_graph_path=" + str(tmp_path / "frozen.pb"),
           "--data_dir=" + str(tmp_path / "data"), "--output_dir=" + str(tmp_path / "out"), "--log_filename=" + str(tmp_path / "log.txt"),
           "--save_results=false"]
    p = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    mm = re.search(r"Model Average \[set5\] PSNR:([0-9.]+)", p.stdout)
    assert mm and abs(float(mm.group(1)) - g["models"]["L2_x2"]["set5_mean"]) <= 1e-3, p.stdout


def test_frozen_reader_on_a_graph_shaped_like_the_reference_freeze_tool_output():
    """tests/golden/frozen_L2_ref*.pb (make_ref_frozen.py): the inference sub-graph of the reference's own MetaGraphDef for its
    shipped L2 checkpoint with every VariableV2 turned into a Const of the same name -- what helper/custom_freeze_graph.py:14-61
    writes -- and the same with the ``prefix/`` names DCSCN.py:192-220 (load_graph) resolves.  The reader must return exactly
    the checkpoint's tensors from both."""
    from dcscn_amd import ckpt, frozen
    ref = ckpt.load_checkpoint(REF_CKPT)
    for fname, prefix in (("frozen_L2_ref.pb", ""), ("frozen_L2_ref_prefixed.pb", "prefix/")):
        path = os.path.join(GOLDEN, fname)
        nodes = frozen.read_graph_nodes(path)
        names = {n[0]: n[1] for n in nodes}
        assert names[prefix + "x"] == names[prefix + "x2"] == "Placeholder" and names[prefix + "output"] in ("Add", "AddV2")
        assert names[prefix + "CNN1/conv_W"] == "Const" and names[prefix + "CNN1/conv_W/read"] == "Identity"
        assert not any(op in ("VariableV2", "Variable") for op in names.values())
        got = frozen.read_frozen_graph(path, prefix=prefix)
        for k, v in ref.items():
            assert k in got and got[k].shape == v.shape and np.array_equal(got[k], v), k


def test_frozen_reader_pads_a_truncated_float_val_with_its_last_value(tmp_path):
    """TensorFlow drops the repeated tail of a tensor stored as float_val: 1 < len(float_val) < size is legal."""
    from dcscn_amd import frozen
    from dcscn_amd.frozen import _ld, _attr, _shape_proto
    import struct
    vals = [1.0, 2.0, 3.0]
    tensor = b"\x08\x01" + _ld(2, _shape_proto((2, 4))) + _ld(5, struct.pack("<3f", *vals))
    node = _ld(1, _ld(1, b"CNN1/conv_B") + _ld(2, b"Const") + _attr("dtype", b"\x30\x01") + _attr("value", _ld(8, tensor)))
    pb = tmp_path / "t.pb"
    pb.write_bytes(node)
    t = frozen.read_frozen_graph(str(pb))
    assert np.array_equal(t["CNN1/conv_B"], np.array([[1, 2, 3, 3], [3, 3, 3, 3]], np.float32))
