"""Generates tests/golden/frozen_L2_ref.pb and frozen_L2_ref_prefixed.pb: a frozen GraphDef shaped like the output of the
REFERENCE's own freeze tool for its shipped ``dcscn_L2_F4to4_PS_R1F4`` checkpoint, without TensorFlow.

helper/custom_freeze_graph.py:14-61 imports the checkpoint's MetaGraphDef, restores the variables and calls
``graph_util.convert_variables_to_constants(sess, graph_def, ["output"])``: the sub-graph the output node depends on is
kept, every VariableV2 in it becomes a Const of the same name (attr ``dtype``, attr ``value`` = TensorProto with
``tensor_content``), its ``<name>/read`` Identity and every op node (Conv2D, Add, the PReLU arithmetic, ConcatV2,
DepthToSpace, the int32 / float helper Consts) stay as they are.  This script does exactly that on the wire format: NodeDefs
are copied byte for byte from /root/reference/models/<name>.ckpt.meta, variables are replaced with the tensors of the
checkpoint (read with dcscn-super-resolution_amd/ckpt.py).  The second file carries the names DCSCN.py:192-220 (load_graph:
``tf.import_graph_def(graph_def, name="prefix")``, then ``prefix/x:0``, ``prefix/x2:0``, ``prefix/output:0``) sees, i.e. what a
re-export of the imported graph holds.

    python tests/golden/make_ref_frozen.py [/root/reference/models]      (build container only; the GPU box uses the committed files)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from dcscn_amd import ckpt  # noqa: E402
from dcscn_amd.frozen import _ld, _attr, _shape_proto  # noqa: E402
from make_ref_graphs import _fields  # noqa: E402

MODEL = "dcscn_L2_F4to4_PS_R1F4"


def node_defs(meta_path):
    with open(meta_path, "rb") as f:
        buf = f.read()
    graph = None
    for f_, w, v in _fields(buf):
        if f_ == 2:
            graph = v
    out = []
    for f_, w, v in _fields(graph):
        if f_ != 1:
            continue
        name = op = None
        inputs = []
        for f2, w2, v2 in _fields(v):
            if f2 == 1:
                name = bytes(v2).decode()
            elif f2 == 2:
                op = bytes(v2).decode()
            elif f2 == 3:
                inputs.append(bytes(v2).decode())
        out.append((name, op, inputs, bytes(v)))
    return out


def rename(raw, prefix):
    """NodeDef with `prefix` in front of its name and of every input (control inputs keep their ^)."""
    out = bytearray()
    for f, w, v in _fields(raw):
        if f == 1:
            out += _ld(1, prefix.encode() + bytes(v))
        elif f == 3:
            s = bytes(v)
            out += _ld(3, (b"^" + prefix.encode() + s[1:]) if s.startswith(b"^") else prefix.encode() + s)
        elif w == 2:
            out += _ld(f, bytes(v))
        else:
            raise ValueError("unexpected NodeDef field %d wire %d" % (f, w))
    return bytes(out)


def main(models_dir):
    base = os.path.join(models_dir, MODEL + ".ckpt")
    tensors = ckpt.load_checkpoint(base)
    nodes = node_defs(base + ".meta")
    by_name = {n[0]: n for n in nodes}
    keep, stack = set(), ["output"]
    while stack:
        n = stack.pop()
        if n in keep:
            continue
        keep.add(n)
        for i in by_name[n][2]:
            stack.append(i.lstrip("^").split(":")[0])
    dtype_float = b"\x30\x01"
    for prefix, fname in (("", "frozen_L2_ref.pb"), ("prefix/", "frozen_L2_ref_prefixed.pb")):
        out = bytearray()
        n_const = 0
        for name, op, inputs, raw in nodes:
            if name not in keep:
                continue
            if op in ("VariableV2", "Variable"):
                a = np.asarray(tensors[name], dtype="<f4")
                tensor = b"\x08\x01" + _ld(2, _shape_proto(a.shape)) + _ld(4, a.tobytes(order="C"))
                raw = _ld(1, name.encode()) + _ld(2, b"Const") + _attr("dtype", dtype_float) + _attr("value", _ld(8, tensor))
                n_const += 1
            out += _ld(1, rename(raw, prefix) if prefix else raw)
        out += _ld(4, b"\x08\x1a")
        path = os.path.join(HERE, fname)
        with open(path, "wb") as f:
            f.write(bytes(out))
        print("%s: %d nodes kept of %d, %d variables frozen, %d bytes" % (fname, len(keep), len(nodes), n_const, len(out)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/models")
