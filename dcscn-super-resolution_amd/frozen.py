"""Frozen TensorFlow graphs (``.pb`` GraphDef with the variables turned into Const nodes) without TensorFlow.

The reference evaluates from such a file with ``--frozenInference`` (evaluate.py:50-51 -> DCSCN.py:192-220 load_graph:
``tf.import_graph_def`` and a lookup of ``x``, ``x2``, ``output``), produced by helper/custom_freeze_graph.py /
helper/optimize_for_inference.py from a checkpoint.  Freezing keeps every variable under its checkpoint name
(``CNN1/conv_W`` becomes a Const node of that name, its ``/read`` Identity stays), so for this library the file is just
another weight container: ``read_frozen_graph`` returns ``{variable name: float32 ndarray}``; the topology is still
given by the flags, exactly as for a checkpoint, and the engine rejects a file whose tensors do not fit.

``write_frozen_graph`` is the TensorFlow-free counterpart of the freeze step: Placeholder nodes ``x`` / ``x2`` /
``dropout_keep_rate`` (the names load_graph resolves, DCSCN.py:208-212) and one Const + ``/read`` Identity per variable.
It does not emit the conv nodes -- the file carries weights, not an executable graph (documented in INTEGRATION.md).

Only what such files contain is parsed: NodeDef{name=1, op=2, input=3, attr=5}, AttrValue{type=6, shape=7, tensor=8},
TensorProto{dtype=1, tensor_shape=2, tensor_content=4, float_val=5}; DT_FLOAT tensors only.
"""

import struct

import numpy as np

from .ckpt import CheckpointError, _proto_fields, _parse_shape, _put_varint

_DT_FLOAT = 1


def _parse_tensor(buf):
    dtype, shape, content, floats = 0, (), None, []
    for f, w, v in _proto_fields(buf):
        if f == 1 and w == 0:
            dtype = v
        elif f == 2 and w == 2:
            shape = _parse_shape(v)
        elif f == 4 and w == 2:
            content = bytes(v)
        elif f == 5 and w == 5:
            floats.append(struct.unpack("<f", v)[0])
        elif f == 5 and w == 2:
            floats.extend(struct.unpack("<%df" % (len(v) // 4), v))
    if dtype != _DT_FLOAT:
        return None
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if content is not None:
        if len(content) != 4 * count:
            raise CheckpointError("Const tensor: %d bytes for shape %s" % (len(content), shape))
        return np.frombuffer(content, dtype="<f4").reshape(shape).astype(np.float32)
    if len(floats) == count:
        return np.asarray(floats, np.float32).reshape(shape)
    if 0 < len(floats) < count:
        # TensorFlow stores a tensor whose tail repeats one value with that tail truncated (one float_val = a constant fill):
        # the missing entries equal the last one given
        return np.concatenate([np.asarray(floats, np.float32), np.full(count - len(floats), floats[-1], np.float32)]).reshape(shape)
    if not floats and count:
        return np.zeros(shape, np.float32)
    raise CheckpointError("Const tensor: %d float_val entries for shape %s" % (len(floats), shape))


def read_graph_nodes(path):
    """``[(name, op, [inputs], {attr name: raw AttrValue bytes})]`` of a binary GraphDef."""
    with open(path, "rb") as f:
        buf = f.read()
    nodes = []
    try:
        for f_, w, v in _proto_fields(buf):
            if f_ != 1 or w != 2:
                continue
            name = op = ""
            inputs, attrs = [], {}
            for f2, w2, v2 in _proto_fields(v):
                if f2 == 1:
                    name = bytes(v2).decode("utf-8")
                elif f2 == 2:
                    op = bytes(v2).decode("utf-8")
                elif f2 == 3:
                    inputs.append(bytes(v2).decode("utf-8"))
                elif f2 == 5:
                    key, val = None, b""
                    for f3, w3, v3 in _proto_fields(v2):
                        if f3 == 1:
                            key = bytes(v3).decode("utf-8")
                        elif f3 == 2:
                            val = bytes(v3)
                    attrs[key] = val
            nodes.append((name, op, inputs, attrs))
    except (IndexError, struct.error, UnicodeDecodeError) as exc:
        raise CheckpointError("%s is not a binary GraphDef: %s" % (path, exc))
    if not nodes:
        raise CheckpointError("%s holds no graph nodes" % path)
    return nodes


def read_frozen_graph(path, prefix=""):
    """``{variable name: float32 ndarray}`` of every float Const node (``prefix`` -- e.g. "prefix/" after an
    import_graph_def round trip -- is stripped).  Raises CheckpointError if the graph still holds variables."""
    tensors = {}
    for name, op, _, attrs in read_graph_nodes(path):
        if op in ("VariableV2", "Variable", "VarHandleOp"):
            raise CheckpointError("%s is not frozen: node %s is a %s" % (path, name, op))
        if op != "Const" or "value" not in attrs:
            continue
        tensor = None
        for f, w, v in _proto_fields(attrs["value"]):
            if f == 8 and w == 2:
                tensor = _parse_tensor(v)
        if tensor is None:
            continue
        if prefix and name.startswith(prefix):
            name = name[len(prefix):]
        tensors[name] = tensor
    return tensors


def _ld(field, payload):
    return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def _attr(key, value):
    return _ld(5, _ld(1, key.encode()) + _ld(2, value))


def _shape_proto(shape):
    return b"".join(_ld(2, b"\x08" + _put_varint(int(s))) for s in shape)


def write_frozen_graph(path, tensors):
    """Binary GraphDef: placeholders x / x2 / dropout_keep_rate + one Const (and its ``/read`` Identity) per variable."""
    out = bytearray()
    dtype_float = b"\x30\x01"                                                 # AttrValue.type = DT_FLOAT
    for name, shape in (("x", (-1, -1, -1, 1)), ("x2", (-1, -1, -1, 1)), ("dropout_keep_rate", ())):
        dims = b"".join(_ld(2, b"\x08" + _put_varint(s & 0xFFFFFFFFFFFFFFFF)) for s in shape)
        out += _ld(1, _ld(1, name.encode()) + _ld(2, b"Placeholder") + _attr("dtype", dtype_float) + _attr("shape", _ld(7, dims)))
    for name in sorted(tensors):
        a = np.asarray(tensors[name], dtype="<f4")
        tensor = b"\x08\x01" + _ld(2, _shape_proto(a.shape)) + _ld(4, a.tobytes(order="C"))
        out += _ld(1, _ld(1, name.encode()) + _ld(2, b"Const") + _attr("dtype", dtype_float) + _attr("value", _ld(8, tensor)))
        out += _ld(1, _ld(1, (name + "/read").encode()) + _ld(2, b"Identity") + _ld(3, name.encode()) + _attr("T", dtype_float))
    out += _ld(4, b"\x08\x1a")                                                # VersionDef.producer = 26 (TF 1.x era)
    with open(path, "wb") as f:
        f.write(bytes(out))
    return path
