"""TensorFlow-free reader for TF "V2 checkpoint" tensor bundles.

The reference restores weights with ``tf.train.Saver.restore`` (helper/tf_graph.py:263-280)
from ``models/<name>.ckpt.{index,data-00000-of-00001}``.  TensorFlow is not part of this stack,
so the two files are parsed directly:

* ``.index``  -- a LevelDB-style sorted table.  48-byte footer = two block handles
  (metaindex, index; each ``varint64 offset, varint64 size``), zero padding, and the 8-byte magic
  ``0xdb4775248b80fb57``.  A block is a run of prefix-compressed entries
  (``varint shared, varint non_shared, varint value_len, key_suffix, value``) followed by a
  restart array ``uint32[n], uint32 n``; on disk every block is trailed by a 1-byte compression
  tag (0 = raw, the only kind the reference's checkpoints use) and a 4-byte CRC.
  Index-block values are handles of data blocks; data-block keys are variable names and values
  are serialized ``BundleEntryProto`` messages.  The key ``""`` holds the ``BundleHeaderProto``.
* ``.data-00000-of-00001`` -- raw little-endian tensors at ``[offset, offset + size)``.

Only what the DCSCN checkpoints contain is supported: float32 tensors, a single shard,
uncompressed blocks.  Anything else raises ``CheckpointError`` (never silently mis-reads).
"""

import os
import struct

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
_FOOTER_LEN = 48
_DT_FLOAT = 1


class CheckpointError(Exception):
    pass


def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _block_entries(buf, offset, size):
    """Yield (key, value) from one table block (without its 5-byte trailer)."""
    if offset + size + 5 > len(buf):
        raise CheckpointError("block handle outside file")
    if buf[offset + size] != 0:
        raise CheckpointError("compressed table blocks are not supported")
    block = buf[offset:offset + size]
    if size < 4:
        raise CheckpointError("block too small")
    n_restarts = struct.unpack_from("<I", block, size - 4)[0]
    limit = size - 4 - 4 * n_restarts
    if limit < 0:
        raise CheckpointError("corrupt restart array")
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        value = bytes(block[pos:pos + value_len])
        pos += value_len
        yield key, value


def _proto_fields(buf):
    """Minimal protobuf wire-format walker: yields (field_number, wire_type, value)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            value, pos = _varint(buf, pos)
        elif wire == 1:
            value = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            length, pos = _varint(buf, pos)
            value = buf[pos:pos + length]
            pos += length
        elif wire == 5:
            value = buf[pos:pos + 4]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wire)
        yield field, wire, value


def _parse_shape(buf):
    dims = []
    for field, wire, value in _proto_fields(buf):
        if field == 2 and wire == 2:          # TensorShapeProto.dim
            size = 0
            for f2, w2, v2 in _proto_fields(value):
                if f2 == 1 and w2 == 0:       # Dim.size
                    size = v2
            dims.append(size)
        elif field == 3 and wire == 0 and value:
            raise CheckpointError("unknown-rank shapes are not supported")
    return tuple(dims)


def _parse_entry(buf):
    entry = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "sliced": False}
    for field, wire, value in _proto_fields(buf):
        if field == 1 and wire == 0:
            entry["dtype"] = value
        elif field == 2 and wire == 2:
            entry["shape"] = _parse_shape(value)
        elif field == 3 and wire == 0:
            entry["shard"] = value
        elif field == 4 and wire == 0:
            entry["offset"] = value
        elif field == 5 and wire == 0:
            entry["size"] = value
        elif field == 7:
            entry["sliced"] = True
    return entry


def read_index(index_path):
    """Return ``{variable_name: entry_dict}`` for a ``<prefix>.index`` file."""
    with open(index_path, "rb") as f:
        buf = f.read()
    if len(buf) < _FOOTER_LEN:
        raise CheckpointError("index file too small: %s" % index_path)
    footer = buf[-_FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, _FOOTER_LEN - 8)[0] != _TABLE_MAGIC:
        raise CheckpointError("bad table magic in %s" % index_path)
    pos = 0
    _, pos = _varint(footer, pos)            # metaindex offset
    _, pos = _varint(footer, pos)            # metaindex size
    index_off, pos = _varint(footer, pos)
    index_size, pos = _varint(footer, pos)

    entries = {}
    num_shards = None
    for _, handle in _block_entries(buf, index_off, index_size):
        block_off, hpos = _varint(handle, 0)
        block_size, hpos = _varint(handle, hpos)
        for key, value in _block_entries(buf, block_off, block_size):
            if key == b"":
                for field, wire, v in _proto_fields(value):   # BundleHeaderProto
                    if field == 1 and wire == 0:
                        num_shards = v
                    elif field == 2 and wire == 0 and v != 0:
                        raise CheckpointError("big-endian bundles are not supported")
                continue
            entries[key.decode("utf-8")] = _parse_entry(value)
    if num_shards not in (None, 1):
        raise CheckpointError("multi-shard bundles are not supported (num_shards=%s)" % num_shards)
    return entries


def list_variables(prefix):
    """``[(name, shape)]`` of every variable in the checkpoint ``prefix`` (sorted by name)."""
    entries = read_index(prefix + ".index")
    return [(name, entries[name]["shape"]) for name in sorted(entries)]


def has_data(prefix):
    return os.path.isfile(prefix + ".data-00000-of-00001")


def load_checkpoint(prefix, include_optimizer_slots=False):
    """Load every float32 variable of ``<prefix>.index`` / ``.data-00000-of-00001``.

    Adam slot variables (``.../Adam``, ``.../Adam_1``, ``beta1_power``, ``beta2_power``) written by
    the reference's training graph are dropped unless ``include_optimizer_slots``.
    """
    entries = read_index(prefix + ".index")
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.isfile(data_path):
        raise CheckpointError("checkpoint data shard missing: %s" % data_path)
    with open(data_path, "rb") as f:
        blob = f.read()

    tensors = {}
    for name, entry in entries.items():
        if not include_optimizer_slots and is_optimizer_slot(name):
            continue
        if entry["sliced"]:
            raise CheckpointError("partitioned variable %s is not supported" % name)
        if entry["dtype"] != _DT_FLOAT:
            if is_optimizer_slot(name):
                continue
            raise CheckpointError("variable %s has unsupported dtype enum %d" % (name, entry["dtype"]))
        count = int(np.prod(entry["shape"], dtype=np.int64)) if entry["shape"] else 1
        if entry["size"] != 4 * count:
            raise CheckpointError("variable %s: size %d does not match shape %s"
                                  % (name, entry["size"], entry["shape"]))
        if entry["offset"] + entry["size"] > len(blob):
            raise CheckpointError("variable %s lies outside the data shard" % name)
        array = np.frombuffer(blob, dtype="<f4", count=count, offset=entry["offset"])
        tensors[name] = array.reshape(entry["shape"]).astype(np.float32, copy=True)
    return tensors


def is_optimizer_slot(name):
    leaf = name.rsplit("/", 1)[-1]
    return leaf in ("Adam", "Adam_1") or name in ("beta1_power", "beta2_power")


# ------------------------------------------------------------------------------------------------
# writer: tf.train.Saver.save without TensorFlow (helper/tf_graph.py:282-296 save_model)
# ------------------------------------------------------------------------------------------------
# Produces the same two files the reference's checkpoints consist of; written the way TensorFlow's BundleWriter and
# its LevelDB-style table builder write them (sorted keys, prefix compression with a restart point every 16 entries,
# one uncompressed data block per 256 KB, shortest-successor index keys, masked CRC32C of every block and of every
# tensor), so a file written here from the tensors of a reference checkpoint is byte-identical to the original
# (tests/test_checkpoint_writer.py) and TensorFlow's own reader accepts it.

_CRC_TABLE = None
_CRC_LANES = 16384         # segments whose CRCs advance together in _crc32c's vectorised part (measured: 7 MB in 0.06 s; 4096 lanes 1.4 s)


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        table = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            table.append(c)
        _CRC_TABLE = (table, np.asarray(table, dtype=np.uint32))
    return _CRC_TABLE


def _crc_raw(data, state):
    """The CRC register after ``data`` (no initial / final inversion), byte by byte."""
    table = _crc_table()[0]
    for b in data:
        state = table[(state ^ b) & 0xFF] ^ (state >> 8)
    return state


def _crc32c(data, crc=0):
    """CRC-32C (Castagnoli).  Small inputs: the table loop.  Large ones (a tensor is megabytes): the register update is
    linear over GF(2), so the buffer is cut into _CRC_LANES equal segments whose registers advance together in numpy
    (one vector step per byte POSITION, not per byte), and the segment results are chained with the operator "advance the
    register over len(segment) zero bytes", itself tabulated bytewise from its action on the 32 unit vectors."""
    if isinstance(data, memoryview):
        data = data.cast("B") if data.format != "B" or data.ndim != 1 else data      # bytes, whatever the view's element type
    elif not isinstance(data, (bytes, bytearray)):
        data = bytes(data)
    state = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    n = len(data)
    seg = n // _CRC_LANES
    if seg < 16:
        return _crc_raw(data, state) ^ 0xFFFFFFFF
    tnp = _crc_table()[1]
    body = np.frombuffer(data, dtype=np.uint8, count=seg * _CRC_LANES).reshape(_CRC_LANES, seg)
    cols = np.ascontiguousarray(body.T).astype(np.uint32)               # [position][segment]
    regs = np.zeros(_CRC_LANES, dtype=np.uint32)                        # every segment from register 0 (linearity)
    unit = (np.uint32(1) << np.arange(32, dtype=np.uint32))             # the operator's action on the unit vectors
    for pos in range(seg):
        regs = tnp[(regs ^ cols[pos]) & 0xFF] ^ (regs >> 8)
        unit = tnp[unit & 0xFF] ^ (unit >> 8)
    # advance(s) = xor of unit[b] over the set bits b of s, as four 256-entry tables
    adv = []
    for byte in range(4):
        t = [0] * 256
        for v in range(1, 256):
            low = v & -v
            t[v] = t[v ^ low] ^ int(unit[8 * byte + low.bit_length() - 1])
        adv.append(t)
    for r in regs.tolist():
        state = adv[0][state & 0xFF] ^ adv[1][(state >> 8) & 0xFF] ^ adv[2][(state >> 16) & 0xFF] ^ adv[3][state >> 24] ^ r
    return _crc_raw(data[seg * _CRC_LANES:], state) ^ 0xFFFFFFFF


def _masked_crc(data):
    crc = _crc32c(data)
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _put_varint(value):
    out = bytearray()
    while True:
        b = value & 0x7F
        value >>= 7
        if value:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


class _BlockBuilder:
    """LevelDB block: prefix-compressed entries + restart array (table/block_builder.cc)."""

    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.out = bytearray()
        self.restarts = []
        self.last = b""
        self.count = 0

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0:
            self.restarts.append(len(self.out))
        else:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        self.out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size_estimate(self):
        """BlockBuilder::CurrentSizeEstimate: what finish() would return -- the table builder's flush criterion."""
        return len(self.out) + 4 * max(len(self.restarts), 1) + 4

    def finish(self):
        restarts = self.restarts or [0]
        return bytes(self.out) + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))


def _build_block(items, restart_interval):
    b = _BlockBuilder(restart_interval)
    for key, value in items:
        b.add(key, value)
    return b.finish()


def _short_successor(key):
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _shortest_separator(start, limit):
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xFF and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _entry_proto(shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08\x01" + b"\x12" + _put_varint(len(dims)) + dims          # dtype = DT_FLOAT, shape
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


def save_checkpoint(prefix, tensors, block_size=256 << 10):
    """Write ``<prefix>.index`` and ``<prefix>.data-00000-of-00001`` holding ``{name: float32 ndarray}``.

    Variables are stored in sorted-name order, back to back (TensorFlow's BundleWriter layout).  Also refreshes the
    ``checkpoint`` state file next to it the way tf.train.Saver does (model_checkpoint_path points at the new prefix)."""
    directory = os.path.dirname(os.path.abspath(prefix))
    os.makedirs(directory, exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    data = bytearray()
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]                          # BundleHeaderProto: num_shards 1, version.producer 1
    for name in names:
        a = np.asarray(tensors[name], dtype="<f4")                       # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.tobytes(order="C")
        items.append((name.encode("utf-8"), _entry_proto(a.shape, len(data), len(raw), _masked_crc(raw))))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))

    out = bytearray()
    index_items = []

    def emit(builder, next_key):
        block = builder.finish()
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out.extend(block + b"\x00" + struct.pack("<I", _masked_crc(block + b"\x00")))
        last = builder.last
        index_items.append((_shortest_separator(last, next_key) if next_key is not None else _short_successor(last), handle))

    # TableBuilder::Add: a data block is flushed as soon as its encoded size (entries + restart array) reaches block_size
    builder = _BlockBuilder(16)
    for i, (key, value) in enumerate(items):
        builder.add(key, value)
        if builder.size_estimate() >= block_size and i + 1 < len(items):
            emit(builder, items[i + 1][0])
            builder = _BlockBuilder(16)
    emit(builder, None)
    meta = _build_block([], 16)
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(meta + b"\x00" + struct.pack("<I", _masked_crc(meta + b"\x00")))
    index = _build_block(index_items, 1)
    index_handle = _put_varint(len(out)) + _put_varint(len(index))
    out.extend(index + b"\x00" + struct.pack("<I", _masked_crc(index + b"\x00")))
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    base = os.path.basename(prefix)
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix
