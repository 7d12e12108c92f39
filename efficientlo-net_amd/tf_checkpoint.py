"""Reader (and a minimal writer) for TensorFlow "tensor bundle" checkpoints -- the format of the reference's
`pretrained_model/pretrained_model.ckpt.{index,data-00000-of-00001}` (main.py:227-249 `saver.save`, :252 restore).

A bundle is
  <prefix>.index               a LevelDB-style sorted table: data blocks of prefix-compressed (key, value) entries,
                               an index block of block handles, a 48-byte footer (metaindex handle, index handle,
                               padding, magic 0xdb4775248b80fb57).  Key "" holds a BundleHeaderProto
                               (1: num_shards, 2: endianness, 3: version); every other key is a variable name whose
                               value is a BundleEntryProto (1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset,
                               5: size, 6: crc32c of the bytes, masked).
  <prefix>.data-SSSSS-of-NNNNN raw little-endian tensor bytes at (offset, size).
Variable names in the reference's checkpoint are the names `tf_util.VariableStore` uses
(tests/test_host_logic.py checks the table against the decoded index), so `load_into(store, prefix)` needs no map.

No TensorFlow and no protobuf package: the few varint/length-delimited fields are parsed by hand.  The writer
emits uncompressed blocks with one restart point per entry and valid crc32c trailers; it exists for round trips
(tests, `Trainer.save(..., tf_bundle=True)`) and has not been read back by TensorFlow itself (not installed here).
"""
import os
from collections import namedtuple

import numpy as np

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          19: np.float16}
DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}
Entry = namedtuple("Entry", "dtype shape shard_id offset size crc32c")


# --------------------------------------------------------------------------- varints / protobuf wire format
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        elif wire == 1:
            val = int.from_bytes(buf[pos:pos + 8], "little")
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        yield field, wire, val


def _parse_entry(buf):
    dtype, shape, shard, offset, size, crc = 0, [], 0, 0, 0, None
    for f, w, v in _fields(buf):
        if f == 1:
            dtype = v
        elif f == 2 and w == 2:
            for f2, w2, v2 in _fields(v):
                if f2 == 2 and w2 == 2:
                    dim = 0
                    for f3, _w3, v3 in _fields(v2):
                        if f3 == 1:
                            dim = v3
                    shape.append(dim)
        elif f == 3:
            shard = v
        elif f == 4:
            offset = v
        elif f == 5:
            size = v
        elif f == 6:
            crc = v
        elif f == 7:
            raise ValueError("sliced (partitioned) variables are not supported")
    return Entry(dtype, tuple(shape), shard, offset, size, crc)


# --------------------------------------------------------------------------- crc32c (Castagnoli), TF's masking
def _crc_table():
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _crc_table()


def crc32c(data, crc=0):
    crc ^= 0xffffffff
    for b in bytes(data):
        crc = _TABLE[(crc ^ b) & 0xff] ^ (crc >> 8)
    return crc ^ 0xffffffff


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


# --------------------------------------------------------------------------- table reader
def _read_block(data, offset, size, verify):
    if data[offset + size] != 0:
        raise ValueError("compressed table block (type %d): not supported" % data[offset + size])
    if verify:
        stored = int.from_bytes(data[offset + size + 1:offset + size + 5], "little")
        if stored != mask_crc(crc32c(data[offset:offset + size + 1])):
            raise ValueError("table block at %d fails its crc32c" % offset)
    blk = data[offset:offset + size]
    n_restarts = int.from_bytes(blk[-4:], "little")
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(prefix, verify=False):
    """-> (header dict, {variable name: Entry}) of <prefix>.index."""
    data = open(prefix + ".index", "rb").read()
    if len(data) < 48 or int.from_bytes(data[-8:], "little") != MAGIC:
        raise ValueError(prefix + ".index is not a TensorFlow bundle index (bad magic)")
    footer = data[-48:]
    pos = 0
    _mo, pos = _varint(footer, pos)
    _ms, pos = _varint(footer, pos)
    io, pos = _varint(footer, pos)
    isz, pos = _varint(footer, pos)
    header, entries = {"num_shards": 1, "endianness": 0, "version": None}, {}
    for _key, handle in _read_block(data, io, isz, verify):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, val in _read_block(data, off, size, verify):
            if key == b"":
                for f, _w, v in _fields(val):
                    if f == 1:
                        header["num_shards"] = v
                    elif f == 2:
                        header["endianness"] = v
                    elif f == 3:
                        header["version"] = v
            else:
                entries[key.decode()] = _parse_entry(val)
    if header["endianness"] != 0:
        raise ValueError("big-endian bundle")
    return header, entries


def load_checkpoint(prefix, names=None, verify=False):
    """-> {name: ndarray} for `names` (default: every variable) of the bundle <prefix>."""
    header, entries = read_index(prefix, verify)
    shards = {}
    out = {}
    for name in (entries if names is None else names):
        e = entries[name]
        if e.dtype not in DTYPES:
            raise ValueError("%s: unsupported dtype enum %d" % (name, e.dtype))
        if e.shard_id not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, e.shard_id, header["num_shards"])
            if not os.path.exists(path):
                raise FileNotFoundError(path + " (the reference repository ships only the .index of its checkpoint)")
            shards[e.shard_id] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = shards[e.shard_id][e.offset:e.offset + e.size]
        if len(raw) != e.size:
            raise ValueError("%s: data shard is shorter than the index says" % name)
        if verify and e.crc32c is not None and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError("%s: tensor bytes fail their crc32c" % name)
        arr = np.frombuffer(bytes(raw), dtype=DTYPES[e.dtype])
        if arr.size != int(np.prod(e.shape, dtype=np.int64)):
            raise ValueError("%s: %d bytes do not fill shape %s" % (name, e.size, e.shape))
        out[name] = arr.reshape(e.shape).copy()
    return out


def model_variables(entries):
    """The names a model restore needs: no Adam slots, no optimiser scalars, no global step (main.py:148 `batch`)."""
    skip = ("beta1_power", "beta2_power", "Variable")
    return [k for k in entries if "/Adam" not in k and k not in skip]


def load_into(store, prefix, verify=False, strict=True):
    """Restore a `tf_util.VariableStore` (and nothing else) from the bundle <prefix>; returns the names loaded.
    `w_x` / `w_q` (the loss weights, main.py:151-152) are returned under those keys if present but not loaded."""
    _header, entries = read_index(prefix, verify)
    have = store.state_dict()
    wanted = [k for k in model_variables(entries) if k in have]
    missing = [k for k in have if k not in entries]
    if strict and missing:
        raise KeyError("checkpoint lacks %d variables, e.g. %s" % (len(missing), missing[:3]))
    store.load_state_dict(load_checkpoint(prefix, wanted, verify))
    return wanted


# --------------------------------------------------------------------------- writer (single shard)
def _block(entries):
    body, restarts = bytearray(), []
    for key, val in entries:
        restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(val)) + key + val
    for r in restarts or [0]:
        body += r.to_bytes(4, "little")
    body += max(len(restarts), 1).to_bytes(4, "little")
    return bytes(body)


def _proto_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _proto_bytes(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


def save_checkpoint(prefix, tensors, entries_per_block=16):
    """Write {name: array} as <prefix>.index + <prefix>.data-00000-of-00001."""
    names = sorted(tensors, key=lambda s: s.encode())
    blob, records = bytearray(), []
    for name in names:
        arr = np.asarray(tensors[name], order="C")              # (ascontiguousarray would turn 0-d into 1-d)
        if arr.dtype not in DTYPE_CODES:
            raise ValueError("%s: dtype %s has no TensorFlow enum here" % (name, arr.dtype))
        raw = arr.tobytes()
        shape = b"".join(_proto_bytes(2, _proto_varint(1, int(d))) for d in arr.shape)
        val = _proto_varint(1, DTYPE_CODES[arr.dtype]) + _proto_bytes(2, shape)
        if len(blob):
            val += _proto_varint(4, len(blob))
        val += _proto_varint(5, len(raw)) + _put_varint((6 << 3) | 5) + mask_crc(crc32c(raw)).to_bytes(4, "little")
        records.append((name.encode(), val))
        blob += raw
    header = _proto_varint(1, 1) + _proto_bytes(3, _proto_varint(1, 1))           # num_shards 1, version.producer 1
    records = [(b"", header)] + records
    out, index_entries = bytearray(), []

    def emit(block):
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out.extend(block + b"\x00" + mask_crc(crc32c(block + b"\x00")).to_bytes(4, "little"))
        return handle

    for i in range(0, len(records), entries_per_block):
        chunk = records[i:i + entries_per_block]
        index_entries.append((chunk[-1][0], emit(_block(chunk))))
    meta = emit(_block([]))
    index = emit(_block(index_entries))
    footer = meta + index
    out.extend(footer + b"\x00" * (40 - len(footer)) + MAGIC.to_bytes(8, "little"))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(blob))
    return names
