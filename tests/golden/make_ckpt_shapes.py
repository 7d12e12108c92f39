"""Decode variable names + shapes from the reference's TF checkpoint INDEX
(pretrained_model/pretrained_model.ckpt.index; the weights blob is absent) into
tests/golden/ckpt_index_shapes.json.  Authoring container only.

Format: a LevelDB-style table -- 48-byte footer (metaindex handle, index handle,
padding, magic) -> index block -> data blocks of prefix-compressed
(key, value) entries; each value is a BundleEntryProto whose field 2 is a
TensorShapeProto (repeated dim{size}).  Blocks are uncompressed (type byte 0).
"""
import json
import os
import sys


def varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def read_block(data, offset, size):
    assert data[offset + size] == 0, "compressed block"
    blk = data[offset:offset + size]
    n_restarts = int.from_bytes(blk[-4:], "little")
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = varint(blk, pos)
        non_shared, pos = varint(blk, pos)
        vlen, pos = varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def proto_fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = varint(buf, pos)
        elif wire == 2:
            ln, pos = varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]; pos += 4
        elif wire == 1:
            val = buf[pos:pos + 8]; pos += 8
        else:
            raise ValueError(wire)
        yield field, wire, val


def shape_of(entry):
    dims = []
    for f, w, v in proto_fields(entry):
        if f == 2 and w == 2:                       # TensorShapeProto
            for f2, w2, v2 in proto_fields(v):
                if f2 == 2 and w2 == 2:             # Dim
                    size = 0
                    for f3, _w3, v3 in proto_fields(v2):
                        if f3 == 1:
                            size = v3
                    dims.append(size)
    return dims


def main():
    ref = os.environ.get("ELO_REFERENCE_DIR", "/root/reference")
    data = open(os.path.join(ref, "pretrained_model", "pretrained_model.ckpt.index"), "rb").read()
    footer = data[-48:]
    pos = 0
    _mo, pos = varint(footer, pos); _ms, pos = varint(footer, pos)
    io, pos = varint(footer, pos); isz, pos = varint(footer, pos)
    shapes = {}
    for _key, handle in read_block(data, io, isz):
        off, p = varint(handle, 0)
        size, p = varint(handle, p)
        for key, val in read_block(data, off, size):
            name = key.decode()
            if name:                                # "" is the bundle header
                shapes[name] = shape_of(val)
    keep = {k: v for k, v in shapes.items() if "/Adam" not in k and k not in ("beta1_power", "beta2_power")}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ckpt_index_shapes.json")
    with open(out, "w") as f:
        json.dump(dict(sorted(keep.items())), f, indent=0)
    print(len(shapes), "entries,", len(keep), "non-Adam ->", out)


if __name__ == "__main__":
    main()
