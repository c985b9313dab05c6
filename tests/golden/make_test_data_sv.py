"""Generates tests/golden/test_data_sv.npz from the reference's own Avro fixture.

Source: /root/reference/pinot-core/src/test/resources/data/test_data-sv.avro (30 000 rows, no codec), the input of
BaseSingleValueQueriesTest (pinot-core/src/test/java/org/apache/pinot/queries/BaseSingleValueQueriesTest.java:67-131).
Only the 11 columns of that test's schema are kept.  Run in the build container (the reference tree is not present on
the GPU box): python tests/golden/make_test_data_sv.py
"""
import io
import json
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/pinot-core/src/test/resources/data/test_data-sv.avro"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_data_sv.npz")
INT_COLS = ["column1", "column3", "column6", "column7", "column9", "column17", "column18", "daysSinceEpoch"]
STR_COLS = ["column5", "column11", "column12"]


def read_long(buf):
    shift = 0
    acc = 0
    while True:
        b = buf.read(1)[0]
        acc |= (b & 0x7F) << shift
        if not (b & 0x80):
            break
        shift += 7
    return (acc >> 1) ^ -(acc & 1)


def read_bytes(buf):
    n = read_long(buf)
    return buf.read(n)


def read_value(buf, typ):
    if isinstance(typ, list):  # union
        idx = read_long(buf)
        return read_value(buf, typ[idx])
    if isinstance(typ, dict):
        typ = typ["type"]
    if typ == "null":
        return None
    if typ in ("int", "long"):
        return read_long(buf)
    if typ == "string":
        return read_bytes(buf).decode("utf-8")
    if typ == "bytes":
        return read_bytes(buf)
    if typ == "float":
        return struct.unpack("<f", buf.read(4))[0]
    if typ == "double":
        return struct.unpack("<d", buf.read(8))[0]
    if typ == "boolean":
        return buf.read(1)[0] != 0
    raise ValueError(f"unsupported avro type {typ}")


def main():
    data = open(SRC, "rb").read()
    buf = io.BytesIO(data)
    assert buf.read(4) == b"Obj\x01"
    meta = {}
    while True:
        n = read_long(buf)
        if n == 0:
            break
        if n < 0:
            n = -n
            read_long(buf)
        for _ in range(n):
            k = read_bytes(buf).decode()
            meta[k] = read_bytes(buf)
    assert meta.get("avro.codec", b"null") == b"null"
    schema = json.loads(meta["avro.schema"])
    sync = buf.read(16)
    fields = schema["fields"]
    cols = {f["name"]: [] for f in fields}
    while buf.tell() < len(data):
        count = read_long(buf)
        read_long(buf)  # block size
        for _ in range(count):
            for f in fields:
                cols[f["name"]].append(read_value(buf, f["type"]))
        assert buf.read(16) == sync
    n = len(cols["column1"])
    assert n == 30000, n
    out = {}
    for c in INT_COLS:
        assert all(v is not None for v in cols[c])
        out[c] = np.asarray(cols[c], dtype=np.int32)
    for c in STR_COLS:
        assert all(v is not None for v in cols[c])
        out[c] = np.asarray(cols[c], dtype=str)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: (v.dtype.str, len(np.unique(v))) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
