"""TEST INFRASTRUCTURE (oracle): DataTableImplV4, builder and reader, restated in plain Python from the reference — the checker of
pg_result_data_table_v4; never imported by the product path.  PARITY UNPINNED against the reference itself: the reference tree holds no
serialized DataTable bytes (its DataTableSerDeTest builds and reads tables in one JVM) and no JVM exists here; what this file pins is that
the library's bytes are what an independent reading of the builder produces and what an independent reading of the READER decodes.

Builder side:
    DataTableBuilderV4 / BaseDataTableBuilder#setColumn   pinot-core/src/main/java/org/apache/pinot/core/common/datatable/*.java
    DataTableUtils#computeColumnOffsets                   pinot-common/.../common/datatable/DataTableUtils.java:41-65
    DataTableImplV4#toBytes / writeLeadingSections        pinot-common/.../common/datatable/DataTableImplV4.java:422-517
    DataSchema#toBytes                                    pinot-common/.../common/utils/DataSchema.java:118-143
    ObjectSerDeUtils (ObjectType values, serializers)     pinot-core/.../core/common/ObjectSerDeUtils.java:116-167,557-591,733-761,893-1039
Reader side:
    DataTableImplV4(ByteBuffer)                           DataTableImplV4.java:118-200, getInt/getLong/.../getCustomObject :222-330
"""
import struct
from typing import Dict, List, Tuple

VERSION_4 = 4
HEADER_SIZE = 13 * 4
FIXED_4 = ("INT", "FLOAT", "STRING")          # DataTableUtils#computeColumnOffsets: everything else takes 8 bytes

# ObjectSerDeUtils.ObjectType
AVG_PAIR, MIN_MAX_RANGE_PAIR, HYPER_LOG_LOG, INT_SET, LONG_SET, FLOAT_SET, DOUBLE_SET, STRING_SET, BYTES_SET = 4, 5, 6, 9, 15, 16, 17, 18, 19


class AvgPair(tuple):          # (sum: float, count: int)
    pass


class MinMaxRangePair(tuple):  # (min: float, max: float)
    pass


class HyperLogLog:
    def __init__(self, log2m: int, registers: bytes):
        self.log2m, self.registers = log2m, bytes(registers)

    def __eq__(self, o):
        return isinstance(o, HyperLogLog) and (self.log2m, self.registers) == (o.log2m, o.registers)

    def __repr__(self):
        return f"HyperLogLog({self.log2m}, {self.registers[:8].hex()}...)"


class ValueSet:
    """A typed set of values (IntOpenHashSet, ... ObjectOpenHashSet<String>): `kind` is the ObjectType, `values` in serialization order."""
    def __init__(self, kind: int, values):
        self.kind, self.values = kind, list(values)

    def __eq__(self, o):
        return isinstance(o, ValueSet) and self.kind == o.kind and sorted(map(_sort_key, self.values)) == sorted(map(_sort_key, o.values))

    def __repr__(self):
        return f"ValueSet({self.kind}, {self.values[:6]}...)"


def _sort_key(v):
    if isinstance(v, float):
        return struct.pack(">d", v)
    return v if isinstance(v, (bytes, str)) else (v,)


def register_set_words(m: int) -> int:
    """stream-lib RegisterSet.getSizeForCount"""
    bits = m // 6
    if bits == 0:
        return 1
    return bits if bits % 32 == 0 else bits + 1


def serialize_object(o) -> Tuple[int, bytes]:
    if isinstance(o, AvgPair):
        return AVG_PAIR, struct.pack(">dq", o[0], o[1])
    if isinstance(o, MinMaxRangePair):
        return MIN_MAX_RANGE_PAIR, struct.pack(">dd", o[0], o[1])
    if isinstance(o, HyperLogLog):
        m = 1 << o.log2m
        words = register_set_words(m)
        out = [struct.pack(">ii", o.log2m, words * 4)]
        for w in range(words):
            x = 0
            for s in range(6):
                r = w * 6 + s
                if r < m:
                    x |= (o.registers[r] & 0x1F) << (5 * s)
            out.append(struct.pack(">I", x))
        return HYPER_LOG_LOG, b"".join(out)
    if isinstance(o, ValueSet):
        n = struct.pack(">i", len(o.values))
        if o.kind == INT_SET:
            return o.kind, n + b"".join(struct.pack(">i", v) for v in o.values)
        if o.kind == LONG_SET:
            return o.kind, n + b"".join(struct.pack(">q", v) for v in o.values)
        if o.kind == FLOAT_SET:
            return o.kind, n + b"".join(struct.pack(">f", v) for v in o.values)
        if o.kind == DOUBLE_SET:
            return o.kind, n + b"".join(struct.pack(">d", v) for v in o.values)
        if o.kind == STRING_SET:
            return o.kind, n + b"".join(struct.pack(">i", len(v.encode())) + v.encode() for v in o.values)
        if o.kind == BYTES_SET:
            return o.kind, n + b"".join(struct.pack(">i", len(v)) + v for v in o.values)
    raise TypeError(f"no ObjectSerDe for {type(o)}")


def deserialize_object(kind: int, b: bytes):
    if kind == AVG_PAIR:
        return AvgPair(struct.unpack(">dq", b))
    if kind == MIN_MAX_RANGE_PAIR:
        return MinMaxRangePair(struct.unpack(">dd", b))
    if kind == HYPER_LOG_LOG:
        log2m, nbytes = struct.unpack_from(">ii", b, 0)
        words = struct.unpack_from(f">{nbytes // 4}I", b, 8)
        m = 1 << log2m
        return HyperLogLog(log2m, bytes((words[r // 6] >> (5 * (r % 6))) & 0x1F for r in range(m)))
    n = struct.unpack_from(">i", b, 0)[0]
    if kind == INT_SET:
        return ValueSet(kind, struct.unpack_from(f">{n}i", b, 4))
    if kind == LONG_SET:
        return ValueSet(kind, struct.unpack_from(f">{n}q", b, 4))
    if kind == FLOAT_SET:
        return ValueSet(kind, struct.unpack_from(f">{n}f", b, 4))
    if kind == DOUBLE_SET:
        return ValueSet(kind, struct.unpack_from(f">{n}d", b, 4))
    if kind in (STRING_SET, BYTES_SET):
        vals, p = [], 4
        for _ in range(n):
            ln = struct.unpack_from(">i", b, p)[0]
            v = b[p + 4:p + 4 + ln]
            vals.append(v.decode() if kind == STRING_SET else bytes(v))
            p += 4 + ln
        return ValueSet(kind, vals)
    raise ValueError(f"object type {kind}")


def column_offsets(types: List[str]) -> Tuple[List[int], int]:
    offs, size = [], 0
    for t in types:
        offs.append(size)
        size += 4 if t in FIXED_4 else 8
    return offs, size


NULL_TYPE_VALUE = 100   # CustomObject.NULL_TYPE_VALUE: a null OBJECT (DataTableBuilder#setColumn(int, null))
NULL_PLACEHOLDER = {"INT": 0, "LONG": 0, "FLOAT": 0.0, "DOUBLE": 0.0, "STRING": "", "BYTES": b""}   # CommonConstants.NullValuePlaceHolder


def serialize_null_row_ids(row_ids: List[int]) -> bytes:
    """RoaringBitmap#serialize of a bitmap built with add(): portable format, array containers up to 4 096 values, bitmap containers beyond,
    no run containers (nobody calls runOptimize on these)."""
    conts: Dict[int, List[int]] = {}
    for r in row_ids:
        conts.setdefault(r >> 16, []).append(r & 0xFFFF)
    keys = sorted(conts)
    out = struct.pack("<II", 12346, len(keys))
    for k in keys:
        out += struct.pack("<HH", k, len(conts[k]) - 1)
    pos = 8 + 8 * len(keys)
    for k in keys:
        out += struct.pack("<I", pos)
        pos += 8192 if len(conts[k]) > 4096 else 2 * len(conts[k])
    for k in keys:
        vals = conts[k]
        if len(vals) <= 4096:
            out += struct.pack(f"<{len(vals)}H", *vals)
        else:
            words = [0] * 1024
            for v in vals:
                words[v >> 6] |= 1 << (v & 63)
            out += struct.pack("<1024Q", *words)
    return out


def deserialize_null_row_ids(b: bytes) -> List[int]:
    cookie, n = struct.unpack_from("<II", b, 0)
    assert cookie == 12346, cookie
    out, descs = [], [struct.unpack_from("<HH", b, 8 + 4 * i) for i in range(n)]
    offs = [struct.unpack_from("<I", b, 8 + 4 * n + 4 * i)[0] for i in range(n)]
    for (key, card_m1), off in zip(descs, offs):
        card = card_m1 + 1
        if card <= 4096:
            out += [(key << 16) | v for v in struct.unpack_from(f"<{card}H", b, off)]
        else:
            words = struct.unpack_from("<1024Q", b, off)
            out += [(key << 16) | (w * 64 + bit) for w in range(1024) for bit in range(64) if (words[w] >> bit) & 1]
    return out


def build_data_table_v4(names: List[str], types: List[str], rows: List[list], null_handling: bool = False, group_by: bool = True) -> bytes:
    """DataTableBuilderV4: startRow / setColumn per stored type / finishRow / build, then DataTableImplV4#toBytes with no exceptions and no
    metadata.  Row values by column type: INT / LONG ints, FLOAT / DOUBLE floats, STRING str, BYTES bytes, OBJECT one of the classes above.
    null_handling (GroupByResultsBlock.java:196-222, AggregationResultsBlock.java:118-140): a None travels as the type's placeholder with its
    row id in the column's null bitmap (a None OBJECT as NULL_TYPE_VALUE; the aggregation-only block marks it in the bitmap as well, the
    group-by block does not), and every column's bitmap — (position, length) in the fixed-size part, the bytes in the variable-size part —
    follows the rows (DataTableBuilderV4#setNullRowIds)."""
    fixed, var = bytearray(), bytearray()
    dictionary: Dict[str, int] = {}
    null_rows: List[List[int]] = [[] for _ in types]
    for row_id, row in enumerate(rows):
        assert len(row) == len(types)
        for c, (t, v) in enumerate(zip(types, row)):
            if v is None:
                assert null_handling
                if t == "OBJECT":
                    if not group_by:
                        null_rows[c].append(row_id)
                    fixed += struct.pack(">ii", len(var), 0)
                    var += struct.pack(">i", NULL_TYPE_VALUE)
                    continue
                null_rows[c].append(row_id)
                v = NULL_PLACEHOLDER[t]
            if t == "INT":
                fixed += struct.pack(">i", v)
            elif t == "LONG":
                fixed += struct.pack(">q", v)
            elif t == "FLOAT":
                fixed += struct.pack(">f", v)
            elif t == "DOUBLE":
                fixed += struct.pack(">d", v)
            elif t == "STRING":
                fixed += struct.pack(">i", dictionary.setdefault(v, len(dictionary)))
            elif t == "BYTES":
                fixed += struct.pack(">ii", len(var), len(v))
                var += v
            elif t == "OBJECT":
                kind, b = serialize_object(v)
                fixed += struct.pack(">ii", len(var), len(b))
                var += struct.pack(">i", kind) + b
            else:
                raise ValueError(t)
    if null_handling:
        for ids in null_rows:
            fixed += struct.pack(">i", len(var))
            if not ids:
                fixed += struct.pack(">i", 0)
            else:
                b = serialize_null_row_ids(ids)
                fixed += struct.pack(">i", len(b))
                var += b
    exceptions = struct.pack(">i", 0)
    dict_bytes = struct.pack(">i", len(dictionary)) + b"".join(struct.pack(">i", len(s.encode())) + s.encode() for s in dictionary)   # insertion order = id order
    schema = struct.pack(">i", len(names)) + b"".join(struct.pack(">i", len(n.encode())) + n.encode() for n in names) + \
        b"".join(struct.pack(">i", len(t)) + t.encode() for t in types)
    metadata = struct.pack(">i", 0)
    header = [VERSION_4, len(rows), len(names)]
    off = HEADER_SIZE
    for section in (exceptions, dict_bytes, schema, bytes(fixed), bytes(var)):
        header += [off, len(section)]
        off += len(section)
    return struct.pack(">13i", *header) + exceptions + dict_bytes + schema + bytes(fixed) + bytes(var) + struct.pack(">i", len(metadata)) + metadata


def parse_data_table_v4(b: bytes) -> dict:
    """DataTableImplV4(ByteBuffer) and the typed getters: {"names", "types", "rows", "exceptions", "metadata"}"""
    h = struct.unpack_from(">13i", b, 0)
    assert h[0] == VERSION_4, h[0]
    n_rows, n_cols = h[1], h[2]
    (ex_s, ex_n, di_s, di_n, sc_s, sc_n, fx_s, fx_n, va_s, va_n) = h[3:]
    # sections are contiguous and in order
    assert ex_s == HEADER_SIZE and di_s == ex_s + ex_n and sc_s == di_s + di_n and fx_s == sc_s + sc_n and va_s == fx_s + fx_n

    def read_str(p):
        ln = struct.unpack_from(">i", b, p)[0]
        return b[p + 4:p + 4 + ln].decode(), p + 4 + ln

    exceptions, p = {}, ex_s
    n_ex = struct.unpack_from(">i", b, p)[0]
    p += 4
    for _ in range(n_ex):
        code = struct.unpack_from(">i", b, p)[0]
        msg, p = read_str(p + 4)
        exceptions[code] = msg
    strings = []
    if di_n:
        p = di_s
        n = struct.unpack_from(">i", b, p)[0]
        p += 4
        for _ in range(n):
            s, p = read_str(p)
            strings.append(s)
    names, types = [], []
    if sc_n:
        p = sc_s
        n = struct.unpack_from(">i", b, p)[0]
        assert n == n_cols
        p += 4
        for _ in range(n):
            s, p = read_str(p)
            names.append(s)
        for _ in range(n):
            s, p = read_str(p)
            types.append(s)
    offs, row_size = column_offsets(types)
    # DataTableImplV4#getNullRowIds: with null handling, (position, length) of every column's null bitmap follows the rows
    has_nulls = fx_n == n_rows * row_size + 8 * len(types)
    assert has_nulls or fx_n == n_rows * row_size, (fx_n, n_rows, row_size)
    var = b[va_s:va_s + va_n]
    rows = []
    for r in range(n_rows):
        base = fx_s + r * row_size
        row = []
        for t, o in zip(types, offs):
            if t == "INT":
                row.append(struct.unpack_from(">i", b, base + o)[0])
            elif t == "LONG":
                row.append(struct.unpack_from(">q", b, base + o)[0])
            elif t == "FLOAT":
                row.append(struct.unpack_from(">f", b, base + o)[0])
            elif t == "DOUBLE":
                row.append(struct.unpack_from(">d", b, base + o)[0])
            elif t == "STRING":
                row.append(strings[struct.unpack_from(">i", b, base + o)[0]])
            elif t == "BYTES":
                pos, ln = struct.unpack_from(">ii", b, base + o)
                row.append(bytes(var[pos:pos + ln]))
            elif t == "OBJECT":   # getCustomObject: position, size; the object type int sits in front of the bytes
                pos, ln = struct.unpack_from(">ii", b, base + o)
                kind = struct.unpack_from(">i", var, pos)[0]
                row.append(None if kind == NULL_TYPE_VALUE else deserialize_object(kind, bytes(var[pos + 4:pos + 4 + ln])))
            else:
                raise ValueError(t)
        rows.append(row)
    null_row_ids = None
    if has_nulls:
        null_row_ids = []
        for c, t in enumerate(types):
            pos, ln = struct.unpack_from(">ii", b, fx_s + n_rows * row_size + 8 * c)
            ids = deserialize_null_row_ids(bytes(var[pos:pos + ln])) if ln else []
            null_row_ids.append(ids)
            for r in ids:
                rows[r][c] = None
    p = va_s + va_n
    meta_len = struct.unpack_from(">i", b, p)[0]
    assert p + 4 + meta_len == len(b), (p, meta_len, len(b))
    n_meta = struct.unpack_from(">i", b, p + 4)[0]
    return {"names": names, "types": types, "rows": rows, "exceptions": exceptions, "metadata_entries": n_meta, "null_row_ids": null_row_ids}
