"""DataTableImplV4, CPU side: the oracle's builder against its reader (both restated from the reference,
oracle/po_datatable.py), a hand-assembled known answer of the layout, and the sizing rules (DataTableUtils#computeColumnOffsets,
RegisterSet.getSizeForCount).  The library's serializer (pg_result_data_table_v4) is compared with this builder in tests/test_gpu_datatable.py."""
import struct

import pytest

from oracle import po_datatable as dt


def test_known_answer_of_a_two_row_table():
    """One INT key, COUNT (LONG), SUM (DOUBLE): every byte written by hand from DataTableImplV4#writeLeadingSections' order."""
    b = dt.build_data_table_v4(["g", "count(*)", "sum(m)"], ["INT", "LONG", "DOUBLE"], [[7, 3, 1.5], [-2, 1, -0.25]])
    exceptions = bytes.fromhex("00000000")
    dictionary = bytes.fromhex("00000000")
    schema = (bytes.fromhex("00000003") + bytes.fromhex("00000001") + b"g" + bytes.fromhex("00000008") + b"count(*)" + bytes.fromhex("00000006") + b"sum(m)"
              + bytes.fromhex("00000003") + b"INT" + bytes.fromhex("00000004") + b"LONG" + bytes.fromhex("00000006") + b"DOUBLE")
    fixed = (bytes.fromhex("00000007") + bytes.fromhex("0000000000000003") + bytes.fromhex("3ff8000000000000")
             + bytes.fromhex("fffffffe") + bytes.fromhex("0000000000000001") + bytes.fromhex("bfd0000000000000"))
    off = 52
    header = [4, 2, 3]
    for s in (exceptions, dictionary, schema, fixed, b""):
        header += [off, len(s)]
        off += len(s)
    expect = struct.pack(">13i", *header) + exceptions + dictionary + schema + fixed + bytes.fromhex("00000004") + bytes.fromhex("00000000")
    assert b == expect
    assert len(fixed) == 2 * 20      # INT 4 + LONG 8 + DOUBLE 8


def test_builder_and_reader_agree_on_every_column_kind():
    names = ["i", "l", "f", "d", "s", "b", "avg(x)", "minmaxrange(x)", "distinctcount(i)", "distinctcount(l)", "distinctcount(f)", "distinctcount(d)",
             "distinctcount(s)", "distinctcount(b)", "distinctcounthll(x)"]
    types = ["INT", "LONG", "FLOAT", "DOUBLE", "STRING", "BYTES"] + ["OBJECT"] * 9
    regs = bytes((i * 7) % 32 for i in range(256))
    rows = [
        [1, -2**40, 1.5, -2.25, "apple", b"\x00\x01", dt.AvgPair((10.0, 4)), dt.MinMaxRangePair((1.0, 9.0)), dt.ValueSet(dt.INT_SET, [3, -1, 2]),
         dt.ValueSet(dt.LONG_SET, [2**40, 5]), dt.ValueSet(dt.FLOAT_SET, [0.5, 2.0]), dt.ValueSet(dt.DOUBLE_SET, [0.1, -7.0]),
         dt.ValueSet(dt.STRING_SET, ["x", "yy", ""]), dt.ValueSet(dt.BYTES_SET, [b"\xff", b""]), dt.HyperLogLog(8, regs)],
        [-(2**31), 2**62, float("inf"), float("-inf"), "pear", b"", dt.AvgPair((0.0, 0)), dt.MinMaxRangePair((float("inf"), float("-inf"))),
         dt.ValueSet(dt.INT_SET, []), dt.ValueSet(dt.LONG_SET, []), dt.ValueSet(dt.FLOAT_SET, []), dt.ValueSet(dt.DOUBLE_SET, []),
         dt.ValueSet(dt.STRING_SET, []), dt.ValueSet(dt.BYTES_SET, []), dt.HyperLogLog(4, bytes(16))],
        [0, 0, 0.0, 0.0, "apple", b"abc", dt.AvgPair((-1.5, 2)), dt.MinMaxRangePair((-3.0, -3.0)), dt.ValueSet(dt.INT_SET, [7]),
         dt.ValueSet(dt.LONG_SET, [7]), dt.ValueSet(dt.FLOAT_SET, [7.0]), dt.ValueSet(dt.DOUBLE_SET, [7.0]), dt.ValueSet(dt.STRING_SET, ["seven"]),
         dt.ValueSet(dt.BYTES_SET, [b"7"]), dt.HyperLogLog(12, bytes([31]) * 4096)],
    ]
    p = dt.parse_data_table_v4(dt.build_data_table_v4(names, types, rows))
    assert p["names"] == names and p["types"] == types and p["exceptions"] == {} and p["metadata_entries"] == 0
    assert p["rows"] == rows
    # "apple" twice is ONE entry of the string dictionary (ids in first-use order)
    b = dt.build_data_table_v4(names, types, rows)
    d_start, d_len = struct.unpack_from(">ii", b, 5 * 4)
    assert struct.unpack_from(">i", b, d_start)[0] == 2


def test_empty_tables():
    p = dt.parse_data_table_v4(dt.build_data_table_v4(["g", "count(*)"], ["STRING", "LONG"], []))
    assert p["rows"] == [] and p["names"] == ["g", "count(*)"]


@pytest.mark.parametrize("log2m,words", [(4, 3), (6, 11), (8, 43), (10, 171), (12, 683), (14, 2731)])
def test_register_set_sizes(log2m, words):
    """RegisterSet.getSizeForCount: count / 6 words, one more unless that is a multiple of 32; the reference's rawhllresults blobs carry
    0x000000ac = 172 bytes = 43 words for log2m 8."""
    assert dt.register_set_words(1 << log2m) == words
    kind, b = dt.serialize_object(dt.HyperLogLog(log2m, bytes(1 << log2m)))
    assert kind == 6 and struct.unpack_from(">ii", b, 0) == (log2m, words * 4) and len(b) == 8 + 4 * words


def test_column_offsets():
    offs, size = dt.column_offsets(["INT", "STRING", "LONG", "FLOAT", "OBJECT", "DOUBLE", "BYTES"])
    assert offs == [0, 4, 8, 16, 20, 28, 36] and size == 44


def test_null_row_ids_round_trip_and_known_bytes():
    """enableNullHandling: placeholders + per-column null bitmaps behind the rows (DataTableBuilderV4#setNullRowIds, GroupByResultsBlock.java:196-222)"""
    names, types = ["k", "s", "sum(m)", "avg(m)"], ["INT", "STRING", "DOUBLE", "OBJECT"]
    rows = [[7, "a", 1.5, dt.AvgPair((3.0, 2))], [None, None, None, None], [9, "b", None, dt.AvgPair((1.0, 1))]]
    b = dt.build_data_table_v4(names, types, rows, null_handling=True)
    p = dt.parse_data_table_v4(b)
    assert p["rows"] == rows and p["null_row_ids"] == [[1], [1], [1, 2], []]   # a null OBJECT of a group-by block is NULL_TYPE_VALUE, not a bitmap entry
    # the trailer: 4 columns x (position, length); an array container of one value is 8 + 4 + 4 + 2 = 18 bytes
    h = struct.unpack_from(">13i", b, 0)
    fixed = b[h[9]:h[9] + h[10]]
    row_size = 4 + 4 + 8 + 8
    assert len(fixed) == 3 * row_size + 4 * 8
    trailer = struct.unpack_from(">8i", fixed, 3 * row_size)
    assert trailer[1] == 18 and trailer[3] == 18 and trailer[5] == 20 and trailer[7] == 0
    # the aggregation-only block marks a null OBJECT in the bitmap as well (AggregationResultsBlock.java:124-130)
    b = dt.build_data_table_v4(["sum(m)", "avg(m)"], ["DOUBLE", "OBJECT"], [[None, None]], null_handling=True, group_by=False)
    assert dt.parse_data_table_v4(b)["null_row_ids"] == [[0], [0]]
    # no nulls: the trailer is still there, all lengths 0
    b = dt.build_data_table_v4(["k"], ["INT"], [[1], [2]], null_handling=True)
    assert dt.parse_data_table_v4(b)["null_row_ids"] == [[]]


def test_null_bitmap_containers():
    ids = list(range(0, 10000, 2)) + [70000, 70001] + list(range(131072, 131072 + 5000))
    b = dt.serialize_null_row_ids(ids)
    assert dt.deserialize_null_row_ids(b) == ids
    cookie, n = struct.unpack_from("<II", b, 0)
    assert (cookie, n) == (12346, 3)
    # container 0: 5 000 values > 4 096 -> a bitmap container (8 192 bytes); container 1: array of 2; container 2: bitmap
    assert len(b) == 8 + 3 * 4 + 3 * 4 + 8192 + 4 + 8192
    from pinot_amd import formats
    import numpy as np
    assert formats.deserialize_roaring(b).tolist() == ids   # the segment-side reader of the same portable format agrees
