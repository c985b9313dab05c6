"""Bit-sliced range index: RangeIndexBasedFilterOperator over BitSlicedRangeIndexReader
(pinot-core/.../filter/RangeIndexBasedFilterOperator.java:75-145, pinot-segment-local/.../readers/BitSlicedRangeIndexReader.java).
The RangeBitmap byte format is third-party (RoaringBitmap 1.3.0, not in the reference tree) and has no fixture there: the writer
(pinot_amd/formats.write_range_index), the oracle's decoder (oracle/po_rangeindex.c: per-row value reconstruction) and the GPU
leaf (bit-sliced lte algebra over the containers) are three separate readings of the format restated in formats.py, checked
against numpy brute force; the cases are RangeQueriesTest.java:147-200's (the reference runs them with and without a range index)."""
import numpy as np
import pytest

from pinot_amd import formats
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment

N = 150_001


def make_segment(seed=5):
    rng = np.random.default_rng(seed)
    data = {
        "di": rng.integers(0, 5000, N).astype(np.int32),                 # dictionary INT, range index, no inverted index
        "ri": rng.integers(-50_000, 900_000, N).astype(np.int32),        # raw INT
        "rl": rng.integers(-2**40, 2**41, N).astype(np.int64),           # raw LONG
        "rf": rng.normal(0, 100, N).astype(np.float32),                  # raw FLOAT
        "rd": rng.normal(0, 1e6, N),                                     # raw DOUBLE
        "so": np.sort(rng.integers(0, 300, N)).astype(np.int32),         # sorted: the sorted index wins over the range index
        "iv": rng.integers(0, 40, N).astype(np.int32),                   # inverted + range: EQ takes the inverted index, RANGE the range index
        "g": rng.integers(0, 9, N).astype(np.int32),
        "m": rng.integers(0, 1000, N).astype(np.int32),
    }
    data["rf"][::977] = 0.0
    data["rf"][5::1009] = -0.0
    schema = {"di": "INT", "ri": "INT", "rl": "LONG", "rf": "FLOAT", "rd": "DOUBLE", "so": "INT", "iv": "INT", "g": "INT", "m": "INT"}
    host = build_segment("ranges", data, schema, inverted_index_columns=["iv"], no_dictionary_columns=["ri", "rl", "rf", "rd", "m"],
                         range_index_columns=["di", "ri", "rl", "rf", "rd", "so", "iv"])
    return host, data


CASES = [   # (where, numpy predicate, expected numEntriesScannedInFilter as a function of (data, mask))
    ("di BETWEEN 100 AND 2999", lambda d: (d["di"] >= 100) & (d["di"] <= 2999), 0),
    ("di > 4990", lambda d: d["di"] > 4990, 0),
    ("di = 17", lambda d: d["di"] == 17, 0),                               # EQ, no inverted index: the exact range index answers
    ("ri BETWEEN 250000 AND 749999", lambda d: (d["ri"] >= 250000) & (d["ri"] <= 749999), 0),
    ("ri < -49000", lambda d: d["ri"] < -49000, 0),
    ("ri >= 899990", lambda d: d["ri"] >= 899990, 0),
    ("ri > 5000000", lambda d: d["ri"] > 5000000, 0),                     # beyond the column's max: empty
    ("ri = 1234", lambda d: d["ri"] == 1234, 0),
    ("rl BETWEEN -1000000000 AND 40000000000", lambda d: (d["rl"] >= -10**9) & (d["rl"] <= 4 * 10**10), 0),
    ("rl <= -1099511627000", lambda d: d["rl"] <= -1099511627000, 0),
    ("rf BETWEEN -0.5 AND 12.25", lambda d: (d["rf"] >= np.float32(-0.5)) & (d["rf"] <= np.float32(12.25)), 0),
    ("rf > 0", lambda d: d["rf"] > 0, 0),
    ("rf <= 0", lambda d: d["rf"] <= 0, 0),                               # 0.0 and -0.0 share an ordinal
    ("rd BETWEEN -1500000.5 AND 20.125", lambda d: (d["rd"] >= -1500000.5) & (d["rd"] <= 20.125), 0),
    ("rd >= 2500000", lambda d: d["rd"] >= 2500000, 0),
    ("so BETWEEN 10 AND 20", lambda d: (d["so"] >= 10) & (d["so"] <= 20), 0),          # SortedIndexBasedFilterOperator
    ("iv = 7", lambda d: d["iv"] == 7, 0),                                               # InvertedIndexFilterOperator
    ("iv BETWEEN 3 AND 9", lambda d: (d["iv"] >= 3) & (d["iv"] <= 9), 0),               # RANGE skips the inverted index: range index
    ("ri BETWEEN 0 AND 99999 AND m < 500", lambda d: (d["ri"] >= 0) & (d["ri"] <= 99999) & (d["m"] < 500),
     lambda d: int(((d["ri"] >= 0) & (d["ri"] <= 99999)).sum())),                      # the range index restricts the scan of m
    ("di < 50 OR rd < -2000000", lambda d: (d["di"] < 50) | (d["rd"] < -2000000), 0),
    ("NOT ri BETWEEN 0 AND 800000 AND iv IN (1, 2)", lambda d: ~((d["ri"] >= 0) & (d["ri"] <= 800000)) & np.isin(d["iv"], [1, 2]), 0),
]


def check(seg, data, where, pred, entries):
    b = seg.execute(f"SELECT g, COUNT(*), SUM(m) FROM ranges WHERE {where} GROUP BY g")
    mask = pred(data)
    expect = {}
    for g in np.unique(data["g"][mask]):
        sel = mask & (data["g"] == g)
        expect[(int(g),)] = [int(sel.sum()), float(data["m"][sel].astype(np.int64).sum())]
    assert b.rows() == expect, where
    want_entries = entries(data) if callable(entries) else entries
    assert b.stats.num_entries_scanned_in_filter == want_entries, where
    assert b.stats.num_docs_scanned == int(mask.sum())
    f = seg.filter(f"SELECT COUNT(*) FROM ranges WHERE {where}")
    np.testing.assert_array_equal(f.doc_ids(), np.flatnonzero(mask).astype(np.int32))
    return b


def test_range_bitmap_writer_shapes():
    """header fields, mask width, container type codes of the restated format"""
    v = np.array([0, 1, 2, 3, 65535, 7], dtype=np.uint64)
    blob = bytes(formats.write_range_index(v, -5, 65535))
    assert blob[:12] == (2).to_bytes(4, "big") + (-5).to_bytes(8, "big", signed=True)
    assert blob[12:14] == b"\x0d\xf0" and blob[14] == 2 and blob[15] == 16 and blob[16:18] == b"\x01\x00" and blob[18:22] == (6).to_bytes(4, "little")
    assert len(blob) > 24 and blob[22:24] == b"\xff\xff"        # every slice has a row whose bit is clear
    assert formats.fp_ordinal(np.array([0.0, -0.0], dtype=np.float32)).tolist() == [0x80000000, 0x80000000]
    o = formats.fp_ordinal(np.array([-np.inf, -1.5, -0.0, 0.0, 1e-30, 2.0, np.inf, np.nan]))
    assert o[0] == 0 and o[7] == 0 and o[6] == 2**64 - 1 and list(o[1:6]) == sorted(o[1:6])


@pytest.mark.parametrize("where,pred,entries", CASES)
def test_oracle_range_index(oracle_api, where, pred, entries):
    host, data = make_segment()
    seg = NativeSegment(oracle_api, host)
    check(seg, data, where, pred, entries)
    seg.destroy()


def test_oracle_with_and_without_range_index_agree(oracle_api):
    """RangeQueriesTest runs its cases against columns with and without the index: same rows, the scan entries disappear"""
    host, data = make_segment(seed=9)
    bare, _ = make_segment(seed=9)
    for c in bare.columns.values():
        c.range_index = None
    a, b = NativeSegment(oracle_api, host), NativeSegment(oracle_api, bare)
    for where, _pred, _e in CASES:
        q = f"SELECT g, COUNT(*), SUM(m) FROM ranges WHERE {where} GROUP BY g"
        ra, rb = a.execute(q), b.execute(q)
        assert ra.rows() == rb.rows(), where
        assert ra.stats.num_entries_scanned_in_filter <= rb.stats.num_entries_scanned_in_filter
    a.destroy()
    b.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("where,pred,entries", CASES)
def test_gpu_range_index(gpu_api, oracle_api, where, pred, entries):
    host, data = make_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    gb = check(g, data, where, pred, entries)
    ob = o.execute(f"SELECT g, COUNT(*), SUM(m) FROM ranges WHERE {where} GROUP BY g")
    assert gb.rows() == ob.rows() and gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_range_index_container_kinds(gpu_api, oracle_api):
    """clustered values give run containers, sparse ones array containers, dense ones bitmaps — in every slice position"""
    n = 200_003
    i = np.arange(n)
    data = {"runs": (i // 5000).astype(np.int32), "sparse": np.where(i % 997 == 0, 1_000_000 + i, 3).astype(np.int32),
            "wide": (i * 2654435761 % (1 << 31)).astype(np.int32), "g": (i % 3).astype(np.int32)}
    host = build_segment("kinds", data, {"runs": "INT", "sparse": "INT", "wide": "INT", "g": "INT"},
                         no_dictionary_columns=["runs", "sparse", "wide"], range_index_columns=["runs", "sparse", "wide"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for where, mask in (("runs BETWEEN 7 AND 30", (data["runs"] >= 7) & (data["runs"] <= 30)),
                        ("sparse > 1000000", data["sparse"] > 1000000), ("sparse = 3", data["sparse"] == 3),
                        ("wide BETWEEN 1000000000 AND 1500000000", (data["wide"] >= 10**9) & (data["wide"] <= 15 * 10**8)),
                        ("runs = 39 AND wide < 100000000", (data["runs"] == 39) & (data["wide"] < 10**8))):
        q = f"SELECT g, COUNT(*) FROM kinds WHERE {where} GROUP BY g"
        gb, ob = g.execute(q), o.execute(q)
        assert gb.rows() == ob.rows(), where
        assert sum(v[0] for v in gb.rows().values()) == int(mask.sum())
        assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter == 0
    g.destroy()
    o.destroy()
