"""Compressed raw forward indexes (SNAPPY / LZ4 / LZ4_LENGTH_PREFIXED chunks, BaseChunkForwardIndexReader.java:61-111).
CPU: the oracle's decompressors against the reference's own legacy blobs (FixedByteChunkSVForwardIndexTest /
VarByteChunkSVForwardIndexTest backward-compatibility fixtures, copied by tests/golden/make_chunk_fixtures.py) and against
libsnappy / liblz4 (pyarrow) on seeded data.  GPU: columns uploaded compressed and decompressed in HBM (pg_decompress.hip) give
the oracle's query results, on the same fixtures and on seeded data."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from pinot_amd import capi, formats
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import HostColumn, HostSegment, build_column

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXED = [("fixedByteCompressed.v2", 2000, 100.2356), ("fixedByteSVRDoubles.v1", 10009, 0.0)]
CODECS = [formats.CHUNK_COMPRESSION_SNAPPY, formats.CHUNK_COMPRESSION_LZ4, formats.CHUNK_COMPRESSION_LZ4_LENGTH_PREFIXED,
          formats.CHUNK_COMPRESSION_ZSTANDARD, formats.CHUNK_COMPRESSION_GZIP]


def golden_blob(name):
    return np.frombuffer(gzip.open(os.path.join(GOLDEN, name + ".gz"), "rb").read(), dtype=np.uint8)


def legacy_segment(api, name, num_docs):
    col = HostColumn("d", "DOUBLE", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, True, 0, golden_blob(name))
    return NativeSegment(api, HostSegment("legacy", num_docs, {"d": col}))


def sequential_sum(values):
    s = 0.0
    for blk in range(0, len(values), 10000):   # block-at-a-time: holder += (sum of the block in docId order)
        inner = 0.0
        for x in values[blk:blk + 10000]:
            inner += x
        s = inner + s
    return s


def check_legacy(seg, num_docs, start):
    exp = np.arange(num_docs) + start
    b = seg.execute("SELECT COUNT(*), MIN(d), MAX(d) FROM t")
    assert b.aggregation_result() == [num_docs, exp[0], exp[-1]]
    d = seg.filter(f"SELECT COUNT(*) FROM t WHERE d BETWEEN {float(exp[7])!r} AND {float(exp[num_docs - 5])!r}")
    np.testing.assert_array_equal(d.doc_ids(), np.arange(7, num_docs - 4))
    return seg.execute("SELECT SUM(d) FROM t").aggregation_result()[0], exp


@pytest.mark.parametrize("name,num_docs,start", FIXED)
def test_oracle_reads_the_reference_compressed_blobs(oracle_api, name, num_docs, start):
    """reader.getDouble(i) == i + startValue for every doc (testBackwardCompatibilityHelper)"""
    seg = legacy_segment(oracle_api, name, num_docs)
    total, exp = check_legacy(seg, num_docs, start)
    assert total == sequential_sum(exp)
    seg.destroy()


@pytest.mark.parametrize("name,data,n", [("varByteStringsCompressed.v2", [b"abcdefghijk", b"12456887", b"pqrstuv", b"500"], 1000),
                                         ("varByteStrings.v1", [b"abcde", b"fgh", b"ijklmn", b"12345"], 1009)])
def test_oracle_reads_the_reference_compressed_var_byte_blobs(oracle_api, name, data, n):
    blob = golden_blob(name)
    out = C.create_string_buffer(64)
    for i in range(n):
        k = oracle_api.lib.po_read_var_bytes(blob.ctypes.data, blob.nbytes, i, out, 64)
        assert out.raw[:k] == data[i % 4], i


def seeded_columns(n, seed=3):
    rng = np.random.default_rng(seed)
    return {
        "i_runs": (np.repeat(rng.integers(-50, 50, n // 40 + 1), 40)[:n].astype(np.int32), "INT"),     # long back-references
        "i_rand": (rng.integers(-(1 << 31), 1 << 31, n).astype(np.int32), "INT"),                       # incompressible: literals only
        "l_step": ((np.arange(n, dtype=np.int64) * 3 + 1_000_000_007), "LONG"),                         # overlapping matches (offset 8 patterns)
        "f_few": (rng.choice(np.array([0.5, -1.25, 3.0, 1e10], dtype=np.float32), n), "FLOAT"),
        "d_mix": (np.where(rng.random(n) < 0.5, 0.0, rng.standard_normal(n)), "DOUBLE"),
        "i_zero": (np.zeros(n, dtype=np.int32), "INT"),                                                 # one literal + one long match per chunk
    }


@pytest.mark.parametrize("codec", CODECS)
def test_oracle_decompressors_match_libsnappy_and_liblz4(oracle_api, codec):
    n = 12_345
    for name, (vals, dt) in seeded_columns(n).items():
        for version, dpc in ((2, 1000), (3, 777)):
            blob = formats.write_raw_fixed_byte_chunk(vals, dt, version=version, docs_per_chunk=dpc, compression=codec)
            col = HostColumn(name, dt, capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, blob)
            seg = NativeSegment(oracle_api, HostSegment("c", n, {name: col}))
            b = seg.execute(f"SELECT MIN({name}), MAX({name}), SUM({name}) FROM t")
            v = vals.astype(np.float64)
            assert b.aggregation_result()[:2] == [float(v.min()), float(v.max())], (name, version)
            lo, hi = np.sort(vals)[[n // 4, 3 * n // 4]]
            d = seg.filter(f"SELECT COUNT(*) FROM t WHERE {name} BETWEEN {lo.item()!r} AND {hi.item()!r}")
            np.testing.assert_array_equal(d.doc_ids(), np.flatnonzero((vals >= lo) & (vals <= hi)))
            seg.destroy()


def test_oracle_rejects_corrupt_chunks(oracle_api):
    vals = np.arange(5000, dtype=np.int32)
    for codec in CODECS:
        blob = formats.write_raw_fixed_byte_chunk(vals, "INT", compression=codec)[:-3].copy()   # the last chunk is cut short
        col = HostColumn("x", "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, blob)
        with pytest.raises(capi.NativeError):
            NativeSegment(oracle_api, HostSegment("c", 5000, {"x": col}))
    blob = formats.write_raw_fixed_byte_chunk(vals, "INT").copy()
    blob[20:24] = np.frombuffer(np.array([2], dtype=">i4").tobytes(), dtype=np.uint8)   # claims ZSTANDARD: the bytes are no zstd frame
    col = HostColumn("x", "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, blob)
    with pytest.raises(capi.NativeError):
        NativeSegment(oracle_api, HostSegment("c", 5000, {"x": col}))


# ---- HIP path ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,num_docs,start", FIXED)
def test_gpu_reads_the_reference_compressed_blobs(gpu_api, oracle_api, name, num_docs, start):
    g, o = legacy_segment(gpu_api, name, num_docs), legacy_segment(oracle_api, name, num_docs)
    tg, exp = check_legacy(g, num_docs, start)
    to, _ = check_legacy(o, num_docs, start)
    assert abs(tg - to) <= 1e-9 * abs(to)      # SUM over doubles: the GPU adds in a different order (tolerance 1e-9 relative)
    g.destroy()
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("codec", CODECS)
def test_gpu_decompressed_columns_match_oracle(gpu_api, oracle_api, codec):
    n = 300_000
    cols = seeded_columns(n)
    host = HostSegment("cmp_0", n)
    plain = HostSegment("plain_0", n)
    rng = np.random.default_rng(1)
    g1 = rng.integers(0, 50, n).astype(np.int32)
    for name, (vals, dt) in cols.items():
        version, dpc = (3, 777) if name in ("l_step", "f_few") else (2, 1000)
        host.columns[name] = build_column(name, vals, dt, dictionary=False, raw_version=version, chunk_compression=codec, docs_per_chunk=dpc)
        plain.columns[name] = build_column(name, vals, dt, dictionary=False)
    host.columns["g1"] = build_column("g1", g1, "INT")
    plain.columns["g1"] = build_column("g1", g1, "INT")
    g, gp, o = NativeSegment(gpu_api, host), NativeSegment(gpu_api, plain), NativeSegment(oracle_api, host)
    queries = [
        "SELECT g1, COUNT(*), SUM(i_runs), MIN(i_rand), MAX(l_step) FROM t GROUP BY g1 LIMIT 100",
        "SELECT g1, SUM(i_rand), MAX(f_few), MIN(i_zero) FROM t WHERE i_runs BETWEEN -10 AND 10 AND l_step > 1000300000 GROUP BY g1 LIMIT 100",
        "SELECT COUNT(*), SUM(l_step), MIN(f_few), MAX(i_runs) FROM t WHERE i_rand > 0 AND f_few IN (0.5, 3.0)",
        "SELECT COUNT(*), MIN(d_mix), MAX(d_mix) FROM t WHERE d_mix < 0.25",
        "SELECT COUNT(*) FROM t WHERE i_zero = 0 AND l_step BETWEEN 1000000007 AND 1000600007",
    ]
    for q in queries:
        gb, pb, ob = g.execute(q), gp.execute(q), o.execute(q)
        if "GROUP BY" in q:
            assert gb.rows() == ob.rows() == pb.rows(), q
        else:
            assert gb.aggregation_result() == ob.aggregation_result() == pb.aggregation_result(), q
        assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
        # two scans and no index: AndDocIdIterator leapfrogs — the count comes from the iterator automaton over the leaves' bitmaps
        assert gb.stats.stats_exact == 1
        assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, q
    g.destroy()
    gp.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_rejects_corrupt_and_unsupported_chunks(gpu_api):
    vals = np.arange(5000, dtype=np.int32)
    for codec in CODECS:
        blob = formats.write_raw_fixed_byte_chunk(vals, "INT", compression=codec)[:-3].copy()   # the last chunk is cut short
        col = HostColumn("x", "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, blob)
        with pytest.raises(capi.NativeError):
            NativeSegment(gpu_api, HostSegment("c", 5000, {"x": col}))
    blob = formats.write_raw_fixed_byte_chunk(vals, "INT").copy()
    blob[20:24] = np.frombuffer(np.array([5], dtype=">i4").tobytes(), dtype=np.uint8)   # claims GZIP: the bytes are no zlib stream
    col = HostColumn("x", "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, blob)
    with pytest.raises(capi.NativeError):
        NativeSegment(gpu_api, HostSegment("c", 5000, {"x": col}))
