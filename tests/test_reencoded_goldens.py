"""The reference's golden numbers on a re-encoded segment: the same docs with the INT group-by columns stored WITHOUT a dictionary.

InnerSegmentAggregationSingleValueQueriesTest's expectations (`:96-153`) are stated on groups' VALUES, so they hold whatever the
encoding of the key columns: with column1 / column6 / column9 raw, DefaultGroupByExecutor picks
NoDictionarySingleColumnGroupKeyGenerator (one key) or NoDictionaryMultiColumnGroupKeyGenerator (a raw column among several,
`DefaultGroupByExecutor.java:100-118`) instead of DictionaryBasedGroupKeyGenerator — and must return the same groups, the same
intermediate results and the same ExecutionStatistics (the filter columns keep their dictionaries and indexes).  This pins the oracle's
no-dictionary generators — and, through the GPU tests, the virtual dictionaries of the HIP path — to the reference's own numbers."""
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment
from tests.fixtures import SV_FILTER, SV_INVERTED, SV_SCHEMA
from tests.test_oracle_goldens import AGGREGATION_QUERY, check_agg, check_stats

RAW_KEYS = ["column9", "column6"]     # group-by columns only: column1 / column3 / column7 of SV_FILTER keep their scans / indexes


def raw_key_segment(sv_data):
    data = {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in sv_data.items()}
    return build_segment("testTable_rawKeys", data, SV_SCHEMA, inverted_index_columns=[c for c in SV_INVERTED if c not in RAW_KEYS],
                         no_dictionary_columns=RAW_KEYS)


GOLDENS = [
    # (GROUP BY, key, (count, sum1, max3, min6, avg_sum, avg_count), post-filter entries) — unfiltered, then with SV_FILTER
    (" GROUP BY column9", (11270,), (1, 815409257, 1215316262, 1328642550, 788414092, 1), 150000,
     (242920,), (3, 4348938306, 407993712, 296467636, 5803888725, 3), 30645),
    (" GROUP BY column9, column11, column12", (1813102948, "P", "HEuxNvH"), (4, 2062187196, 1988589001, 394608493, 4782388964, 4), 210000,
     (1176631727, "P", "KrNxpdycSiwoRohEiTIlLqDHnx"), (1, 716185211, 489993380, 371110078, 487714191, 1), 42903),
    (" GROUP BY column1, column6, column9, column11, column12", (484569489, 16200443, 1159557463, "P", "MaztCmmxxgguBUxPti"),
     (2, 969138978, 995355481, 16200443, 2222394270, 2), 210000,
     (1318761745, 353175528, 1172307870, "P", "HEuxNvH"), (2, 2637523490, 557154208, 353175528, 2427862396, 2), 42903),
]


def run_goldens(seg):
    for gb, key, agg, post, fkey, fagg, fpost in GOLDENS:
        b = seg.execute(AGGREGATION_QUERY + gb)
        check_stats(b, 30000, 0, post, 30000)
        check_agg(b.rows()[key], *agg)
        b = seg.execute(AGGREGATION_QUERY + SV_FILTER + gb)
        check_stats(b, 6129, 63064, fpost, 30000)
        check_agg(b.rows()[fkey], *fagg)


def test_oracle_raw_keys_reproduce_the_reference_goldens(oracle_api, sv_data):
    seg = NativeSegment(oracle_api, raw_key_segment(sv_data))
    run_goldens(seg)
    seg.destroy()


@pytest.mark.gpu
def test_gpu_raw_keys_reproduce_the_reference_goldens(gpu_api, oracle_api, sv_data):
    host = raw_key_segment(sv_data)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    run_goldens(g)
    for gb, *_ in GOLDENS:          # and every group, not only the asserted one
        for flt in ("", SV_FILTER):
            assert g.execute(AGGREGATION_QUERY + flt + gb).rows() == o.execute(AGGREGATION_QUERY + flt + gb).rows()
    g.destroy()
    o.destroy()


# ---- the same goldens over compressed raw chunks: every ChunkCompressionType must decode to the reference's docs ----------------------
def compressed_segment(sv_data, codec):
    """column1 / column3 / column6 / column7 (the aggregated and filtered INT columns) as raw chunks written with `codec`."""
    from pinot_amd.segment import build_column
    host = raw_key_segment(sv_data)
    for c in ("column1", "column3", "column6", "column7"):
        host.columns[c] = build_column(c, sv_data[c], "INT", dictionary=False, chunk_compression=codec, docs_per_chunk=1000)
    return host


def aggregation_goldens(seg):   # InnerSegmentAggregationSingleValueQueriesTest.java:43-60
    b = seg.execute(AGGREGATION_QUERY)
    assert b.stats.num_docs_scanned == 30000
    check_agg(b.aggregation_result(), 30000, 32317185437847, 2147419555, 1689277, 28175373944314, 30000)
    b = seg.execute(AGGREGATION_QUERY + SV_FILTER)
    assert b.stats.num_docs_scanned == 6129
    check_agg(b.aggregation_result(), 6129, 6875947596072, 999813884, 1980174, 4699510391301, 6129)


@pytest.mark.parametrize("codec", [1, 2, 3, 4, 5], ids=["SNAPPY", "ZSTANDARD", "LZ4", "LZ4_LENGTH_PREFIXED", "GZIP"])
def test_oracle_goldens_over_compressed_chunks(oracle_api, sv_data, codec):
    seg = NativeSegment(oracle_api, compressed_segment(sv_data, codec))
    aggregation_goldens(seg)
    seg.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("codec", [1, 2, 3, 4, 5], ids=["SNAPPY", "ZSTANDARD", "LZ4", "LZ4_LENGTH_PREFIXED", "GZIP"])
def test_gpu_goldens_over_compressed_chunks(gpu_api, sv_data, codec):
    seg = NativeSegment(gpu_api, compressed_segment(sv_data, codec))
    aggregation_goldens(seg)
    seg.destroy()


# ---- the same goldens through a star-tree over (column9, column11, column12) holding count / sum / max / min and the AVG pair ------------
def star_tree_segment(sv_data):
    from pinot_amd import startree
    from tests.fixtures import sv_segment
    host = sv_segment(sv_data)
    startree.add_star_tree(host, ["column9", "column11", "column12"],
                           [("COUNT", "*"), ("SUM", "column1"), ("MAX", "column3"), ("MIN", "column6"), ("AVG", "column7"), ("MINMAXRANGE", "column3")],
                           max_leaf_records=10)
    return host


def star_tree_goldens(seg):
    """InnerSegmentAggregationSingleValueQueriesTest.java:96-133 without the filter (its columns are no star-tree dimensions): the
    pre-aggregated docs must add up to the reference's groups — COUNT, SUM, MAX, MIN and the serialized AvgPair alike."""
    b = seg.execute(AGGREGATION_QUERY + " GROUP BY column9")
    assert b.stats.star_tree_index == 0 and b.stats.num_docs_scanned < 30000
    check_agg(b.rows()[(11270,)], 1, 815409257, 1215316262, 1328642550, 788414092, 1)
    b = seg.execute(AGGREGATION_QUERY + " GROUP BY column9, column11, column12")
    assert b.stats.star_tree_index == 0
    check_agg(b.rows()[(1813102948, "P", "HEuxNvH")], 4, 2062187196, 1988589001, 394608493, 4782388964, 4)
    b = seg.execute(AGGREGATION_QUERY)      # root: one pre-aggregated doc
    assert b.stats.star_tree_index == 0 and b.stats.num_docs_scanned == 1
    check_agg(b.aggregation_result(), 30000, 32317185437847, 2147419555, 1689277, 28175373944314, 30000)
    b = seg.execute("SELECT MINMAXRANGE(column3), AVG(column7) FROM testTable WHERE column11 = 'P'")
    assert b.stats.star_tree_index == 0
    return b.aggregation_result()


def test_oracle_goldens_through_a_star_tree(oracle_api, sv_data):
    from pinot_amd import capi
    from pinot_amd.query import parse_sql
    seg = NativeSegment(oracle_api, star_tree_segment(sv_data))
    star = star_tree_goldens(seg)
    q = parse_sql("SELECT MINMAXRANGE(column3), AVG(column7) FROM testTable WHERE column11 = 'P'")
    q.flags |= capi.QUERY_FLAG_SKIP_STAR_TREE
    assert seg.execute(q).aggregation_result() == star      # the pairs add up to what the plain scan computes
    seg.destroy()


@pytest.mark.gpu
def test_gpu_goldens_through_a_star_tree(gpu_api, oracle_api, sv_data):
    host = star_tree_segment(sv_data)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    assert star_tree_goldens(g) == star_tree_goldens(o)
    g.destroy()
    o.destroy()


# ---- and behind a range index on the filter's range columns ------------------------------------------------------------------------------
def range_index_segment(sv_data):
    data = {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in sv_data.items()}
    return build_segment("testTable_rangeIdx", data, SV_SCHEMA, inverted_index_columns=SV_INVERTED, range_index_columns=["column1", "column3"])


def range_index_goldens(seg):
    """:43-60 with `column1 > 100000000` and `column3 BETWEEN ...` answered by RangeIndexBasedFilterOperator: same docs, same values;
    the two leaves scan nothing, so numEntriesScannedInFilter is what the remaining scan leaves count."""
    b = seg.execute(AGGREGATION_QUERY + SV_FILTER)
    assert b.stats.num_docs_scanned == 6129 and b.stats.num_entries_scanned_in_filter < 63064
    check_agg(b.aggregation_result(), 6129, 6875947596072, 999813884, 1980174, 4699510391301, 6129)
    b = seg.execute(AGGREGATION_QUERY + SV_FILTER + " GROUP BY column9")
    check_agg(b.rows()[(242920,)], 3, 4348938306, 407993712, 296467636, 5803888725, 3)
    return b.stats.num_entries_scanned_in_filter


def test_oracle_goldens_behind_a_range_index(oracle_api, sv_data):
    seg = NativeSegment(oracle_api, range_index_segment(sv_data))
    range_index_goldens(seg)
    seg.destroy()


@pytest.mark.gpu
def test_gpu_goldens_behind_a_range_index(gpu_api, oracle_api, sv_data):
    host = range_index_segment(sv_data)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    assert range_index_goldens(g) == range_index_goldens(o)
    g.destroy()
    o.destroy()


# ---- and with the summed columns widened: column1 as raw DOUBLE (fixed-point digit accumulators), column7 as raw LONG ---------------------
def widened_segment(sv_data):
    import numpy as np
    data = {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in sv_data.items()}
    data["column1"] = np.asarray(sv_data["column1"], dtype=np.float64)     # every INT is an exact double: the sums stay the reference's
    data["column7"] = np.asarray(sv_data["column7"], dtype=np.int64)
    schema = dict(SV_SCHEMA, column1="DOUBLE", column7="LONG")
    return build_segment("testTable_widened", data, schema, inverted_index_columns=[c for c in SV_INVERTED if c != "column7"],
                         no_dictionary_columns=["column1", "column7"])


def widened_goldens(seg):
    aggregation_goldens(seg)
    b = seg.execute(AGGREGATION_QUERY + " GROUP BY column9, column11, column12")
    check_agg(b.rows()[(1813102948, "P", "HEuxNvH")], 4, 2062187196, 1988589001, 394608493, 4782388964, 4)


def test_oracle_goldens_with_double_and_long_sums(oracle_api, sv_data):
    seg = NativeSegment(oracle_api, widened_segment(sv_data))
    widened_goldens(seg)
    seg.destroy()


@pytest.mark.gpu
def test_gpu_goldens_with_double_and_long_sums(gpu_api, sv_data):
    seg = NativeSegment(gpu_api, widened_segment(sv_data))
    widened_goldens(seg)
    seg.destroy()


# ---- DISTINCTCOUNTHLL over RAW columns: values hashed on the fly (MurmurHash.hashLong) instead of through the dictionary ---------------------
def test_oracle_hll_goldens_over_raw_columns(oracle_api, sv_data):
    """InnerSegmentAggregationSingleValueQueriesTest#testDistinctCountHLL's cardinalities (5977 / 23825; 1886 / 4492 behind the filter's
    first predicates are asserted in test_oracle_goldens) do not depend on the encoding: with column1 / column3 stored raw the registers
    come from hashing the values themselves."""
    from pinot_amd.executor import extract_final
    seg = NativeSegment(oracle_api, widened_int_raw_segment(sv_data))
    b = seg.execute("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable")
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [5977, 23825]
    b = seg.execute("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable" + SV_FILTER)   # :268-274
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [1886, 4492]
    seg.destroy()


def widened_int_raw_segment(sv_data):
    data = {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in sv_data.items()}
    return build_segment("testTable_rawHll", data, SV_SCHEMA, inverted_index_columns=SV_INVERTED, no_dictionary_columns=["column1", "column3"])


@pytest.mark.gpu
def test_gpu_hll_goldens_over_raw_columns(gpu_api, oracle_api, sv_data):
    from pinot_amd.executor import extract_final
    host = widened_int_raw_segment(sv_data)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    b = g.execute("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable")
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [5977, 23825]
    b = g.execute("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable" + SV_FILTER)
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [1886, 4492]
    q = "SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3), COUNT(*) FROM testTable GROUP BY column9"
    assert g.execute(q).rows() == o.execute(q).rows()
    g.destroy()
    o.destroy()
