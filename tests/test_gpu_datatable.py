"""pg_result_data_table_v4 against the oracle's DataTableBuilderV4 restatement (oracle/po_datatable.py): the same groups and intermediate
results, byte for byte, and the oracle's READER decodes them back."""
import numpy as np
import pytest

from oracle import po_datatable as dt
from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment

pytestmark = pytest.mark.gpu

SET_KIND = {"INT": dt.INT_SET, "LONG": dt.LONG_SET, "FLOAT": dt.FLOAT_SET, "DOUBLE": dt.DOUBLE_SET, "STRING": dt.STRING_SET, "BYTES": dt.BYTES_SET}
FN = {"COUNT": "count", "SUM": "sum", "MIN": "min", "MAX": "max", "AVG": "avg", "MINMAXRANGE": "minmaxrange", "DISTINCTCOUNT": "distinctcount",
      "DISTINCTCOUNTHLL": "distinctcounthll"}


def make_host(n=60_003, seed=5):
    rng = np.random.default_rng(seed)
    data = {
        "gi": rng.integers(-5, 5, n).astype(np.int32),
        "gl": rng.integers(0, 4, n).astype(np.int64) * 10**12,
        "gf": (rng.integers(0, 3, n) * 0.5).astype(np.float32),
        "gd": (rng.integers(0, 3, n) * -1.25).astype(np.float64),
        "gs": np.array(["ant", "bee", "cat", "dog"], dtype=object)[rng.integers(0, 4, n)],
        "rawk": rng.integers(0, 6, n).astype(np.int32),
        "m": rng.integers(-1000, 1000, n).astype(np.int32),
        "c": rng.integers(0, 50, n).astype(np.int32),
        "cl": rng.integers(0, 30, n).astype(np.int64) * 7,
        "cs": np.array([f"v{i}" for i in range(20)], dtype=object)[rng.integers(0, 20, n)],
        "u": rng.integers(0, 5000, n).astype(np.int32),
    }
    schema = {"gi": "INT", "gl": "LONG", "gf": "FLOAT", "gd": "DOUBLE", "gs": "STRING", "rawk": "INT", "m": "INT", "c": "INT", "cl": "LONG", "cs": "STRING", "u": "INT"}
    return build_segment("dt", data, schema, no_dictionary_columns=["rawk", "m"])


def expected_bytes(block, host):
    q = block.query
    names = list(q.group_by) + [("count(*)" if a.function == "COUNT" else f"{FN[a.function]}({a.column})") for a in q.aggregations]
    types = [host.columns[g].data_type for g in q.group_by]
    cols = block.columns
    for a, col in zip(q.aggregations, cols):
        types.append("LONG" if a.function == "COUNT" else ("DOUBLE" if a.function in ("SUM", "MIN", "MAX") else "OBJECT"))
    rows = []
    keys = block.group_keys if q.group_by else [()]
    for i, k in enumerate(keys):
        row = [float(v) if t in ("FLOAT", "DOUBLE") else (int(v) if t in ("INT", "LONG") else v) for v, t in zip(k, types)]
        for a, col in zip(q.aggregations, cols):
            v = col[i]
            if a.function == "AVG":
                v = dt.AvgPair(v)
            elif a.function == "MINMAXRANGE":
                v = dt.MinMaxRangePair(v)
            elif a.function == "DISTINCTCOUNT":
                v = dt.ValueSet(SET_KIND[host.columns[a.column].data_type], sorted(v))   # ascending dictIds = ascending values
            elif a.function == "DISTINCTCOUNTHLL":
                v = dt.HyperLogLog(a.log2m or 8, v)
            row.append(v)
        rows.append(row)
    return dt.build_data_table_v4(names, types, rows), names, types, rows


QUERIES = [
    "SELECT gi, COUNT(*), SUM(m), MIN(m), MAX(m) FROM dt GROUP BY gi LIMIT 100",
    "SELECT gs, gl, COUNT(*), AVG(m), MINMAXRANGE(m) FROM dt WHERE m > -500 GROUP BY gs, gl LIMIT 100",
    "SELECT gf, gd, COUNT(*), SUM(m) FROM dt GROUP BY gf, gd LIMIT 100",
    "SELECT gs, DISTINCTCOUNT(c), DISTINCTCOUNT(cl), DISTINCTCOUNT(cs), COUNT(*) FROM dt GROUP BY gs LIMIT 100",
    "SELECT gi, DISTINCTCOUNTHLL(u), DISTINCTCOUNTHLL(u, 6) FROM dt WHERE c < 25 GROUP BY gi LIMIT 100",
    "SELECT rawk, COUNT(*), MAX(m) FROM dt GROUP BY rawk LIMIT 100",                 # a raw group key: values, not dictIds
    "SELECT rawk, gs, SUM(m) FROM dt GROUP BY rawk, gs LIMIT 100",
    "SELECT gs, DISTINCTCOUNT(rawk), DISTINCTCOUNT(m), COUNT(*) FROM dt GROUP BY gs LIMIT 100",   # raw columns: typed VALUE sets (IntOpenHashSet)
    "SELECT COUNT(*), SUM(m), AVG(m), DISTINCTCOUNT(c), DISTINCTCOUNTHLL(u) FROM dt WHERE gi >= 0",   # AggregationResultsBlock: one row
    "SELECT gi, COUNT(*) FROM dt WHERE m > 5000 GROUP BY gi LIMIT 100",               # no groups at all
]


@pytest.fixture(scope="module")
def seg(gpu_api):
    host = make_host()
    s = NativeSegment(gpu_api, host)
    yield s, host
    s.destroy()


@pytest.mark.parametrize("sql", QUERIES)
def test_data_table_bytes_equal_the_builder(seg, sql):
    s, host = seg
    nr = s.execute_native(parse_sql(sql), keep_device_table=False)
    got = nr.data_table_v4()
    block = nr.block()
    want, names, types, rows = expected_bytes(block, host)
    assert got == want
    p = dt.parse_data_table_v4(got)
    assert p["names"] == names and p["types"] == types and p["rows"] == rows
    nr.free()


def test_data_table_of_a_merged_result(gpu_api):
    """Two segments over the same dictionaries folded in the library: the table is the merged one's."""
    host_a, host_b = make_host(30_001, seed=1), make_host(30_001, seed=1)
    a, b = NativeSegment(gpu_api, host_a), NativeSegment(gpu_api, host_b)
    q = parse_sql("SELECT gs, COUNT(*), SUM(m), DISTINCTCOUNTHLL(u) FROM dt GROUP BY gs LIMIT 100")
    ra, rb = a.execute_native(q), b.execute_native(q)
    single = dt.parse_data_table_v4(ra.data_table_v4())
    ra.merge(rb)
    merged = dt.parse_data_table_v4(ra.data_table_v4())
    assert [r[0] for r in merged["rows"]] == [r[0] for r in single["rows"]]
    for m, s1 in zip(merged["rows"], single["rows"]):
        assert m[1] == 2 * s1[1] and m[2] == 2 * s1[2] and m[3] == s1[3]    # counts and sums double, registers (max of equals) stay
    want, *_ = expected_bytes(ra.block(), host_a)
    assert ra.data_table_v4() == want
    for r in (ra, rb):
        r.free()
    a.destroy()
    b.destroy()


def test_final_values_are_refused(seg):
    s, _ = seg
    q = parse_sql("SELECT gi, DISTINCTCOUNTHLL(u) FROM dt GROUP BY gi LIMIT 100")
    q.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    nr = s.execute_native(q, keep_device_table=False)
    with pytest.raises(capi.NativeError):
        nr.data_table_v4()
    nr.free()


# ---- enableNullHandling: placeholders, NULL_TYPE_VALUE objects and the columns' null bitmaps behind the rows ---------------------------------
NULL_QUERIES = [
    "SELECT gi, COUNT(*), SUM(m), MIN(m), AVG(m) FROM dt GROUP BY gi LIMIT 100",          # gi and m hold nulls: NULL key, NULL results
    "SELECT gs, gi, COUNT(*), MAX(m), MINMAXRANGE(m) FROM dt GROUP BY gs, gi LIMIT 100",   # a NULL STRING key ("" placeholder in the dictionary)
    "SELECT COUNT(*), SUM(m), AVG(m), MAX(m) FROM dt WHERE c > 1000",                      # one row of NULLs (and a 0)
    "SELECT gl, COUNT(*), SUM(u) FROM dt GROUP BY gl LIMIT 100",                           # null handling, no null anywhere: empty bitmaps
    "SELECT gi, COUNT(*) FROM dt WHERE c > 1000 GROUP BY gi LIMIT 100",                    # no rows
    "SELECT gi, COUNT(m), COUNT(*), SUM(m) FROM dt GROUP BY gi LIMIT 100",                 # COUNT(col): "count(m)" in the schema, non-null values counted
]


def test_data_table_with_null_vectors(gpu_api):
    from pinot_amd import capi, formats
    host = make_host()
    n = host.total_docs
    rng = np.random.default_rng(5)
    for c, frac in (("gi", 0.1), ("gs", 0.2), ("m", 0.3)):
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(np.flatnonzero(rng.random(n) < frac)), dtype=np.uint8)
    s = NativeSegment(gpu_api, host)
    for sql in NULL_QUERIES:
        q = parse_sql(sql)
        q.flags |= capi.QUERY_FLAG_NULL_HANDLING
        nr = s.execute_native(q, keep_device_table=False)
        got = nr.data_table_v4()
        block = nr.block()
        # enableNullHandling: COUNT(col) keeps its argument in the column name (CountAggregationFunction.java:64-66)
        names = list(q.group_by) + [("count(*)" if a.function == "COUNT" and not a.column else f"{FN.get(a.function, 'count')}({a.column})") for a in q.aggregations]
        types = [host.columns[g].data_type for g in q.group_by] + \
            ["LONG" if a.function == "COUNT" else ("DOUBLE" if a.function in ("SUM", "MIN", "MAX") else "OBJECT") for a in q.aggregations]
        rows = []
        keys = block.group_keys if q.group_by else [()]
        for i, k in enumerate(keys):
            row = [None if v is None else (float(v) if t in ("FLOAT", "DOUBLE") else (int(v) if t in ("INT", "LONG") else v)) for v, t in zip(k, types)]
            for a, col in zip(q.aggregations, block.columns):
                v = col[i]
                if v is not None and a.function == "AVG":
                    v = dt.AvgPair(v)
                elif v is not None and a.function == "MINMAXRANGE":
                    v = dt.MinMaxRangePair(v)
                row.append(v)
            rows.append(row)
        want = dt.build_data_table_v4(names, types, rows, null_handling=True, group_by=bool(q.group_by))
        assert got == want, sql
        p = dt.parse_data_table_v4(got)
        assert p["rows"] == rows and p["null_row_ids"] is not None, sql
        nr.free()
    s.destroy()
