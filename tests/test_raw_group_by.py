"""GROUP BY one no-dictionary INT / LONG column: the reference's NoDictionarySingleColumnGroupKeyGenerator
(pinot-core/.../query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:53-90,241-265) maps raw values to group
ids in docId order, trimmed at numGroupsLimit.  The oracle is checked against numpy here; the HIP path (64-bit keys in the hash
group-by) against the oracle in the gpu tests below."""
import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment


def raw_key_segment(n=150_000, seed=11):
    rng = np.random.default_rng(seed)
    ki = rng.integers(-3000, 3000, n).astype(np.int32)
    ki[rng.integers(0, n, 50)] = np.iinfo(np.int32).min
    ki[rng.integers(0, n, 50)] = np.iinfo(np.int32).max
    kl = (rng.integers(-2000, 2000, n).astype(np.int64) * 0x1_0000_0001_7) ^ 0x55
    kl[rng.integers(0, n, 40)] = np.iinfo(np.int64).min
    kl[rng.integers(0, n, 40)] = np.iinfo(np.int64).max - 1
    kw = rng.integers(0, 1 << 40, n).astype(np.int64)            # nearly all distinct: numGroupsLimit trims
    data = {
        "ki": ki, "kl": kl, "kw": kw,
        "f": rng.integers(0, 20, n).astype(np.int32),
        "r": rng.integers(0, 100_000, n).astype(np.int32),
        "m": rng.integers(-1000, 1 << 20, n).astype(np.int32),
        "ml": rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64),
        "md": rng.integers(-(1 << 30), 1 << 30, n) * 0.25,      # dyadic: double sums are exact in any order
    }
    schema = {"ki": "INT", "kl": "LONG", "kw": "LONG", "f": "INT", "r": "INT", "m": "INT", "ml": "LONG", "md": "DOUBLE"}
    host = build_segment("rawKeys_0", data, schema, inverted_index_columns=["f"], no_dictionary_columns=["ki", "kl", "kw", "r", "m", "ml", "md"])
    return host, data


RAW_GROUP_QUERIES = [
    ("SELECT ki, COUNT(*), SUM(m), MAX(ml) FROM rawKeys GROUP BY ki LIMIT 100000", None),
    ("SELECT kl, COUNT(*), MIN(m), AVG(md) FROM rawKeys GROUP BY kl LIMIT 100000", None),
    ("SELECT kl, SUM(ml), MINMAXRANGE(m) FROM rawKeys WHERE f IN (1, 4, 7) AND r BETWEEN 20000 AND 70000 GROUP BY kl LIMIT 100000", None),
    ("SELECT ki, MAX(md) FROM rawKeys WHERE r < 500 GROUP BY ki LIMIT 100000", None),
    ("SELECT kw, COUNT(*), SUM(m) FROM rawKeys GROUP BY kw LIMIT 1000000", 1_000_000),
    ("SELECT kw, COUNT(*), SUM(m) FROM rawKeys GROUP BY kw LIMIT 1000000", None),        # default numGroupsLimit (100 000) trims
    ("SELECT kw, SUM(ml) FROM rawKeys WHERE f = 3 GROUP BY kw LIMIT 1000000", 2500),
    ("SELECT ki, COUNT(*) FROM rawKeys WHERE r > 1000000 GROUP BY ki LIMIT 10", None),   # nothing matches
]


def numpy_groups(data, key, mask, limit):
    """value -> docIds of the first `limit` distinct values in docId order (the generator's trimming)."""
    docs = np.flatnonzero(mask)
    vals = data[key][docs]
    uniq, first = np.unique(vals, return_index=True)
    keep = set(uniq[np.argsort(first)][:limit].tolist())
    return {int(v): docs[vals == v] for v in keep} if len(keep) <= 20000 else keep


def test_oracle_raw_group_by_matches_numpy(oracle_api):
    host, data = raw_key_segment(30_000)
    o = NativeSegment(oracle_api, host)
    b = o.execute("SELECT ki, COUNT(*), SUM(m), MAX(ml) FROM rawKeys GROUP BY ki LIMIT 100000")
    want = numpy_groups(data, "ki", np.ones(30_000, bool), 100_000)
    rows = b.rows()
    assert sorted(k[0] for k in rows) == sorted(want)
    for v, docs in want.items():
        assert rows[(v,)] == [len(docs), float(data["m"][docs].astype(np.float64).sum()), float(data["ml"][docs].max())]
    # filter + LONG keys + numGroupsLimit: the first 700 distinct values in docId order survive
    q = parse_sql("SELECT kl, COUNT(*), MIN(m) FROM rawKeys WHERE f IN (1, 4, 7) GROUP BY kl LIMIT 100000")
    q.num_groups_limit = 700
    b = o.execute(q)
    mask = np.isin(data["f"], [1, 4, 7])
    want = numpy_groups(data, "kl", mask, 700)
    rows = b.rows()
    assert len(rows) == 700 and b.stats.num_groups_limit_reached
    assert sorted(k[0] for k in rows) == sorted(want)
    for v, docs in want.items():
        assert rows[(v,)] == [len(docs), float(data["m"][docs].min())]
    o.destroy()


def test_oracle_rejects_other_raw_group_by_shapes(oracle_api):
    host, _ = raw_key_segment(2_000)
    o = NativeSegment(oracle_api, host)
    for q in ("SELECT ki, kl, COUNT(*) FROM rawKeys GROUP BY ki, kl LIMIT 10", "SELECT md, COUNT(*) FROM rawKeys GROUP BY md LIMIT 10"):
        with pytest.raises(capi.NativeError):
            o.execute(q)
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("q,limit", RAW_GROUP_QUERIES)
def test_gpu_raw_group_by_matches_oracle(gpu_api, oracle_api, q, limit):
    host, _ = raw_key_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    qg, qo = parse_sql(q), parse_sql(q)
    if limit:
        qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    gr, orr = gb.rows(), ob.rows()
    assert sorted(gr) == sorted(orr)
    for k in orr:
        assert gr[k] == orr[k], (k, gr[k], orr[k])
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
    assert gb.stats.num_entries_scanned_post_filter == ob.stats.num_entries_scanned_post_filter
    assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_raw_group_by_rejections(gpu_api):
    host, data = raw_key_segment(5_000)
    g = NativeSegment(gpu_api, host)
    for q in ("SELECT ki, kl, COUNT(*) FROM rawKeys GROUP BY ki, kl LIMIT 10", "SELECT md, COUNT(*) FROM rawKeys GROUP BY md LIMIT 10"):
        with pytest.raises(capi.NativeError):
            g.execute(q)
    g.destroy()
    # Long.MAX_VALUE is the hash table's empty marker: refused, not answered wrongly
    data = dict(data)
    data["kl"] = data["kl"].copy()
    data["kl"][17] = np.iinfo(np.int64).max
    schema = {"kl": "LONG", "m": "INT"}
    host2 = build_segment("rawKeys_1", {"kl": data["kl"], "m": data["m"]}, schema, no_dictionary_columns=["kl", "m"])
    g = NativeSegment(gpu_api, host2)
    with pytest.raises(capi.NativeError):
        g.execute("SELECT kl, COUNT(*) FROM rawKeys GROUP BY kl LIMIT 10")
    g.destroy()
