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
    kf = rng.choice(np.array([0.0, -0.0, 1.5, -2.25, 3.0e10, np.nan, np.inf, -np.inf, 1e-40], dtype=np.float32), n)   # -0.0 / 0.0 are two keys, NaN one
    kf[rng.integers(0, n, 30)] = np.frombuffer(np.array([0x7FC00001, 0xFFC12345], dtype=np.uint32).tobytes(), dtype=np.float32)[rng.integers(0, 2, 30)]  # other NaN payloads
    kd = rng.integers(-300, 300, n) * 0.1                        # non-dyadic doubles: the key is the exact bit pattern
    kd[rng.integers(0, n, 60)] = rng.choice(np.array([np.nan, -0.0, 0.0, 1e300]), 60)
    data = {
        "ki": ki, "kl": kl, "kw": kw, "kf": kf, "kd": kd,
        "f": rng.integers(0, 20, n).astype(np.int32),
        "r": rng.integers(0, 100_000, n).astype(np.int32),
        "m": rng.integers(-1000, 1 << 20, n).astype(np.int32),
        "ml": rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64),
        "md": rng.integers(-(1 << 30), 1 << 30, n) * 0.25,      # dyadic: double sums are exact in any order
    }
    schema = {"ki": "INT", "kl": "LONG", "kw": "LONG", "kf": "FLOAT", "kd": "DOUBLE", "f": "INT", "r": "INT", "m": "INT", "ml": "LONG", "md": "DOUBLE"}
    host = build_segment("rawKeys_0", data, schema, inverted_index_columns=["f"], no_dictionary_columns=["ki", "kl", "kw", "kf", "kd", "r", "m", "ml", "md"])
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


# FLOAT / DOUBLE raw keys (Float2Int / Double2IntOpenHashMap) and any group-by with a raw column among several
# (NoDictionaryMultiColumnGroupKeyGenerator.java:60-130): (query, numGroupsLimit)
MULTI_RAW_QUERIES = [
    ("SELECT kf, COUNT(*), SUM(m) FROM rawKeys GROUP BY kf LIMIT 100000", None),
    ("SELECT kd, COUNT(*), MAX(ml) FROM rawKeys WHERE f IN (1, 4, 7) GROUP BY kd LIMIT 100000", None),
    ("SELECT ki, kf, COUNT(*), SUM(m) FROM rawKeys GROUP BY ki, kf LIMIT 1000000", 1_000_000),
    ("SELECT f, kd, COUNT(*), MIN(m), AVG(md) FROM rawKeys WHERE r BETWEEN 20000 AND 70000 GROUP BY f, kd LIMIT 100000", None),
    ("SELECT kl, f, ki, SUM(ml) FROM rawKeys GROUP BY kl, f, ki LIMIT 1000000", 1_000_000),
    ("SELECT ki, kl, COUNT(*) FROM rawKeys GROUP BY ki, kl LIMIT 1000000", 3000),          # numGroupsLimit trims in docId order
    ("SELECT kd, kf, COUNT(*) FROM rawKeys WHERE r > 1000000 GROUP BY kd, kf LIMIT 10", None),   # nothing matches
]


def key_bits(data, col):
    """The key the fastutil maps compare: the value for INT / LONG, floatToIntBits / doubleToLongBits for FLOAT / DOUBLE."""
    v = data[col]
    if v.dtype == np.float32:
        b = v.view(np.uint32).astype(np.int64)
        return np.where(np.isnan(v), 0x7FC00000, b)
    if v.dtype == np.float64:
        b = v.view(np.int64)
        return np.where(np.isnan(v), 0x7FF8000000000000, b)
    return v.astype(np.int64)


def key_repr(data, col, bits):
    """executor._key_repr of the value behind `bits`."""
    v = data[col]
    if v.dtype == np.float32:
        x = float(np.array([bits], dtype=np.int64).astype(np.uint32).view(np.float32)[0])
    elif v.dtype == np.float64:
        x = float(np.array([bits], dtype=np.int64).view(np.float64)[0])
    else:
        return int(bits)
    if x != x:
        return "NaN"
    if x == 0.0:
        return "-0.0" if np.signbit(x) else 0.0
    return x


def test_oracle_multi_column_raw_keys_match_numpy(oracle_api):
    host, data = raw_key_segment(40_000)
    o = NativeSegment(oracle_api, host)
    for cols, mask, limit in ((["kf"], np.ones(40_000, bool), 100_000), (["ki", "kf"], np.ones(40_000, bool), 100_000),
                              (["f", "kd"], (data["r"] >= 20000) & (data["r"] <= 70000), 100_000), (["ki", "kl"], np.ones(40_000, bool), 900)):
        where = " WHERE r BETWEEN 20000 AND 70000" if not mask.all() else ""
        q = parse_sql(f"SELECT {', '.join(cols)}, COUNT(*), SUM(m) FROM rawKeys{where} GROUP BY {', '.join(cols)} LIMIT 1000000")
        q.num_groups_limit = limit
        b = o.execute(q)
        docs = np.flatnonzero(mask)
        keys = np.stack([key_bits(data, c)[docs] if c != "f" else data["f"][docs].astype(np.int64) for c in cols], axis=1)
        want, order = {}, []
        for d, k in zip(docs, map(tuple, keys)):
            if k not in want:
                if len(order) >= limit:
                    continue
                want[k] = [0, 0.0]
                order.append(k)
            want[k][0] += 1
            want[k][1] += float(data["m"][d])
        rows = b.rows()
        assert len(rows) == len(want), cols
        assert bool(b.stats.num_groups_limit_reached) == (len(want) >= limit)
        for k, (cnt, sm) in want.items():
            kk = tuple(key_repr(data, c, kb) if c != "f" else int(kb) for c, kb in zip(cols, k))
            assert rows[kk] == [cnt, sm], (cols, kk)
    o.destroy()


def test_oracle_dictionary_string_with_raw_int_keys(oracle_api):
    host = build_segment("s_0", {"s": np.array(["a", "b", "a"]), "m": np.arange(3, dtype=np.int32)}, {"s": "STRING", "m": "INT"},
                         no_dictionary_columns=["m"])
    o = NativeSegment(oracle_api, host)
    assert len(o.execute("SELECT s, m, COUNT(*) FROM s GROUP BY s, m LIMIT 10").rows()) == 3   # dictionary STRING + raw INT: fine
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
@pytest.mark.parametrize("q,limit", MULTI_RAW_QUERIES)
def test_gpu_virtual_dictionary_group_by_matches_oracle(gpu_api, oracle_api, q, limit):
    """Raw FLOAT / DOUBLE keys and raw columns among several group-by columns run through the columns' virtual dictionaries
    (pg_vdict.hip): same groups, same values, same trimming as the oracle's tuple map."""
    host, _ = raw_key_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    qg, qo = parse_sql(q), parse_sql(q)
    if limit:
        qg.num_groups_limit = qo.num_groups_limit = limit
    for _ in range(2):   # the second run takes the cached dictionary and the cached plan
        gb, ob = g.execute(qg), o.execute(qo)
        gr, orr = gb.rows(), ob.rows()
        assert len(gr) == len(orr)
        assert set(gr) == set(orr)
        for k in orr:
            assert gr[k] == orr[k], (k, gr[k], orr[k])
        assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
        assert gb.stats.num_entries_scanned_post_filter == ob.stats.num_entries_scanned_post_filter
        assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_raw_long_max_value_key(gpu_api):
    host, data = raw_key_segment(5_000)
    # Long.MAX_VALUE is the hash table's empty marker once biased: such a column is grouped through its virtual dictionary instead
    data = dict(data)
    data["kl"] = data["kl"].copy()
    data["kl"][17] = np.iinfo(np.int64).max
    schema = {"kl": "LONG", "m": "INT"}
    host2 = build_segment("rawKeys_1", {"kl": data["kl"], "m": data["m"]}, schema, no_dictionary_columns=["kl", "m"])
    g = NativeSegment(gpu_api, host2)
    from tests.oracle_binding import load_oracle
    o = NativeSegment(load_oracle(), host2)
    sql = "SELECT kl, COUNT(*), SUM(m) FROM rawKeys GROUP BY kl LIMIT 100000"
    assert g.execute(sql).rows() == o.execute(sql).rows()
    assert (np.iinfo(np.int64).max,) in g.execute(sql).rows()
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_hash_group_by_retries_with_more_buckets(gpu_api, oracle_api, gpu_knobs):
    """A hash bucket whose distinct keys overflow its LDS table is met with four times as many buckets, not with an error after the
    work: start the nearly-all-distinct key column at 16 buckets (PG_HASH_FIRST_BUCKETS), far too few for 150 000 keys."""
    host, _ = raw_key_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    gpu_knobs(PG_HASH_FIRST_BUCKETS="16")
    qg, qo = parse_sql("SELECT kw, COUNT(*), SUM(m) FROM rawKeys GROUP BY kw LIMIT 1000000"), parse_sql("SELECT kw, COUNT(*), SUM(m) FROM rawKeys GROUP BY kw LIMIT 1000000")
    qg.num_groups_limit = qo.num_groups_limit = 1_000_000
    gb, ob = g.execute(qg), o.execute(qo)
    assert gb.rows() == ob.rows() and len(gb.rows()) > 140_000
    g.destroy()
    o.destroy()
