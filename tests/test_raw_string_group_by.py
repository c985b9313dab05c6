"""GROUP BY raw (no-dictionary) STRING / BYTES columns: NoDictionarySingleColumnGroupKeyGenerator's Object2IntOpenHashMap
(pinot-core/.../query/aggregation/groupby/NoDictionarySingleColumnGroupKeyGenerator.java:132-140) and
NoDictionaryMultiColumnGroupKeyGenerator's on-the-fly dictionaries (:60-130): values -> group ids in docId order, trimmed at
numGroupsLimit.  The oracle is checked against a brute force over the rows (no reference golden groups a raw STRING column); the HIP
path — a virtual dictionary built on the device from 64-bit hashes of the values, verified byte for byte (pg_vdict.hip) — against
the oracle."""
import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import HostSegment, build_column

WORDS = ["", "a", "ab", "abc", "Zürich", "東京", "x" * 300, "tab\tsep", "nul\x00inside", "trailing "]


def make(n, seed=5):
    rng = np.random.default_rng(seed)
    s = [WORDS[int(i)] if i < len(WORDS) else f"key-{int(i) * 7919 % 1000003}" for i in rng.integers(0, 400, n)]
    b = [bytes(rng.integers(0, 256, int(k)).astype(np.uint8)) if k else b"" for k in rng.integers(0, 4, n)]   # many collisions on short values
    wide = [f"u{int(i)}" for i in rng.integers(0, 1 << 40, n)]                                                    # nearly all distinct
    data = {"s": s, "b": b, "wide": wide,
            "d": rng.integers(0, 9, n).astype(np.int32), "ri": rng.integers(-50, 50, n).astype(np.int32),
            "m": rng.integers(-1000, 1000, n).astype(np.int32), "f": rng.integers(0, 20, n).astype(np.int32)}
    host = HostSegment("rawStr_0", n)
    host.columns["s"] = build_column("s", s, "STRING", dictionary=False, docs_per_chunk=777)
    host.columns["b"] = build_column("b", b, "BYTES", dictionary=False, raw_version=3)
    host.columns["wide"] = build_column("wide", wide, "STRING", dictionary=False)
    host.columns["d"] = build_column("d", data["d"], "INT", inverted=True)
    host.columns["ri"] = build_column("ri", data["ri"], "INT", dictionary=False)
    host.columns["m"] = build_column("m", data["m"], "INT", dictionary=False)
    host.columns["f"] = build_column("f", data["f"], "INT", inverted=True)
    return host, data


def brute(data, keys, mask, limit=100_000):
    out, admitted = {}, 0
    n = len(data["m"])
    for i in range(n):
        if not mask[i]:
            continue
        k = tuple(data[c][i] if not isinstance(data[c][i], np.generic) else data[c][i].item() for c in keys)
        if k not in out:
            if admitted >= limit:
                continue
            out[k] = [0, 0.0, -np.inf]
            admitted += 1
        a = out[k]
        a[0] += 1
        a[1] += float(data["m"][i])
        a[2] = max(a[2], float(data["m"][i]))
    return out


CASES = [
    ("SELECT s, COUNT(*), SUM(m), MAX(m) FROM rawStr GROUP BY s LIMIT 100000", ["s"], lambda d: np.ones(len(d["m"]), bool), None),
    ("SELECT b, COUNT(*), SUM(m), MAX(m) FROM rawStr WHERE f IN (1, 2, 3) GROUP BY b LIMIT 100000", ["b"], lambda d: np.isin(d["f"], [1, 2, 3]), None),
    ("SELECT s, d, COUNT(*), SUM(m), MAX(m) FROM rawStr GROUP BY s, d LIMIT 100000", ["s", "d"], lambda d: np.ones(len(d["m"]), bool), None),
    ("SELECT d, s, ri, COUNT(*), SUM(m), MAX(m) FROM rawStr WHERE m > 0 GROUP BY d, s, ri LIMIT 100000", ["d", "s", "ri"], lambda d: d["m"] > 0, None),
    ("SELECT b, s, COUNT(*), SUM(m), MAX(m) FROM rawStr WHERE f = 7 GROUP BY b, s LIMIT 100000", ["b", "s"], lambda d: d["f"] == 7, None),
    ("SELECT wide, COUNT(*), SUM(m), MAX(m) FROM rawStr GROUP BY wide LIMIT 1000000", ["wide"], lambda d: np.ones(len(d["m"]), bool), 500),   # trimmed in docId order
    ("SELECT s, COUNT(*), SUM(m), MAX(m) FROM rawStr WHERE m > 5000 GROUP BY s LIMIT 10", ["s"], lambda d: d["m"] > 5000, None),            # nothing matches
]


@pytest.mark.parametrize("sql,keys,mask,limit", CASES)
def test_oracle_matches_brute_force(oracle_api, sql, keys, mask, limit):
    host, data = make(6000)
    o = NativeSegment(oracle_api, host)
    q = parse_sql(sql)
    if limit:
        q.num_groups_limit = limit
    b = o.execute(q)
    want = brute(data, keys, mask(data), limit or 100_000)
    rows = b.rows()
    assert set(rows) == set(want)
    for k, (c, s, mx) in want.items():
        assert rows[k] == [c, s, mx], k
    if limit:
        assert b.stats.num_groups_limit_reached
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 777, 6000, 120_000])
def test_gpu_matches_oracle(gpu_api, oracle_api, n):
    host, _ = make(n, seed=n)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql, _, _, limit in CASES:
        q = parse_sql(sql)
        if limit:
            q.num_groups_limit = limit
        gb, ob = g.execute(q), o.execute(q)
        assert gb.rows() == ob.rows(), sql
        for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs",
                  "num_groups_limit_reached"):
            assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql)
    g.destroy()
    o.destroy()
