"""Floating SUM parity (north_star: within 1 ulp of the reference) and LONG SUMs that cannot wrap.

The reference adds every value to a double in docId order (SumAggregationFunction.java:160-179).  Its result is a function of
that order: the oracle (same order) differs from the exact sum by up to n * 2^-53 * sum|x|.  The GPU path keeps SUMs exact in
fixed point (pinot_amd/csrc/pg_fixed_point.h) and rounds once, so the assertions are
    |gpu - exact| <= 1 ulp(exact)            exact = math.fsum over the group's values (correctly rounded exact sum)
    |gpu - oracle| <= n * 2^-52 * sum|x|     the reference's own drift bound
on data that is NOT exactly representable, through every aggregation route (LDS table, no GROUP BY, radix partitions, hashed
raw keys) and source encoding (raw / dictionary FLOAT / DOUBLE / LONG).  LONG sums beyond 2^63 must equal the exact integer sum
rounded once (the reference sums longs as doubles and cannot wrap)."""
import math

import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment

pytestmark = pytest.mark.gpu
N = 150_007


@pytest.fixture(scope="module")
def seg(gpu_api, oracle_api):
    rng = np.random.default_rng(77)
    dd_vals = np.sort(rng.normal(0, 1e4, 500))
    df_vals = np.sort(rng.uniform(-50, 50, 300).astype(np.float32))
    lq_vals = np.sort(rng.integers(-2**62, 2**62, 400, dtype=np.int64))
    data = {
        "k": rng.integers(0, 2000, N).astype(np.int32),
        "k2": rng.integers(0, 7, N).astype(np.int32),
        "k3": rng.integers(0, 40, N).astype(np.int32),
        "r": rng.integers(0, 40_000, N).astype(np.int32),
        "inv": rng.integers(0, 5, N).astype(np.int32),
        "dm": rng.normal(0, 1e3, N),                                    # raw DOUBLE, mixed signs, nothing representable
        "dp": rng.lognormal(0, 3, N),                                   # raw DOUBLE, positive, 10 decades of magnitude
        "fm": rng.uniform(-1, 1, N).astype(np.float32),                 # raw FLOAT
        "dd": dd_vals[rng.integers(0, 500, N)],                         # dictionary DOUBLE
        "df": df_vals[rng.integers(0, 300, N)],                         # dictionary FLOAT
        "lbig": rng.integers(-2**62, 2**62, N, dtype=np.int64),         # raw LONG: sums wrap int64
        "lpos": rng.integers(2**61, 2**62, N, dtype=np.int64),          # raw LONG, all positive: far beyond 2^63
        "lq": lq_vals[rng.integers(0, 400, N)],                         # dictionary LONG, wide
    }
    schema = {"k": "INT", "k2": "INT", "k3": "INT", "r": "INT", "inv": "INT", "dm": "DOUBLE", "dp": "DOUBLE", "fm": "FLOAT",
              "dd": "DOUBLE", "df": "FLOAT", "lbig": "LONG", "lpos": "LONG", "lq": "LONG"}
    host = build_segment("sums", data, schema, inverted_index_columns=["inv"],
                         no_dictionary_columns=["r", "dm", "dp", "fm", "lbig", "lpos"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o, data
    g.destroy()
    o.destroy()


def exact_sum(values):
    """correctly rounded exact sum: math.fsum for floats, Python integers for longs"""
    if values.dtype.kind in "iu":
        return float(sum(int(v) for v in values))      # int -> float conversion rounds once (nearest-even)
    return math.fsum(float(v) for v in values)


def check(g, o, data, sql, group_cols, sums, where=None, stride=1):
    gb, ob = g.execute(sql), o.execute(sql)
    grows, orows = gb.rows(), ob.rows()
    assert sorted(grows) == sorted(orows)
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
    sel = np.ones(N, dtype=bool) if where is None else where(data)
    keys = np.stack([data[c][sel] for c in group_cols], axis=1) if group_cols else np.zeros((int(sel.sum()), 0), dtype=np.int64)
    order = np.lexsort(keys.T[::-1]) if group_cols else np.arange(keys.shape[0])
    sk = keys[order]
    bounds = np.flatnonzero(np.any(np.diff(sk, axis=0) != 0, axis=1)) + 1 if group_cols and len(sk) else np.array([], dtype=int)
    starts = np.concatenate([[0], bounds]).astype(int)
    ends = np.concatenate([bounds, [len(sk)]]).astype(int)
    assert len(starts) == len(grows) or not group_cols
    worst = 0.0
    for s, e in list(zip(starts, ends))[::stride]:     # (stride > 1: the exact-sum check samples the groups; the row comparison above never does)
        key = tuple(int(x) for x in sk[s]) if group_cols else ()
        idx = order[s:e]
        gv, ov = grows[key], orows[key]
        for pos, col, is_avg in sums:
            vals = data[col][sel][idx]
            want = exact_sum(vals)
            got = gv[pos][0] if is_avg else gv[pos]
            ref = ov[pos][0] if is_avg else ov[pos]
            if is_avg:
                assert gv[pos][1] == ov[pos][1] == len(idx)
            tol = math.ulp(want) if want != 0 else 0.0
            assert abs(got - want) <= tol, (sql, key, col, got, want)
            drift = len(idx) * 2.0**-52 * float(np.abs(vals.astype(np.float64)).sum())
            assert abs(got - ref) <= drift, (sql, key, col, got, ref)
            worst = max(worst, abs(ref - want) / (math.ulp(want) or 1.0))
    return gb, worst


def test_lds_table_narrow_keys(seg):
    g, o, data = seg
    gb, worst = check(g, o, data, "SELECT k2, SUM(dm), SUM(fm), AVG(dd), SUM(df), SUM(lbig), COUNT(*), SUM(dp) FROM sums GROUP BY k2",
                      ["k2"], [(0, "dm", False), (1, "fm", False), (2, "dd", True), (3, "df", False), (4, "lbig", False), (6, "dp", False)])
    assert worst > 1.0      # the reference's order really drifts by more than an ulp on this data: the exact sum is the honest target


def test_no_group_by(seg):
    g, o, data = seg
    check(g, o, data, "SELECT SUM(dm), SUM(lbig), AVG(fm), SUM(lpos), SUM(lq) FROM sums WHERE inv IN (1, 3)", [],
          [(0, "dm", False), (1, "lbig", False), (2, "fm", True), (3, "lpos", False), (4, "lq", False)],
          where=lambda d: np.isin(d["inv"], [1, 3]))
    check(g, o, data, "SELECT SUM(lpos), SUM(dp) FROM sums", [], [(0, "lpos", False), (1, "dp", False)])


def test_lds_table_wide_keys_behind_scans(seg):
    g, o, data = seg
    check(g, o, data, "SELECT k, SUM(dm), SUM(lbig) FROM sums WHERE r < 30000 AND fm > -0.5 GROUP BY k LIMIT 5000", ["k"],
          [(0, "dm", False), (1, "lbig", False)], where=lambda d: (d["r"] < 30000) & (d["fm"] > -0.5))


def test_radix_partitions(seg):
    g, o, data = seg
    gb, _ = check(g, o, data, "SELECT k, k3, SUM(dm), SUM(lpos), COUNT(*) FROM sums GROUP BY k, k3 LIMIT 1000000", ["k", "k3"],
                  [(0, "dm", False), (1, "lpos", False)], stride=5)
    assert gb.stats.kernel.decode().startswith("pg_radix")
    check(g, o, data, "SELECT k, k3, SUM(fm), AVG(dd) FROM sums WHERE inv != 2 GROUP BY k, k3 LIMIT 1000000", ["k", "k3"],
          [(0, "fm", False), (1, "dd", True)], where=lambda d: d["inv"] != 2, stride=5)


def test_hashed_raw_keys(seg):
    g, o, data = seg
    gb, _ = check(g, o, data, "SELECT r, SUM(dm), SUM(fm), SUM(lbig) FROM sums GROUP BY r LIMIT 100000", ["r"],
                  [(0, "dm", False), (1, "fm", False), (2, "lbig", False)])
    assert gb.stats.kernel.decode() == "pg_hash_group_by"


def test_non_finite_values_keep_ieee_semantics(gpu_api, oracle_api):
    """a column holding NaN / Inf is summed in IEEE double like the reference: NaN and infinities propagate"""
    rng = np.random.default_rng(5)
    n = 20_011
    x = rng.integers(-100, 100, n).astype(np.float64)
    y = x.copy()
    x[[7, 5000]] = np.inf
    y[[3]] = np.nan
    z = x.copy()
    z[[9000]] = -np.inf
    data = {"g": rng.integers(0, 4, n).astype(np.int32), "x": x, "y": y, "z": z}
    host = build_segment("nf", data, {"g": "INT", "x": "DOUBLE", "y": "DOUBLE", "z": "DOUBLE"}, no_dictionary_columns=["x", "y", "z"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for q in ("SELECT SUM(x), SUM(y), SUM(z) FROM nf", "SELECT g, SUM(x), SUM(y), SUM(z) FROM nf GROUP BY g"):
        gr, orr = g.execute(q).rows(), o.execute(q).rows()
        assert sorted(gr) == sorted(orr)
        for k in orr:
            for a, b in zip(gr[k], orr[k]):
                assert (math.isnan(a) and math.isnan(b)) or a == b, (q, k, a, b)
    g.destroy()
    o.destroy()
