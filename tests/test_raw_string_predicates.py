"""Predicates over a raw (no-dictionary) single-value STRING column (VERDICT r5 missing #6: `pg_plan.cpp` refused them).  The reference
evaluates them with its raw-value evaluators — value.equals / set.contains / String#compareTo against the bounds
(EqualsPredicateEvaluatorFactory.java, InPredicateEvaluatorFactory.java, RangePredicateEvaluatorFactory.java) — inside a
ScanBasedFilterOperator that visits (and counts) every candidate doc.  The oracle restates those evaluators; the GPU library applies them
once per DISTINCT value of the column's virtual dictionary (pg_vdict.hip) and scans the ids.  The oracle against a Python brute force here,
the library against the oracle under `-m gpu`."""
import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import HostSegment, build_column

SMILE = "\U0001F600"     # a supplementary character: a surrogate pair in UTF-16, lead byte F0 in UTF-8
PRIVATE = ""       # U+E000: lead byte EE in UTF-8 — ABOVE the surrogates in UTF-16, BELOW F0 in UTF-8: the one place the two orders disagree
# (String.compareTo orders UTF-16 code units)
WORDS = ["ant", "bee", "bee!", "cat", "", "dog", "Dog", "eel", "zebra", "été", "￮", SMILE, PRIVATE, "x"]
STATS = ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs")


def table(n, seed):
    rng = np.random.default_rng(seed)
    s = [WORDS[i] for i in rng.integers(0, len(WORDS), n)]
    g = rng.integers(0, 5, n)
    m = rng.integers(-100, 100, n)
    inv = rng.integers(0, 4, n)
    seg = HostSegment("t", n)
    seg.columns["s"] = build_column("s", s, "STRING", dictionary=False)
    seg.columns["g"] = build_column("g", g.tolist(), "INT")
    seg.columns["m"] = build_column("m", m.tolist(), "INT", dictionary=False)
    seg.columns["inv"] = build_column("inv", inv.tolist(), "INT", inverted=True)
    return seg, s, g, m, inv


def utf16(x):
    return x.encode("utf-16-be")     # bytewise order of UTF-16BE = order of the code units = String.compareTo


CASES = [   # (WHERE clause, the predicate on a value)
    ("s = 'bee'", lambda v: v == "bee"),
    ("s != 'bee'", lambda v: v != "bee"),
    ("s = 'nothing'", lambda v: False),                               # not a value of the column: still a full scan
    ("s IN ('ant', 'zebra', 'nothing', '')", lambda v: v in ("ant", "zebra", "")),
    ("s NOT IN ('ant', 'dog')", lambda v: v not in ("ant", "dog")),
    ("s BETWEEN 'b' AND 'dog'", lambda v: utf16("b") <= utf16(v) <= utf16("dog")),
    ("s > 'cat'", lambda v: utf16(v) > utf16("cat")),
    ("s <= 'Dog'", lambda v: utf16(v) <= utf16("Dog")),
    (f"s >= '{PRIVATE}'", lambda v: utf16(v) >= utf16(PRIVATE)),      # the smiley's surrogates sort BELOW U+E000
    (f"s < '{SMILE}'", lambda v: utf16(v) < utf16(SMILE)),
]


@pytest.mark.parametrize("n", [1, 257, 3000])
def test_oracle_against_brute_force(oracle_api, n):
    host, s, g, m, inv = table(n, seed=n)
    o = NativeSegment(oracle_api, host)
    for where, pred in CASES:
        b = o.execute(f"SELECT COUNT(*), SUM(m) FROM t WHERE {where}")
        hit = np.array([pred(v) for v in s], dtype=bool)
        assert b.rows()[()] == [int(hit.sum()), float(m[hit].sum())], where
        assert b.stats.num_entries_scanned_in_filter == n, where            # a raw evaluator never folds: every doc is visited
        # restricted by an inverted-index leaf: only its candidates are visited
        b = o.execute(f"SELECT g, COUNT(*) FROM t WHERE inv = 1 AND {where} GROUP BY g LIMIT 10")
        want = {}
        for k in np.unique(g[hit & (inv == 1)]):
            want[(int(k),)] = [int((hit & (inv == 1) & (g == k)).sum())]
        assert b.rows() == want, where
        if (inv == 1).any():
            assert b.stats.num_entries_scanned_in_filter == int((inv == 1).sum()), where
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 257, 2049, 70_001, 300_007])
def test_gpu_matches_oracle(gpu_api, oracle_api, n):
    host, *_ = table(n, seed=n)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    shapes = ["SELECT COUNT(*), SUM(m) FROM t WHERE {w}",
              "SELECT g, COUNT(*), MAX(m) FROM t WHERE inv IN (1, 2) AND {w} GROUP BY g LIMIT 10",
              "SELECT g, SUM(m) FROM t WHERE ({w} OR m > 90) AND inv != 3 GROUP BY g LIMIT 10",
              "SELECT COUNT(*) FROM t WHERE NOT ({w})"]
    for where, _ in CASES:
        for shape in shapes:
            q = parse_sql(shape.format(w=where))
            q.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
            gb, ob = g.execute(q), o.execute(q)
            assert gb.rows() == ob.rows(), (shape, where)
            for f in STATS:
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (shape, where, f)
    q = parse_sql("SELECT COUNT(*) FROM t WHERE s = 'ant'")     # what stays refused
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    with pytest.raises(capi.NativeError) as e:
        g.execute(q)
    assert e.value.status == capi.PG_ERR_UNSUPPORTED
    g.destroy()
    o.destroy()
