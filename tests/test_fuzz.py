"""Randomised differential tests: seeded query generator (tests/fuzz_queries.py) over a segment with every column kind on the
path.  CPU: the oracle's filter against a brute-force numpy evaluation, its groups and SUMs against numpy.  GPU: the HIP path
against the oracle — results, ExecutionStatistics, numGroupsLimit flag — on several hundred queries spanning every aggregation
mode the planner can choose."""
import os

import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from tests.fuzz_queries import Gen, clone, describe, eval_filter, fuzz_segment


@pytest.fixture(scope="module")
def fuzz():
    return fuzz_segment()


def check_against_brute_force(b, q, data, nulls, n, what):
    """One oracle result against a numpy evaluation of the same query; returns the number of aggregations whose groups were checked."""
    checked_groups = 0
    mask = eval_filter(q.filter, data, nulls, n) if q.filter else np.ones(n, bool)
    non_scan = b.stats.num_docs_scanned != int(mask.sum()) and not q.filter   # NonScanBased answers scan nothing
    if not non_scan:
        assert b.stats.num_docs_scanned == int(mask.sum()), what
    if not q.group_by:
        for a, r in zip(q.aggregations, b.aggregation_result()):
            if a.function == "COUNT":
                assert r == int(mask.sum()), what
            elif a.function == "SUM":
                assert r == float(data[a.column][mask].astype(np.float64).sum()), what
            elif a.function == "MIN" and mask.any():
                assert r == float(data[a.column][mask].min()), what
            elif a.function == "MAX" and mask.any():
                assert r == float(data[a.column][mask].max()), what
            elif a.function == "AVG":
                assert r == (float(data[a.column][mask].astype(np.float64).sum()), int(mask.sum())), what
            elif a.function == "MINMAXRANGE":
                v = data[a.column][mask]
                assert r == ((float(v.min()), float(v.max())) if mask.any() else (float("inf"), float("-inf"))), what
            elif a.function == "DISTINCTCOUNT":
                col = data[a.column].astype(str) if data[a.column].dtype == object else data[a.column]
                assert r == frozenset(np.unique(col[mask]).tolist()), what
            elif a.function == "DISTINCTCOUNTHLL":   # registers after hll.offer(value) for every matching doc (numpy restatement)
                from pinot_amd.startree import hll_registers
                dt = "LONG" if data[a.column].dtype == np.int64 else "INT"
                assert r == bytes(hll_registers(data[a.column][mask], dt, a.log2m or 8)), what
        return 0
    limit = q.num_groups_limit or 100_000
    docs = np.flatnonzero(mask)
    code = np.zeros(len(docs), dtype=np.int64)
    uniques = []
    for gcol in q.group_by:
        col = data[gcol].astype(str) if data[gcol].dtype == object else data[gcol]
        uq, inv = np.unique(col[docs], return_inverse=True)
        uniques.append(uq)
        code = code * len(uq) + inv
    ucode, first_pos, inv = np.unique(code, return_index=True, return_inverse=True)
    order = np.argsort(first_pos)                 # groups in order of first appearance (docId order)
    kept = set(order[:limit].tolist())

    def key_of(c):
        parts = []
        for uq in reversed(uniques):
            c, r = divmod(c, len(uq))
            parts.append(uq[r].item() if hasattr(uq[r], "item") else uq[r])
        return tuple(reversed(parts))
    want = {key_of(int(ucode[i])): i for i in kept}
    rows = b.rows()
    assert set(rows) == set(want), what
    if len(ucode) > limit:
        assert b.stats.num_groups_limit_reached, what
    for j, a in enumerate(q.aggregations):
        if a.function in ("COUNT", "SUM", "AVG"):
            vals = np.ones(len(docs)) if a.function == "COUNT" else data[a.column][docs].astype(np.float64)
            acc = np.bincount(inv, weights=vals, minlength=len(ucode))
            cnt = np.bincount(inv, minlength=len(ucode))
            for k, i in want.items():
                exp = int(acc[i]) if a.function == "COUNT" else float(acc[i]) if a.function == "SUM" else (float(acc[i]), int(cnt[i]))
                assert rows[k][j] == exp, (what, k)
            checked_groups += 1
        elif a.function in ("MIN", "MAX", "MINMAXRANGE") and len(docs):
            v = data[a.column][docs].astype(np.float64)
            lo = np.full(len(ucode), np.inf)
            hi = np.full(len(ucode), -np.inf)
            np.minimum.at(lo, inv, v)
            np.maximum.at(hi, inv, v)
            for k, i in want.items():
                exp = float(lo[i]) if a.function == "MIN" else float(hi[i]) if a.function == "MAX" else (float(lo[i]), float(hi[i]))
                assert rows[k][j] == exp, (what, k)
        elif a.function == "DISTINCTCOUNT" and len(ucode) <= 3000:
            col = data[a.column].astype(str) if data[a.column].dtype == object else data[a.column]
            sets = {}
            for g, v in zip(inv.tolist(), col[docs].tolist()):
                sets.setdefault(g, set()).add(v)
            for k, i in want.items():
                assert rows[k][j] == frozenset(sets[i]), (what, k)
    return checked_groups


def test_oracle_matches_brute_force(oracle_api, fuzz):
    host, data, nulls = fuzz
    n = host.total_docs
    o = NativeSegment(oracle_api, host)
    gen = Gen(data, seed=101)
    checked_groups = 0
    for i in range(150):
        q = gen.query()
        checked_groups += check_against_brute_force(o.execute(clone(q)), q, data, nulls, n, f"#{i} {describe(q)}")
    assert checked_groups > 40
    o.destroy()


def test_oracle_raw_key_groups_match_brute_force(oracle_api, fuzz):
    """GROUP BY over no-dictionary columns — FLOAT / DOUBLE keys, raw columns among several keys (the reference's
    NoDictionaryMultiColumnGroupKeyGenerator) — against numpy, numGroupsLimit trimming included."""
    host, data, nulls = fuzz
    n = host.total_docs
    o = NativeSegment(oracle_api, host)
    gen = Gen(data, seed=303)
    checked_groups = 0
    for i in range(60):
        q = gen.raw_key_query()
        checked_groups += check_against_brute_force(o.execute(clone(q)), q, data, nulls, n, f"#{i} {describe(q)}")
    assert checked_groups > 20
    o.destroy()


def _compare(gb, ob, what):
    gr, orr = gb.rows(), ob.rows()
    assert set(gr) == set(orr), what
    for k in orr:
        assert gr[k] == orr[k], (what, k, gr[k], orr[k])
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned, what
    assert gb.stats.num_entries_scanned_post_filter == ob.stats.num_entries_scanned_post_filter, what
    assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached, what
    assert gb.stats.stats_exact == 1, what
    assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, what


# The -m gpu suite runs inside a time limit of the driver's: the default number of random queries per test is 60 % of what the tests were
# written with (PG_FUZZ_SCALE=1 for the full count, larger for a soak; PG_FUZZ_SEED_BASE for other queries)
FUZZ_SCALE = float(os.environ.get("PG_FUZZ_SCALE", "0.6"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,with_valid_docs", [(1, False), (2, False), (3, True), (4, False), (5, True), (6, False)])
def test_gpu_matches_oracle_on_random_queries(gpu_api, oracle_api, fuzz, seed, with_valid_docs):
    # seed 6: 2.5 M docs (1 221 wave tiles: every workgroup of the chip busy, radix slices and hash buckets in the hundreds)
    host, data, nulls = fuzz if seed != 6 else fuzz_segment(2_500_000, seed=6)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    if with_valid_docs:
        valid = np.flatnonzero(np.random.default_rng(seed).random(host.total_docs) < 0.8)
        g.set_queryable_doc_ids(valid)
        o.set_queryable_doc_ids(valid)
    gen = Gen(data, seed=int(os.environ.get("PG_FUZZ_SEED_BASE", "1000")) + seed)   # another base = another 1 280 queries
    unsupported, mismatches = [], []
    n_queries = int((200 if seed != 6 else 80) * FUZZ_SCALE)
    for i in range(n_queries):
        q = gen.query()
        what = f"seed {seed} #{i} {describe(q)}"
        try:
            ob = o.execute(clone(q))
        except capi.NativeError as e:
            with pytest.raises(capi.NativeError):   # what the reference rejects, the GPU path rejects too
                g.execute(clone(q))
            continue
        try:
            gb = g.execute(clone(q))
        except capi.NativeError as e:
            assert e.status == capi.PG_ERR_UNSUPPORTED, (what, e)
            unsupported.append((what, str(e)))
            continue
        try:
            _compare(gb, ob, what)
            if i % 3 == 0:              # the filter-only entry point (FilterOperator + DocIdSetOperator): same docIds
                gd, od = g.filter(clone(q)), o.filter(clone(q))
                assert gd.cardinality() == od.cardinality(), what
                np.testing.assert_array_equal(gd.doc_ids(), od.doc_ids(), err_msg=what)
        except AssertionError as e:     # keep going: one run reports every disagreement
            mismatches.append(str(e)[:600])
    assert not mismatches, "\n".join(mismatches[:12])
    assert len(unsupported) <= n_queries // 5, unsupported
    g.destroy()
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,with_valid_docs", [(0, False), (1, True)])
def test_gpu_matches_oracle_on_random_queries_under_null_handling(gpu_api, oracle_api, fuzz, seed, with_valid_docs):
    """The same random queries with enableNullHandling: `ci` (dictionary, inverted index: a filter and group-by column) and `r` (raw: a filter
    column and an aggregation argument) hold nulls.  The oracle evaluates getTrues / getNulls / getFalses and skips nulls doc at a time; the HIP
    path rewrites the filter tree and joins IS [NOT] NULL partitions: two constructions of the same semantics."""
    host, data, nulls = fuzz
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    if with_valid_docs:
        valid = np.flatnonzero(np.random.default_rng(seed).random(host.total_docs) < 0.8)
        g.set_queryable_doc_ids(valid)
        o.set_queryable_doc_ids(valid)
    gen = Gen(data, seed=7000 + seed)
    unsupported, mismatches, with_nulls = [], [], 0
    n_queries = int(160 * FUZZ_SCALE)
    for i in range(n_queries):
        q = gen.query()
        what = f"null handling, seed {seed} #{i} {describe(q)}"
        oq, gq = clone(q), clone(q)
        oq.flags |= capi.QUERY_FLAG_NULL_HANDLING
        gq.flags |= capi.QUERY_FLAG_NULL_HANDLING
        try:
            ob = o.execute(oq)
        except capi.NativeError:
            with pytest.raises(capi.NativeError):
                g.execute(gq)
            continue
        try:
            gb = g.execute(gq)
        except capi.NativeError as e:
            assert e.status == capi.PG_ERR_UNSUPPORTED, (what, e)
            unsupported.append((what, str(e)))
            continue
        try:
            gr, orr = gb.rows(), ob.rows()
            assert gr == orr, what
            assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned, what
            with_nulls += any(None in k for k in orr) or any(v is None for row in orr.values() for v in row)
            if i % 3 == 0:
                np.testing.assert_array_equal(g.filter(clone(q), null_handling=True).doc_ids(), o.filter(clone(q), null_handling=True).doc_ids(), err_msg=what)
        except AssertionError as e:
            mismatches.append(str(e)[:600])
    assert not mismatches, "\n".join(mismatches[:12])
    assert len(unsupported) <= n_queries // 5, unsupported
    assert with_nulls > 10
    g.destroy()
    o.destroy()


# ---- star-tree: random queries over the dimensions; star-tree answer == plain answer (BaseStarTreeV2Test's differential) ------
def _star_query(rng):
    from pinot_amd.query import AggregationSpec, FilterContext, Predicate, QueryContext, UNBOUNDED
    dims = {"h1": 16, "h2": 10, "h3": 10, "h4": 8}

    def pred(col):
        card = dims[col]
        kind = rng.choice(["EQ", "NOT_EQ", "IN", "NOT_IN", "RANGE"])
        if kind in ("EQ", "NOT_EQ"):
            return FilterContext.pred(Predicate(kind, col, [str(int(rng.integers(-1, card + 1)))]))
        if kind in ("IN", "NOT_IN"):
            return FilterContext.pred(Predicate(kind, col, [str(int(v)) for v in rng.integers(0, card, int(rng.integers(1, 4)))]))
        a, b = sorted(int(v) for v in rng.integers(-1, card + 1, 2))
        return FilterContext.pred(Predicate("RANGE", col, [], str(a), str(b) if rng.random() < 0.7 else UNBOUNDED,
                                            bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and rng.random() < 0.7))
    q = QueryContext(table="gpuBench")
    kids = []
    for col in rng.choice(list(dims), size=int(rng.integers(0, 4)), replace=False):
        x = rng.random()
        if x < 0.6:
            kids.append(pred(col))
        elif x < 0.8:       # OR within one column: the only OR a star-tree takes (StarTreeUtils#extractOrClausePredicates)
            kids.append(FilterContext.or_([pred(col) for _ in range(int(rng.integers(2, 4)))]))
        else:
            kids.append(FilterContext.not_(pred(col)))
    if len(kids) == 1:
        q.filter = kids[0]
    elif kids:
        q.filter = FilterContext.and_(kids)
    q.group_by = [str(c) for c in rng.choice(list(dims), size=int(rng.integers(0, 4)), replace=False)]
    q.has_group_by = bool(q.group_by)
    for _ in range(int(rng.integers(1, 4))):
        fn = str(rng.choice(["COUNT", "SUM", "MIN", "MAX", "DISTINCTCOUNTHLL"]))
        q.aggregations.append(AggregationSpec(fn, None if fn == "COUNT" else ("u" if fn == "DISTINCTCOUNTHLL" else "m")))
    q.limit = 100_000
    return q


def test_oracle_star_tree_equals_plain_on_random_queries(oracle_api):
    from tests.fixtures import synth_star_segment
    o = NativeSegment(oracle_api, synth_star_segment())
    rng = np.random.default_rng(77)
    used = 0
    for i in range(150):
        q = _star_query(rng)
        what = f"#{i} {describe(q)}"
        star = o.execute(clone(q))
        plain_q = clone(q)
        plain_q.flags |= capi.QUERY_FLAG_SKIP_STAR_TREE
        plain = o.execute(plain_q)
        assert star.rows() == plain.rows(), what
        used += star.stats.star_tree_index == 0
    assert used > 100      # nearly every generated query is fit for the tree
    o.destroy()


@pytest.mark.gpu
def test_gpu_star_tree_matches_oracle_on_random_queries(gpu_api, oracle_api):
    from tests.fixtures import synth_star_segment
    host = synth_star_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    rng = np.random.default_rng(78)
    mismatches = []
    for i in range(int(200 * FUZZ_SCALE)):
        q = _star_query(rng)
        what = f"#{i} {describe(q)}"
        gb, ob = g.execute(clone(q)), o.execute(clone(q))
        try:
            assert gb.stats.star_tree_index == ob.stats.star_tree_index, what
            _compare(gb, ob, what)
        except AssertionError as e:
            mismatches.append(str(e)[:600])
    assert not mismatches, "\n".join(mismatches[:12])
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_raw_key_groups_match_oracle_on_random_queries(gpu_api, oracle_api, fuzz):
    """The raw-key generator (FLOAT / DOUBLE keys, raw and dictionary columns mixed, numGroupsLimit): virtual dictionaries on the
    device against the oracle's tuple map — groups, values, statistics."""
    host, data, nulls = fuzz
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    gen = Gen(data, seed=int(os.environ.get("PG_FUZZ_SEED_BASE", "1000")) + 77)
    mismatches = []
    for i in range(int(120 * FUZZ_SCALE)):
        q = gen.raw_key_query()
        what = f"raw keys #{i} {describe(q)}"
        gb, ob = g.execute(clone(q)), o.execute(clone(q))
        try:
            _compare(gb, ob, what)
        except AssertionError as e:
            mismatches.append(str(e)[:600])
    assert not mismatches, "\n".join(mismatches[:12])
    g.destroy()
    o.destroy()
