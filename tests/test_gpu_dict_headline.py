"""Config 3 in Pinot's DEFAULT encoding: the scan column and the value column are
dictionary-encoded — fixed-bit dictId streams (FixedBitSVForwardIndexReaderV2.java:65-99), the range predicate is a dictId interval
(RangePredicateEvaluatorFactory.java:126-167), SUM / MIN / MAX read dictionary.get(dictId) (DataFetcher.java:335-386).  The pg_fast_dictrange_s
family (pg_kernels_specd.hip) runs these plans: _a computes the value of an arithmetic dictionary, _g gathers it, _r reads a raw value column
next to a dictionary-encoded scan column.  Results and ExecutionStatistics equal the oracle's at the four sizes of
test_wave_specialised_variant_matches_oracle (fewer tiles than workgroups, odd and even stage counts, a ragged last tile), and the rows over
the identity dictionaries equal the rows of the raw-column query over the same docs."""
import os

import numpy as np
import pytest

from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql

pytestmark = pytest.mark.gpu

COLUMNS = ["c_inv1", "c_inv2", "r_int", "g1", "g2", "m", "r_int_d", "m_d", "r_int_s", "m_s"]
knobs_off = not (os.environ.get("PG_NO_SPECD") or os.environ.get("PG_SPECD_NO_AFFINE") or os.environ.get("PG_NO_DENSE_FUSED") or os.environ.get("PG_FORCE_INTERPRETER"))
IDX = "c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)"

QUERIES = [
    (synth.QUERY_CFG3_DICT, "pg_fast_dictrange_s_a"),
    (synth.QUERY_NORTH_STAR_DICT, "pg_fast_dictrange_s_a"),
    (synth.QUERY_CFG3_SPARSE, "pg_fast_dictrange_s_g"),
    (f"SELECT g2, COUNT(*), MIN(m_s), MAX(m_s), SUM(m_s) FROM gpuBench WHERE {IDX} AND r_int_s BETWEEN 750000 AND 2249999 "
     "GROUP BY g2 ORDER BY g2 LIMIT 10000", "pg_fast_dictrange_s_g"),
    (f"SELECT g1, g2, COUNT(*), SUM(m_s) FROM gpuBench WHERE {IDX} AND r_int_s BETWEEN 750000 AND 2249999 "
     "GROUP BY g1, g2 ORDER BY g1, g2 LIMIT 10000", "pg_fast_dictrange_s_g"),
    # one side raw, the other dictionary-encoded
    (f"SELECT g1, SUM(m_d), MAX(m_d) FROM gpuBench WHERE {IDX} AND r_int BETWEEN 250000 AND 749999 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_a"),
    (f"SELECT g1, SUM(m), MIN(m) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_r"),
    (f"SELECT g2, g1, COUNT(*), SUM(m_d) FROM gpuBench WHERE {IDX} AND r_int_s > 1000000 GROUP BY g2, g1 ORDER BY g2, g1 LIMIT 10000", "pg_fast_dictrange_s_a"),
    # complemented posting groups, an equality (a one-dictId interval), ranges open at one end, a value outside the dictionary
    ("SELECT g1, COUNT(*), MIN(m_d), MAX(m_d), SUM(m_d) FROM gpuBench WHERE c_inv1 NOT IN (0, 7) AND c_inv2 = 1 AND r_int_d BETWEEN 100 AND 900000 "
     "GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_a"),
    ("SELECT g1, SUM(m_s) FROM gpuBench WHERE c_inv1 IN (1, 2, 3, 4, 5) AND r_int_d < 10 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_g"),
    ("SELECT g1, SUM(m_d), COUNT(*) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_d = 4711 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_a"),
    ("SELECT g1, SUM(m_d) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_s = 14135 GROUP BY g1 ORDER BY g1 LIMIT 1000", None),   # 14135 may be no dictionary value
    ("SELECT g1, MAX(m_s) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_d >= 999990 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_fast_dictrange_s_g"),
    ("SELECT g1, SUM(m_d) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_d BETWEEN 2000000 AND 3000000 GROUP BY g1 LIMIT 1000", None),   # empty interval
    # the family's other filter shapes: a lone range scan, inverted-index leaves only, no filter
    ("SELECT g1, g2, SUM(m_d) FROM gpuBench WHERE r_int_d BETWEEN 250000 AND 749999 GROUP BY g1, g2 ORDER BY g1, g2 LIMIT 10000", "pg_specd_scan_a"),
    ("SELECT g1, SUM(m_s), COUNT(*) FROM gpuBench WHERE r_int_s < 21 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_scan_g"),
    ("SELECT g1, SUM(m), COUNT(*) FROM gpuBench WHERE r_int_d > 500000 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_scan_r"),
    ("SELECT g1, SUM(m_d), MAX(m_d) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_scan_a"),
    (f"SELECT g1, SUM(m_d) FROM gpuBench WHERE {IDX} GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_index_a"),
    ("SELECT g2, g1, MAX(m_s), COUNT(*) FROM gpuBench WHERE c_inv1 NOT IN (3, 4) GROUP BY g2, g1 ORDER BY g2, g1 LIMIT 10000", "pg_specd_index_g"),
    ("SELECT g1, SUM(m_d), MAX(m_d) FROM gpuBench GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_none_a"),
    ("SELECT g1, g2, COUNT(*), MIN(m_s) FROM gpuBench GROUP BY g1, g2 ORDER BY g1, g2 LIMIT 10000", "pg_specd_none_g"),
    # a scan over a <= 8-bit dictionary column in front of a dictionary-encoded value
    ("SELECT g1, SUM(m_d) FROM gpuBench WHERE g2 BETWEEN 10 AND 30 GROUP BY g1 ORDER BY g1 LIMIT 1000", "pg_specd_scan_a"),
    # no GROUP BY (AggregationOperator's shapes): the same kernels with zero group columns — the slot is the lane's replica
    (f"SELECT SUM(m_d), MAX(m_d) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999", "pg_fast_dictrange_s_a"),
    (f"SELECT COUNT(*), MIN(m_s), SUM(m_s) FROM gpuBench WHERE {IDX} AND r_int_s BETWEEN 750000 AND 2249999", "pg_fast_dictrange_s_g"),
    (f"SELECT SUM(m), MIN(m) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999", "pg_fast_dictrange_s_r"),
    ("SELECT SUM(m_d), COUNT(*) FROM gpuBench WHERE r_int_d BETWEEN 250000 AND 749999", "pg_specd_scan_a"),
    ("SELECT AVG(m_d), MINMAXRANGE(m_d) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999", "pg_specd_scan_a"),
    (f"SELECT SUM(m_s), MAX(m_s) FROM gpuBench WHERE {IDX}", "pg_specd_index_g"),
    ("SELECT SUM(m_d) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_d BETWEEN 2000000 AND 3000000", None),   # empty interval: no doc, a NULL-less zero row
]
STATS = ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs")


def _as_s(name):
    """the shared-stage frame's kernel (pg_kernels_specw.hip) under the name of its independent-wavefront twin"""
    return name.replace("pg_fast_dictrange_w", "pg_fast_dictrange_s").replace("pg_specw_", "pg_specd_").replace("_dma", "")


# the family's variants: "s" — the product's: independent wavefronts (pg_kernels_specd.hip), the headline shape's columns by LDS-DMA
# (pg_fast_dictrange_s_*_dma) where the planner finds room for two column areas per strip; "n" — PG_SPECD_NO_DMA: register-staged everywhere;
# "w" — PG_SPECW=1: the shared-stage frame (pg_fast_dictrange_w, pg_kernels_specw.hip) wherever two stage buffers fit — measured slower, kept parity-green
FRAME_ENV = {"s": {}, "n": {"PG_SPECD_NO_DMA": "1"}, "w": {"PG_SPECW": "1"}}


@pytest.fixture(scope="module", params=[("s", n) for n in (1, 2049, 70_001, 700_001, 9_030_011)] + [("n", n) for n in (2049, 700_001)] +
                [("w", n) for n in (2049, 700_001, 3_000_017)], ids=lambda p: f"{p[0]}-{p[1]}")
def pair(request, gpu_api, oracle_api):
    frame, n = request.param
    before = {k: os.environ.get(k) for k in ("PG_SPECW", "PG_SPECD_NO_DMA")}
    for k in before:
        os.environ.pop(k, None)
    os.environ.update(FRAME_ENV[frame])
    gpu_api.call("options_reload")
    host = synth.generate_segment(n, segment_index=3, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    g.frame = frame
    yield g, o
    g.destroy()
    o.destroy()
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    gpu_api.call("options_reload")


@pytest.mark.parametrize("sql,kernel", QUERIES)
def test_dictionary_encoded_headline_matches_oracle(pair, sql, kernel):
    g, o = pair
    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
    gb, ob = g.execute(qc), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    check = kernel and knobs_off and gb.stats.num_total_docs >= 700_001   # small segments keep sparse (CSR) postings: the interpreted leaves
    if check:
        ran = gb.stats.kernel.decode()
        assert (ran if g.frame == "n" else _as_s(ran)) == kernel
        if g.frame == "w" and sql in (synth.QUERY_CFG3_DICT, synth.QUERY_CFG3_SPARSE):
            assert ran.startswith("pg_fast_dictrange_w_")   # config 3's table leaves room for two stage buffers (the north star's 2-key table does not)
        if g.frame == "s" and sql in (synth.QUERY_CFG3_DICT, synth.QUERY_CFG3_SPARSE) and not os.environ.get("PG_SPECD_NO_DMA") and not os.environ.get("PG_SPECW"):
            assert ran.endswith("_dma")                     # ... and for two column areas per strip
    gb2 = g.execute(qc)   # the plan's second execution (cached plan, observed rates): the same kernel, the same answer
    assert gb2.rows() == ob.rows()
    if check:
        assert gb2.stats.kernel.decode() == gb.stats.kernel.decode()


def test_identity_dictionaries_give_the_raw_columns_rows(pair):
    """r_int_d / m_d hold the docs of r_int / m: the dictionary-encoded query returns the raw query's rows (and statistics)."""
    g, _ = pair
    for raw, enc in ((synth.QUERY_CFG3, synth.QUERY_CFG3_DICT), (synth.QUERY_NORTH_STAR, synth.QUERY_NORTH_STAR_DICT)):
        a, b = g.execute(raw), g.execute(enc)
        assert a.rows() == b.rows()
        for f in STATS:
            assert getattr(a.stats, f) == getattr(b.stats, f), f


@pytest.mark.parametrize("n", [2049, 700_001, 3_000_017])
def test_dictionary_encoded_headline_behind_an_upsert_snapshot(gpu_api, oracle_api, n):
    """FilterPlanNode.run's outer AND with queryableDocIds: the snapshot is ANDed in after the scan's candidates have been counted."""
    host = synth.generate_segment(n, segment_index=5, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    rng = np.random.default_rng(n)
    for keep in (0.9, 0.5, 0.02):
        ids = np.flatnonzero(rng.random(n) < keep)
        g.set_queryable_doc_ids(ids)
        o.set_queryable_doc_ids(ids)
        for sql, kernel in ((synth.QUERY_CFG3_DICT, "pg_fast_dictrange_st_a"), (synth.QUERY_CFG3_SPARSE, "pg_fast_dictrange_st_g"),
                            (synth.QUERY_NORTH_STAR_DICT, "pg_fast_dictrange_st_a"),
                            ("SELECT g1, SUM(m_d) FROM gpuBench WHERE r_int_d BETWEEN 250000 AND 749999 GROUP BY g1 LIMIT 1000", None),
                            ("SELECT g1, SUM(m_s), MAX(m_s) FROM gpuBench GROUP BY g1 LIMIT 1000", None),
                            (f"SELECT SUM(m_d), MAX(m_d), COUNT(*) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999", "pg_fast_dictrange_st_a"),
                            ("SELECT SUM(m_s) FROM gpuBench", None),
                            (f"SELECT COUNT(*) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999", None),   # filter only behind the snapshot
                            ("SELECT COUNT(*) FROM gpuBench WHERE r_int_s < 900000", None)):
            gb, ob = g.execute(sql), o.execute(sql)
            assert gb.rows() == ob.rows(), (sql, keep)
            for f in STATS:
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql, keep)
            if kernel and knobs_off and n >= 65536 and keep >= 0.5:   # dense snapshots are bitmap containers: the arithmetic (dense) form
                assert _as_s(gb.stats.kernel.decode()).replace("pg_fast_dictrange_wt", "pg_fast_dictrange_st") == kernel, (sql, keep)
    g.destroy()
    o.destroy()


# ---- no GROUP BY, no filter over one dictionary-encoded INT column: pg_nogroup_da (arithmetic dictionary) / pg_nogroup_dg (gathered) -----------
NOGROUP = [
    ("SELECT SUM(m_d), MIN(m_d), MAX(m_d), COUNT(*) FROM gpuBench", "pg_nogroup_da"),
    ("SELECT AVG(m_d), MINMAXRANGE(m_d) FROM gpuBench", "pg_nogroup_da"),
    ("SELECT SUM(m_s), MIN(m_s), MAX(m_s), COUNT(*) FROM gpuBench", "pg_nogroup_dg"),
    ("SELECT SUM(r_int_s) FROM gpuBench", "pg_nogroup_dg"),
    ("SELECT SUM(g1), MAX(g1), MIN(g1) FROM gpuBench", "pg_nogroup_da"),          # a 7-bit column
    ("SELECT SUM(c_inv1), COUNT(*) FROM gpuBench", "pg_nogroup_da"),
    ("SELECT SUM(m_d), SUM(m) FROM gpuBench", None),                               # two columns: not this kernel
]


@pytest.mark.parametrize("sql,kernel", NOGROUP)
def test_no_group_by_over_a_dictionary_encoded_column(pair, sql, kernel):
    g, o = pair
    gb, ob = g.execute(sql), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    ran = gb.stats.kernel.decode()
    if kernel and not os.environ.get("PG_NO_SCAN_PIPE") and not os.environ.get("PG_FORCE_INTERPRETER"):
        # no index involved: every segment size takes it (a gathered dictionary of <= 36 K values — the small segments' — from its copy in LDS)
        assert ran == kernel or (kernel == "pg_nogroup_dg" and ran == "pg_nogroup_dl"), sql
    elif not kernel:
        assert not ran.startswith("pg_nogroup_d")


@pytest.mark.parametrize("n", [1, 511, 513, 2049, 300_007])
def test_no_group_by_dictionary_widths(gpu_api, oracle_api, n):
    """every dictId width the template is instantiated for that a segment of this size can hold, arithmetic and not"""
    from pinot_amd.segment import build_segment
    rng = np.random.default_rng(n)
    data, schema = {}, {}
    for card in (2, 5, 17, 130, 300, 1000, 5000, 20000, 70000, 200000):
        if card > max(n, 2):
            continue
        ids = rng.integers(0, card, n)
        ids[: min(card, n)] = np.arange(min(card, n))                                             # every dictionary value occurs
        data[f"a{card}"] = (ids * 3 - 7).astype(np.int32)                                          # value = -7 + 3 x dictId
        data[f"s{card}"] = np.sort(rng.choice(4_000_000, card, replace=False) - 2_000_000)[ids].astype(np.int32)   # no arithmetic form
        schema[f"a{card}"] = schema[f"s{card}"] = "INT"
    host = build_segment("widths", data, schema)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for name, values in data.items():
        sql = f"SELECT SUM({name}), MIN({name}), MAX({name}), COUNT(*), AVG({name}) FROM widths"
        gb, ob = g.execute(sql), o.execute(sql)
        assert gb.rows() == ob.rows(), name
        v = values.astype(np.int64)
        assert gb.aggregation_result()[:4] == [float(v.sum()), float(v.min()), float(v.max()), n], name
        if not os.environ.get("PG_NO_SCAN_PIPE") and not os.environ.get("PG_FORCE_INTERPRETER"):
            card = len(np.unique(values))
            if card == 1:
                want = "pg_nogroup_dl"      # one value: no step to speak of — gathered (from LDS)
            elif name.startswith("a") or card == 2:
                want = "pg_nogroup_da"      # (any two values are an arithmetic dictionary)
            else:
                want = "pg_nogroup_dl" if card <= 36 * 1024 else "pg_nogroup_dg"
            assert gb.stats.kernel.decode() == want, (name, gb.stats.kernel)
    g.destroy(); o.destroy()


# ---- filter only over a dictionary-encoded scan column: pg_dictrange_fo ----------------------------------------------------------------------------
FILTER_ONLY = [
    (f"SELECT COUNT(*) FROM gpuBench WHERE {IDX} AND r_int_d BETWEEN 250000 AND 749999", True),
    ("SELECT COUNT(*) FROM gpuBench WHERE r_int_d BETWEEN 250000 AND 749999", True),
    ("SELECT COUNT(*) FROM gpuBench WHERE r_int_s > 1000000", True),
    ("SELECT COUNT(*) FROM gpuBench WHERE c_inv1 NOT IN (0, 7) AND g2 BETWEEN 10 AND 30", True),          # a 6-bit column behind complemented postings
    ("SELECT COUNT(*) FROM gpuBench WHERE c_inv2 = 1 AND r_int_d = 4711", True),                          # a one-dictId interval
    ("SELECT COUNT(*) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int_d BETWEEN 2000000 AND 3000000", False),   # an empty interval
    ("SELECT COUNT(*) FROM gpuBench WHERE r_int_d < 500000 AND m_d > 100", False),                        # two scans: the chain kernel
]


@pytest.mark.parametrize("sql,ours", FILTER_ONLY)
def test_filter_only_over_a_dictionary_encoded_column(pair, sql, ours):
    g, o = pair
    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
    gb, ob = g.execute(qc), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    ran = gb.stats.kernel.decode()
    if ours and gb.stats.num_total_docs > 1 and not os.environ.get("PG_NO_SCAN_PIPE") and not os.environ.get("PG_FORCE_INTERPRETER"):
        assert ran == "pg_dictrange_fo", (sql, ran)
    # the docId set of the same filter (FilterOperator -> DocIdSetOperator): the match words the kernel writes
    assert g.filter(sql).doc_ids().tolist() == o.filter(sql).doc_ids().tolist()
