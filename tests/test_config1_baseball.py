"""BASELINE config 1: the Quickstart `baseballStats` table (97 889 rows), `SELECT SUM(runs) ... GROUP BY teamID`.
The reference pins only the row count (BasicAuthBatchIntegrationTest.java:176-177); the quickstart's sample queries
(pinot-tools/.../Quickstart.java:109-131: count, SUM(runs) grouped, `yearID = 2000`, `yearID >= 2000`) are checked against
numpy here (oracle) and against the oracle on the GPU.  Data: tests/golden/baseball_stats.npz (tests/golden/make_baseball_stats.py)."""
import os

import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "baseball_stats.npz")
QUERIES = [
    "SELECT COUNT(*) FROM baseballStats",
    "SELECT teamID, SUM(runs) FROM baseballStats GROUP BY teamID LIMIT 1000",
    "SELECT teamID, SUM(runs), COUNT(*), MAX(homeRuns) FROM baseballStats WHERE yearID = 2000 GROUP BY teamID LIMIT 1000",
    "SELECT league, teamID, SUM(hits), MIN(yearID), MAX(yearID) FROM baseballStats WHERE yearID >= 2000 GROUP BY league, teamID LIMIT 1000",
    "SELECT yearID, SUM(runs), AVG(hits), DISTINCTCOUNT(teamID) FROM baseballStats WHERE league IN ('AL', 'NL') AND runs > 0 GROUP BY yearID LIMIT 1000",
    "SELECT SUM(runs), SUM(hits), SUM(homeRuns) FROM baseballStats WHERE teamID = 'BOS' OR teamID = 'NYA'",
]


def baseball_segment():
    z = np.load(GOLDEN)
    data = {k: (z[k].astype(object) if z[k].dtype.kind == "U" else z[k]) for k in z.files}
    schema = {"teamID": "STRING", "league": "STRING", "yearID": "INT", "runs": "INT", "hits": "INT", "homeRuns": "INT"}
    # quickstart table config: metrics are raw-friendly but dictionary-encoded by default; teamID carries the inverted index
    host = build_segment("baseballStats_OFFLINE_0", data, schema, inverted_index_columns=["teamID", "league"])
    return host, {k: z[k] for k in z.files}


def test_config1_oracle_matches_numpy(oracle_api):
    host, d = baseball_segment()
    assert host.total_docs == 97889
    o = NativeSegment(oracle_api, host)
    assert o.execute(QUERIES[0]).aggregation_result() == [97889]
    rows = o.execute(QUERIES[1]).rows()
    teams, inv = np.unique(d["teamID"], return_inverse=True)
    sums = np.bincount(inv, weights=d["runs"].astype(np.float64))
    assert rows == {(str(t),): [float(s)] for t, s in zip(teams, sums)}
    m = d["yearID"] == 2000
    rows = o.execute(QUERIES[2]).rows()
    for t in np.unique(d["teamID"][m]):
        k = m & (d["teamID"] == t)
        assert rows[(str(t),)] == [float(d["runs"][k].sum()), int(k.sum()), float(d["homeRuns"][k].max())]
    assert len(rows) == len(np.unique(d["teamID"][m]))
    k = (d["teamID"] == "BOS") | (d["teamID"] == "NYA")
    assert o.execute(QUERIES[5]).aggregation_result() == [float(d[c][k].sum()) for c in ("runs", "hits", "homeRuns")]
    o.destroy()


@pytest.mark.gpu
def test_config1_gpu_matches_oracle(gpu_api, oracle_api):
    host, _ = baseball_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for q in QUERIES:
        gb, ob = g.execute(q), o.execute(q)
        assert gb.rows() == ob.rows(), q
        assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
        assert gb.stats.num_entries_scanned_post_filter == ob.stats.num_entries_scanned_post_filter
        assert gb.stats.stats_exact == 1
        assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, q
    g.destroy()
    o.destroy()
