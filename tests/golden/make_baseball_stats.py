"""BASELINE config 1 (pinot-tools Quickstart `baseballStats`): the raw CSV is not in the reference tree, the same 97 889 rows are
the test resource of the parquet input format (ParquetRecordReaderTest).  Extracts the columns the quickstart queries group and
aggregate on into tests/golden/baseball_stats.npz.  Run in the build container (needs /root/reference and pyarrow).
Schema types follow pinot-tools/src/main/resources/examples/batch/baseballStats/baseballStats_offline_table_config / schema:
teamID, league STRING dimensions; yearID INT dimension; runs, hits, homeRuns INT metrics."""
import os

import numpy as np
import pyarrow.parquet as pq

SRC = "/root/reference/pinot-plugins/pinot-input-format/pinot-parquet/src/test/resources/baseballStats.snappy.parquet"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "baseball_stats.npz")

if __name__ == "__main__":
    t = pq.read_table(SRC).select(["teamID", "league", "yearID", "runs", "hits", "homeRuns"]).to_pandas()
    out = {"teamID": t["teamID"].to_numpy().astype("U3"), "league": t["league"].fillna("").to_numpy().astype("U2")}
    for c in ("yearID", "runs", "hits", "homeRuns"):
        out[c] = t[c].fillna("0").astype(np.int64).to_numpy().astype(np.int32)
    np.savez_compressed(DST, **out)
    print(len(t), {k: v.dtype for k, v in out.items()}, os.path.getsize(DST))
