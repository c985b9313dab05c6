"""Segments the reference's own query tests build, rebuilt in Pinot's byte layouts."""
import numpy as np

from pinot_amd.segment import build_segment

# BaseSingleValueQueriesTest.java:73-106 — schema + inverted index columns
SV_SCHEMA = {
    "column1": "INT", "column3": "INT", "column5": "STRING", "column6": "INT", "column7": "INT", "column9": "INT",
    "column11": "STRING", "column12": "STRING", "column17": "INT", "column18": "INT", "daysSinceEpoch": "INT",
}
SV_INVERTED = ["column6", "column7", "column11", "column17", "column18"]
SV_FILTER = (" WHERE column1 > 100000000"
             " AND column3 BETWEEN 20000000 AND 1000000000"
             " AND column5 = 'gFuH'"
             " AND (column6 < 500000000 OR column11 NOT IN ('t', 'P'))"
             " AND daysSinceEpoch = 126164076")


def sv_segment(sv_data, name="testTable_126164076_167572854"):
    data = {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in sv_data.items()}
    return build_segment(name, data, SV_SCHEMA, inverted_index_columns=SV_INVERTED)
