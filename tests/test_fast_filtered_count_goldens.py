"""FastFilteredCountTest (pinot-core/src/test/java/org/apache/pinot/queries/FastFilteredCountTest.java:104-113 the table, :147-308 the cases):
`select count(*)` over sorted / inverted / range-indexed columns and their AND / OR / NOT combinations with the counts the reference expects —
the shapes FastFilteredCountOperator answers from bitmap cardinalities (BitmapCollection).  The cases are extracted from the reference's
source by tools/gen_fast_filtered_count_golden.py into tests/golden/fast_filtered_count_cases.json (TEXT_MATCH / JSON_MATCH cases dropped)."""
import json
import os

import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "fast_filtered_count_cases.json")))["cases"]
NUM_RECORDS, BUCKET_SIZE = 1000, 8


def reference_table(range_index):
    i = np.arange(NUM_RECORDS)
    data = {"class": (i % BUCKET_SIZE).astype(np.int32), "sorted": i.astype(np.int32), "intRangeCol": (NUM_RECORDS - i).astype(np.int32)}
    # the test's table: sorted column, inverted indexes on class and sorted, a range index on intRangeCol
    return build_segment("testTable_0", data, {k: "INT" for k in data}, inverted_index_columns=["class", "sorted"],
                         no_dictionary_columns=["intRangeCol"] if range_index else [], range_index_columns=["intRangeCol"] if range_index else [])


def check(api, range_index):
    seg = NativeSegment(api, reference_table(range_index))
    assert len(CASES) >= 29
    for c in CASES:
        b = seg.execute(c["query"])
        assert b.aggregation_result() == [c["expected"]], c["query"]
        assert seg.filter(c["query"]).cardinality() == c["expected"], c["query"]
    seg.destroy()


@pytest.mark.parametrize("range_index", [False, True])
def test_oracle_reproduces_fast_filtered_count_test(oracle_api, range_index):
    check(oracle_api, range_index)


@pytest.mark.gpu
@pytest.mark.parametrize("range_index", [False, True])
def test_gpu_reproduces_fast_filtered_count_test(gpu_api, range_index):
    check(gpu_api, range_index)
