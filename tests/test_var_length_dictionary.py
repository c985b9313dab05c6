"""Variable-length STRING dictionaries (useVarLengthDictionary: VarLengthValueWriter.java:78-130, VarLengthValueReader.java; recognised by the
".vl;" magic like BaseImmutableDictionary.java:58-66): a column registered with one answers exactly like the same column with the
fixed-width padded dictionary — single-value and multi-value, filters (EQ / IN / NOT IN / RANGE), group keys, DISTINCTCOUNT."""
import copy

import numpy as np
import pytest

from pinot_amd import formats
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_mv_column, build_segment

WORDS = ["a", "bb", "kiwi", "a-much-longer-value-than-the-others", "zebra", "", "näïve", "bb2"]

QUERIES = [
    "SELECT COUNT(*), SUM(m) FROM vl WHERE s = 'kiwi'",
    "SELECT COUNT(*) FROM vl WHERE s IN ('a', 'zebra', 'nope', '')",
    "SELECT COUNT(*), MAX(m) FROM vl WHERE s NOT IN ('bb', 'bb2')",
    "SELECT COUNT(*) FROM vl WHERE s BETWEEN 'b' AND 'l'",
    "SELECT s, COUNT(*), SUM(m) FROM vl GROUP BY s LIMIT 100",
    "SELECT g, DISTINCTCOUNT(s), COUNT(*) FROM vl WHERE s != 'a' GROUP BY g LIMIT 100",
    "SELECT ms, COUNT(*) FROM vl WHERE ms = 'zebra' OR s = 'a' GROUP BY ms LIMIT 100",
    "SELECT s, ms, COUNT(*), MIN(m) FROM vl WHERE ms NOT IN ('kiwi') GROUP BY s, ms LIMIT 1000",
]


def segments(n=20_011, seed=3):
    rng = np.random.default_rng(seed)
    data = {"s": np.array(WORDS, dtype=object)[rng.integers(0, len(WORDS), n)], "g": rng.integers(0, 5, n).astype(np.int32),
            "m": rng.integers(-100, 100, n).astype(np.int32)}
    fixed = build_segment("vl", data, {"s": "STRING", "g": "INT", "m": "INT"}, inverted_index_columns=["s"], no_dictionary_columns=["m"])
    fixed.columns["ms"] = build_mv_column("ms", [[WORDS[int(v)] for v in rng.integers(0, len(WORDS), int(rng.integers(1, 4)))] for _ in range(n)], "STRING")
    var = copy.copy(fixed)
    var.columns = dict(fixed.columns)
    for c in ("s", "ms"):
        col = copy.copy(fixed.columns[c])
        col.dictionary = formats.write_var_length_string_dictionary(col.dict_values)
        col.dict_bytes_per_value = 0        # ColumnMetadata's lengthOfEachEntry means nothing for a variable-length dictionary
        var.columns[c] = col
    return fixed, var


def check(api_a, api_b=None):
    fixed, var = segments()
    a, b = NativeSegment(api_a, fixed), NativeSegment(api_b or api_a, var)
    for q in QUERIES:
        ra, rb = a.execute(q), b.execute(q)
        assert ra.rows() == rb.rows(), q
        assert ra.stats.num_docs_scanned == rb.stats.num_docs_scanned and ra.stats.num_entries_scanned_in_filter == rb.stats.num_entries_scanned_in_filter
    a.destroy()
    b.destroy()


def test_oracle_reads_variable_length_dictionaries(oracle_api):
    check(oracle_api)


@pytest.mark.gpu
def test_gpu_reads_variable_length_dictionaries(gpu_api, oracle_api):
    check(gpu_api)               # variable == fixed on the device
    check(oracle_api, gpu_api)   # and the device's variable-length column == the oracle's fixed one
