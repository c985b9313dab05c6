"""Segments written by Pinot itself: the reference's v1-layout fixtures paddingNull / paddingPercent / paddingOld
(pinot-core/src/test/resources/data/*.tar.gz, 5 docs, one file per index; copied by tests/golden/make_padding_segments.py).
They pin the byte formats against reference-written bytes: big-endian sorted dictionaries (INT / LONG / FLOAT / padded STRING),
the MSB-first fixed-bit forward index with bitsPerElement = PinotDataBitSet.getNumBitsPerValue(cardinality - 1), and the
metadata the loader reads.  Our writers must reproduce those bytes exactly; the oracle and the HIP path must answer queries on
the loaded segment."""
import os
import tarfile

import numpy as np
import pytest

from pinot_amd import formats, segment_dir
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import decode_column

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["paddingNull", "paddingPercent", "paddingOld"]


def load(name, tmp_path):
    tarfile.open(os.path.join(GOLDEN, name + ".tar.gz")).extractall(tmp_path)
    d = os.path.join(tmp_path, name)
    return segment_dir.load_segment_v1_dir(d), segment_dir.read_properties(os.path.join(d, "metadata.properties")), d


@pytest.mark.parametrize("name", NAMES)
def test_reference_written_bytes_round_trip_through_our_writers(name, tmp_path):
    seg, props, d = load(name, str(tmp_path))
    assert seg.total_docs == 5 and not seg.skipped and sorted(seg.columns) == ["age", "name", "outgoingName1", "percent"]
    pad = "\0" if name == "paddingNull" else "%"
    for cname, col in seg.columns.items():
        assert col.bits_per_value == formats.num_bits_per_value(col.cardinality - 1)          # PinotDataBitSet.getNumBitsPerValue
        ids = decode_column(col, 5, dict_ids=True)
        assert ids.min() >= 0 and ids.max() < col.cardinality and len(set(ids.tolist())) == col.cardinality
        np.testing.assert_array_equal(formats.pack_fixed_bit(ids, col.bits_per_value), col.forward_index)   # FixedBitSVForwardIndexWriter bytes
        if col.data_type == "STRING":
            padded = b"".join(v.encode().ljust(col.dict_bytes_per_value, pad.encode()) for v in col.dict_values)
            assert padded == bytes(col.dictionary)
            assert sorted(padded[i * 9:(i + 1) * 9] for i in range(col.cardinality)) == [padded[i * 9:(i + 1) * 9] for i in range(col.cardinality)]
        else:
            assert col.dict_values == sorted(col.dict_values)
            np.testing.assert_array_equal(formats.write_numeric_dictionary(np.array(col.dict_values), col.data_type), col.dictionary)
    t = decode_column(seg.columns["outgoingName1"], 5)
    assert (int(t.min()), int(t.max())) == (int(props["segment.start.time"][0]), int(props["segment.end.time"][0])) == (246, 902)


def check_queries(api, tmp_path):
    seg, _, _ = load("paddingNull", str(tmp_path))
    s = NativeSegment(api, seg)
    age = decode_column(seg.columns["age"], 5)
    name = decode_column(seg.columns["name"], 5)
    t = decode_column(seg.columns["outgoingName1"], 5)
    assert s.execute("SELECT COUNT(*), MIN(outgoingName1), MAX(outgoingName1), SUM(age) FROM myTable WHERE age > 0").aggregation_result() == [
        5, 246.0, 902.0, float(age.sum())]
    rows = s.execute("SELECT name, COUNT(*), MAX(age) FROM myTable GROUP BY name LIMIT 10").rows()
    assert rows == {(n,): [int((name == n).sum()), float(age[name == n].max())] for n in ("lynda", "lynda 2.0")}
    assert s.execute("SELECT COUNT(*) FROM myTable WHERE name = 'lynda' AND outgoingName1 BETWEEN 300 AND 1000").aggregation_result() == [
        int(((name == "lynda") & (t >= 300) & (t <= 1000)).sum())]
    pct = decode_column(seg.columns["percent"], 5)
    want = float(pct[pct < 800].astype(np.float64).sum())
    got = s.execute("SELECT SUM(percent) FROM myTable WHERE percent < 800").aggregation_result()[0]
    assert abs(got - want) <= np.spacing(want)      # floating SUM: within 1 ulp (BASELINE.json), the order of the adds is free
    s.destroy()


def test_oracle_queries_a_reference_written_segment(oracle_api, tmp_path):
    check_queries(oracle_api, tmp_path)


@pytest.mark.gpu
def test_gpu_queries_a_reference_written_segment(gpu_api, tmp_path):
    check_queries(gpu_api, tmp_path)
