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


# ---- the reference's star-tree fixture as a queryable segment ---------------------------------------------------------
def airline_star_segment():
    """tests/golden/startree_airline (a star-tree built by the reference) + a parent segment reconstructed from its base
    docs: every base doc repeated count__* times with ArrDelay = its max__ArrDelay (the aggregates of the reconstruction
    equal the original's).  The fixture holds no dictionaries, so the dimensions get identity INT dictionaries
    (value == dictId)."""
    import json
    import os
    from pinot_amd import formats, startree
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "startree_airline")
    meta = json.load(open(os.path.join(golden, "meta.json")))
    blob = np.fromfile(os.path.join(golden, "star_tree_index"), dtype=np.uint8)
    im = meta["index_map"]

    def entry(col, kind):
        off, size = im[f"0.{col}.{kind}.OFFSET"], im[f"0.{col}.{kind}.SIZE"]
        return blob[off:off + size].copy()
    n = meta["total_docs"]
    dims = meta["split_order"]
    names, nodes = formats.read_star_tree(entry("null", "STAR_TREE"))
    dim_ids = np.stack([formats.unpack_fixed_bit(entry(d, "FORWARD_INDEX"), meta["columns"][d]["bitsPerElement"], n)
                        for d in dims], axis=1)
    h = formats.parse_raw_fixed_byte_chunk_header(entry("count__*", "FORWARD_INDEX"))
    counts = np.frombuffer(bytes(entry("count__*", "FORWARD_INDEX")), dtype=">i8", count=n, offset=h["raw_data_start"]).astype(np.int64)
    maxes = np.frombuffer(bytes(entry("max__ArrDelay", "FORWARD_INDEX")), dtype=">f8", count=n, offset=h["raw_data_start"])
    root_kids = nodes[nodes[0][5]:nodes[0][6] + 1]
    n_base = int(max(k[3] for k in root_kids if k[1] != -1))
    rep = counts[:n_base]
    data = {d: np.repeat(dim_ids[:n_base, j], rep).astype(np.int32) for j, d in enumerate(dims)}
    data["ArrDelay"] = np.repeat(maxes[:n_base].astype(np.int64), rep).astype(np.int32)
    # shuffle the parent docs (a real segment is not sorted by the split order)
    perm = np.random.default_rng(11).permutation(len(data["ArrDelay"]))
    data = {k: v[perm] for k, v in data.items()}
    # ArrDelay as a raw column: with a dictionary, match-all MIN/MAX queries would go to NonScanBasedAggregationOperator
    seg = build_segment(meta["segment_name"], data, {d: "INT" for d in dims + ["ArrDelay"]}, no_dictionary_columns=["ArrDelay"])
    for d in dims:   # identity dictionaries of the fixture's cardinality, so that dictIds are the fixture's
        card = meta["columns"][d]["cardinality"]
        col = seg.columns[d]
        col.cardinality = card
        col.bits_per_value = formats.num_bits_per_value(card - 1)
        col.dictionary = formats.write_numeric_dictionary(np.arange(card, dtype=np.int32), "INT")
        col.dict_values = list(range(card))
        col.forward_index = formats.pack_fixed_bit(data[d], col.bits_per_value)
        col.fwd_encoding = 0
        col.is_sorted = False
    st = startree.HostStarTree(
        n, dims, [entry(d, "FORWARD_INDEX") for d in dims],
        [startree.StarTreePair("COUNT", "*", entry("count__*", "FORWARD_INDEX"), counts.tolist()),
         startree.StarTreePair("MAX", "ArrDelay", entry("max__ArrDelay", "FORWARD_INDEX"), maxes.tolist())],
        entry("null", "STAR_TREE"), meta["max_leaf_records"], dim_ids, n_base)
    seg.star_trees.append(st)
    return seg, meta


def synth_star_segment(num_docs=40_000, max_leaf_records=64, skip=("h3",), seed_index=3):
    """gpuBench docs (BASELINE config 5 columns + a raw metric) with a star-tree over (h1, h2, h3, h4) holding
    count__*, sum__m, min__m, max__m and distinctCountHLL__u — built by pinot_amd.startree (the builder the reference
    fixture pins)."""
    from pinot_amd import startree, synth
    seg = synth.generate_segment(num_docs, segment_index=seed_index, columns=["h1", "h2", "h3", "h4", "u", "m", "g2"],
                                 native=False)
    startree.add_star_tree(seg, ["h1", "h2", "h3", "h4"],
                           [("COUNT", "*"), ("SUM", "m"), ("MIN", "m"), ("MAX", "m"), ("DISTINCTCOUNTHLL", "u")],
                           max_leaf_records=max_leaf_records, skip_star_node_creation=skip)
    return seg


def synth_star_pairs_segment(num_docs=30_000, max_leaf_records=100):
    """A star-tree holding the two 16-byte BYTES pairs: avg__m (AvgPair: sum, count) and minMaxRange__m (MinMaxRangePair)."""
    from pinot_amd import startree, synth
    seg = synth.generate_segment(num_docs, segment_index=5, columns=["h1", "h2", "h4", "m", "g2"], native=False)
    startree.add_star_tree(seg, ["h1", "h2", "h4"], [("COUNT", "*"), ("AVG", "m"), ("MINMAXRANGE", "m"), ("SUM", "m")],
                           max_leaf_records=max_leaf_records)
    return seg


STAR_PAIR_QUERIES = [
    ("SELECT COUNT(*), AVG(m), MINMAXRANGE(m) FROM gpuBench", True),
    ("SELECT h1, AVG(m), SUM(m) FROM gpuBench WHERE h2 IN (1, 2, 5) GROUP BY h1", True),
    ("SELECT h4, h2, MINMAXRANGE(m), COUNT(*) FROM gpuBench WHERE h1 BETWEEN 3 AND 11 GROUP BY h4, h2", True),
    ("SELECT AVG(m) FROM gpuBench WHERE h1 = 2 AND h4 != 3", True),
    ("SELECT h2, AVG(m), MAX(m) FROM gpuBench GROUP BY h2", False),        # max__m is not in this tree
]


SYNTH_STAR_QUERIES = [
    ("SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY h1, h2, h3, h4 LIMIT 20000", True),   # config 5
    ("SELECT COUNT(*), SUM(m), MIN(m), MAX(m), DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h2 = 3", True),
    ("SELECT h1, COUNT(*), SUM(m), DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY h1", True),
    ("SELECT h4, SUM(m), MAX(m) FROM gpuBench WHERE h1 IN (1, 2, 3) AND h3 BETWEEN 2 AND 7 GROUP BY h4", True),
    ("SELECT h2, h4, COUNT(*), MIN(m) FROM gpuBench WHERE h1 != 5 AND h4 > 2 GROUP BY h2, h4", True),
    ("SELECT COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h3 = 4 AND h4 = 1", True),
    ("SELECT h3, COUNT(*) FROM gpuBench WHERE (h2 = 1 OR h2 > 7) AND NOT h4 IN (0, 7) GROUP BY h3", True),
    ("SELECT SUM(m) FROM gpuBench WHERE h1 = 99", True),            # no matching dictId... the regular filter is already empty
    ("SELECT h1, DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h4 NOT IN (1, 2) GROUP BY h1", True),
    ("SELECT g2, COUNT(*) FROM gpuBench GROUP BY g2", False),       # g2 is not a star-tree dimension
    ("SELECT COUNT(*), AVG(m) FROM gpuBench WHERE h1 = 2", False),  # avg__m is not in the tree
    ("SELECT h1, DISTINCTCOUNT(u) FROM gpuBench GROUP BY h1", False),
]
