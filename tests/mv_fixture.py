"""A small table with multi-value columns and a brute-force evaluator over its rows, in the style of
DictionaryBasedGroupKeyGeneratorTest.java:163-200 (build random rows, run the real operators, compare with a plain loop).

The reference tree holds no multi-value AVRO fixture (test_data-mv.avro is absent), so the oracle's multi-value paths are exercised
over random rows against this brute force (and by the reader / writer round trips in test_host_formats.py).  Reference-held NUMBERS for
multi-value group-by and the *MV functions do exist — MultiValueRawQueriesTest builds a formulaic table with hard-coded expectations —
and are asserted in tests/test_mv_reference_goldens.py (round 4; round 3 wrongly stated there were none)."""
import math

import numpy as np

from pinot_amd.segment import HostSegment, build_column, build_mv_column, build_raw_mv_column

WORDS = ["ant", "bee", "cat", "dog", "eel", "fox", "gnu", "hen", "ibis", "jay", "koi", "lynx"]


def make_rows(n, seed=7, empty_rows=True):
    rng = np.random.default_rng(seed)
    rows = []
    for d in range(n):
        k1 = int(rng.integers(0 if empty_rows else 1, 5))
        k2 = int(rng.integers(1, 4))
        rows.append({
            "s1": int(rng.integers(0, 7)),                                   # single-value dictionary INT
            "s2": WORDS[int(rng.integers(0, 5))],                            # single-value dictionary STRING
            "m": int(rng.integers(-1000, 1000)),                             # raw INT metric
            "mv1": [int(v) for v in rng.integers(0, 40, k1)],                # multi-value INT (duplicates within a doc happen), inverted index
            "mv2": [WORDS[int(v)] for v in rng.integers(0, len(WORDS), k2)],  # multi-value STRING, scan only
            "mv3": [int(v) * 1000003 for v in rng.integers(0, 9, int(rng.integers(1, 3)))],   # multi-value LONG
            "mvh": [int(v) for v in rng.integers(0, 30000, int(rng.integers(1, 4)))],          # multi-value INT, many distinct values
        })
    return rows


def build(rows, name="mvTable") -> HostSegment:
    seg = HostSegment(name, len(rows))
    seg.columns["s1"] = build_column("s1", [r["s1"] for r in rows], "INT", inverted=True)
    seg.columns["s2"] = build_column("s2", [r["s2"] for r in rows], "STRING")
    seg.columns["m"] = build_column("m", [r["m"] for r in rows], "INT", dictionary=False)
    seg.columns["mv1"] = build_mv_column("mv1", [r["mv1"] for r in rows], "INT", inverted=True)
    seg.columns["mv2"] = build_mv_column("mv2", [r["mv2"] for r in rows], "STRING")
    seg.columns["mv3"] = build_mv_column("mv3", [r["mv3"] for r in rows], "LONG")
    seg.columns["mvh"] = build_mv_column("mvh", [r["mvh"] for r in rows], "INT")
    return seg


def build_with_raw_twins(rows, name="mvTable", compression=(0, 3, 4, 2, 5)) -> HostSegment:
    """The same table plus raw (no-dictionary) twins of the numeric multi-value columns, FixedByteChunkMVForwardIndexReader's format
    under every ChunkCompressionType the path reads: r1 = mv1 (INT), r3 = mv3 (LONG), rh = mvh (INT), rf / rd = mv1 / 4 as FLOAT / DOUBLE.
    Every query over a raw twin must return what the dictionary column returns (MultiValueRawQueriesTest's own criterion)."""
    seg = build(rows, name)
    seg.columns["r1"] = build_raw_mv_column("r1", [r["mv1"] for r in rows], "INT", compression=compression[0])
    seg.columns["r3"] = build_raw_mv_column("r3", [r["mv3"] for r in rows], "LONG", compression=compression[1])
    seg.columns["rh"] = build_raw_mv_column("rh", [r["mvh"] for r in rows], "INT", compression=compression[2])
    seg.columns["rf"] = build_raw_mv_column("rf", [[v / 4.0 for v in r["mv1"]] for r in rows], "FLOAT", compression=compression[3])
    seg.columns["rd"] = build_raw_mv_column("rd", [[v / 4.0 for v in r["mv1"]] for r in rows], "DOUBLE", compression=compression[4])
    seg.columns["fd"] = build_mv_column("fd", [[v / 4.0 for v in r["mv1"]] for r in rows], "DOUBLE")   # dictionary twin of rf / rd
    seg.columns["rs"] = build_raw_mv_column("rs", [r["mv2"] for r in rows], "STRING")   # VarByteChunkMVForwardIndexReader: raw twin of mv2
    return seg


INT_DEFAULT = -(1 << 31)


def values_of(row, col):
    v = row[col]
    if isinstance(v, list):
        return v if v else [INT_DEFAULT]     # the segment creator stores the default null value for an empty entry
    return [v]


def brute_force(rows, where, group_by, aggs):
    """`where(row) -> bool`; `group_by`: column names (multi-value columns expand to every combination, repeats kept);
    `aggs`: (function, column) pairs.  Returns {key tuple: [intermediate results]} like ResultsBlock.rows()."""
    out = {}
    for r in rows:
        if not where(r):
            continue
        keys = [()]
        for g in group_by:
            keys = [k + (v,) for k in keys for v in values_of(r, g)]
        for k in keys:
            acc = out.setdefault(k, [None] * len(aggs))
            for i, (fn, col) in enumerate(aggs):
                vals = values_of(r, col) if col else []
                if fn == "COUNT":
                    acc[i] = (acc[i] or 0) + 1
                elif fn == "COUNTMV":
                    acc[i] = (acc[i] or 0) + len(vals)
                elif fn in ("SUM", "SUMMV"):
                    acc[i] = (acc[i] or 0.0) + float(sum(vals))
                elif fn in ("MIN", "MINMV"):
                    acc[i] = min([acc[i]] + [float(v) for v in vals]) if acc[i] is not None else float(min(vals))
                elif fn in ("MAX", "MAXMV"):
                    acc[i] = max([acc[i]] + [float(v) for v in vals]) if acc[i] is not None else float(max(vals))
                elif fn in ("AVG", "AVGMV"):
                    s, c = acc[i] or (0.0, 0)
                    acc[i] = (s + float(sum(vals)), c + len(vals))
                elif fn in ("MINMAXRANGE", "MINMAXRANGEMV"):
                    lo, hi = acc[i] or (math.inf, -math.inf)
                    acc[i] = (min(lo, float(min(vals))), max(hi, float(max(vals))))
                elif fn in ("DISTINCTCOUNT", "DISTINCTCOUNTMV"):
                    acc[i] = (acc[i] or frozenset()) | frozenset(vals)
                else:
                    raise ValueError(fn)
    return out
