"""Worker of tests/test_gpu_malformed_buffers.py (a process of its own: a crash is the finding, not the end of the suite).  Builds one small
valid segment with every index kind the boundary parses — fixed-bit forward indexes, dictionaries (fixed and variable length), Roaring
inverted indexes, raw fixed-byte chunks (PASS_THROUGH / SNAPPY / LZ4 / LZ4_LENGTH_PREFIXED / ZSTANDARD / GZIP), raw var-byte chunks,
multi-value forward indexes (fixed-bit, MV_ENTRY_DICT, raw), a sorted column, a RangeBitmap, a null vector, a star-tree — then registers
MUTATED copies (truncated at many lengths, bytes flipped in the header and at random) on fresh segments.  Every call must return PG_OK,
PG_ERR_INVALID_ARGUMENT or PG_ERR_UNSUPPORTED; after a PG_OK a query over the column must answer or refuse, never crash or hang.  Prints
one JSON line with the tallies."""
import copy
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pinot_amd import capi, formats, startree   # noqa: E402
from pinot_amd.executor import NativeSegment   # noqa: E402
from pinot_amd.segment import HostSegment, add_range_index, build_segment   # noqa: E402

OK_STATUSES = {capi.PG_OK, capi.PG_ERR_INVALID_ARGUMENT, capi.PG_ERR_UNSUPPORTED}
N = 6_000


def base_segment():
    rng = np.random.default_rng(17)
    data = {
        "d7": rng.integers(0, 100, N).astype(np.int32),                       # 7-bit dictIds, 100 values: room above the cardinality
        "d20": rng.integers(0, 700_000, N).astype(np.int32),
        "ds": np.array([f"v{v:04d}" for v in rng.integers(0, 300, N)], dtype=object),
        "srt": np.sort(rng.integers(0, 50, N)).astype(np.int32),
        "ri": rng.integers(-1000, 1000, N).astype(np.int32),
        "rl": rng.integers(-10**12, 10**12, N).astype(np.int64),
        "rd": rng.normal(size=N).astype(np.float64),
        "rs": np.array([f"s{v}" * (1 + v % 3) for v in rng.integers(0, 500, N)], dtype=object),
    }
    schema = {"d7": "INT", "d20": "INT", "ds": "STRING", "srt": "INT", "ri": "INT", "rl": "LONG", "rd": "DOUBLE", "rs": "STRING"}
    host = build_segment("fuzz", data, schema, no_dictionary_columns=["ri", "rl", "rd", "rs"], inverted_index_columns=["d7", "ds", "srt"])
    return host, data


def mutations(buf: np.ndarray, rng, n_random=24):
    """Truncations and byte flips of one buffer (bytes objects would alias: every mutation is a fresh array)."""
    n = len(buf)
    cuts = sorted({0, 1, 3, 4, 7, 8, 12, 16, 20, 27, 28, 31, 32, 40, 64, n // 4, n // 2, n - 9, n - 4, n - 1} & set(range(0, n)))
    for c in cuts:
        yield f"cut@{c}", buf[:c].copy()
    head = min(n, 96)
    for i in range(0, head, 1 if head <= 48 else 2):
        for flip in (0xFF, 0x80, 0x01):
            m = buf.copy()
            m[i] ^= flip
            yield f"flip@{i}^{flip:02x}", m
    for k in range(n_random):
        m = buf.copy()
        for _ in range(1 + k % 4):
            m[int(rng.integers(0, n))] = int(rng.integers(0, 256))
        yield f"rand{k}", m


def main():
    import torch  # noqa: F401
    api = capi.gpu_api()
    api.call("init", 0)
    host, data = base_segment()
    rng = np.random.default_rng(3)
    tally = {"registrations": 0, "ok": 0, "invalid": 0, "unsupported": 0, "queries": 0, "query_refused": 0}
    budget = int(os.environ.get("FUZZ_MAX_MUTATIONS", "100000"))

    last = {"what": ""}

    def note(what):   # FUZZ_TRACE=1: the case in flight goes to stderr (a crash then names it)
        last["what"] = what
        if os.environ.get("FUZZ_TRACE"):
            print(what, file=sys.stderr, flush=True)

    def classify(status):
        assert status in OK_STATUSES, f"status {status}: {api.last_error()}"
        tally["registrations"] += 1
        tally["ok" if status == capi.PG_OK else ("invalid" if status == capi.PG_ERR_INVALID_ARGUMENT else "unsupported")] += 1

    def try_column(col, queries):
        seg = NativeSegment(api, HostSegment("fuzz_one", N))
        d = col.desc()
        status = api.f("segment_add_column")(seg.handle, C.byref(d))
        classify(status)
        if status == capi.PG_OK:
            seg.host.columns[col.name] = col
            for q in queries:
                try:
                    seg.execute(q)
                    tally["queries"] += 1
                except capi.NativeError as e:
                    assert e.status in OK_STATUSES, (q, e)
                    tally["query_refused"] += 1
                except Exception:   # noqa: BLE001  (decoding a garbage result on the Python side is not the library's concern)
                    tally["queries"] += 1
        seg.destroy()

    for name, col in host.columns.items():
        if col.has_dictionary:
            qs = [f"SELECT {name}, COUNT(*) FROM t GROUP BY {name} LIMIT 100000", f"SELECT COUNT(*) FROM t WHERE {name} >= '0'" if col.data_type == "STRING" else f"SELECT COUNT(*) FROM t WHERE {name} > 5"]
        elif col.data_type in ("STRING", "BYTES"):
            qs = [f"SELECT {name}, COUNT(*) FROM t GROUP BY {name} LIMIT 100000"]
        else:
            qs = [f"SELECT COUNT(*), SUM({name}), MAX({name}) FROM t WHERE {name} > 0"]
        for attr in ("forward_index", "dictionary", "inverted_index"):
            buf = getattr(col, attr)
            if buf is None or len(buf) == 0:
                continue
            for what, m in mutations(np.asarray(buf, dtype=np.uint8), rng):
                if tally["registrations"] >= budget:
                    break
                c2 = copy.copy(col)
                setattr(c2, attr, m)
                try_column(c2, qs)
        # metadata that disagrees with the buffers
        for field, values in (("cardinality", (0, 1, col.cardinality + 1, 2**31 - 1)), ("bits_per_value", (0, 1, 33, 64)), ("dict_bytes_per_value", (0, 1, 3, 1 << 20))):
            for v in values:
                c2 = copy.copy(col)
                setattr(c2, field, v)
                try_column(c2, qs[:1])
    # compressed raw chunks: every codec, mutated payloads reach the device / host decoders
    vals = data["ri"]
    for comp_name, comp in (("SNAPPY", 1), ("LZ4", 3), ("LZ4_LENGTH_PREFIXED", 4), ("ZSTANDARD", 2), ("GZIP", 5)):
        try:
            blob = formats.write_raw_fixed_byte_chunk(vals, "INT", version=2, compression=comp)
        except Exception:   # noqa: BLE001  (a codec this host cannot write)
            continue
        base = copy.copy(host.columns["ri"])
        for what, m in mutations(np.asarray(blob, dtype=np.uint8), rng, n_random=60):
            c2 = copy.copy(base)
            c2.forward_index = m
            try_column(c2, ["SELECT COUNT(*), SUM(ri) FROM t WHERE ri > 0"])
    # multi-value columns (round 5): FixedBitMVForwardIndexReader's header + row-start bitmap + entries, the raw chunked
    # multi-value formats, their inverted indexes — mutated buffers, then queries whose kernels walk the entries
    from tests import mv_fixture as mvf
    mv_host = mvf.build_with_raw_twins(mvf.make_rows(1500, seed=5))
    mv_n = mv_host.total_docs

    def try_mv_column(col, queries):
        seg = NativeSegment(api, HostSegment("fuzz_mv", mv_n))
        d = col.desc()
        status = api.f("segment_add_column")(seg.handle, C.byref(d))
        classify(status)
        if status == capi.PG_OK:
            seg.host.columns[col.name] = col
            for q in queries:
                try:
                    seg.execute(q)
                    tally["queries"] += 1
                except capi.NativeError as e:
                    assert e.status in OK_STATUSES, (q, e)
                    tally["query_refused"] += 1
                except Exception:   # noqa: BLE001
                    tally["queries"] += 1
        seg.destroy()
    for name in ("mv1", "mv2", "mv3", "r1", "r3", "rd", "rs"):
        col = mv_host.columns[name]
        numeric = col.data_type != "STRING"
        qs = [f"SELECT COUNT(*) FROM t WHERE {name} = " + ("7" if numeric else "'cat'"), f"SELECT COUNTMV({name}), COUNT(*) FROM t"]
        if col.has_dictionary:
            qs.append(f"SELECT {name}, COUNT(*) FROM t GROUP BY {name} LIMIT 100000")
        for attr in ("forward_index", "dictionary", "inverted_index"):
            buf = getattr(col, attr)
            if buf is None or len(buf) == 0:
                continue
            for what, m in mutations(np.asarray(buf, dtype=np.uint8), rng, n_random=16):
                c2 = copy.copy(col)
                setattr(c2, attr, m)
                note(f"mv {name}.{attr} {what}")
                try_mv_column(c2, qs)
        for field, values in (("cardinality", (0, 1, col.cardinality + 1)), ("bits_per_value", (0, 33)), ("total_number_of_entries", (0, 1, 2**31 - 1))):
            if not hasattr(col, field):
                continue
            for v in values:
                c2 = copy.copy(col)
                setattr(c2, field, v)
                note(f"mv {name}.{field} = {v}")
                try_mv_column(c2, qs[:2])
    # star-tree buffers through pg_segment_add_star_tree: the serialized tree (StarTreeV2 node records), the dimensions' forward indexes, the
    # function-column pairs' raw forward indexes — then the query the tree answers
    from pinot_amd import startree, synth
    st_parent = synth.generate_segment(20_000, segment_index=1, columns=list(synth.CFG5_COLUMNS), native=False)
    startree.add_star_tree(st_parent, ["h1", "h2", "h3", "h4"], [("COUNT", "*"), ("DISTINCTCOUNTHLL", "u")], max_leaf_records=100)
    tree = st_parent.star_trees[0]
    st_parent.star_trees = []
    st_queries = [synth.QUERY_CFG5, "SELECT h1, COUNT(*) FROM t WHERE h2 = 3 GROUP BY h1"]

    def try_star_tree(t2):
        seg = NativeSegment(api, st_parent)
        d = t2.desc()
        status = api.f("segment_add_star_tree")(seg.handle, C.byref(d))
        classify(status)
        if status == capi.PG_OK:
            for q in st_queries:
                try:
                    seg.execute(q)
                    tally["queries"] += 1
                except capi.NativeError as e:
                    assert e.status in OK_STATUSES, (q, e)
                    tally["query_refused"] += 1
                except Exception:   # noqa: BLE001
                    tally["queries"] += 1
        seg.destroy()
    for what, m in mutations(np.asarray(tree.star_tree, dtype=np.uint8), rng, n_random=60):
        t2 = copy.copy(tree)
        t2.star_tree = m
        note(f"star-tree nodes {what}")
        try_star_tree(t2)
    for di in range(len(tree.dimension_forward_indexes)):
        for what, m in mutations(np.asarray(tree.dimension_forward_indexes[di], dtype=np.uint8), rng, n_random=8):
            t2 = copy.copy(tree)
            t2.dimension_forward_indexes = list(tree.dimension_forward_indexes)
            t2.dimension_forward_indexes[di] = m
            note(f"star-tree dim {di} {what}")
            try_star_tree(t2)
    for pi in range(len(tree.pairs)):
        for what, m in mutations(np.asarray(tree.pairs[pi].forward_index, dtype=np.uint8), rng, n_random=12):
            t2 = copy.copy(tree)
            t2.pairs = [copy.copy(pp) for pp in tree.pairs]
            t2.pairs[pi].forward_index = m
            note(f"star-tree pair {pi} {what}")
            try_star_tree(t2)
    for field, values in (("num_docs", (0, 1, tree.num_docs + 1, 2**31 - 1)), ("max_leaf_records", (0, 1))):
        for v in values:
            t2 = copy.copy(tree)
            setattr(t2, field, v)
            note(f"star-tree {field} = {v}")
            try_star_tree(t2)
    # bitmaps and the range index through their setters
    seg = NativeSegment(api, host)
    nulls = np.frombuffer(formats.serialize_roaring(np.arange(0, N, 7, dtype=np.int64)), dtype=np.uint8)
    rcol = add_range_index(copy.copy(host.columns["ri"]), data["ri"])
    for setter, blob, args in (("segment_set_null_vector", nulls, (b"ri",)), ("segment_set_queryable_doc_ids", nulls, ()),
                               ("segment_set_range_index", np.asarray(rcol.range_index, dtype=np.uint8), (b"ri",))):
        for what, m in mutations(blob, rng, n_random=40):
            status = api.f(setter)(seg.handle, *args, m.ctypes.data, m.nbytes)
            classify(status)
            if status == capi.PG_OK:
                try:
                    seg.execute("SELECT COUNT(*), SUM(ri) FROM t WHERE ri > 10")
                    tally["queries"] += 1
                except capi.NativeError as e:
                    assert e.status in OK_STATUSES, e
                    tally["query_refused"] += 1
    seg.destroy()
    print(json.dumps(tally))


if __name__ == "__main__":
    main()
