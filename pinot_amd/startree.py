"""Star-tree index (StarTreeV2) on the host side: the bytes the executor registers with `pg_segment_add_star_tree`, and a
builder that produces them the way the reference's builder does (tests and the cfg-5 benchmark need star-trees; segment
*creation* is otherwise out of scope).

Builder = OnHeapSingleTreeBuilder / BaseSingleTreeBuilder (pinot-segment-local/.../startree/v2/builder/
BaseSingleTreeBuilder.java:306-462, OnHeapSingleTreeBuilder.java:60-170):
  1. sortAndAggregateSegmentRecords: sort the segment's docs by the dimensions in split order, merge equal dimension
     tuples with the ValueAggregators → the *base* star-tree docs;
  2. constructStarTree: per node, one child per distinct value of the next dimension (docs are sorted, so children are
     ranges); if the node has > 1 child and the dimension is not in skipStarNodeCreation, a star child whose docs are the
     node's docs with that dimension removed, re-sorted and re-aggregated (appended at the end of the doc space); recurse
     into every child with more than maxLeafRecords docs;
  3. createAggregatedDocs: every node gets an aggregated doc (a one-doc leaf is its own; a node with a star child shares
     the star child's; otherwise a new doc appended at the end);
  4. serializeTree: BFS, children sorted by dimension value (star = -1 first), little-endian (formats.write_star_tree).
Children live in a java.util.HashMap<Integer, TreeNode> in the reference and are *iterated in HashMap order* during 2. and
3., which decides the docId order of the appended docs; `_java_hashmap_order` reproduces that order (bucket = (h ^ h>>>16)
& (capacity-1), insertion order inside a bucket, capacity 16·2^k with size <= 0.75·capacity; treeified buckets — >= 8
colliding keys — are not modelled).  tests/test_startree.py rebuilds the reference's own fixture
(tests/golden/startree_airline) from its base docs and compares every byte.

Star is stored as dictId 0 in the dimension forward indexes (StarTreeV2Constants.STAR_IN_FORWARD_INDEX).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi, formats

STAR = -1                      # StarTreeNode.ALL
STAR_IN_FORWARD_INDEX = 0      # StarTreeV2Constants.java:39

# AggregationFunctionType#getName of the functions the path stores (AggregationFunctionColumnPair#toColumnName)
PAIR_FUNCTION_NAMES = {"COUNT": "count", "SUM": "sum", "MIN": "min", "MAX": "max", "DISTINCTCOUNTHLL": "distinctCountHLL",
                       "AVG": "avg", "MINMAXRANGE": "minMaxRange"}
PAIR_VALUE_TYPES = {"COUNT": "LONG", "SUM": "DOUBLE", "MIN": "DOUBLE", "MAX": "DOUBLE", "DISTINCTCOUNTHLL": "BYTES",
                    "AVG": "BYTES", "MINMAXRANGE": "BYTES"}   # AvgPair (sum, count) / MinMaxRangePair (min, max): 16 bytes
DEFAULT_LOG2M = 8              # CommonConstants.Helix.DEFAULT_HYPERLOGLOG_LOG2M


def pair_column_name(function: str, column: str) -> str:
    return f"{PAIR_FUNCTION_NAMES[function]}__{column}"


def parse_pair(name: str):
    fn, col = name.split("__", 1)
    for k, v in PAIR_FUNCTION_NAMES.items():
        if v.lower() == fn.lower():
            return k, col
    raise ValueError(f"unsupported function-column pair {name}")


@dataclass
class StarTreePair:
    function: str                      # COUNT / SUM / MIN / MAX / DISTINCTCOUNTHLL
    column: str                        # "*" for COUNT
    forward_index: np.ndarray          # raw forward index bytes (LONG / DOUBLE fixed-byte chunks, BYTES var-byte chunks)
    values: Optional[list] = None      # decoded aggregates per star-tree doc (tests only)

    @property
    def name(self) -> str:
        return pair_column_name(self.function, self.column)

    @property
    def data_type(self) -> str:
        return PAIR_VALUE_TYPES[self.function]


@dataclass
class HostStarTree:
    num_docs: int
    dimensions: List[str]
    dimension_forward_indexes: List[np.ndarray]
    pairs: List[StarTreePair]
    star_tree: np.ndarray
    max_leaf_records: int = 10000
    dim_dict_ids: Optional[np.ndarray] = None   # [num_docs, n_dims] decoded dictIds (tests only)
    n_base_docs: int = 0
    _keep: list = field(default_factory=list)

    def desc(self) -> "capi.PgStarTreeDesc":
        n_d, n_p = len(self.dimensions), len(self.pairs)
        names = (C.c_char_p * n_d)(*[d.encode() for d in self.dimensions])
        fwd = (capi.PgBuffer * n_d)(*[capi.np_buffer(b) for b in self.dimension_forward_indexes])
        pairs = (capi.PgStarTreePair * n_p)()
        cols = [p.column.encode() for p in self.pairs]
        for i, p in enumerate(self.pairs):
            pairs[i] = capi.PgStarTreePair(capi.AGG_FUNCTIONS[p.function], capi.DATA_TYPES[p.data_type], cols[i],
                                           capi.np_buffer(p.forward_index))
        d = capi.PgStarTreeDesc(self.num_docs, n_d, n_p, self.max_leaf_records, names, fwd, pairs,
                                capi.np_buffer(self.star_tree))
        self._keep = [names, fwd, pairs, cols, d]
        return d

    def nbytes(self) -> int:
        return (sum(b.nbytes for b in self.dimension_forward_indexes) + sum(p.forward_index.nbytes for p in self.pairs)
                + self.star_tree.nbytes)


# ----------------------------------------------------------------------------------------------------------------------
# HyperLogLog value aggregation (DistinctCountHLLValueAggregator.java:36-80): registers as uint8 arrays
# ----------------------------------------------------------------------------------------------------------------------
def murmur_hash_long(v: np.ndarray) -> np.ndarray:
    """stream-lib MurmurHash.hashLong on int64 values → uint32 hashes."""
    m = np.uint32(0x5BD1E995)
    v = np.asarray(v, dtype=np.int64).view(np.uint64)
    with np.errstate(over="ignore"):
        k = (v & np.uint64(0xFFFFFFFF)).astype(np.uint32) * m
        k ^= k >> np.uint32(24)
        h = k * m                                   # h = 0 ^ k*m
        k = (v >> np.uint64(32)).astype(np.uint32) * m
        k ^= k >> np.uint32(24)
        h = h * m
        h ^= k * m
        h ^= h >> np.uint32(13)
        h = h * m
        h ^= h >> np.uint32(15)
    return h


def hll_index_rank(hashes: np.ndarray, log2m: int):
    x = np.asarray(hashes, dtype=np.uint32)
    j = (x >> np.uint32(32 - log2m)).astype(np.int64)
    with np.errstate(over="ignore"):
        w = (x << np.uint32(log2m)) | np.uint32((1 << (log2m - 1)) + 1)
    # numberOfLeadingZeros(w) + 1; w != 0 always
    nlz = 31 - np.floor(np.log2(w.astype(np.float64))).astype(np.int64)
    return j, (nlz + 1).astype(np.uint8)


def hll_registers(values: np.ndarray, data_type: str, log2m: int = DEFAULT_LOG2M) -> np.ndarray:
    """HyperLogLog(log2m) after offer(v) for every value (Integer/Long → hashLong(value); Float → raw int bits; Double →
    raw long bits)."""
    if data_type in ("INT", "LONG"):
        as_long = np.asarray(values, dtype=np.int64)
    elif data_type == "FLOAT":
        as_long = np.asarray(values, dtype=np.float32).view(np.int32).astype(np.int64)
    elif data_type == "DOUBLE":
        as_long = np.asarray(values, dtype=np.float64).view(np.int64)
    else:
        raise ValueError("HLL over STRING values is not needed by the builder")
    j, r = hll_index_rank(murmur_hash_long(as_long), log2m)
    regs = np.zeros(1 << log2m, dtype=np.uint8)
    np.maximum.at(regs, j, r)
    return regs


class _Agg:
    """ValueAggregator: getInitialAggregatedValue / applyRawValue / applyAggregatedValue / cloneAggregatedValue."""

    def __init__(self, function: str, data_type: str = "INT", log2m: int = DEFAULT_LOG2M):
        self.function, self.data_type, self.log2m = function, data_type, log2m

    def initial(self, raw):
        f = self.function
        if f == "COUNT":
            return 1
        if f == "DISTINCTCOUNTHLL":
            return hll_registers(np.asarray([raw]), self.data_type, self.log2m)
        if f == "AVG":            # AvgValueAggregator.java:41-47: AvgPair(value, 1)
            return (float(raw), 1)
        if f == "MINMAXRANGE":    # MinMaxRangeValueAggregator: MinMaxRangePair(value, value)
            return (float(raw), float(raw))
        return float(raw)

    def apply_raw(self, agg, raw):
        f = self.function
        if f == "COUNT":
            return agg + 1
        if f == "SUM":
            return agg + float(raw)
        if f == "MIN":
            return min(agg, float(raw))
        if f == "MAX":
            return max(agg, float(raw))
        if f == "AVG":
            return (agg[0] + float(raw), agg[1] + 1)
        if f == "MINMAXRANGE":
            return (min(agg[0], float(raw)), max(agg[1], float(raw)))
        return np.maximum(agg, hll_registers(np.asarray([raw]), self.data_type, self.log2m))

    def apply_aggregated(self, a, b):
        f = self.function
        if f in ("COUNT", "SUM"):
            return a + b
        if f == "MIN":
            return min(a, b)
        if f == "MAX":
            return max(a, b)
        if f == "AVG":
            return (a[0] + b[0], a[1] + b[1])
        if f == "MINMAXRANGE":
            return (min(a[0], b[0]), max(a[1], b[1]))
        return np.maximum(a, b)

    def clone(self, a):
        return a.copy() if isinstance(a, np.ndarray) else a


def _java_hashmap_order(keys: Sequence[int]) -> List[int]:
    """Indices of `keys` (given in insertion order) in java.util.HashMap<Integer, ?> iteration order."""
    n = len(keys)
    cap = 16
    while n > (cap * 3) // 4:
        cap *= 2

    def bucket(k):
        h = k & 0xFFFFFFFF
        return (h ^ (h >> 16)) & (cap - 1)
    return sorted(range(n), key=lambda i: (bucket(keys[i]), i))


class _Node:
    __slots__ = ("dimension_id", "dimension_value", "start", "end", "aggregated", "children")

    def __init__(self):
        self.dimension_id = -1
        self.dimension_value = -1
        self.start = -1
        self.end = -1
        self.aggregated = -1
        self.children = None     # list of _Node in HashMap iteration order


class StarTreeBuilder:
    """Stage 2–4 of the reference builder over already sorted + aggregated base records."""

    def __init__(self, n_dims: int, aggs: List[_Agg], max_leaf_records: int, skip_star_dims: Sequence[int] = ()):
        self.n_dims = n_dims
        self.aggs = aggs
        self.max_leaf = max_leaf_records
        self.skip = set(skip_star_dims)
        self.dims: List[List[int]] = []
        self.metrics: List[list] = []
        self.n_nodes = 0

    # -- record store --------------------------------------------------------------------------------------------------
    def append(self, dims: List[int], metrics: list):
        self.dims.append(dims)
        self.metrics.append(metrics)

    @property
    def num_docs(self):
        return len(self.dims)

    def _merge_star_record(self, agg, doc):
        if agg is None:
            return [list(self.dims[doc]), [a.clone(v) for a, v in zip(self.aggs, self.metrics[doc])]]
        agg[1] = [a.apply_aggregated(x, y) for a, x, y in zip(self.aggs, agg[1], self.metrics[doc])]
        return agg

    def _merge_record(self, agg, rec):
        if agg is None:
            return [list(rec[0]), [a.clone(v) for a, v in zip(self.aggs, rec[1])]]
        agg[1] = [a.apply_aggregated(x, y) for a, x, y in zip(self.aggs, agg[1], rec[1])]
        return agg

    def _new_node(self):
        self.n_nodes += 1
        return _Node()

    # -- constructStarTree -------------------------------------------------------------------------------------------------
    def _construct(self, node: _Node, start: int, end: int):
        child_dim = node.dimension_id + 1
        if child_dim == self.n_dims:
            return
        children, keys = [], []
        node_start = start
        value = self.dims[start][child_dim]
        for i in range(start + 1, end):                       # constructNonStarNodes
            v = self.dims[i][child_dim]
            if v != value:
                c = self._new_node()
                c.dimension_id, c.dimension_value, c.start, c.end = child_dim, value, node_start, i
                children.append(c)
                keys.append(value)
                node_start, value = i, v
        c = self._new_node()
        c.dimension_id, c.dimension_value, c.start, c.end = child_dim, value, node_start, end
        children.append(c)
        keys.append(value)
        if child_dim not in self.skip and len(children) > 1:  # constructStarNode
            s = self._new_node()
            s.dimension_id, s.dimension_value, s.start = child_dim, STAR, self.num_docs
            docs = sorted(range(start, end), key=lambda d: self.dims[d][child_dim + 1:])   # stable, like Arrays.sort
            cur, nxt = docs[0], None
            nxt = self._merge_star_record(None, cur)
            nxt[0][child_dim] = STAR_IN_FORWARD_INDEX
            for d in docs[1:]:
                if self.dims[d][child_dim + 1:] != self.dims[cur][child_dim + 1:]:
                    self.append(nxt[0], nxt[1])
                    cur = d
                    nxt = self._merge_star_record(None, d)
                    nxt[0][child_dim] = STAR_IN_FORWARD_INDEX
                else:
                    nxt = self._merge_star_record(nxt, d)
            self.append(nxt[0], nxt[1])
            s.end = self.num_docs
            children.append(s)
            keys.append(STAR)
        node.children = [children[i] for i in _java_hashmap_order(keys)]
        for child in node.children:
            if child.end - child.start > self.max_leaf:
                self._construct(child, child.start, child.end)

    # -- createAggregatedDocs -----------------------------------------------------------------------------------------------
    def _aggregate(self, node: _Node):
        if node.children is None:
            if node.start == node.end - 1:
                node.aggregated = node.start
                return [self.dims[node.start], self.metrics[node.start]]
            agg = None
            for d in range(node.start, node.end):
                agg = self._merge_star_record(agg, d)
        else:
            star = [c for c in node.children if c.dimension_value == STAR]
            if star:
                agg = None
                for c in node.children:
                    r = self._aggregate(c)
                    if c.dimension_value == STAR:
                        agg = r
                        node.aggregated = c.aggregated
                return agg
            agg = None
            for c in node.children:
                agg = self._merge_record(agg, self._aggregate(c))
        for i in range(node.dimension_id + 1, self.n_dims):
            agg[0][i] = STAR_IN_FORWARD_INDEX
        node.aggregated = self.num_docs
        self.append(agg[0], agg[1])
        return agg

    # -- serializeTree ----------------------------------------------------------------------------------------------------------
    def _serialize(self, root: _Node) -> np.ndarray:
        rows = []
        queue = [root]
        qi = 0
        while qi < len(queue):
            node = queue[qi]
            if node.children is None:
                first = last = -1
            else:
                kids = sorted(node.children, key=lambda c: c.dimension_value)
                first = len(queue)                      # currentNodeId + queue.size() + 1 with the head already removed
                last = first + len(kids) - 1
                queue.extend(kids)
            rows.append([node.dimension_id, node.dimension_value, node.start, node.end, node.aggregated, first, last])
            qi += 1
        return np.asarray(rows, dtype=np.int32)

    def build(self):
        root = self._new_node()
        self._construct(root, 0, self.num_docs)
        self._aggregate(root)
        return self._serialize(root)


def _sort_and_aggregate(dim_ids: np.ndarray, metric_raw: List[Optional[np.ndarray]], aggs: List[_Agg]):
    """sortAndAggregateSegmentRecords: returns (base dims [n_base, D], per-metric aggregated value lists)."""
    n, n_dims = dim_ids.shape
    order = np.lexsort(tuple(dim_ids[:, j] for j in range(n_dims - 1, -1, -1)))   # stable
    sd = dim_ids[order]
    new_group = np.ones(n, dtype=bool)
    new_group[1:] = np.any(sd[1:] != sd[:-1], axis=1)
    starts = np.flatnonzero(new_group)
    ends = np.append(starts[1:], n)
    base_dims = sd[starts]
    out = []
    for a, raw in zip(aggs, metric_raw):
        if a.function == "COUNT":
            out.append([int(e - s) for s, e in zip(starts, ends)])
            continue
        rv = np.asarray(raw)[order]
        vals = []
        if a.function == "DISTINCTCOUNTHLL":
            j, r = hll_index_rank(murmur_hash_long(_as_long(rv, a.data_type)), a.log2m)
            m = 1 << a.log2m
            gid = np.repeat(np.arange(len(starts)), ends - starts)
            regs = np.zeros(len(starts) * m, dtype=np.uint8)
            np.maximum.at(regs, gid * m + j, r)
            vals = [regs[g * m:(g + 1) * m].copy() for g in range(len(starts))]
        else:
            for s, e in zip(starts, ends):
                acc = a.initial(rv[s])
                for k in range(s + 1, e):
                    acc = a.apply_raw(acc, rv[k])
                vals.append(acc)
        out.append(vals)
    return base_dims, out


def _as_long(values: np.ndarray, data_type: str) -> np.ndarray:
    if data_type in ("INT", "LONG"):
        return np.asarray(values, dtype=np.int64)
    if data_type == "FLOAT":
        return np.asarray(values, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.asarray(values, dtype=np.float64).view(np.int64)


def _write_pair(function: str, values: list, log2m: int) -> np.ndarray:
    if function == "COUNT":
        return formats.write_raw_fixed_byte_chunk(np.asarray(values, dtype=np.int64), "LONG")
    if function == "DISTINCTCOUNTHLL":
        blobs = [formats.serialize_hll(v, log2m) for v in values]
        return formats.write_raw_var_byte_chunk(blobs, longest_entry=8 + 4 * ((1 << log2m) // 6 + (0 if (1 << log2m) % 6 == 0 else 1)))
    if function == "AVG":           # AvgPair#toBytes: big-endian double sum, long count
        import struct
        return formats.write_raw_var_byte_chunk([struct.pack(">dq", v[0], v[1]) for v in values], longest_entry=16)
    if function == "MINMAXRANGE":   # MinMaxRangePair#toBytes: big-endian double min, double max
        import struct
        return formats.write_raw_var_byte_chunk([struct.pack(">dd", v[0], v[1]) for v in values], longest_entry=16)
    return formats.write_raw_fixed_byte_chunk(np.asarray(values, dtype=np.float64), "DOUBLE")


def build_from_base_records(dimensions: Sequence[str], cardinalities: Sequence[int], base_dims: np.ndarray,
                            pairs: Sequence[tuple], base_metrics: List[list], max_leaf_records: int = 10000,
                            skip_star_node_creation: Sequence[str] = (), value_types: Optional[Dict[str, str]] = None,
                            log2m: int = DEFAULT_LOG2M) -> HostStarTree:
    """`pairs`: [(function, column)]; `base_metrics[i]`: the aggregated value of pair i for every base record."""
    value_types = value_types or {}
    aggs = [_Agg(f, value_types.get(c, "INT"), log2m) for f, c in pairs]
    skip = [list(dimensions).index(d) for d in skip_star_node_creation]
    b = StarTreeBuilder(len(dimensions), aggs, max_leaf_records, skip)
    for i in range(base_dims.shape[0]):
        b.append([int(x) for x in base_dims[i]], [m[i] for m in base_metrics])
    n_base = b.num_docs
    nodes = b.build()
    dim_arr = np.asarray(b.dims, dtype=np.int32).reshape(b.num_docs, len(dimensions))
    fwd = [formats.pack_fixed_bit(dim_arr[:, j], formats.num_bits_per_value(card - 1))
           for j, card in enumerate(cardinalities)]
    host_pairs = []
    for i, (f, c) in enumerate(pairs):
        vals = [m[i] for m in b.metrics]
        host_pairs.append(StarTreePair(f, c, _write_pair(f, vals, log2m), vals))
    return HostStarTree(b.num_docs, list(dimensions), fwd, host_pairs, formats.write_star_tree(dimensions, nodes),
                        max_leaf_records, dim_arr, n_base)


def build_star_tree(dim_dict_ids: Dict[str, np.ndarray], cardinalities: Dict[str, int],
                    metric_values: Dict[str, np.ndarray], metric_types: Dict[str, str], dimensions: Sequence[str],
                    pairs: Sequence[tuple], max_leaf_records: int = 10000,
                    skip_star_node_creation: Sequence[str] = (), log2m: int = DEFAULT_LOG2M) -> HostStarTree:
    """Full builder over a segment: `dim_dict_ids[d]` = dictIds of dimension d per doc, `metric_values[c]` = the values
    (stored type `metric_types[c]`) of metric column c per doc."""
    dims = np.stack([np.asarray(dim_dict_ids[d], dtype=np.int32) for d in dimensions], axis=1)
    aggs = [_Agg(f, metric_types.get(c, "INT"), log2m) for f, c in pairs]
    raws = [None if f == "COUNT" else metric_values[c] for f, c in pairs]
    base_dims, base_metrics = _sort_and_aggregate(dims, raws, aggs)
    return build_from_base_records(dimensions, [cardinalities[d] for d in dimensions], base_dims, pairs, base_metrics,
                                   max_leaf_records, skip_star_node_creation, metric_types, log2m)


def add_star_tree(host_segment, dimensions: Sequence[str], pairs: Sequence[tuple], max_leaf_records: int = 10000,
                  skip_star_node_creation: Sequence[str] = (), decoded: Optional[Dict[str, np.ndarray]] = None,
                  log2m: int = DEFAULT_LOG2M) -> HostStarTree:
    """Builds a star-tree over a HostSegment and appends it to `host_segment.star_trees`.  `decoded[c]` may supply the
    per-doc values of a column; otherwise they are decoded from the segment's own forward index / dictionary."""
    from .segment import decode_column
    decoded = dict(decoded or {})
    dim_ids, cards = {}, {}
    for d in dimensions:
        col = host_segment.columns[d]
        assert col.has_dictionary, "star-tree dimensions are dictionary encoded"
        dim_ids[d] = decode_column(col, host_segment.total_docs, dict_ids=True)
        cards[d] = col.cardinality
    mvals, mtypes = {}, {}
    for f, c in pairs:
        if f == "COUNT":
            continue
        col = host_segment.columns[c]
        mtypes[c] = col.data_type
        mvals[c] = decoded[c] if c in decoded else decode_column(col, host_segment.total_docs)
    st = build_star_tree(dim_ids, cards, mvals, mtypes, dimensions, pairs, max_leaf_records, skip_star_node_creation, log2m)
    host_segment.star_trees.append(st)
    return st
