"""Host-side mirror of the reference's operator / plan-maker interface for the hot path, on top of the C ABI.

Reference classes mirrored (same names, argument meaning and error behaviour; all under pinot-core/src/main/java/org/
apache/pinot/core/):
  plan/maker/PlanMaker.java:37-67, InstancePlanMakerImplV2.java:275-294   → GpuInstancePlanMaker.make_segment_plan_node
  operator/query/GroupByOperator.java:100-140, AggregationOperator.java   → GpuGroupByOperator / GpuAggregationOperator
  operator/blocks/results/GroupByResultsBlock.java, AggregationResultsBlock → GroupByResultsBlock / AggregationResultsBlock
  operator/ExecutionStatistics.java                                        → ExecutionStatistics
  operator/combine/GroupByCombineOperator.java:102-165,191-222 + data/table/IndexedTable.java:90-120
                                                                           → GroupByCombineOperator (host merge)
The Java plug-in is the production host; this module is what the tests and bench.py drive.
`NativeSegment` works over any library exporting the pinot_gpu.h entry points; the product code only ever passes
capi.gpu_api() — the oracle binding lives in tests/.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .query import CQuery, QueryContext, parse_sql
from .segment import HostSegment


@dataclass
class ExecutionStatistics:
    num_docs_scanned: int
    num_entries_scanned_in_filter: int
    num_entries_scanned_post_filter: int
    num_total_docs: int


class NativeSegment:
    """IndexSegment handle: the host segment's buffers registered with (and, for the GPU library, pinned in HBM by) a
    native library.  `destroy()` mirrors IndexSegment#destroy."""

    def __init__(self, api: capi.NativeApi, host: HostSegment, device: Optional[int] = None):
        self.api = api
        self.host = host
        self.handle = C.c_void_p()
        self.device = device
        if device is None:       # the default device (pg_init)
            api.call("segment_create", host.name.encode(), host.total_docs, C.byref(self.handle))
        else:                    # segment -> GPU map of a multi-GPU server process
            api.call("segment_create_on_device", host.name.encode(), host.total_docs, int(device), C.byref(self.handle))
        self._descs = []
        for col in host.columns.values():
            self._register(col)
        for st in getattr(host, "star_trees", []):
            self.add_star_tree(st)

    def _register(self, col):
        d = col.desc()
        self._descs.append(d)
        self.api.call("segment_add_column", self.handle, C.byref(d))
        nv = getattr(col, "null_vector", None)
        if nv is not None:
            self.api.call("segment_set_null_vector", self.handle, col.name.encode(), nv.ctypes.data, nv.nbytes)
        ri = getattr(col, "range_index", None)
        if ri is not None:
            self.api.call("segment_set_range_index", self.handle, col.name.encode(), ri.ctypes.data, ri.nbytes)

    def set_queryable_doc_ids(self, doc_ids):
        """SegmentContext#getQueryableDocIdsSnapshot (upsert validDocIds): ascending docIds, or None to clear."""
        if doc_ids is None:
            self.api.call("segment_set_queryable_doc_ids", self.handle, None, 0)
            return
        from . import formats
        blob = np.frombuffer(formats.serialize_roaring(np.asarray(doc_ids, dtype=np.int64)), dtype=np.uint8)
        self.api.call("segment_set_queryable_doc_ids", self.handle, blob.ctypes.data, blob.nbytes)

    def add_column(self, col, keep_host_buffers: bool = True):
        """Registers one more column (streaming upload of big segments: the GPU library copies the bytes into HBM during
        the call, so the caller may drop the host buffers afterwards with keep_host_buffers=False)."""
        self._register(col)
        self.host.columns[col.name] = col
        if not keep_host_buffers:
            col.forward_index = None
            col.inverted_index = None
            self._descs.pop()

    def add_star_tree(self, star_tree):
        """StarTreeLoaderUtils#loadStarTreeV2: the dimensions must already be registered columns of this segment."""
        d = star_tree.desc()
        self.api.call("segment_add_star_tree", self.handle, C.byref(d))

    @property
    def total_docs(self) -> int:
        return self.host.total_docs

    def device_bytes(self) -> int:
        out = C.c_uint64()
        self.api.call("segment_device_bytes", self.handle, C.byref(out))
        return out.value

    def destroy(self):
        if self.handle:
            self.api.call("segment_destroy", self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- FilterOperator / DocIdSetOperator ------------------------------------------------------------------------
    def filter(self, q, null_handling: bool = False) -> "DocIdSet":
        qc = parse_sql(q) if isinstance(q, str) else q
        cq = CQuery(qc)
        h = C.c_void_p()
        if null_handling or (qc.flags & capi.QUERY_FLAG_NULL_HANDLING):   # enableNullHandling=true: three-valued filter
            self.api.call("filter_exec_flags", self.handle, cq.filter_ptr(), capi.QUERY_FLAG_NULL_HANDLING, C.byref(h))
        else:
            self.api.call("filter_exec", self.handle, cq.filter_ptr(), C.byref(h))
        return DocIdSet(self.api, h)

    # -- GroupByOperator / AggregationOperator ---------------------------------------------------------------------
    def execute(self, q, profile: bool = False) -> "ResultsBlock":
        qc = parse_sql(q) if isinstance(q, str) else q
        if profile:
            qc.flags |= capi.QUERY_FLAG_PROFILE
        cached = getattr(qc, "_cquery", None)     # the C structs of a QueryContext are reusable across executions
        key = (qc.flags, qc.num_groups_limit, qc.limit, qc.min_segment_group_trim_size)
        if cached is None or cached[0] != key:
            cached = (key, CQuery(qc))
            qc._cquery = cached
        cq = cached[1]
        h = C.c_void_p()
        self.api.call("query_exec", self.handle, cq.ptr(), C.byref(h))
        try:
            return ResultsBlock.from_native(self.api, h, qc, self.host)
        finally:
            self.api.call("result_free", h)

    def execute_native(self, q, keep_device_table: bool = True, cancel: "Optional[CancelToken]" = None) -> "NativeResult":
        """Runs the query and keeps the native result handle (and, with `keep_device_table`, its dense accumulator table in HBM)
        so that it can be merged in the library: NativeResult.merge / .all_reduce, then .block()."""
        qc = parse_sql(q) if isinstance(q, str) else q
        flags = qc.flags | (capi.QUERY_FLAG_KEEP_DEVICE_TABLE if keep_device_table else 0)
        cache = getattr(qc, "_cquery_native", None)
        key = (flags, qc.num_groups_limit, qc.limit, qc.min_segment_group_trim_size)
        if cache is None or cache[0] != key:
            saved = qc.flags
            qc.flags = flags
            cache = (key, CQuery(qc))
            qc.flags = saved
            qc._cquery_native = cache
        h = C.c_void_p()
        self.api.call("query_exec_cancellable", self.handle, cache[1].ptr(), cancel.handle if cancel is not None else None, C.byref(h))
        return NativeResult(self.api, h, qc, self.host)


class CancelToken:
    """pg_cancel_t: the interrupting thread calls request(); the executing thread's query returns PG_ERR_CANCELLED
    (EarlyTerminationException, BaseOperator.java:44-46)."""

    def __init__(self, api: capi.NativeApi):
        self.api = api
        self.handle = C.c_void_p()
        api.call("cancel_create", C.byref(self.handle))

    def request(self):
        self.api.call("cancel_request", self.handle)

    def reset(self):
        self.api.call("cancel_reset", self.handle)

    def destroy(self):
        if self.handle:
            self.api.call("cancel_destroy", self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Comm:
    """pg_comm_t: an RCCL communicator owned by the library (one per GPU)."""

    def __init__(self, api: capi.NativeApi, handle):
        self.api = api
        self.handle = handle

    @staticmethod
    def unique_id(api: capi.NativeApi) -> bytes:
        buf = C.create_string_buffer(capi.COMM_UNIQUE_ID_BYTES)
        api.call("comm_get_unique_id", buf)
        return buf.raw

    @staticmethod
    def init_rank(api: capi.NativeApi, device: int, world: int, rank: int, unique_id: bytes) -> "Comm":
        assert len(unique_id) == capi.COMM_UNIQUE_ID_BYTES
        h = C.c_void_p()
        api.call("comm_init_rank", device, world, rank, C.create_string_buffer(unique_id, len(unique_id)), C.byref(h))
        return Comm(api, h)

    @staticmethod
    def init_all(api: capi.NativeApi, devices: Sequence[int]) -> "List[Comm]":
        n = len(devices)
        devs = (C.c_int32 * n)(*devices)
        out = (C.c_void_p * n)()
        api.call("comm_init_all", n, devs, out)
        return [Comm(api, C.c_void_p(out[i])) for i in range(n)]

    def world_size(self) -> int:
        n = C.c_int32()
        self.api.call("comm_world_size", self.handle, C.byref(n))
        return n.value

    def destroy(self):
        if self.handle:
            self.api.call("comm_destroy", self.handle)
            self.handle = C.c_void_p()


class NativeResult:
    """A pg_result_t kept alive for merging inside the library (GroupByCombineOperator over segments sharing their key
    space): merge() folds another segment's result of the same query on the same GPU into this one, all_reduce() merges
    across the GPUs of a communicator over RCCL; block() materialises the (merged) intermediate results."""

    def __init__(self, api: capi.NativeApi, handle, qc: QueryContext, host: HostSegment):
        self.api = api
        self.handle = handle
        self.qc = qc
        self.host = host

    def merge(self, other: "NativeResult") -> "NativeResult":
        self.api.call("result_merge", self.handle, other.handle)
        return self

    def all_reduce(self, comm: Comm) -> "NativeResult":
        self.api.call("result_all_reduce", self.handle, comm.handle)
        return self

    def block(self) -> "ResultsBlock":
        return ResultsBlock.from_native(self.api, self.handle, self.qc, self.host)

    def stats(self) -> capi.PgExecStats:
        st = capi.PgExecStats()
        self.api.call("result_stats", self.handle, C.byref(st))
        return st

    def data_table_v4(self) -> bytes:
        """The (merged) intermediate results as DataTableImplV4 bytes (pg_result_data_table_v4)."""
        size = C.c_int64(0)
        self.api.call("result_data_table_v4", self.handle, None, 0, C.byref(size))
        buf = (C.c_uint8 * max(size.value, 1))()
        self.api.call("result_data_table_v4", self.handle, buf, size.value, C.byref(size))
        return bytes(buf[:size.value])

    def free(self):
        if self.handle:
            self.api.call("result_free", self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DocIdSet:
    def __init__(self, api: capi.NativeApi, handle):
        self.api = api
        self.handle = handle

    def cardinality(self) -> int:
        out = C.c_int64()
        self.api.call("docidset_cardinality", self.handle, C.byref(out))
        return out.value

    def words(self) -> np.ndarray:
        n = C.c_int64()
        self.api.call("docidset_num_words", self.handle, C.byref(n))
        out = np.zeros(n.value, dtype=np.uint64)
        self.api.call("docidset_copy_words", self.handle, out.ctypes.data, n.value)
        return out

    def doc_ids(self) -> np.ndarray:
        n = self.cardinality()
        out = np.zeros(n, dtype=np.int32)
        self.api.call("docidset_copy_docids", self.handle, out.ctypes.data, n)
        return out

    def stats(self) -> capi.PgExecStats:
        s = capi.PgExecStats()
        self.api.call("docidset_stats", self.handle, C.byref(s))
        return s

    def free(self):
        if self.handle:
            self.api.call("docidset_free", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _key_repr(v):
    """Group key component as a dict key: floats keep their sign of zero (-0.0 and 0.0 are two groups in the reference's
    Float2Int / Double2Int maps, but equal as Python floats) and NaN compares equal to itself."""
    if isinstance(v, float):
        if v != v:
            return "NaN"
        if v == 0.0:
            return "-0.0" if math.copysign(1.0, v) < 0 else 0.0
    return v


class ResultsBlock:
    """GroupByResultsBlock / AggregationResultsBlock: intermediate results per group key.

    `rows` maps the decoded group key tuple (dictionary values, as GroupKeyGenerator#getGroupKeys yields them) to the
    list of intermediate results, one per aggregation: COUNT → int, SUM/MIN/MAX → float, AVG → (sum, count),
    MINMAXRANGE → (min, max), DISTINCTCOUNT → frozenset of values, DISTINCTCOUNTHLL → bytes of 2^log2m registers.
    The native arrays are kept as numpy arrays (`arrays`, `group_dict_ids`); the Python-object views (`group_keys`,
    `columns`) are built on first use so that the benchmark loop does not pay for boxing."""

    def __init__(self):
        self.query: Optional[QueryContext] = None
        self.group_dict_ids: Optional[np.ndarray] = None   # [n_group_cols, n_groups]
        self.arrays: List[tuple] = []                       # per aggregation: (kind, component arrays...)
        self.stats: Optional[capi.PgExecStats] = None
        self._host: Optional[HostSegment] = None
        self._group_keys: Optional[List[tuple]] = None
        self._columns: Optional[List[list]] = None

    @staticmethod
    def from_native(api: capi.NativeApi, h, qc: QueryContext, host: HostSegment) -> "ResultsBlock":
        rb = ResultsBlock()
        rb.query = qc
        rb._host = host
        n = C.c_int32()
        api.call("result_num_groups", h, C.byref(n))
        ng = n.value
        ngb = len(qc.group_by)
        ids = np.zeros((ngb, ng), dtype=np.int32)
        rb.group_values = None
        rb.group_value_columns = {}      # group-by column index -> the groups' values (no-dictionary columns)
        for j in range(ngb):
            kt = C.c_int32()
            api.call("result_group_key_type", h, j, C.byref(kt))
            if kt.value == capi.GROUP_KEY_LONG_VALUES:   # a no-dictionary group-by column: the groups' values themselves
                vals = np.zeros(ng, dtype=np.int64)
                api.call("result_group_values_long", h, j, vals.ctypes.data, ng)
                rb.group_value_columns[j] = vals
            elif kt.value == capi.GROUP_KEY_DOUBLE_VALUES:
                vals = np.zeros(ng, dtype=np.float64)
                api.call("result_group_values_double", h, j, vals.ctypes.data, ng)
                rb.group_value_columns[j] = vals
            elif kt.value == capi.GROUP_KEY_BYTES_VALUES:   # a raw STRING / BYTES group-by column: the groups' byte strings
                total = C.c_uint64()
                api.call("result_group_values_bytes_size", h, j, C.byref(total))
                offs = np.zeros(ng + 1, dtype=np.int64)
                blob = np.zeros(max(int(total.value), 1), dtype=np.uint8)
                api.call("result_group_values_bytes", h, j, offs.ctypes.data, ng + 1, blob.ctypes.data, int(total.value))
                raw = blob.tobytes()
                vals = np.empty(ng, dtype=object)
                is_string = host.columns[qc.group_by[j]].data_type == "STRING"
                for i in range(ng):
                    v = raw[offs[i]:offs[i + 1]]
                    vals[i] = v.decode("utf-8") if is_string else v
                rb.group_value_columns[j] = vals
            else:
                api.call("result_group_dict_ids", h, j, ids[j].ctypes.data, ng)
        if ngb == 1 and 0 in rb.group_value_columns and rb.group_value_columns[0].dtype == np.int64:
            rb.group_values = rb.group_value_columns[0]
        rb.group_dict_ids = ids
        for a, spec in enumerate(qc.aggregations):
            kind = C.c_int32()
            api.call("result_kind_of", h, a, C.byref(kind))
            k = kind.value
            if k == capi.RESULT_LONG:
                out = np.zeros(ng, dtype=np.int64)
                api.call("result_longs", h, a, 0, out.ctypes.data, ng)
                rb.arrays.append((k, out))
            elif k == capi.RESULT_DOUBLE:
                out = np.zeros(ng, dtype=np.float64)
                api.call("result_doubles", h, a, 0, out.ctypes.data, ng)
                rb.arrays.append((k, out))
            elif k == capi.RESULT_AVG_PAIR:
                s = np.zeros(ng, dtype=np.float64)
                c = np.zeros(ng, dtype=np.int64)
                api.call("result_doubles", h, a, 0, s.ctypes.data, ng)
                api.call("result_longs", h, a, 0, c.ctypes.data, ng)
                rb.arrays.append((k, s, c))
            elif k == capi.RESULT_MINMAX_PAIR:
                lo = np.zeros(ng, dtype=np.float64)
                hi = np.zeros(ng, dtype=np.float64)
                api.call("result_doubles", h, a, 0, lo.ctypes.data, ng)
                api.call("result_doubles", h, a, 1, hi.ctypes.data, ng)
                rb.arrays.append((k, lo, hi))
            elif k == capi.RESULT_DICTID_SET:
                sizes = np.zeros(ng, dtype=np.int32)
                api.call("result_set_sizes", h, a, sizes.ctypes.data, ng)
                total = int(sizes.sum())
                flat = np.zeros(max(total, 1), dtype=np.int32)
                api.call("result_set_dict_ids", h, a, flat.ctypes.data, total)
                rb.arrays.append((k, sizes, flat[:total]))
            elif k == capi.RESULT_VALUE_SET:   # DISTINCTCOUNT over a raw column: the values themselves
                sizes = np.zeros(ng, dtype=np.int32)
                api.call("result_set_sizes", h, a, sizes.ctypes.data, ng)
                total = int(sizes.sum())
                floating = host.columns[spec.column].data_type in ("FLOAT", "DOUBLE")
                flat = np.zeros(max(total, 1), dtype=np.float64 if floating else np.int64)
                api.call("result_set_values_double" if floating else "result_set_values_long", h, a, flat.ctypes.data, total)
                rb.arrays.append((k, sizes, flat[:total]))
            elif k == capi.RESULT_HLL:
                m = 1 << (spec.log2m or 8)
                regs = np.empty(max(ng * m, 1), dtype=np.uint8)
                api.call("result_hll_registers", h, a, regs.ctypes.data, ng * m)
                rb.arrays.append((k, regs[:ng * m].reshape(ng, m)))
            else:
                raise RuntimeError(f"unknown result kind {k}")
        # enableNullHandling: which results / group keys are NULL (all zero without the flag: not asked for then)
        rb.agg_nulls, rb.key_nulls = {}, {}
        if qc.flags & capi.QUERY_FLAG_NULL_HANDLING:
            for a in range(len(qc.aggregations)):
                f = np.zeros(max(ng, 1), dtype=np.uint8)
                api.call("result_agg_nulls", h, a, f.ctypes.data, ng)
                if f[:ng].any():
                    rb.agg_nulls[a] = f[:ng].astype(bool)
            for j in range(ngb):
                f = np.zeros(max(ng, 1), dtype=np.uint8)
                api.call("result_group_key_nulls", h, j, f.ctypes.data, ng)
                if f[:ng].any():
                    rb.key_nulls[j] = f[:ng].astype(bool)
        st = capi.PgExecStats()
        api.call("result_stats", h, C.byref(st))
        rb.stats = st
        return rb

    @property
    def num_groups(self) -> int:
        return self.group_dict_ids.shape[1] if self.query.group_by else 1

    @property
    def group_keys(self) -> List[tuple]:
        if self._group_keys is None and getattr(self, "group_values", None) is not None:
            kn = getattr(self, "key_nulls", {}).get(0)
            self._group_keys = [(None if (kn is not None and kn[i]) else int(v),) for i, v in enumerate(self.group_values)]
        if self._group_keys is None:
            ids = self.group_dict_ids
            vcols = getattr(self, "group_value_columns", {})
            ng = self.num_groups
            per_col = []
            for j, g in enumerate(self.query.group_by):
                if j in vcols:   # raw values; a NaN key is one group: give it a key that compares equal to itself
                    v = vcols[j]
                    per_col.append(list(v) if v.dtype == object else ([int(x) for x in v] if v.dtype == np.int64 else [float(x) for x in v]))
                else:
                    dv = self._host.columns[g].dict_values
                    per_col.append([dv[ids[j, i]] for i in range(ng)])
            kn = getattr(self, "key_nulls", {})
            self._group_keys = [tuple(None if (j in kn and kn[j][i]) else _key_repr(per_col[j][i]) for j in range(len(per_col))) for i in range(ng)]
        return self._group_keys

    @property
    def columns(self) -> List[list]:
        if self._columns is None:
            cols = []
            for spec, arr in zip(self.query.aggregations, self.arrays):
                k = arr[0]
                if k == capi.RESULT_LONG:
                    cols.append([int(v) for v in arr[1]])
                elif k == capi.RESULT_DOUBLE:
                    cols.append([float(v) for v in arr[1]])
                elif k == capi.RESULT_AVG_PAIR:
                    cols.append([(float(x), int(y)) for x, y in zip(arr[1], arr[2])])
                elif k == capi.RESULT_MINMAX_PAIR:
                    cols.append([(float(x), float(y)) for x, y in zip(arr[1], arr[2])])
                elif k == capi.RESULT_DICTID_SET:
                    dv = self._host.columns[spec.column].dict_values
                    col, pos = [], 0
                    for sz in arr[1]:
                        col.append(frozenset(dv[d] for d in arr[2][pos:pos + sz]))
                        pos += sz
                    cols.append(col)
                elif k == capi.RESULT_VALUE_SET:
                    col, pos = [], 0
                    for sz in arr[1]:
                        col.append(frozenset(arr[2][pos:pos + sz].tolist()))
                        pos += sz
                    cols.append(col)
                else:
                    cols.append([bytes(r) for r in arr[1]])
            for a, flags in getattr(self, "agg_nulls", {}).items():   # enableNullHandling: a NULL result
                cols[a] = [None if flags[i] else v for i, v in enumerate(cols[a])]
            self._columns = cols
        return self._columns

    # -- conveniences ----------------------------------------------------------------------------------------------
    def execution_statistics(self) -> ExecutionStatistics:
        s = self.stats
        return ExecutionStatistics(s.num_docs_scanned, s.num_entries_scanned_in_filter,
                                   s.num_entries_scanned_post_filter, s.num_total_docs)

    def rows(self) -> Dict[tuple, list]:
        return {k: [col[i] for col in self.columns] for i, k in enumerate(self.group_keys)}

    def aggregation_result(self) -> list:
        """AggregationResultsBlock#getResults for a query without GROUP BY."""
        assert not self.query.group_by
        return [col[0] for col in self.columns]


# ----------------------------------------------------------------------------------------------------------------------
# HyperLogLog cardinality (final result extraction; stream-lib HyperLogLog#cardinality)
# ----------------------------------------------------------------------------------------------------------------------
def hll_cardinality(registers: bytes) -> int:
    m = len(registers)
    if m == 16:
        alpha_mm = 0.673 * m * m
    elif m == 32:
        alpha_mm = 0.697 * m * m
    elif m == 64:
        alpha_mm = 0.709 * m * m
    else:
        alpha_mm = (0.7213 / (1 + 1.079 / m)) * m * m
    s = 0.0
    zeros = 0.0
    for v in registers:
        s += 1.0 / (1 << v)
        if v == 0:
            zeros += 1
    est = alpha_mm * (1 / s)
    if est <= 2.5 * m:
        if zeros == 0:   # linearCounting: m * log(m / 0.0) = Infinity, Math.round(Infinity) = Long.MAX_VALUE
            return (1 << 63) - 1
        return int(math.floor(m * math.log(m / zeros) + 0.5))
    return int(math.floor(est + 0.5))


def hll_merge(a: bytes, b: bytes) -> bytes:
    return bytes(max(x, y) for x, y in zip(a, b))


# ----------------------------------------------------------------------------------------------------------------------
# GroupByCombineOperator + IndexedTable: merge per-segment intermediate results keyed by *values*
# (dictionaries are per segment; GroupByCombineOperator.java:132-147 decodes before upserting).
# ----------------------------------------------------------------------------------------------------------------------
def merge_intermediate(function: str, a, b):
    """AggregationFunction#merge for the functions on the path (the *MV forms merge like their single-value forms)."""
    function = capi.MV_TO_SV_FUNCTION.get(function, function)
    if function in ("COUNT", "SUM"):
        return a + b
    if function == "MIN":
        return b if b < a else a
    if function == "MAX":
        return b if b > a else a
    if function == "AVG":
        return (a[0] + b[0], a[1] + b[1])
    if function == "MINMAXRANGE":
        return (min(a[0], b[0]), max(a[1], b[1]))
    if function == "DISTINCTCOUNT":
        return a | b
    if function == "DISTINCTCOUNTHLL":
        return hll_merge(a, b)
    raise ValueError(function)


def extract_final(function: str, v):
    """AggregationFunction#extractFinalResult."""
    function = capi.MV_TO_SV_FUNCTION.get(function, function)
    if function == "AVG":
        return v[0] / v[1] if v[1] else float("-inf")
    if function == "MINMAXRANGE":
        return v[1] - v[0]
    if function == "DISTINCTCOUNT":
        return len(v)
    if function == "DISTINCTCOUNTHLL":
        return hll_cardinality(v)
    return v


class GroupByCombineOperator:
    """Host-side merge of per-segment (per-GPU) results: IndexedTable#upsert semantics (key → record, merge per
    aggregation function).  The RCCL variant for identical key spaces lives in pinot_amd.distributed."""

    def __init__(self, blocks: Sequence[ResultsBlock]):
        self.blocks = list(blocks)

    def merge(self) -> Dict[tuple, list]:
        table: Dict[tuple, list] = {}
        fns = [a.function for a in self.blocks[0].query.aggregations]
        for b in self.blocks:
            for key, vals in b.rows().items():
                cur = table.get(key)
                if cur is None:
                    table[key] = list(vals)
                else:
                    table[key] = [merge_intermediate(f, x, y) for f, x, y in zip(fns, cur, vals)]
        return table

    def final(self) -> Dict[tuple, list]:
        fns = [a.function for a in self.blocks[0].query.aggregations]
        return {k: [extract_final(f, v) for f, v in zip(fns, vals)] for k, vals in self.merge().items()}


class GpuInstancePlanMaker:
    """PlanMaker for the accelerated path.  `make_segment_plan_node(segment, query)` returns a callable plan node whose
    run() yields the operator result, or raises NativeError(PG_ERR_UNSUPPORTED) so the caller can fall back to the
    default plan — the contract the Java GpuInstancePlanMaker follows."""

    def __init__(self):
        self.api = capi.gpu_api()

    def load_segment(self, host: HostSegment) -> NativeSegment:
        return NativeSegment(self.api, host)

    def make_segment_plan_node(self, segment: NativeSegment, query):
        qc = parse_sql(query) if isinstance(query, str) else query
        cq = CQuery(qc)
        self.api.call("query_supported", segment.handle, cq.ptr())
        return _SegmentPlanNode(segment, qc)


class _SegmentPlanNode:
    def __init__(self, segment: NativeSegment, qc: QueryContext):
        self.segment = segment
        self.qc = qc

    def run(self) -> ResultsBlock:
        return self.segment.execute(self.qc)
