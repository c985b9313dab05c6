"""ctypes mirror of include/pinot_gpu.h.

`NativeApi(lib, prefix)` binds either the product library (libpinot_gpu.so, prefix "pg_") or — in tests only — the CPU
oracle (oracle/_build/liboracle.so, prefix "po_"), which exports the same entry points over the same structs so that
parity tests feed both sides byte-identical segments and queries.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

PG_ABI_VERSION = 4

# pg_status
PG_OK = 0
PG_ERR_INVALID_ARGUMENT = -1
PG_ERR_UNSUPPORTED = -2
PG_ERR_DEVICE = -3
PG_ERR_OUT_OF_MEMORY = -4
PG_ERR_NOT_FOUND = -5
PG_ERR_CANCELLED = -6
PG_ERR_INTERNAL = -7

DATA_TYPES = {"INT": 0, "LONG": 1, "FLOAT": 2, "DOUBLE": 3, "STRING": 4, "BYTES": 5}
FWD_DICT_FIXED_BIT = 0
FWD_RAW_FIXED_BYTE_CHUNK = 1
FWD_DICT_SORTED = 2
FWD_DICT_FIXED_BIT_MV = 3
FWD_RAW_VAR_BYTE_CHUNK = 4
FWD_RAW_MV_FIXED_BYTE_CHUNK = 5
FWD_RAW_MV_VAR_BYTE_CHUNK = 6

FILTER_AND, FILTER_OR, FILTER_NOT, FILTER_PREDICATE, FILTER_CONSTANT_TRUE, FILTER_CONSTANT_FALSE = range(6)
PRED_EQ, PRED_NOT_EQ, PRED_IN, PRED_NOT_IN, PRED_RANGE, PRED_IS_NULL, PRED_IS_NOT_NULL = range(7)
AGG_FUNCTIONS = {"COUNT": 0, "SUM": 1, "MIN": 2, "MAX": 3, "AVG": 4, "DISTINCTCOUNT": 5, "DISTINCTCOUNTHLL": 6,
                 "MINMAXRANGE": 7, "COUNTMV": 8, "SUMMV": 9, "MINMV": 10, "MAXMV": 11, "AVGMV": 12, "MINMAXRANGEMV": 13,
                 "DISTINCTCOUNTMV": 14, "DISTINCTCOUNTHLLMV": 15}
MV_TO_SV_FUNCTION = {"COUNTMV": "COUNT", "SUMMV": "SUM", "MINMV": "MIN", "MAXMV": "MAX", "AVGMV": "AVG", "MINMAXRANGEMV": "MINMAXRANGE",
                     "DISTINCTCOUNTMV": "DISTINCTCOUNT", "DISTINCTCOUNTHLLMV": "DISTINCTCOUNTHLL"}
RESULT_LONG, RESULT_DOUBLE, RESULT_AVG_PAIR, RESULT_MINMAX_PAIR, RESULT_DICTID_SET, RESULT_HLL, RESULT_VALUE_SET = range(7)

QUERY_FLAG_PROFILE = 0x1
QUERY_FLAG_SKIP_STAR_TREE = 0x2
QUERY_FLAG_KEEP_DEVICE_TABLE = 0x4
QUERY_FLAG_APPROX_FILTER_STATS = 0x8
QUERY_FLAG_EXACT_FILTER_STATS = 0x10
QUERY_FLAG_FINAL_DISTINCT = 0x20
QUERY_FLAG_NULL_HANDLING = 0x40
COMM_UNIQUE_ID_BYTES = 128
GROUP_KEY_DICT_IDS, GROUP_KEY_LONG_VALUES, GROUP_KEY_DOUBLE_VALUES, GROUP_KEY_BYTES_VALUES = 0, 1, 2, 3


class PgBuffer(C.Structure):
    _fields_ = [("addr", C.c_void_p), ("size", C.c_uint64)]


class PgColumnDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("data_type", C.c_int32),
        ("fwd_encoding", C.c_int32),
        ("has_dictionary", C.c_int32),
        ("cardinality", C.c_int32),
        ("bits_per_value", C.c_int32),
        ("is_sorted", C.c_int32),
        ("dict_bytes_per_value", C.c_int32),
        ("total_number_of_entries", C.c_int32),
        ("forward_index", PgBuffer),
        ("dictionary", PgBuffer),
        ("inverted_index", PgBuffer),
    ]


class PgStarTreePair(C.Structure):
    _fields_ = [("function", C.c_int32), ("data_type", C.c_int32), ("column", C.c_char_p), ("forward_index", PgBuffer)]


class PgStarTreeDesc(C.Structure):
    _fields_ = [
        ("num_docs", C.c_int32),
        ("n_dimensions", C.c_int32),
        ("n_pairs", C.c_int32),
        ("max_leaf_records", C.c_int32),
        ("dimensions", C.POINTER(C.c_char_p)),
        ("dimension_forward_indexes", C.POINTER(PgBuffer)),
        ("pairs", C.POINTER(PgStarTreePair)),
        ("star_tree", PgBuffer),
    ]


class PgFilterNode(C.Structure):
    pass


PgFilterNode._fields_ = [
    ("type", C.c_int32),
    ("n_children", C.c_int32),
    ("children", C.POINTER(PgFilterNode)),
    ("predicate_type", C.c_int32),
    ("n_values", C.c_int32),
    ("column", C.c_char_p),
    ("values", C.POINTER(C.c_char_p)),
    ("lower", C.c_char_p),
    ("upper", C.c_char_p),
    ("lower_inclusive", C.c_int32),
    ("upper_inclusive", C.c_int32),
]


class PgAggSpec(C.Structure):
    _fields_ = [("function", C.c_int32), ("log2m", C.c_int32), ("column", C.c_char_p)]


ORDER_BY_GROUP_KEY, ORDER_BY_AGGREGATION = 0, 1


class PgOrderBy(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("ascending", C.c_int32), ("nulls_last", C.c_int32)]


class PgQuery(C.Structure):
    _fields_ = [
        ("filter", C.POINTER(PgFilterNode)),
        ("n_group_by", C.c_int32),
        ("n_aggregations", C.c_int32),
        ("group_by_columns", C.POINTER(C.c_char_p)),
        ("aggregations", C.POINTER(PgAggSpec)),
        ("num_groups_limit", C.c_int32),
        ("max_initial_result_holder_capacity", C.c_int32),
        ("flags", C.c_int32),
        ("n_order_by", C.c_int32),
        ("order_by", C.POINTER(PgOrderBy)),
        ("limit", C.c_int32),
        ("min_segment_group_trim_size", C.c_int32),
    ]


class PgExecStats(C.Structure):
    _fields_ = [
        ("num_docs_scanned", C.c_int64),
        ("num_entries_scanned_in_filter", C.c_int64),
        ("num_entries_scanned_post_filter", C.c_int64),
        ("num_total_docs", C.c_int64),
        ("num_groups_limit_reached", C.c_int32),
        ("stats_exact", C.c_int32),
        ("device_ms_total", C.c_float),
        ("device_ms_filter", C.c_float),
        ("device_ms_aggregate", C.c_float),
        ("device_ms_reduce", C.c_float),
        ("host_ms_plan", C.c_float),
        ("host_ms_total", C.c_float),
        ("algorithmic_bytes", C.c_int64),
        ("kernel", C.c_char * 32),
        ("star_tree_index", C.c_int32),
        ("filter_stats_path", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class NativeError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[{status}] {message}")
        self.status = status
        self.message = message


REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PG_GPU_LIB: measurement knob (kernel variants built next to the product library by tools/build_variants.sh)
GPU_LIB_PATH = os.environ.get("PG_GPU_LIB") or os.path.join(REPO_ROOT, "pinot_amd", "csrc", "libpinot_gpu.so")

# every symbol include/pinot_gpu.h declares (checked by the "not gpu" suite against the built library)
ABI_SYMBOLS = [
    "abi_version", "init", "device_count", "last_error",
    "segment_create", "segment_add_column", "segment_add_star_tree", "segment_set_null_vector", "segment_set_range_index", "segment_set_queryable_doc_ids", "segment_num_docs", "segment_device_bytes", "segment_destroy",
    "filter_exec", "filter_exec_flags", "docidset_cardinality", "docidset_num_words", "docidset_copy_words", "docidset_copy_docids",
    "docidset_stats", "docidset_free",
    "query_supported", "query_exec",
    "result_num_groups", "result_group_dict_ids", "result_group_key_type", "result_group_values_long", "result_group_values_double",
    "result_group_values_bytes_size", "result_group_values_bytes", "result_kind_of", "result_doubles", "result_longs",
    "result_set_sizes", "result_set_dict_ids", "result_set_values_long", "result_set_values_double", "result_hll_registers", "result_agg_nulls", "result_group_key_nulls", "result_stats", "result_free",
]
# entry points only the product library has (the CPU oracle is one segment, one thread, no devices): multi-GPU placement,
# cancellation, the dense cross-segment merge and its RCCL communicators
GPU_ONLY_SYMBOLS = [
    "segment_create_on_device", "segment_device",
    "cancel_create", "cancel_request", "cancel_reset", "cancel_destroy", "query_exec_cancellable",
    "result_merge", "result_all_reduce", "result_data_table_v4",
    "comm_get_unique_id", "comm_init_rank", "comm_init_all", "comm_world_size", "comm_destroy",
    "options_reload",
]


class NativeApi:
    """Thin typed wrapper over a shared library exporting the pinot_gpu.h entry points under `prefix`."""

    def __init__(self, path: str, prefix: str = "pg_"):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL if prefix == "pg_" else C.RTLD_LOCAL)
        for sym in ABI_SYMBOLS + (GPU_ONLY_SYMBOLS if prefix == "pg_" else []):
            fn = getattr(self.lib, prefix + sym)  # AttributeError => missing export
            fn.restype = C.c_int32
        if prefix == "pg_":
            self.f("segment_create_on_device").argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
            self.f("segment_device").argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
            self.f("cancel_create").argtypes = [C.POINTER(C.c_void_p)]
            self.f("cancel_request").argtypes = [C.c_void_p]
            self.f("cancel_reset").argtypes = [C.c_void_p]
            self.f("cancel_destroy").argtypes = [C.c_void_p]
            self.f("query_exec_cancellable").argtypes = [C.c_void_p, C.POINTER(PgQuery), C.c_void_p, C.POINTER(C.c_void_p)]
            self.f("result_merge").argtypes = [C.c_void_p, C.c_void_p]
            self.f("result_all_reduce").argtypes = [C.c_void_p, C.c_void_p]
            self.f("result_data_table_v4").argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
            self.f("comm_get_unique_id").argtypes = [C.c_void_p]
            self.f("comm_init_rank").argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
            self.f("comm_init_all").argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
            self.f("comm_world_size").argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
            self.f("comm_destroy").argtypes = [C.c_void_p]
            self.f("options_reload").argtypes = []
        self.f("last_error").argtypes = [C.c_char_p, C.c_size_t]
        self.f("init").argtypes = [C.c_int32]
        self.f("segment_create").argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
        self.f("segment_add_column").argtypes = [C.c_void_p, C.POINTER(PgColumnDesc)]
        self.f("segment_add_star_tree").argtypes = [C.c_void_p, C.POINTER(PgStarTreeDesc)]
        self.f("segment_num_docs").argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self.f("segment_device_bytes").argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self.f("segment_destroy").argtypes = [C.c_void_p]
        self.f("filter_exec").argtypes = [C.c_void_p, C.POINTER(PgFilterNode), C.POINTER(C.c_void_p)]
        self.f("filter_exec_flags").argtypes = [C.c_void_p, C.POINTER(PgFilterNode), C.c_int32, C.POINTER(C.c_void_p)]
        self.f("docidset_cardinality").argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self.f("docidset_num_words").argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self.f("docidset_copy_words").argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self.f("docidset_copy_docids").argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self.f("docidset_stats").argtypes = [C.c_void_p, C.POINTER(PgExecStats)]
        self.f("docidset_free").argtypes = [C.c_void_p]
        self.f("query_supported").argtypes = [C.c_void_p, C.POINTER(PgQuery)]
        self.f("query_exec").argtypes = [C.c_void_p, C.POINTER(PgQuery), C.POINTER(C.c_void_p)]
        self.f("result_num_groups").argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self.f("result_group_dict_ids").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("segment_set_null_vector").argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        self.f("segment_set_queryable_doc_ids").argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self.f("segment_set_range_index").argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        self.f("result_group_key_type").argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        self.f("result_group_values_long").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_group_values_double").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_group_values_bytes_size").argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
        self.f("result_group_values_bytes").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
        self.f("result_kind_of").argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        self.f("result_doubles").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_longs").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_set_sizes").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_set_dict_ids").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        self.f("result_set_values_long").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        self.f("result_set_values_double").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        self.f("result_hll_registers").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        self.f("result_agg_nulls").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_group_key_nulls").argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        self.f("result_stats").argtypes = [C.c_void_p, C.POINTER(PgExecStats)]
        self.f("result_free").argtypes = [C.c_void_p]

    def f(self, name: str):
        return getattr(self.lib, self.prefix + name)

    def last_error(self) -> str:
        buf = C.create_string_buffer(4096)
        self.f("last_error")(buf, 4096)
        return buf.value.decode("utf-8", "replace")

    def check(self, status: int) -> None:
        if status != PG_OK:
            raise NativeError(status, self.last_error())

    def call(self, name: str, *args) -> None:
        self.check(self.f(name)(*args))


def np_buffer(arr: Optional[np.ndarray]) -> PgBuffer:
    if arr is None or arr.size == 0:
        return PgBuffer(None, 0)
    assert arr.dtype == np.uint8 and arr.flags["C_CONTIGUOUS"]
    return PgBuffer(arr.ctypes.data, arr.nbytes)


_gpu_api: Optional[NativeApi] = None


def gpu_api() -> NativeApi:
    """Loads libpinot_gpu.so. Fails loudly when the HIP extension has not been built (no CPU fallback exists)."""
    global _gpu_api
    if _gpu_api is None:
        if not os.path.exists(GPU_LIB_PATH):
            raise RuntimeError(
                f"{GPU_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pinot_amd/csrc`. There is no CPU fallback for the product path.")
        _gpu_api = NativeApi(GPU_LIB_PATH, "pg_")
    return _gpu_api
