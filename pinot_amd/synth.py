"""Synthetic `gpuBench` table (BASELINE.md §2 / SURVEY.md §8d): a pure function of (seed, column, docId).

    h = splitmix64(splitmix64(seed ^ salt(column)) + docId);   value = ((h >> 32) * range) >> 32

`generate_segment()` uses the multi-threaded native writer (pinot_amd/csrc/synth/pg_synth.cpp → libpinot_synth.so) when it
is built and falls back to the numpy restatement below for small segments (tests cross-check the two byte for byte).
Dictionary columns have the identity dictionary [0, cardinality), so dictId == value.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional

import numpy as np

from . import capi, formats
from .segment import HostColumn, HostSegment

SEED_BASE = 0x50494E4F54  # "PINOT"; segment s uses SEED_BASE ^ s


@dataclass(frozen=True)
class SynthColumn:
    name: str
    kind: str            # "dict" (fixed-bit dictIds, INT identity dictionary) or "raw" (INT, PASS_THROUGH chunks)
    range: int           # cardinality for dict columns, exclusive upper bound of values for raw columns
    inverted: bool = False
    like: str = ""       # non-empty: the column repeats that column's values doc for doc (its dictionary-encoded twin)
    dictionary: str = "identity"   # "identity": value = dictId (what the segment creator builds when every value of [0, range) occurs);
                                   # "sparse": a sorted INT dictionary that is NOT arithmetic (sparse_dictionary below): values are gathered

    @property
    def salt(self) -> int:
        h = 1469598103934665603
        for ch in (self.like or self.name).encode():
            h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h


GPU_BENCH_COLUMNS: List[SynthColumn] = [
    SynthColumn("c_inv1", "dict", 8, inverted=True),
    SynthColumn("c_inv2", "dict", 4, inverted=True),
    SynthColumn("r_int", "raw", 1_000_000),
    SynthColumn("g1", "dict", 100),
    SynthColumn("g2", "dict", 50),
    SynthColumn("m", "raw", 1 << 20),
    SynthColumn("h1", "dict", 16),
    SynthColumn("h2", "dict", 10),
    SynthColumn("h3", "dict", 10),
    SynthColumn("h4", "dict", 8),
    SynthColumn("u", "dict", 1_000_000),
    # config 3 in Pinot's DEFAULT encoding (DictionaryIndexConfig.java:32: every column has a dictionary unless the table config says otherwise):
    # the same docs as r_int / m, as 20-bit dictId streams.  With every value of the range present the sorted dictionary is [0, range).
    SynthColumn("r_int_d", "dict", 1_000_000, like="r_int"),
    SynthColumn("m_d", "dict", 1 << 20, like="m"),
    # ... and with dictionaries that are not arithmetic progressions: the range is a dictId interval found by binary search, SUM / MAX gather
    SynthColumn("r_int_s", "dict", 1_000_000, like="r_int", dictionary="sparse"),
    SynthColumn("m_s", "dict", 1 << 20, like="m", dictionary="sparse"),
]
GPU_BENCH = {c.name: c for c in GPU_BENCH_COLUMNS}

# BASELINE.md queries
QUERY_CFG2 = "SELECT COUNT(*) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999"
QUERY_CFG3 = ("SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) "
              "AND r_int BETWEEN 250000 AND 749999 GROUP BY g1 ORDER BY g1 LIMIT 1000")
QUERY_NORTH_STAR = ("SELECT g1, g2, SUM(m) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) "
                    "AND r_int BETWEEN 250000 AND 749999 GROUP BY g1, g2 ORDER BY g1, g2 LIMIT 10000")
QUERY_CFG5 = ("SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY h1, h2, h3, h4 LIMIT 20000")
CFG3_COLUMNS = ["c_inv1", "c_inv2", "r_int", "g1", "g2", "m"]
# config 3 / north-star over the dictionary-encoded twins (identity dictionaries: the rows equal QUERY_CFG3's / QUERY_NORTH_STAR's)
QUERY_CFG3_DICT = QUERY_CFG3.replace("r_int", "r_int_d").replace("(m)", "(m_d)")
QUERY_NORTH_STAR_DICT = QUERY_NORTH_STAR.replace("r_int", "r_int_d").replace("(m)", "(m_d)")
# ... and over the sparse dictionaries: value = sparse_dictionary(card)[dictId] ~ 3 x dictId, so the same dictId interval is [750000, 2249999]
QUERY_CFG3_SPARSE = ("SELECT g1, SUM(m_s), MAX(m_s) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) "
                     "AND r_int_s BETWEEN 750000 AND 2249999 GROUP BY g1 ORDER BY g1 LIMIT 1000")
CFG3_DICT_COLUMNS = ["c_inv1", "c_inv2", "r_int_d", "g1", "g2", "m_d"]
CFG3_SPARSE_COLUMNS = ["c_inv1", "c_inv2", "r_int_s", "g1", "g2", "m_s"]
CFG5_COLUMNS = ["h1", "h2", "h3", "h4", "u"]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def values_numpy(col: SynthColumn, seed: int, n: int, start: int = 0) -> np.ndarray:
    key = _splitmix64(np.array([(seed ^ col.salt) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        h = _splitmix64(key + np.arange(start, start + n, dtype=np.uint64))
    return (((h >> np.uint64(32)) * np.uint64(col.range)) >> np.uint64(32)).astype(np.int32)


_SYNTH_LIB_PATH = os.path.join(capi.REPO_ROOT, "pinot_amd", "csrc", "libpinot_synth.so")
_lib = None


def synth_lib():
    global _lib
    if _lib is None and os.path.exists(_SYNTH_LIB_PATH):
        lib = C.CDLL(_SYNTH_LIB_PATH)
        lib.pgs_default_threads.restype = C.c_int
        lib.pgs_fill_values.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
        lib.pgs_fill_values.restype = None
        lib.pgs_fill_fixed_bit.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
        lib.pgs_fill_fixed_bit.restype = None
        if hasattr(lib, "pgs_fill_fixed_bit_from"):
            lib.pgs_fill_fixed_bit_from.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
            lib.pgs_fill_fixed_bit_from.restype = None
        lib.pgs_raw_int_index.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int]
        lib.pgs_raw_int_index.restype = C.c_int64
        lib.pgs_inverted_begin.argtypes = [C.c_int64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int64)]
        lib.pgs_inverted_begin.restype = C.c_void_p
        lib.pgs_inverted_fill.argtypes = [C.c_void_p, C.c_void_p]
        lib.pgs_inverted_fill.restype = None
        lib.pgs_inverted_end.argtypes = [C.c_void_p]
        lib.pgs_inverted_end.restype = None
        _lib = lib
    return _lib


def sparse_dictionary(card: int) -> np.ndarray:
    """A sorted INT dictionary that is no arithmetic progression: 3 x dictId + one pseudo-random bit."""
    ids = np.arange(card, dtype=np.uint64)
    return (3 * ids + (((ids * np.uint64(0x9E3779B1)) >> np.uint64(31)) & np.uint64(1))).astype(np.int32)


def _dictionary_values(col: SynthColumn) -> np.ndarray:
    return sparse_dictionary(col.range) if col.dictionary == "sparse" else np.arange(col.range, dtype=np.int32)


def _identity_dictionary(card: int) -> np.ndarray:
    return formats.write_numeric_dictionary(np.arange(card, dtype=np.int32), "INT")


def _dictionary(col: SynthColumn):
    values = _dictionary_values(col)
    return formats.write_numeric_dictionary(values, "INT"), (list(range(col.range)) if col.dictionary == "identity" else values.tolist())


def _column_numpy(col: SynthColumn, seed: int, n: int, raw_version: int) -> HostColumn:
    vals = values_numpy(col, seed, n)
    if col.kind == "raw":
        fwd = formats.write_raw_fixed_byte_chunk(vals, "INT", version=raw_version)
        return HostColumn(col.name, "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, fwd)
    bits = formats.num_bits_per_value(col.range - 1)
    fwd = formats.pack_fixed_bit(vals, bits)
    inv = formats.write_inverted_index(vals, col.range) if col.inverted else None
    dict_bytes, dict_values = _dictionary(col)
    return HostColumn(col.name, "INT", capi.FWD_DICT_FIXED_BIT, True, col.range, bits, False, 4, fwd, dict_bytes, inv, dict_values)


def _column_native(lib, col: SynthColumn, seed: int, n: int, raw_version: int, threads: int) -> HostColumn:
    if col.kind == "raw":
        size = lib.pgs_raw_int_index(None, n, seed, col.salt, col.range, raw_version, 1000, threads)
        fwd = np.empty(size, dtype=np.uint8)
        lib.pgs_raw_int_index(fwd.ctypes.data, n, seed, col.salt, col.range, raw_version, 1000, threads)
        return HostColumn(col.name, "INT", capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, fwd)
    bits = formats.num_bits_per_value(col.range - 1)
    fwd = np.zeros((n * bits + 7) // 8, dtype=np.uint8)
    lib.pgs_fill_fixed_bit(fwd.ctypes.data, n, bits, seed, col.salt, col.range, threads)
    inv = None
    if col.inverted:
        total = C.c_int64()
        h = lib.pgs_inverted_begin(n, col.range, seed, col.salt, threads, C.byref(total))
        inv = np.empty(total.value, dtype=np.uint8)
        lib.pgs_inverted_fill(h, inv.ctypes.data)
        lib.pgs_inverted_end(h)
    dict_bytes, dict_values = _dictionary(col)
    return HostColumn(col.name, "INT", capi.FWD_DICT_FIXED_BIT, True, col.range, bits, False, 4, fwd, dict_bytes, inv, dict_values)


def generate_doc_range(first_doc: int, num_docs: int, segment_index: int = 0, columns: Optional[Iterable[str]] = None, threads: int = 0,
                       name: Optional[str] = None) -> HostSegment:
    """The docs [first_doc, first_doc + num_docs) of gpuBench segment `segment_index` as a segment of their own (dictionary columns only:
    the doc-sharded oracle runs of the full-size tests).  Values are a pure function of (segment, column, docId)."""
    seed = SEED_BASE ^ segment_index
    lib = synth_lib()
    seg = HostSegment(name or f"gpuBench_{segment_index}_from_{first_doc}", num_docs)
    for col in [GPU_BENCH[c] for c in (columns or [])]:
        assert col.kind != "raw" and not col.inverted, "generate_doc_range: dictionary columns without an inverted index"
        bits = formats.num_bits_per_value(col.range - 1)
        if lib is not None and hasattr(lib, "pgs_fill_fixed_bit_from"):
            fwd = np.zeros((num_docs * bits + 7) // 8, dtype=np.uint8)
            lib.pgs_fill_fixed_bit_from(fwd.ctypes.data, first_doc, num_docs, bits, seed, col.salt, col.range, threads if threads > 0 else lib.pgs_default_threads())
        else:
            fwd = formats.pack_fixed_bit(values_numpy(col, seed, num_docs, start=first_doc), bits)
        dict_bytes, dict_values = _dictionary(col)
        seg.columns[col.name] = HostColumn(col.name, "INT", capi.FWD_DICT_FIXED_BIT, True, col.range, bits, False, 4, fwd, dict_bytes, None, dict_values)
    return seg


def generate_segment(num_docs: int, segment_index: int = 0, columns: Optional[Iterable[str]] = None,
                     native: Optional[bool] = None, threads: int = 0, name: Optional[str] = None) -> HostSegment:
    """Builds gpuBench segment `segment_index` with `num_docs` docs. Raw columns above 2^29 docs use chunk-offset
    version 3 (8-byte offsets), as BaseChunkForwardIndexWriter requires beyond 2 GB."""
    seed = SEED_BASE ^ segment_index
    cols = [GPU_BENCH[c] for c in (columns or [c.name for c in GPU_BENCH_COLUMNS])]
    lib = synth_lib() if native in (None, True) else None
    if native is True and lib is None:
        raise RuntimeError(f"{_SYNTH_LIB_PATH} is not built")
    raw_version = 2 if num_docs * 4 + 28 + 4 * ((num_docs + 999) // 1000) <= 0x7FFFFFFF else 3
    seg = HostSegment(name or f"gpuBench_{segment_index}", num_docs)
    if lib is not None and threads <= 0:
        threads = lib.pgs_default_threads()
    for col in cols:
        if lib is not None:
            seg.columns[col.name] = _column_native(lib, col, seed, num_docs, raw_version, threads)
        else:
            seg.columns[col.name] = _column_numpy(col, seed, num_docs, raw_version)
    return seg
