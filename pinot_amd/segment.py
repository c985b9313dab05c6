"""Host-side segment model: the bytes of one immutable Pinot segment's index entries, column by column.

`build_segment()` plays the role SegmentIndexCreationDriverImpl plays in the reference's tests
(pinot-core/src/test/.../queries/BaseSingleValueQueriesTest.java:107-131): turn rows into dictionaries, forward
indexes and inverted indexes in Pinot's byte layouts.  Only what the hot path reads is produced.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import capi, formats

NUMERIC_NP = {"INT": np.int32, "LONG": np.int64, "FLOAT": np.float32, "DOUBLE": np.float64}


@dataclass
class HostColumn:
    name: str
    data_type: str                      # stored type: INT / LONG / FLOAT / DOUBLE / STRING
    fwd_encoding: int
    has_dictionary: bool
    cardinality: int
    bits_per_value: int
    is_sorted: bool
    dict_bytes_per_value: int
    forward_index: np.ndarray
    dictionary: Optional[np.ndarray] = None
    inverted_index: Optional[np.ndarray] = None
    dict_values: Optional[list] = None  # decoded dictionary (python objects / numpy scalars) for result decoding
    null_vector: Optional[np.ndarray] = None  # NullValueVectorReader: serialized RoaringBitmap of the null docIds (uint8 array)
    range_index: Optional[np.ndarray] = None  # BitSlicedRangeIndexReader bytes (the column's `range_index` entry)
    total_number_of_entries: int = 0    # multi-value columns: values over all docs (ColumnMetadata#getTotalNumberOfEntries)
    _name_bytes: bytes = b""

    def desc(self) -> capi.PgColumnDesc:
        self._name_bytes = self.name.encode()
        return capi.PgColumnDesc(
            name=self._name_bytes,
            data_type=capi.DATA_TYPES[self.data_type],
            fwd_encoding=self.fwd_encoding,
            has_dictionary=int(self.has_dictionary),
            cardinality=self.cardinality,
            bits_per_value=self.bits_per_value,
            is_sorted=int(self.is_sorted),
            dict_bytes_per_value=self.dict_bytes_per_value,
            total_number_of_entries=int(self.total_number_of_entries),
            forward_index=capi.np_buffer(self.forward_index),
            dictionary=capi.np_buffer(self.dictionary),
            inverted_index=capi.np_buffer(self.inverted_index),
        )


@dataclass
class HostSegment:
    name: str
    total_docs: int
    columns: Dict[str, HostColumn] = field(default_factory=dict)
    star_trees: list = field(default_factory=list)      # List[startree.HostStarTree] (IndexSegment#getStarTrees)

    def nbytes(self) -> int:
        t = 0
        for c in self.columns.values():
            for b in (c.forward_index, c.dictionary, c.inverted_index):
                if b is not None:
                    t += b.nbytes
        return t


def add_range_index(col: HostColumn, values) -> HostColumn:
    """BitSlicedRangeIndexCreator over the column: dictIds for dictionary columns, value - min for raw INT / LONG, FPOrdering
    ordinals for raw FLOAT / DOUBLE (the creator's two constructors, BitSlicedRangeIndexCreator.java:55-75)."""
    if col.has_dictionary:
        ids = decode_column(col, len(values), dict_ids=True).astype(np.uint64)
        col.range_index = formats.write_range_index(ids, 0, col.cardinality - 1)
    elif col.data_type in ("INT", "LONG"):
        v = np.asarray(values, dtype=np.int64)
        mn, mx = (int(v.min()), int(v.max())) if len(v) else (0, 0)
        col.range_index = formats.write_range_index((v - mn).astype(np.uint64), mn, mx - mn)
    else:
        v = np.asarray(values, dtype=NUMERIC_NP[col.data_type])
        col.range_index = formats.write_range_index(formats.fp_ordinal(v), 0, 0xFFFFFFFF if col.data_type == "FLOAT" else (1 << 64) - 1)
    return col


def build_column(name: str, values, data_type: str, *, dictionary: bool = True, inverted: bool = False,
                 raw_version: int = 2, run_compress: bool = True, chunk_compression: int = 0,
                 docs_per_chunk: int = 1000) -> HostColumn:
    if data_type in ("STRING", "BYTES") and not dictionary:
        # raw var-byte column (VarByteChunkForwardIndexWriter, PASS_THROUGH): a GROUP BY key at most on this path
        blobs = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in values]
        fwd = formats.write_raw_var_byte_chunk(blobs, version=raw_version, docs_per_chunk=docs_per_chunk)
        return HostColumn(name, data_type, capi.FWD_RAW_VAR_BYTE_CHUNK, False, 0, 0, False, 0, fwd)
    if data_type == "STRING":
        vals = np.asarray(values, dtype=object)
        uq, inv = np.unique(vals.astype(str), return_inverse=True)  # Java String.compareTo == code-point order for BMP/ASCII test data
        uniq = [str(v) for v in uq.tolist()]
        dict_ids = inv.astype(np.int32)
        dict_buf, width = formats.write_string_dictionary(uniq)
        dict_values = list(uniq)
    else:
        vals = np.ascontiguousarray(values, dtype=NUMERIC_NP[data_type])
        if not dictionary:
            fwd = formats.write_raw_fixed_byte_chunk(vals, data_type, version=raw_version, docs_per_chunk=docs_per_chunk,
                                                     compression=chunk_compression)
            return HostColumn(name, data_type, capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0,
                              bool(np.all(vals[1:] >= vals[:-1])) if len(vals) else True, 0, fwd)
        uniq, dict_ids = np.unique(vals, return_inverse=True)
        dict_ids = dict_ids.astype(np.int32)
        dict_buf = formats.write_numeric_dictionary(uniq, data_type)
        width = formats._WIDTHS[data_type]
        dict_values = uniq.tolist()
    n = len(dict_ids)
    card = len(dict_values)
    is_sorted = bool(np.all(dict_ids[1:] >= dict_ids[:-1])) if n else True
    bits = formats.num_bits_per_value(card - 1)
    if is_sorted:
        # SingleValueSortedForwardIndexCreator: the sorted index is forward and inverted index at once
        fwd = formats.write_sorted_index(dict_ids, card)
        enc = capi.FWD_DICT_SORTED
        inv = None
    else:
        fwd = formats.pack_fixed_bit(dict_ids, bits)
        enc = capi.FWD_DICT_FIXED_BIT
        inv = formats.write_inverted_index(dict_ids, card, run_compress) if inverted else None
    return HostColumn(name, data_type, enc, True, card, bits, is_sorted, width, fwd, dict_buf, inv, dict_values)


def build_raw_mv_column(name: str, rows: Sequence[Sequence], data_type: str, *, compression: int = 0, version: int = 2) -> HostColumn:
    """A raw (no-dictionary) multi-value column of INT / LONG / FLOAT / DOUBLE (MultiValueFixedByteRawIndexCreator); an empty row takes
    the default null value like the segment creator does.  `dict_values` (sorted distinct values) serves the CHECK side only: the oracle
    reports group keys of such a column as ids of its internal dictionary."""
    if data_type == "STRING":   # VarByteChunkMVForwardIndexReader (MultiValueVarByteRawIndexCreator); the default null value is "null"
        rows = [[str(v) for v in r] if len(r) else ["null"] for r in rows]
        fwd = formats.write_raw_mv_var_byte_chunk(rows, version=version)
        flat_s = [v for r in rows for v in r]
        col = HostColumn(name, data_type, capi.FWD_RAW_MV_VAR_BYTE_CHUNK, False, 0, 0, False, 0, fwd, None, None, sorted(set(flat_s)))
        col.total_number_of_entries = len(flat_s)
        return col
    default = {"INT": -(1 << 31), "LONG": -(1 << 63), "FLOAT": float("-inf"), "DOUBLE": float("-inf")}[data_type]
    rows = [list(r) if len(r) else [default] for r in rows]
    fwd = formats.write_raw_mv_fixed_byte_chunk(rows, data_type, version=version, compression=compression)
    flat = np.ascontiguousarray([v for r in rows for v in r], dtype=NUMERIC_NP[data_type])
    col = HostColumn(name, data_type, capi.FWD_RAW_MV_FIXED_BYTE_CHUNK, False, 0, 0, False, 0, fwd, None, None, np.unique(flat).tolist())
    col.total_number_of_entries = int(len(flat))
    return col


def build_mv_column(name: str, rows: Sequence[Sequence], data_type: str, *, inverted: bool = False, run_compress: bool = True) -> HostColumn:
    """A multi-value dictionary column (MultiValueUnsortedForwardIndexCreator + BitmapInvertedIndexWriter): `rows[doc]` is the doc's
    values, duplicates and order kept; an empty row takes the default null value like the segment creator does
    (FieldSpec#getDefaultNullValue: Integer.MIN_VALUE / Long.MIN_VALUE / -inf / "null")."""
    default = {"INT": -(1 << 31), "LONG": -(1 << 63), "FLOAT": float("-inf"), "DOUBLE": float("-inf"), "STRING": "null"}[data_type]
    rows = [list(r) if len(r) else [default] for r in rows]
    lengths = np.array([len(r) for r in rows], dtype=np.int64)
    flat = [v for r in rows for v in r]
    if data_type == "STRING":
        uq, inv = np.unique(np.asarray(flat, dtype=object).astype(str), return_inverse=True)
        dict_values = [str(v) for v in uq.tolist()]
        dict_buf, width = formats.write_string_dictionary(dict_values)
    else:
        uq, inv = np.unique(np.ascontiguousarray(flat, dtype=NUMERIC_NP[data_type]), return_inverse=True)
        dict_values = uq.tolist()
        dict_buf, width = formats.write_numeric_dictionary(uq, data_type), formats._WIDTHS[data_type]
    dict_ids = inv.astype(np.int32)
    card = len(dict_values)
    bits = formats.num_bits_per_value(card - 1)
    fwd = formats.write_fixed_bit_mv(dict_ids, lengths, bits)
    inv_buf = formats.write_inverted_index_mv(dict_ids, np.repeat(np.arange(len(rows)), lengths), card, run_compress) if inverted else None
    col = HostColumn(name, data_type, capi.FWD_DICT_FIXED_BIT_MV, True, card, bits, False, width, fwd, dict_buf, inv_buf, dict_values)
    col.total_number_of_entries = int(lengths.sum())
    return col


def decode_mv_column(col: HostColumn, num_docs: int):
    """Check reader: per doc the list of dictIds."""
    ids, starts = formats.read_fixed_bit_mv(col.forward_index, num_docs, col.total_number_of_entries, col.bits_per_value)
    return [ids[starts[d]:starts[d + 1]] for d in range(num_docs)]


def build_segment(name: str, data: Dict[str, Sequence], schema: Dict[str, str], *,
                  inverted_index_columns: Iterable[str] = (), no_dictionary_columns: Iterable[str] = (),
                  range_index_columns: Iterable[str] = (), raw_version: int = 2, run_compress: bool = True) -> HostSegment:
    inv = set(inverted_index_columns)
    nodict = set(no_dictionary_columns)
    rng_cols = set(range_index_columns)
    total = None
    seg = HostSegment(name, 0)
    for col, dtype in schema.items():
        vals = data[col]
        if total is None:
            total = len(vals)
        assert len(vals) == total, f"column {col}: {len(vals)} rows != {total}"
        seg.columns[col] = build_column(col, vals, dtype, dictionary=col not in nodict, inverted=col in inv,
                                        raw_version=raw_version, run_compress=run_compress)
        if col in rng_cols:
            add_range_index(seg.columns[col], vals)
    seg.total_docs = int(total or 0)
    return seg


def decode_column(col: HostColumn, num_docs: int, dict_ids: bool = False) -> np.ndarray:
    """Check reader: the per-doc dictIds (dict_ids=True) or values of a column, decoded from its index bytes."""
    if col.has_dictionary:
        if col.fwd_encoding == capi.FWD_DICT_SORTED:
            pairs = np.frombuffer(bytes(col.forward_index), dtype=">i4").reshape(-1, 2)
            ids = np.zeros(num_docs, dtype=np.int32)
            for d, (s, e) in enumerate(pairs):
                ids[s:e + 1] = d
        else:
            ids = formats.unpack_fixed_bit(col.forward_index, col.bits_per_value, num_docs)
        if dict_ids:
            return ids
        return np.asarray(col.dict_values)[ids]
    assert not dict_ids
    h = formats.parse_raw_fixed_byte_chunk_header(col.forward_index)
    return np.frombuffer(bytes(col.forward_index), dtype=formats._BE_DTYPES[col.data_type], count=num_docs,
                         offset=h["raw_data_start"]).astype(NUMERIC_NP[col.data_type])
