"""Reader (and a test writer) for Pinot's v3 segment directory: `metadata.properties`, `index_map`, `columns.psf`, and the
star-tree files `star_tree_index` / `star_tree_index_map`.

  columns.psf   every index entry = 8-byte magic 0xdeadbeefdeafbead + the index bytes, big-endian file; located by
                `<column>.<index>.startOffset` / `.size` in index_map, the size INCLUDING the marker
                (pinot-segment-local/.../segment/store/SingleFileIndexDirectory.java:72-73,170-196,285-306;
                key names pinot-segment-spi/.../V1Constants.java and ColumnIndexUtils.java)
  metadata      `column.<name>.<key>` (cardinality, dataType, bitsPerElement, lengthOfEachEntry, isSorted, hasDictionary,
                isSingleValues, ...), `segment.total.docs`, `segment.name`, `startree.v2.*`
                (V1Constants.MetadataKeys, pinot-segment-spi/.../index/startree/StarTreeV2Constants.java:41-60)
  star-tree     star_tree_index_map keys `<tree>.<column>.<INDEX_TYPE>.<OFFSET|SIZE>`, no markers inside star_tree_index
                (pinot-segment-local/.../startree/v2/store/StarTreeIndexMapUtils.java)
Reader dispatch follows ForwardIndexReaderFactory.java:74-109: dictionary + sorted → SortedIndexReaderImpl; dictionary SV →
FixedBitSVForwardIndexReaderV2; raw fixed-width SV → FixedByteChunkSVForwardIndexReader.  Multi-value columns, variable-length
dictionaries and ZSTANDARD / GZIP raw chunks are outside the hot path and are skipped (listed in `HostSegment.skipped`).

Pinned by the reference's own segment metadata (tests/golden/startree_airline/segment_meta.json: the index_map and column
metadata of a segment the reference built): every entry size follows the layouts this module assumes.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi, formats, startree
from .segment import HostColumn, HostSegment

MAGIC_MARKER = 0xDEADBEEFDEAFBEAD
INDEX_FILE, INDEX_MAP_FILE, METADATA_FILE = "columns.psf", "index_map", "metadata.properties"
STAR_TREE_INDEX_FILE, STAR_TREE_INDEX_MAP_FILE = "star_tree_index", "star_tree_index_map"
_FIXED_WIDTH = {"INT": 4, "LONG": 8, "FLOAT": 4, "DOUBLE": 8}
_STORED_TYPE = {"BOOLEAN": "INT", "TIMESTAMP": "LONG", "JSON": "STRING"}   # FieldSpec.DataType#getStoredType


def stored_type(data_type: str) -> str:
    return _STORED_TYPE.get(data_type, data_type)


def read_properties(path: str) -> Dict[str, List[str]]:
    """Commons-configuration style `key = value` lines; repeated keys accumulate (list properties)."""
    out: Dict[str, List[str]] = {}
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line.strip() or line.lstrip().startswith(("#", "!")) or "=" not in line:
                continue
            k, v = line.split("=", 1)
            out.setdefault(k.strip(), []).append(v.strip())
    return out


def parse_index_map(props: Dict[str, List[str]]) -> Dict[Tuple[str, str], Tuple[int, int]]:
    """{(column, index type): (startOffset, size)}; column names may contain dots, the last two components never do."""
    starts, sizes = {}, {}
    for k, v in props.items():
        m = re.match(r"^(.*)\.([^.]+)\.(startOffset|size)$", k)
        if not m:
            continue
        (starts if m.group(3) == "startOffset" else sizes)[(m.group(1), m.group(2))] = int(v[0])
    return {key: (starts[key], sizes[key]) for key in starts if key in sizes}


def column_metadata(props: Dict[str, List[str]]) -> Dict[str, Dict[str, str]]:
    cols: Dict[str, Dict[str, str]] = {}
    for k, v in props.items():
        if not k.startswith("column."):
            continue
        name, _, key = k[len("column."):].rpartition(".")
        cols.setdefault(name, {})[key] = v[0]
    return cols


def expected_entry_sizes(meta: Dict[str, str], total_docs: int) -> Dict[str, Optional[int]]:
    """Sizes (marker included) the layouts on the path imply for a column's dictionary / forward_index entries; None where the
    layout is outside the hot path (multi-value, raw var-length ...)."""
    out: Dict[str, Optional[int]] = {}
    card = int(meta.get("cardinality", 0))
    dt = stored_type(meta.get("dataType", ""))
    has_dict = meta.get("hasDictionary", "true") == "true"
    sv = meta.get("isSingleValues", "true") == "true"
    if has_dict:
        width = _FIXED_WIDTH.get(dt, int(meta.get("lengthOfEachEntry", 0)))
        out["dictionary"] = 8 + card * width
    if not sv:
        out["forward_index"] = None
    elif has_dict and meta.get("isSorted", "false") == "true":
        out["forward_index"] = 8 + card * 8
    elif has_dict:
        out["forward_index"] = 8 + (total_docs * int(meta["bitsPerElement"]) + 7) // 8
    else:
        out["forward_index"] = None
    return out


def _decode_dictionary(buf: np.ndarray, data_type: str, card: int, width: int, padding: str) -> list:
    if data_type in _FIXED_WIDTH:
        return np.frombuffer(bytes(buf), dtype=formats._BE_DTYPES[data_type], count=card).tolist()
    raw = bytes(buf)
    pad = padding.encode("utf-8")[:1] or b"\0"
    return [raw[i * width:(i + 1) * width].rstrip(pad).decode("utf-8") for i in range(card)]


def load_segment_v1_dir(path: str) -> HostSegment:
    """The v1 layout (one file per index; V1Constants.Indexes: `<col>.dict`, `<col>.sv.unsorted.fwd`, `<col>.sv.sorted.fwd`,
    `<col>.bitmap.inv`; ColumnIndexDirectory / FilePerIndexDirectory) — what SegmentV1V2ToV3FormatConverter packs into
    columns.psf.  The index bytes are the same as in v3, without the 8-byte marker."""
    props = read_properties(os.path.join(path, METADATA_FILE))
    total_docs = int(props["segment.total.docs"][0])
    padding = props.get("segment.padding.character", ["%"])[0]     # V1Constants.Str.LEGACY_STRING_PAD_CHAR when absent
    padding = "\0" if padding in ("\\u0000", "\\\\u0000", "") else padding
    seg = HostSegment(props.get("segment.name", ["segment"])[0], total_docs)
    seg.skipped = {}

    def entry(col: str, ext: str) -> Optional[np.ndarray]:
        f = os.path.join(path, col + ext)
        return np.fromfile(f, dtype=np.uint8) if os.path.exists(f) else None

    for name, m in column_metadata(props).items():
        dt = stored_type(m.get("dataType", ""))
        card = int(m.get("cardinality", 0))
        if m.get("isSingleValues", "true") != "true" or m.get("hasDictionary", "true") != "true":
            seg.skipped[name] = "multi-value or raw column"
            continue
        is_sorted = m.get("isSorted", "false") == "true"
        fwd = entry(name, ".sv.sorted.fwd" if is_sorted else ".sv.unsorted.fwd")
        dbuf = entry(name, ".dict")
        width = _FIXED_WIDTH.get(dt, int(m.get("lengthOfEachEntry", 0)))
        if fwd is None or dbuf is None or dbuf.size != card * width:
            seg.skipped[name] = "missing index files"
            continue
        stored = dt if dt in _FIXED_WIDTH else "STRING"
        bits = int(m.get("bitsPerElement", formats.num_bits_per_value(card - 1)))
        seg.columns[name] = HostColumn(name, stored, capi.FWD_DICT_SORTED if is_sorted else capi.FWD_DICT_FIXED_BIT, True, card, bits,
                                       is_sorted, width, fwd, dbuf, None if is_sorted else entry(name, ".bitmap.inv"),
                                       _decode_dictionary(dbuf, dt, card, width, padding))
    return seg


def load_segment_dir(path: str) -> HostSegment:
    """ImmutableSegmentLoader.load for the parts the path reads.  `path` is the segment directory or its `v3/` child."""
    if os.path.isdir(os.path.join(path, "v3")):
        path = os.path.join(path, "v3")
    props = read_properties(os.path.join(path, METADATA_FILE))
    imap = parse_index_map(read_properties(os.path.join(path, INDEX_MAP_FILE)))
    psf = np.fromfile(os.path.join(path, INDEX_FILE), dtype=np.uint8)
    total_docs = int(props["segment.total.docs"][0])
    padding = props.get("segment.padding.character", ["\\u0000"])[0]
    padding = "\0" if padding in ("\\u0000", "") else padding
    seg = HostSegment(props.get("segment.name", ["segment"])[0], total_docs)
    seg.skipped = {}

    def entry(col: str, kind: str) -> Optional[np.ndarray]:
        loc = imap.get((col, kind))
        if loc is None:
            return None
        start, size = loc
        marker, = struct.unpack_from(">Q", psf, start)
        if marker != MAGIC_MARKER:
            raise ValueError(f"Inconsistent data read. Index data file {INDEX_FILE} is possibly corrupted ({col}.{kind})")
        return psf[start + 8:start + size].copy()

    for name, m in column_metadata(props).items():
        dt = stored_type(m.get("dataType", ""))
        card = int(m.get("cardinality", 0))
        has_dict = m.get("hasDictionary", "true") == "true"
        if m.get("isSingleValues", "true") != "true":
            seg.skipped[name] = "multi-value column"
            continue
        fwd = entry(name, "forward_index")
        if fwd is None:
            seg.skipped[name] = "no forward index"
            continue
        if has_dict:
            dbuf = entry(name, "dictionary")
            width = _FIXED_WIDTH.get(dt, int(m.get("lengthOfEachEntry", 0)))
            if dbuf is None or dbuf.size != card * width:
                seg.skipped[name] = "variable-length or missing dictionary"
                continue
            stored = dt if dt in _FIXED_WIDTH else "STRING"
            is_sorted = m.get("isSorted", "false") == "true"
            bits = int(m.get("bitsPerElement", formats.num_bits_per_value(card - 1)))
            seg.columns[name] = HostColumn(
                name, stored, capi.FWD_DICT_SORTED if is_sorted else capi.FWD_DICT_FIXED_BIT, True, card, bits, is_sorted,
                width, fwd, dbuf, None if is_sorted else entry(name, "inverted_index"),
                _decode_dictionary(dbuf, dt, card, width, padding))
        else:
            if dt not in _FIXED_WIDTH:
                seg.skipped[name] = "raw variable-length column"
                continue
            h = formats.parse_raw_fixed_byte_chunk_header(fwd)
            if h["compression"] not in (formats.CHUNK_COMPRESSION_PASS_THROUGH, formats.CHUNK_COMPRESSION_SNAPPY,
                                        formats.CHUNK_COMPRESSION_LZ4, formats.CHUNK_COMPRESSION_LZ4_LENGTH_PREFIXED):
                seg.skipped[name] = f"compressed raw chunks (type {h['compression']})"
                continue
            seg.columns[name] = HostColumn(name, dt, capi.FWD_RAW_FIXED_BYTE_CHUNK, False, 0, 0,
                                           m.get("isSorted", "false") == "true", 0, fwd)

    for name, col in seg.columns.items():   # StandardIndexes.NULL_VALUE_VECTOR_ID: one RoaringBitmap of the null docIds
        col.null_vector = entry(name, "nullvalue_vector")
        ri = entry(name, "range_index")   # StandardIndexes.RANGE_ID; only the exact bit-sliced index (BitSlicedRangeIndexCreator.VERSION = 2)
        if ri is not None and len(ri) >= 4 and int.from_bytes(bytes(ri[:4]), "big") == 2:
            col.range_index = ri

    n_trees = int(props.get("startree.v2.count", ["0"])[0])
    if n_trees and os.path.exists(os.path.join(path, STAR_TREE_INDEX_FILE)):
        blob = np.fromfile(os.path.join(path, STAR_TREE_INDEX_FILE), dtype=np.uint8)
        smap = read_properties(os.path.join(path, STAR_TREE_INDEX_MAP_FILE))
        for t in range(n_trees):
            def sentry(col, kind):
                off, size = int(smap[f"{t}.{col}.{kind}.OFFSET"][0]), int(smap[f"{t}.{col}.{kind}.SIZE"][0])
                return blob[off:off + size].copy()
            pre = f"startree.v2.{t}."
            dims = props[pre + "split.order"]
            if any(d not in seg.columns for d in dims):
                continue
            pairs = []
            for pname in props[pre + "function.column.pairs"]:
                fn, col = startree.parse_pair(pname)
                pairs.append(startree.StarTreePair(fn, col, sentry(pname, "FORWARD_INDEX")))
            seg.star_trees.append(startree.HostStarTree(
                int(props[pre + "total.docs"][0]), list(dims), [sentry(d, "FORWARD_INDEX") for d in dims], pairs,
                sentry("null", "STAR_TREE"), int(props.get(pre + "max.leaf.records", ["10000"])[0])))
    return seg


def write_segment_dir(seg: HostSegment, path: str, padding: str = "\0") -> None:
    """Test writer: lays a HostSegment out as a v3 directory (SegmentV1V2ToV3FormatConverter's result), entries in column order
    dictionary → forward_index → inverted_index like SingleFileIndexDirectory appends them."""
    out = os.path.join(path, "v3")
    os.makedirs(out, exist_ok=True)
    blob = bytearray()
    lines = []
    meta = [f"segment.name = {seg.name}", f"segment.total.docs = {seg.total_docs}", "segment.index.version = v3",
            "segment.padding.character = \\u0000"]

    def put(col, kind, buf):
        if buf is None:
            return
        start = len(blob)
        blob.extend(struct.pack(">Q", MAGIC_MARKER))
        blob.extend(bytes(buf))
        lines.append(f"{col}.{kind}.startOffset = {start}")
        lines.append(f"{col}.{kind}.size = {len(blob) - start}")

    for name, c in seg.columns.items():
        put(name, "dictionary", c.dictionary if c.has_dictionary else None)
        put(name, "forward_index", c.forward_index)
        put(name, "inverted_index", c.inverted_index)
        put(name, "nullvalue_vector", c.null_vector)
        put(name, "range_index", getattr(c, "range_index", None))
        p = f"column.{name}."
        meta += [p + f"cardinality = {c.cardinality}", p + f"totalDocs = {seg.total_docs}",
                 p + f"dataType = {c.data_type}", p + f"bitsPerElement = {c.bits_per_value}",
                 p + f"lengthOfEachEntry = {c.dict_bytes_per_value if c.data_type == 'STRING' else 0}",
                 p + f"isSorted = {'true' if c.is_sorted else 'false'}",
                 p + f"hasDictionary = {'true' if c.has_dictionary else 'false'}", p + "isSingleValues = true"]
    if seg.star_trees:
        sblob, slines = bytearray(), []
        meta.append(f"startree.v2.count = {len(seg.star_trees)}")
        for t, st in enumerate(seg.star_trees):
            def sput(col, kind, buf):
                slines.append(f"{t}.{col}.{kind}.OFFSET = {len(sblob)}")
                slines.append(f"{t}.{col}.{kind}.SIZE = {len(bytes(buf))}")
                sblob.extend(bytes(buf))
            sput("null", "STAR_TREE", st.star_tree)
            for d, b in zip(st.dimensions, st.dimension_forward_indexes):
                sput(d, "FORWARD_INDEX", b)
            for p in st.pairs:
                sput(p.name, "FORWARD_INDEX", p.forward_index)
            pre = f"startree.v2.{t}."
            meta.append(pre + f"total.docs = {st.num_docs}")
            meta += [pre + f"split.order = {d}" for d in st.dimensions]
            meta += [pre + f"function.column.pairs = {p.name}" for p in st.pairs]
            meta.append(pre + f"max.leaf.records = {st.max_leaf_records}")
        with open(os.path.join(out, STAR_TREE_INDEX_FILE), "wb") as f:
            f.write(bytes(sblob))
        with open(os.path.join(out, STAR_TREE_INDEX_MAP_FILE), "w") as f:
            f.write("\n".join(slines) + "\n")
    with open(os.path.join(out, INDEX_FILE), "wb") as f:
        f.write(bytes(blob))
    with open(os.path.join(out, INDEX_MAP_FILE), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(out, METADATA_FILE), "w") as f:
        f.write("\n".join(meta) + "\n")
