"""Writers (and a few check readers) for the Pinot v3 index-entry byte layouts the hot path reads.

The reference builds segments with SegmentIndexCreationDriverImpl; segment *creation* is out of scope,
but the executor consumes the exact bytes of these index entries, so tests and the benchmark need to produce them.
Every writer cites the reference writer/reader whose layout it follows.  All multi-byte fields are big-endian unless
noted (RoaringBitmap payloads are little-endian inside the big-endian file).
"""
from __future__ import annotations

import struct
from typing import Iterable, List, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------------------------------------
# Fixed-bit forward index (dictIds)
#   writer: pinot-segment-local/.../io/writer/impl/FixedBitSVForwardIndexWriter.java:42-45  (length = ceil(N*bits/8))
#   bit order: pinot-segment-local/.../io/util/PinotDataBitSet.java:74-97 (MSB first inside a big-endian byte stream)
# ----------------------------------------------------------------------------------------------------------------------


def num_bits_per_value(max_value: int) -> int:
    """PinotDataBitSet.getNumBitsPerValue (pinot-segment-local/.../io/util/PinotDataBitSet.java:61-72)."""
    if max_value <= 1:
        return 1
    return int(max_value).bit_length()


def pack_fixed_bit(values: np.ndarray, bits: int) -> np.ndarray:
    """Packs non-negative ints into the MSB-first bit stream. Returns uint8 array of ceil(N*bits/8) bytes."""
    assert 1 <= bits <= 31
    values = np.ascontiguousarray(values, dtype=np.uint32)
    n = values.shape[0]
    total_bytes = (n * bits + 7) // 8
    out = np.zeros(total_bytes, dtype=np.uint8)
    if n == 0:
        return out
    chunk = 1 << 21  # multiple of 8 => every chunk starts on a byte boundary
    for start in range(0, n, chunk):
        v = values[start:start + chunk]
        # the 32 bits of every value MSB first, of which the low `bits` are kept
        b = np.unpackbits(v.astype(">u4").view(np.uint8).reshape(-1, 4), axis=1)[:, 32 - bits:].reshape(-1)
        packed = np.packbits(b)  # big-endian bit order within bytes
        off = (start * bits) // 8
        out[off:off + packed.shape[0]] = packed
    return out


def unpack_fixed_bit(buf: np.ndarray, bits: int, n: int) -> np.ndarray:
    """Check reader (PinotDataBitSet.readInt semantics) used by tests only."""
    b = np.unpackbits(np.ascontiguousarray(buf, dtype=np.uint8))[: n * bits].reshape(n, bits)
    weights = (1 << np.arange(bits - 1, -1, -1, dtype=np.int64))
    return (b.astype(np.int64) * weights).sum(axis=1).astype(np.int32)


# ----------------------------------------------------------------------------------------------------------------------
# Raw fixed-byte chunk forward index, PASS_THROUGH
#   writer: pinot-segment-local/.../io/writer/impl/BaseChunkForwardIndexWriter.java:131-165 (7-int header, offsets)
#           FixedByteChunkForwardIndexWriter.java:49-101
#   reader: .../readers/forward/BaseChunkForwardIndexReader.java:61-111, FixedByteChunkSVForwardIndexReader.java:53-94
# ----------------------------------------------------------------------------------------------------------------------
CHUNK_COMPRESSION_PASS_THROUGH = 0  # ChunkCompressionType.PASS_THROUGH.getValue()
CHUNK_COMPRESSION_SNAPPY, CHUNK_COMPRESSION_LZ4, CHUNK_COMPRESSION_LZ4_LENGTH_PREFIXED = 1, 3, 4
CHUNK_COMPRESSION_ZSTANDARD, CHUNK_COMPRESSION_GZIP = 2, 5


def compress_chunk(chunk: bytes, compression: int) -> bytes:
    """Test writer for ChunkCompressor#compress: libsnappy / liblz4 through pyarrow (raw snappy, raw LZ4 block; the
    LZ4_LENGTH_PREFIXED form of lz4-java's LZ4CompressorWithLength puts the little-endian decompressed length in front)."""
    if compression == CHUNK_COMPRESSION_PASS_THROUGH:
        return chunk
    import pyarrow as pa
    if compression == CHUNK_COMPRESSION_SNAPPY:
        return pa.compress(chunk, codec="snappy", asbytes=True)
    if compression == CHUNK_COMPRESSION_LZ4:
        return pa.compress(chunk, codec="lz4_raw", asbytes=True)
    if compression == CHUNK_COMPRESSION_LZ4_LENGTH_PREFIXED:
        return struct.pack("<i", len(chunk)) + pa.compress(chunk, codec="lz4_raw", asbytes=True)
    if compression == CHUNK_COMPRESSION_ZSTANDARD:   # ZstandardCompressor: one zstd frame per chunk
        return pa.compress(chunk, codec="zstd", asbytes=True)
    if compression == CHUNK_COMPRESSION_GZIP:        # GzipCompressor: java.util.zip.Deflater (zlib stream) + big-endian uncompressed length
        import zlib
        return zlib.compress(bytes(chunk)) + struct.pack(">i", len(chunk))
    raise ValueError(f"chunk compression type {compression}")

_BE_DTYPES = {"INT": ">i4", "LONG": ">i8", "FLOAT": ">f4", "DOUBLE": ">f8"}
_WIDTHS = {"INT": 4, "LONG": 8, "FLOAT": 4, "DOUBLE": 8}


def write_raw_fixed_byte_chunk(values: np.ndarray, data_type: str, version: int = 2,
                               docs_per_chunk: int = 1000, compression: int = CHUNK_COMPRESSION_PASS_THROUGH) -> np.ndarray:
    """Header(version, numChunks, numDocsPerChunk, sizeOfEntry, totalDocs, compressionType, dataHeaderStart=28),
    chunk start offsets (int for v2, long for v3+), then the big-endian values back to back — or, with a compression type,
    every chunk compressed on its own (BaseChunkForwardIndexWriter#writeChunk :179-198)."""
    assert version in (2, 3, 4)
    width = _WIDTHS[data_type]
    n = int(values.shape[0])
    if compression != CHUNK_COMPRESSION_PASS_THROUGH:
        assert version in (2, 3)
        num_chunks = (n + docs_per_chunk - 1) // docs_per_chunk
        off_size = 4 if version == 2 else 8
        data = np.ascontiguousarray(values).astype(_BE_DTYPES[data_type]).tobytes()
        step = docs_per_chunk * width
        chunks = [compress_chunk(data[i * step:(i + 1) * step], compression) for i in range(num_chunks)]
        pos = 7 * 4 + num_chunks * off_size
        starts = []
        for ch in chunks:
            starts.append(pos)
            pos += len(ch)
        header = struct.pack(">7i", version, num_chunks, docs_per_chunk, width, n, compression, 28)
        off_bytes = np.asarray(starts, dtype=np.int64).astype(">i4" if off_size == 4 else ">i8").tobytes()
        return np.frombuffer(header + off_bytes + b"".join(chunks), dtype=np.uint8)
    if version >= 4 and (docs_per_chunk & (docs_per_chunk - 1)) != 0:
        docs_per_chunk = 1 << (docs_per_chunk - 1).bit_length()  # normalizeDocsPerChunk
    num_chunks = (n + docs_per_chunk - 1) // docs_per_chunk
    off_size = 4 if version == 2 else 8
    header_size = 7 * 4 + num_chunks * off_size
    chunk_bytes = docs_per_chunk * width
    header = struct.pack(">7i", version, num_chunks, docs_per_chunk, width, n, CHUNK_COMPRESSION_PASS_THROUGH, 28)
    offs = header_size + np.arange(num_chunks, dtype=np.int64) * chunk_bytes
    if off_size == 4:
        assert num_chunks == 0 or offs[-1] <= 0x7FFFFFFF, "Integer overflow detected (use raw version 3 or 4)"
        off_bytes = offs.astype(">i4").tobytes()
    else:
        off_bytes = offs.astype(">i8").tobytes()
    data = np.ascontiguousarray(values).astype(_BE_DTYPES[data_type]).tobytes()
    out = np.frombuffer(header + off_bytes + data, dtype=np.uint8)
    return out


def parse_raw_fixed_byte_chunk_header(buf: np.ndarray) -> dict:
    version, num_chunks, docs_per_chunk, size_of_entry = struct.unpack(">4i", bytes(buf[:16]))
    if version > 1:
        total_docs, compression, data_header_start = struct.unpack(">3i", bytes(buf[16:28]))
    else:
        total_docs, compression, data_header_start = -1, 1, 16
    off_size = 4 if version <= 2 else 8
    raw_data_start = data_header_start + num_chunks * off_size
    return dict(version=version, num_chunks=num_chunks, docs_per_chunk=docs_per_chunk, size_of_entry=size_of_entry,
                total_docs=total_docs, compression=compression, data_header_start=data_header_start,
                raw_data_start=raw_data_start)


# ----------------------------------------------------------------------------------------------------------------------
# Dictionaries: sorted big-endian fixed-width values
#   reader: pinot-segment-local/.../segment/index/readers/BaseImmutableDictionary.java:45-58, IntDictionary.java:28-80
#   STRING: fixed-width entries padded with 0 bytes (FixedByteValueReaderWriter#getUnpaddedString strips them)
# ----------------------------------------------------------------------------------------------------------------------


def write_numeric_dictionary(sorted_values: np.ndarray, data_type: str) -> np.ndarray:
    return np.frombuffer(np.ascontiguousarray(sorted_values).astype(_BE_DTYPES[data_type]).tobytes(), dtype=np.uint8)


def write_string_dictionary(sorted_values: Sequence[str]) -> Tuple[np.ndarray, int]:
    enc = [s.encode("utf-8") for s in sorted_values]
    width = max([len(e) for e in enc] + [1])
    out = np.zeros((len(enc), width), dtype=np.uint8)
    for i, e in enumerate(enc):
        out[i, : len(e)] = np.frombuffer(e, dtype=np.uint8)
    return out.reshape(-1), width


# ----------------------------------------------------------------------------------------------------------------------
# Sorted forward index: (startDocId, endDocId) inclusive pairs, 2 BE ints per dictId
#   reader: pinot-segment-local/.../segment/index/readers/sorted/SortedIndexReaderImpl.java:35-60
# ----------------------------------------------------------------------------------------------------------------------


def write_sorted_index(dict_ids: np.ndarray, cardinality: int) -> np.ndarray:
    n = dict_ids.shape[0]
    starts = np.searchsorted(dict_ids, np.arange(cardinality), side="left")
    ends = np.searchsorted(dict_ids, np.arange(cardinality), side="right") - 1
    pairs = np.stack([starts, ends], axis=1).astype(">i4")
    assert n == 0 or (ends[-1] == n - 1 and starts[0] == 0)
    return np.frombuffer(pairs.tobytes(), dtype=np.uint8)


# ----------------------------------------------------------------------------------------------------------------------
# RoaringBitmap portable serialization (RoaringBitmap 1.3.0, pom.xml:798-802; source not in the reference tree).
# Public format spec: cookie 12346 (no run containers) | 12347 (+ (size-1)<<16, run-flag bitset); descriptive header
# (key u16, cardinality-1 u16); offset header (always for 12346, only when size >= 4 for 12347); containers:
# array = sorted u16, bitmap = 1024 LE u64, run = n_runs u16 then (start u16, length-1 u16) pairs.
# Container choice follows RoaringBitmapWriter.writer() defaults used by the inverted index creators
# (OnHeapBitmapInvertedIndexCreator.java:41-42): array if cardinality <= 4096 else bitmap, then runOptimize():
# run container iff 2 + 4*n_runs < current serialized size.
# ----------------------------------------------------------------------------------------------------------------------
SERIAL_COOKIE_NO_RUNCONTAINER = 12346
SERIAL_COOKIE = 12347
NO_OFFSET_THRESHOLD = 4
ARRAY_MAX = 4096


def _container_payload(lows: np.ndarray, run_compress: bool) -> Tuple[int, bytes]:
    """lows: sorted unique uint16 values of one container. Returns (kind, bytes): kind 0=array 1=bitmap 2=run."""
    card = int(lows.shape[0])
    l32 = lows.astype(np.int32)
    if card > 1:
        breaks = np.flatnonzero(np.diff(l32) != 1)
        n_runs = int(breaks.shape[0]) + 1
    else:
        breaks = np.zeros(0, dtype=np.int64)
        n_runs = 1
    size_now = 2 * card if card <= ARRAY_MAX else 8192
    if run_compress and 2 + 4 * n_runs < size_now:
        starts = np.concatenate([[0], breaks + 1])
        ends = np.concatenate([breaks, [card - 1]])
        rs = l32[starts]
        rl = l32[ends] - rs
        pairs = np.stack([rs, rl], axis=1).astype("<u2")
        return 2, struct.pack("<H", n_runs) + pairs.tobytes()
    if card <= ARRAY_MAX:
        return 0, lows.astype("<u2").tobytes()
    bits = np.zeros(65536, dtype=np.uint8)
    bits[l32] = 1
    words = np.packbits(bits, bitorder="little").view("<u8")
    return 1, words.tobytes()


def serialize_roaring(doc_ids: np.ndarray, run_compress: bool = True) -> bytes:
    """doc_ids: sorted unique non-negative int docIds."""
    doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int64)
    if doc_ids.shape[0] == 0:
        return struct.pack("<II", SERIAL_COOKIE_NO_RUNCONTAINER, 0)
    highs = (doc_ids >> 16).astype(np.int64)
    keys, starts = np.unique(highs, return_index=True)
    bounds = list(starts) + [doc_ids.shape[0]]
    kinds: List[int] = []
    payloads: List[bytes] = []
    cards: List[int] = []
    for i in range(len(keys)):
        lows = (doc_ids[bounds[i]:bounds[i + 1]] & 0xFFFF).astype(np.uint16)
        kind, payload = _container_payload(lows, run_compress)
        kinds.append(kind)
        payloads.append(payload)
        cards.append(int(lows.shape[0]))
    size = len(keys)
    has_run = any(k == 2 for k in kinds)
    parts: List[bytes] = []
    if has_run:
        parts.append(struct.pack("<I", SERIAL_COOKIE | ((size - 1) << 16)))
        flags = np.zeros((size + 7) // 8, dtype=np.uint8)
        for i, k in enumerate(kinds):
            if k == 2:
                flags[i >> 3] |= 1 << (i & 7)
        parts.append(flags.tobytes())
    else:
        parts.append(struct.pack("<II", SERIAL_COOKIE_NO_RUNCONTAINER, size))
    desc = np.empty((size, 2), dtype="<u2")
    desc[:, 0] = keys.astype(np.uint16)
    desc[:, 1] = (np.asarray(cards, dtype=np.int64) - 1).astype(np.uint16)
    parts.append(desc.tobytes())
    header_len = sum(len(p) for p in parts)
    with_offsets = (not has_run) or size >= NO_OFFSET_THRESHOLD
    if with_offsets:
        header_len += 4 * size
        offs = np.empty(size, dtype="<u4")
        pos = header_len
        for i, p in enumerate(payloads):
            offs[i] = pos
            pos += len(p)
        parts.append(offs.tobytes())
    parts.extend(payloads)
    return b"".join(parts)


def deserialize_roaring(blob: bytes) -> np.ndarray:
    """Check reader (tests only): returns sorted docIds."""
    mv = memoryview(blob)
    (cookie,) = struct.unpack_from("<I", mv, 0)
    pos = 4
    if (cookie & 0xFFFF) == SERIAL_COOKIE:
        size = (cookie >> 16) + 1
        flags = np.frombuffer(mv[pos:pos + (size + 7) // 8], dtype=np.uint8)
        pos += (size + 7) // 8
        has_run = True
    else:
        assert cookie == SERIAL_COOKIE_NO_RUNCONTAINER, cookie
        (size,) = struct.unpack_from("<I", mv, pos)
        pos += 4
        flags = None
        has_run = False
    desc = np.frombuffer(mv[pos:pos + 4 * size], dtype="<u2").reshape(size, 2)
    pos += 4 * size
    if (not has_run) or size >= NO_OFFSET_THRESHOLD:
        pos += 4 * size
    out = []
    for i in range(size):
        key = int(desc[i, 0])
        card = int(desc[i, 1]) + 1
        is_run = has_run and ((flags[i >> 3] >> (i & 7)) & 1)
        if is_run:
            (n_runs,) = struct.unpack_from("<H", mv, pos)
            pos += 2
            pairs = np.frombuffer(mv[pos:pos + 4 * n_runs], dtype="<u2").reshape(n_runs, 2).astype(np.int64)
            pos += 4 * n_runs
            vals = np.concatenate([np.arange(s, s + l + 1) for s, l in pairs]) if n_runs else np.zeros(0, np.int64)
        elif card > ARRAY_MAX:
            words = np.frombuffer(mv[pos:pos + 8192], dtype=np.uint8)
            pos += 8192
            vals = np.flatnonzero(np.unpackbits(words, bitorder="little")).astype(np.int64)
        else:
            vals = np.frombuffer(mv[pos:pos + 2 * card], dtype="<u2").astype(np.int64)
            pos += 2 * card
        out.append(vals + (key << 16))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------------
# Bitmap inverted index: (cardinality+1) BE uint32 absolute offsets then the serialized bitmaps
#   writer: pinot-segment-local/.../segment/creator/impl/inv/BitmapInvertedIndexWriter.java:36-104
#   reader: pinot-segment-local/.../segment/index/readers/BitmapInvertedIndexReader.java:45-62
# ----------------------------------------------------------------------------------------------------------------------


# ----------------------------------------------------------------------------------------------------------------------
# Bit-sliced range index: BitSlicedRangeIndexCreator (pinot-segment-local/.../creator/impl/inv/BitSlicedRangeIndexCreator.java
# :38-133) = big-endian {int version 2, long min} + a RoaringBitmap `RangeBitmap` (third-party, little-endian; format restated
# from the published RoaringBitmap 1.3.0 source, no fixture of it exists in the reference tree):
#   u16 cookie 0xF00D, u8 base 2, u8 sliceCount, u16 maxKey (2^16-row chunks), u32 maxRid (rows), maxKey x mask[(sliceCount+7)/8],
#   then per chunk and per slice present in its mask: u8 type (0 bitmap / 1 run / 2 array) + the container
#   (bitmap: 8192 B; run: u16 n + n x (u16 start, u16 length-1); array: u16 n + n x u16).
# Slice i holds the rows whose value has bit i CLEAR.  Values: dictIds (dictionary columns), value - min (raw INT / LONG),
# FPOrdering.ordinalOf (FLOAT / DOUBLE).
# ----------------------------------------------------------------------------------------------------------------------
def fp_ordinal(values: np.ndarray) -> np.ndarray:
    """FPOrdering.ordinalOf for float32 / float64 arrays -> uint64 with the same total (unsigned) order."""
    if values.dtype == np.float32:
        bits = values.view(np.uint32).astype(np.uint64)
        sign, top, allm = np.uint64(0x80000000), np.uint64(0x80000000), np.uint64(0xFFFFFFFF)
    else:
        bits = values.astype(np.float64).view(np.uint64)
        sign, top, allm = np.uint64(1 << 63), np.uint64(1 << 63), np.uint64(0xFFFFFFFFFFFFFFFF)
    neg = (bits & sign) != 0
    out = np.where(neg, np.where(bits == top, top, (~bits) & allm), bits ^ sign)
    out = np.where(np.isposinf(values), allm, out)
    out = np.where(np.isneginf(values) | np.isnan(values), np.uint64(0), out)
    return out.astype(np.uint64)


def write_range_index(stored: np.ndarray, min_value: int, max_stored: int) -> np.ndarray:
    """`stored`: uint64 per-row values as the appender receives them (already minus min / ordinals); max_stored: the appender's
    maxValue (cardinality - 1, max - min, 0xFFFFFFFF or 2^64 - 1)."""
    stored = np.ascontiguousarray(stored, dtype=np.uint64)
    n = int(stored.shape[0])
    slice_count = max(1, int(max_stored).bit_length())       # 64 - numberOfLeadingZeros(maxValue | 1)
    n_keys = (n + 65535) >> 16
    bytes_per_mask = (slice_count + 7) >> 3
    masks = bytearray()
    containers = bytearray()
    for key in range(n_keys):
        chunk = stored[key << 16:(key + 1) << 16]
        mask = 0
        for i in range(slice_count):
            rows = np.flatnonzero(((chunk >> np.uint64(i)) & np.uint64(1)) == 0).astype(np.uint16)
            if rows.shape[0] == 0:
                continue
            mask |= 1 << i
            kind, payload = _container_payload(rows, True)        # Container#runOptimize picks the smallest form
            if kind == 1:
                containers += b"\x00" + payload
            elif kind == 2:
                containers += b"\x01" + payload                   # payload starts with the u16 run count
            else:
                containers += b"\x02" + struct.pack("<H", int(rows.shape[0])) + payload
        masks += int(mask).to_bytes(bytes_per_mask, "little")
    blob = struct.pack(">iq", 2, int(min_value)) + struct.pack("<HBBHI", 0xF00D, 2, slice_count, n_keys, n) + bytes(masks) + bytes(containers)
    return np.frombuffer(blob, dtype=np.uint8).copy()


# ----------------------------------------------------------------------------------------------------------------------
# Multi-value dictionary forward index
#   writer: pinot-segment-local/.../io/writer/impl/FixedBitMVForwardIndexWriter.java:76-157 (chunk offset header | row-start bitmap |
#           bit-packed dictIds; docsPerChunk = ceil(2048 / (float)(totalNumValues / numDocs)), the division an integer one)
#   reader: .../readers/forward/FixedBitMVForwardIndexReader.java:57-76
# ----------------------------------------------------------------------------------------------------------------------


def mv_docs_per_chunk(num_docs: int, total_values: int) -> int:
    avg = total_values // num_docs   # int / int in both the writer and the reader
    return int(np.ceil(np.float32(2048) / np.float32(avg)))


def write_fixed_bit_mv(dict_ids: np.ndarray, lengths: np.ndarray, bits: int) -> np.ndarray:
    """`dict_ids`: the dictIds of all docs back to back; `lengths`: values per doc (>= 1 each, as the segment creator guarantees)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n, total = len(lengths), int(lengths.sum())
    assert n > 0 and total == len(dict_ids) and int(lengths.min()) >= 1
    per_chunk = mv_docs_per_chunk(n, total)
    starts = np.zeros(n, dtype=np.int64)
    np.cumsum(lengths[:-1], out=starts[1:])
    header = starts[::per_chunk].astype(">i4").tobytes()
    assert len(header) == 4 * ((n + per_chunk - 1) // per_chunk)
    bitmap = np.zeros((total + 7) // 8 * 8, dtype=np.uint8)
    bitmap[starts] = 1
    return np.frombuffer(header + np.packbits(bitmap).tobytes()[:(total + 7) // 8] + pack_fixed_bit(dict_ids, bits).tobytes(), dtype=np.uint8)


def write_fixed_bit_mv_entry_dict(dict_ids: np.ndarray, lengths: np.ndarray, bits: int) -> np.ndarray:
    """The MV_ENTRY_DICT forward index (FixedBitMVEntryDictForwardIndexWriter.java:80-130): distinct entry lists get ids in first-seen order;
    header (magic 0xffabcdef, short version 1, byte bitsPerValue, byte bitsPerId, int uniqueEntries, int totalValues, int offsetBufferOffset,
    int valueBufferOffset), then the docs' ids, the entries' start offsets (unique + 1) and the entries' dictIds as MSB-first bit arrays."""
    lengths = np.asarray(lengths, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lengths)])
    entry_ids, entries, ids = {}, [], []
    for d in range(len(lengths)):
        e = tuple(int(v) for v in dict_ids[starts[d]:starts[d + 1]])
        if e not in entry_ids:
            entry_ids[e] = len(entries)
            entries.append(e)
        ids.append(entry_ids[e])
    n_unique, n_total = len(entries), sum(len(e) for e in entries)
    bits_id, bits_off = num_bits_per_value(n_unique - 1), num_bits_per_value(n_total)
    offs = np.concatenate([[0], np.cumsum([len(e) for e in entries])]).astype(np.int64)
    id_buf = pack_fixed_bit(np.asarray(ids), bits_id).tobytes()
    off_buf = pack_fixed_bit(offs, bits_off).tobytes()
    val_buf = pack_fixed_bit(np.asarray([v for e in entries for v in e]), bits).tobytes()
    off_at = 24 + len(id_buf)
    val_at = off_at + len(off_buf)
    header = struct.pack(">IhBBiiii", 0xFFABCDEF, 1, bits, bits_id, n_unique, n_total, off_at, val_at)
    return np.frombuffer(header + id_buf + off_buf + val_buf, dtype=np.uint8)


def read_fixed_bit_mv(buf: np.ndarray, num_docs: int, total_values: int, bits: int):
    """Check reader: (dictIds back to back, start offset per doc + total)."""
    per_chunk = mv_docs_per_chunk(num_docs, total_values)
    off = 4 * ((num_docs + per_chunk - 1) // per_chunk)
    nb = (total_values + 7) // 8
    starts = np.flatnonzero(np.unpackbits(np.asarray(buf[off:off + nb], dtype=np.uint8))[:total_values])
    assert len(starts) == num_docs
    ids = unpack_fixed_bit(np.asarray(buf[off + nb:], dtype=np.uint8), bits, total_values)
    return ids, np.append(starts, total_values).astype(np.int64)


def write_inverted_index_mv(dict_ids: np.ndarray, doc_of_value: np.ndarray, cardinality: int, run_compress: bool = True) -> np.ndarray:
    """BitmapInvertedIndexWriter over a multi-value column: per dictId the docs holding it at least once
    (OffHeapBitmapInvertedIndexCreator#add(int[] dictIds, int length): the bitmaps absorb a doc's repeated values)."""
    pairs = np.unique(np.stack([np.asarray(dict_ids, dtype=np.int64), np.asarray(doc_of_value, dtype=np.int64)], axis=1), axis=0)
    bounds = np.searchsorted(pairs[:, 0], np.arange(cardinality + 1), side="left")
    blobs = [serialize_roaring(pairs[bounds[d]:bounds[d + 1], 1], run_compress) for d in range(cardinality)]
    offsets = np.empty(cardinality + 1, dtype=np.int64)
    pos = (cardinality + 1) * 4
    for d, b in enumerate(blobs):
        offsets[d] = pos
        pos += len(b)
    offsets[cardinality] = pos
    return np.frombuffer(offsets.astype(">u4").tobytes() + b"".join(blobs), dtype=np.uint8)


def write_inverted_index(dict_ids: np.ndarray, cardinality: int, run_compress: bool = True) -> np.ndarray:
    order = np.argsort(dict_ids, kind="stable")
    sorted_ids = dict_ids[order]
    bounds = np.searchsorted(sorted_ids, np.arange(cardinality + 1), side="left")
    blobs = []
    for d in range(cardinality):
        docs = order[bounds[d]:bounds[d + 1]]  # ascending because the sort is stable
        blobs.append(serialize_roaring(docs, run_compress))
    offsets = np.empty(cardinality + 1, dtype=np.int64)
    pos = (cardinality + 1) * 4
    for d, b in enumerate(blobs):
        offsets[d] = pos
        pos += len(b)
    offsets[cardinality] = pos
    assert pos <= 0xFFFFFFFF, "inverted index larger than 4 GB (BitmapInvertedIndexReader offsets are unsigned int)"
    head = offsets.astype(">u4").tobytes()
    return np.frombuffer(head + b"".join(blobs), dtype=np.uint8)


# ----------------------------------------------------------------------------------------------------------------------
# Raw var-byte chunk forward index (BYTES / STRING), PASS_THROUGH, writer versions 2 and 3
#   writer: pinot-segment-local/.../io/writer/impl/VarByteChunkForwardIndexWriter.java:46-158 (chunk = numDocsPerChunk int
#           offsets relative to the chunk start, 0 for the absent rows of the last chunk, then the values back to back)
#           BaseChunkForwardIndexWriter.java:131-198 (same 7-int header as the fixed-byte format; sizeOfEntry = longest value)
#   reader: .../readers/forward/VarByteChunkSVForwardIndexReader.java:158-217
# ----------------------------------------------------------------------------------------------------------------------


def write_raw_var_byte_chunk(values: Sequence[bytes], version: int = 2, docs_per_chunk: int = 1000,
                             longest_entry: int = None) -> np.ndarray:
    assert version in (2, 3)
    n = len(values)
    longest = max([len(v) for v in values] + [0]) if longest_entry is None else longest_entry
    num_chunks = (n + docs_per_chunk - 1) // docs_per_chunk
    off_size = 4 if version == 2 else 8
    header_size = 7 * 4 + num_chunks * off_size
    chunks = []
    for c in range(num_chunks):
        vals = values[c * docs_per_chunk:(c + 1) * docs_per_chunk]
        offs = np.zeros(docs_per_chunk, dtype=">i4")
        pos = docs_per_chunk * 4
        for i, v in enumerate(vals):
            offs[i] = pos
            pos += len(v)
        chunks.append(offs.tobytes() + b"".join(vals))
    starts, pos = [], header_size
    for ch in chunks:
        starts.append(pos)
        pos += len(ch)
    header = struct.pack(">7i", version, num_chunks, docs_per_chunk, longest, n, CHUNK_COMPRESSION_PASS_THROUGH, 28)
    off_bytes = np.asarray(starts, dtype=np.int64).astype(">i4" if off_size == 4 else ">i8").tobytes()
    return np.frombuffer(header + off_bytes + b"".join(chunks), dtype=np.uint8)


def write_raw_mv_fixed_byte_chunk(rows: Sequence[Sequence], data_type: str, version: int = 2, docs_per_chunk: int = 1000,
                                  compression: int = 0) -> np.ndarray:
    """Raw (no-dictionary) multi-value forward index of a fixed-width type: MultiValueFixedByteRawIndexCreator ->
    VarByteChunkForwardIndexWriter#putIntMV ... (.../creator/impl/fwd/MultiValueFixedByteRawIndexCreator.java:77-84): the var-byte chunk
    layout whose value of a doc is ArraySerDeUtils.serialize…ArrayWithLength = big-endian int numValues + the values big-endian;
    lengthOfLongestEntry = 4 + maxNumberOfMultiValueElements x size.  `compression`: a ChunkCompressionType applied chunk by chunk."""
    np_t = {"INT": ">i4", "LONG": ">i8", "FLOAT": ">f4", "DOUBLE": ">f8"}[data_type]
    values = [struct.pack(">i", len(r)) + np.asarray(r, dtype=np_t).tobytes() for r in rows]
    longest = 4 + max(len(r) for r in rows) * np.dtype(np_t).itemsize
    if compression == 0:
        return write_raw_var_byte_chunk(values, version=version, docs_per_chunk=docs_per_chunk, longest_entry=longest)
    plain = bytes(write_raw_var_byte_chunk(values, version=version, docs_per_chunk=docs_per_chunk, longest_entry=longest))
    h = parse_raw_fixed_byte_chunk_header(np.frombuffer(plain, dtype=np.uint8))
    off_size = 4 if version == 2 else 8
    fmt = ">i" if off_size == 4 else ">q"
    starts = [struct.unpack_from(fmt, plain, h["data_header_start"] + i * off_size)[0] for i in range(h["num_chunks"])] + [len(plain)]
    chunks = [compress_chunk(plain[starts[i]:starts[i + 1]], compression) for i in range(h["num_chunks"])]
    header_size = 28 + h["num_chunks"] * off_size
    pos, offs = header_size, []
    for ch in chunks:
        offs.append(pos)
        pos += len(ch)
    header = struct.pack(">7i", version, h["num_chunks"], docs_per_chunk, longest, len(rows), compression, 28)
    off_bytes = np.asarray(offs, dtype=np.int64).astype(">i4" if off_size == 4 else ">i8").tobytes()
    return np.frombuffer(header + off_bytes + b"".join(chunks), dtype=np.uint8)


def write_var_length_string_dictionary(values: Sequence[str]) -> np.ndarray:
    """A variable-length STRING dictionary (SegmentDictionaryCreator with useVarLengthDictionary -> VarLengthValueWriter.java:78-130): ".vl;",
    int version 1, int numValues, int dataSectionStartOffset = 16, numValues + 1 absolute offsets, the UTF-8 values; `values` sorted."""
    enc = [v.encode("utf-8") for v in values]
    n = len(enc)
    pos = 16 + 4 * (n + 1)
    offs = []
    for e in enc:
        offs.append(pos)
        pos += len(e)
    offs.append(pos)
    return np.frombuffer(b".vl;" + struct.pack(">iii", 1, n, 16) + b"".join(struct.pack(">i", o) for o in offs) + b"".join(enc), dtype=np.uint8)


def write_raw_mv_var_byte_chunk(rows: Sequence[Sequence[str]], version: int = 2, docs_per_chunk: int = 1000) -> np.ndarray:
    """Raw (no-dictionary) multi-value STRING forward index: MultiValueVarByteRawIndexCreator -> VarByteChunkForwardIndexWriter#putStringMV: the
    var-byte chunk layout whose value of a doc is ArraySerDeUtils.serializeStringArray (.../utils/ArraySerDeUtils.java:282-292) = int numValues,
    numValues int lengths, the UTF-8 bytes."""
    values = []
    for r in rows:
        enc = [v.encode("utf-8") for v in r]
        values.append(struct.pack(">i", len(enc)) + b"".join(struct.pack(">i", len(e)) for e in enc) + b"".join(enc))
    return write_raw_var_byte_chunk(values, version=version, docs_per_chunk=docs_per_chunk, longest_entry=max(len(v) for v in values))


def read_raw_var_byte_chunk(buf: np.ndarray) -> List[bytes]:
    """Check reader: VarByteChunkSVForwardIndexReader#getBytesUncompressed for every docId."""
    h = parse_raw_fixed_byte_chunk_header(buf)
    assert h["compression"] == CHUNK_COMPRESSION_PASS_THROUGH
    b = bytes(buf)
    off_size = 4 if h["version"] <= 2 else 8
    fmt = ">i" if off_size == 4 else ">q"
    starts = [struct.unpack_from(fmt, b, h["data_header_start"] + i * off_size)[0] for i in range(h["num_chunks"])]
    out = []
    dpc = h["docs_per_chunk"]
    for doc in range(h["total_docs"]):
        c, r = divmod(doc, dpc)
        cs = starts[c]
        s = cs + struct.unpack_from(">i", b, cs + 4 * r)[0]
        chunk_end = starts[c + 1] if c + 1 < h["num_chunks"] else len(b)
        if r == dpc - 1:
            e = chunk_end
        else:
            nxt = struct.unpack_from(">i", b, cs + 4 * (r + 1))[0]
            e = chunk_end if nxt == 0 else cs + nxt
        out.append(b[s:e])
    return out


# ----------------------------------------------------------------------------------------------------------------------
# Serialized HyperLogLog (stream-lib 2.9.8 HyperLogLog#getBytes, wrapped by ObjectSerDeUtils.HYPER_LOG_LOG_SER_DE —
# pinot-core/.../common/ObjectSerDeUtils.java:733-767): BE int log2m, BE int 4*words, then the RegisterSet words (BE ints),
# 6 five-bit registers per word, register i at bits 5*(i%6) of word i/6.  SURVEY.md §9; header pinned by
# pinot-core/src/test/resources/data/rawhllresults.txt (00000008 000000ac).
# ----------------------------------------------------------------------------------------------------------------------


def serialize_hll(registers: np.ndarray, log2m: int) -> bytes:
    m = 1 << log2m
    regs = np.asarray(registers, dtype=np.uint32)
    assert regs.shape[0] == m
    n_words = m // 6 + (0 if m % 6 == 0 else 1)      # RegisterSet.getSizeForCount
    padded = np.zeros(n_words * 6, dtype=np.uint32)
    padded[:m] = regs
    words = np.zeros(n_words, dtype=np.uint32)
    for k in range(6):
        words |= padded[k::6] << np.uint32(5 * k)
    return struct.pack(">ii", log2m, n_words * 4) + words.astype(">u4").tobytes()


def deserialize_hll(blob: bytes) -> Tuple[int, np.ndarray]:
    log2m, nbytes = struct.unpack_from(">ii", blob, 0)
    words = np.frombuffer(blob, dtype=">u4", count=nbytes // 4, offset=8).astype(np.uint32)
    m = 1 << log2m
    regs = np.zeros(len(words) * 6, dtype=np.uint8)
    for k in range(6):
        regs[k::6] = (words >> np.uint32(5 * k)) & 0x1F
    return log2m, regs[:m]


# ----------------------------------------------------------------------------------------------------------------------
# Star-tree index file (`star_tree.index`), LITTLE-endian ("Backward-compatible: star-tree file is always little-endian",
#   pinot-segment-local/.../startree/StarTreeBuilderUtils.java:114-250; reader OffHeapStarTree.java:38-85,
#   OffHeapStarTreeNode.java:30-37).  Nodes are written in BFS order, children sorted by dimension value (star = -1 first).
# ----------------------------------------------------------------------------------------------------------------------
STAR_TREE_MAGIC = 0xBADDA55B00DAD00D
STAR_TREE_VERSION = 1
STAR_NODE_FIELDS = ("dimension_id", "dimension_value", "start_doc_id", "end_doc_id", "aggregated_doc_id",
                    "first_child_id", "last_child_id")


def write_star_tree(dimensions: Sequence[str], nodes: np.ndarray) -> np.ndarray:
    """`nodes`: int32 [numNodes, 7] in BFS order (STAR_NODE_FIELDS)."""
    nodes = np.ascontiguousarray(nodes, dtype="<i4")
    header_size = 20 + sum(8 + len(d.encode("utf-8")) for d in dimensions) + 4
    parts = [struct.pack("<Qiii", STAR_TREE_MAGIC, STAR_TREE_VERSION, header_size, len(dimensions))]
    for i, d in enumerate(dimensions):
        e = d.encode("utf-8")
        parts.append(struct.pack("<ii", i, len(e)) + e)
    parts.append(struct.pack("<i", nodes.shape[0]))
    assert sum(len(p) for p in parts) == header_size
    return np.frombuffer(b"".join(parts) + nodes.tobytes(), dtype=np.uint8)


def read_star_tree(buf) -> Tuple[List[str], np.ndarray]:
    b = bytes(buf)
    magic, version, header_size, n_dims = struct.unpack_from("<Qiii", b, 0)
    if magic != STAR_TREE_MAGIC:
        raise ValueError("Invalid magic marker in star-tree data buffer")
    if version != STAR_TREE_VERSION:
        raise ValueError("Invalid version in star-tree data buffer")
    off = 20
    names = [None] * n_dims
    for _ in range(n_dims):
        dim_id, n = struct.unpack_from("<ii", b, off)
        off += 8
        names[dim_id] = b[off:off + n].decode("utf-8")
        off += n
    n_nodes, = struct.unpack_from("<i", b, off)
    off += 4
    if off != header_size:
        raise ValueError("Error loading star-tree, header length mis-match")
    if off + n_nodes * 28 != len(b):
        raise ValueError("Error loading star-tree, buffer size mis-match")
    nodes = np.frombuffer(b, dtype="<i4", count=n_nodes * 7, offset=off).astype(np.int32).reshape(n_nodes, 7)
    return names, nodes
