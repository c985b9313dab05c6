/**
 * ImmutableSegment -> pg_segment_t: walks the segment's DataSources once (keyed by segment name + CRC), registers every
 * column's index buffers by (address, size) — the bytes are copied into HBM during the call — and spreads segments
 * round-robin over the configured GPUs (pg_segment_create_on_device: the segment -> GPU map).  Segments holding something the
 * library refuses (raw multi-value columns are skipped: a query touching one is refused; unsupported chunk codecs per column) are marked
 * "Java plan only" (handle 0).  Upsert snapshots: SegmentContext#getQueryableDocIdsSnapshot is handed over only when its
 * identity changed (pg_segment_set_queryable_doc_ids).  Release: IndexSegment#destroy -> release(segment).
 * Buffer sources per pg_column_desc field: INTEGRATION.md §3.
 */
package org.apache.pinot.gpu;

import java.util.Map;
import java.util.concurrent.ConcurrentHashMap;
import java.util.concurrent.atomic.AtomicInteger;
import org.apache.pinot.segment.spi.ColumnMetadata;
import org.apache.pinot.segment.spi.ImmutableSegment;
import org.apache.pinot.segment.spi.SegmentContext;
import org.apache.pinot.segment.spi.index.StandardIndexes;
import org.apache.pinot.segment.spi.memory.PinotDataBuffer;
import org.apache.pinot.segment.spi.store.SegmentDirectory;
import org.roaringbitmap.buffer.MutableRoaringBitmap;

public final class GpuSegmentRegistry {
  private static final class Entry {
    long _handle;
    Object _lastSnapshot;
  }

  private final int[] _devices;
  private final AtomicInteger _next = new AtomicInteger();
  private final Map<String, Entry> _entries = new ConcurrentHashMap<>();
  private final Map<Long, Integer> _deviceOfHandle = new ConcurrentHashMap<>();

  public GpuSegmentRegistry(int[] devices) {
    _devices = devices;
  }

  public long handleFor(ImmutableSegment segment, SegmentContext context) {
    String key = segment.getSegmentName() + ":" + segment.getSegmentMetadata().getCrc();
    Entry e = _entries.computeIfAbsent(key, k -> register(segment));
    if (e._handle != 0) {
      MutableRoaringBitmap snapshot = context.getQueryableDocIdsSnapshot();
      synchronized (e) {
        if (snapshot != e._lastSnapshot) {   // upsert validDocIds / queryableDocIds changed since the last query on this segment
          if (snapshot == null) {
            PinotGpu.segmentSetQueryableDocIds(e._handle, 0, 0);
          } else {
            java.nio.ByteBuffer b = java.nio.ByteBuffer.allocateDirect(snapshot.serializedSizeInBytes());
            snapshot.serialize(b);
            PinotGpu.segmentSetQueryableDocIds(e._handle, GpuBuffers.address(b), b.capacity());
          }
          e._lastSnapshot = snapshot;
        }
      }
    }
    return e._handle;
  }

  private Entry register(ImmutableSegment segment) {
    Entry e = new Entry();
    int device = _devices[Math.floorMod(_next.getAndIncrement(), _devices.length)];
    long h = PinotGpu.segmentCreate(segment.getSegmentName(), segment.getSegmentMetadata().getTotalDocs(), device);
    try (SegmentDirectory.Reader reader = GpuBuffers.readerOf(segment)) {
      for (String column : segment.getPhysicalColumnNames()) {
        ColumnMetadata md = segment.getSegmentMetadata().getColumnMetadataFor(column);
        if (!md.isSingleValue() && !md.hasDictionary() && md.getDataType().getStoredType().name().equals("BYTES")) {
          continue;   // raw multi-value BYTES columns stay with the Java plan: a query touching one is refused (unknown column)
        }
        PinotDataBuffer fwd = reader.getIndexFor(column, StandardIndexes.forward());
        PinotDataBuffer dict = md.hasDictionary() ? reader.getIndexFor(column, StandardIndexes.dictionary()) : null;
        PinotDataBuffer inv = reader.hasIndexFor(column, StandardIndexes.inverted()) ? reader.getIndexFor(column, StandardIndexes.inverted()) : null;
        // pg_fwd_encoding: 3 = FixedBitMVForwardIndexReader (dictionary-encoded multi-value column, ForwardIndexReaderFactory.java:92-96)
        // 4 = VarByteChunkSVForwardIndexReader (raw STRING / BYTES: a GROUP BY key at most), 1 = FixedByteChunkSVForwardIndexReader
        boolean varByte = !md.hasDictionary() && !md.getDataType().getStoredType().isFixedWidth();
        // 5 = FixedByteChunkMVForwardIndexReader (raw multi-value INT / LONG / FLOAT / DOUBLE, ForwardIndexReaderFactory.java:104-108):
        // read once at registration into a dictionary-encoded twin (pg_segment.cpp), group keys come back as values
        // 6 = VarByteChunkMVForwardIndexReader (raw multi-value STRING), read the same way
        int fwdEncoding = !md.isSingleValue() ? (md.hasDictionary() ? 3 : md.getDataType().getStoredType().isFixedWidth() ? 5 : 6)
            : md.isSorted() && md.hasDictionary() ? 2 : md.hasDictionary() ? 0 : varByte ? 4 : 1;
        try {
          PinotGpu.segmentAddColumn(h, column, GpuBuffers.storedType(md), fwdEncoding, md.hasDictionary(), md.getCardinality(),
              md.getBitsPerElement(), md.isSorted(), GpuBuffers.dictionaryBytesPerValue(md),
              md.isSingleValue() ? 0 : md.getTotalNumberOfEntries(), GpuBuffers.address(fwd), fwd.size(), GpuBuffers.dictionaryValuesAddress(dict), GpuBuffers.dictionaryValuesSize(dict, md),
              inv == null ? 0 : GpuBuffers.address(inv), inv == null ? 0 : inv.size());
        } catch (RuntimeException perColumn) {
          // Multi-value and var-byte columns come in layouts the library does not read (V4 / V5 var-byte chunks, CLP; the MV_ENTRY_DICT
          // forward index — ForwardIndexReaderFactory.java:82-86 — is read since round 4): such a column is SKIPPED, not fatal — the segment keeps serving every query that does not touch it, and a query that
          // does is refused by pg_query_supported (unknown column) and answered by the Java plan.  (Round 3 rethrew here: one such
          // column made every query of the segment fail, ADVICE r3.)  Single-value fixed-width columns keep the strict behaviour below.
          if (md.isSingleValue() && !varByte) {
            throw perColumn;
          }
          continue;
        }
        if (reader.hasIndexFor(column, StandardIndexes.range())) {
          // DataSource#getRangeIndex: RANGE predicates then take RangeIndexBasedFilterOperator's place in the plan
          // (FilterOperatorUtils.java:99-131); a legacy (version 1) index is refused by the library: the column keeps its scan leaf
          PinotDataBuffer range = reader.getIndexFor(column, StandardIndexes.range());
          try {
            PinotGpu.segmentSetRangeIndex(h, column, GpuBuffers.address(range), range.size());
          } catch (UnsupportedOperationException legacy) {
            // BitSlicedRangeIndexReader.java:41-58 reads version 2 only; RangeIndexReaderImpl (version 1) answers on the Java side
          }
        }
        if (reader.hasIndexFor(column, StandardIndexes.nullValueVector())) {
          PinotDataBuffer nulls = reader.getIndexFor(column, StandardIndexes.nullValueVector());
          PinotGpu.segmentSetNullVector(h, column, GpuBuffers.address(nulls), nulls.size());
        }
      }
      GpuBuffers.registerStarTrees(h, segment, reader);   // one PinotGpu.segmentAddStarTree per IndexSegment#getStarTrees() entry
      e._handle = h;
      _deviceOfHandle.put(h, device);
    } catch (UnsupportedOperationException unsupported) {   // e.g. ZSTANDARD chunks: this segment keeps the Java plan
      PinotGpu.segmentDestroy(h);
      e._handle = 0;
    } catch (Exception other) {
      PinotGpu.segmentDestroy(h);
      throw new RuntimeException(other);
    }
    return e;
  }

  /** The GPU a registered segment's columns are pinned on (pg_segment_create_on_device). */
  public int deviceOf(long handle) {
    Integer device = _deviceOfHandle.get(handle);
    return device == null ? _devices[0] : device;
  }

  public void release(ImmutableSegment segment) {
    Entry e = _entries.remove(segment.getSegmentName() + ":" + segment.getSegmentMetadata().getCrc());
    if (e != null && e._handle != 0) {
      _deviceOfHandle.remove(e._handle);
      PinotGpu.segmentDestroy(e._handle);
    }
  }
}
