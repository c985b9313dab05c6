/**
 * PinotDataBuffer plumbing of the GPU plug-in: native addresses of index buffers, the dictionary arithmetic pg_column_desc needs, and
 * the registration of a segment's star-trees.
 *
 * Addresses.  PinotDataBuffer has no public address accessor (pinot-segment-spi/.../memory/PinotDataBuffer.java:375-700: typed
 * get/put, view, toDirectByteBuffer); every implementation the default factory creates (PinotDataBuffer.java:162-196) is one
 * contiguous off-heap or mmap'ed region, and toDirectByteBuffer(offset, size) (:654-676) returns a zero-copy direct ByteBuffer over
 * it — so the address of the buffer is the JNI GetDirectBufferAddress of a one-byte view at offset 0 (PinotGpu.directBufferAddress).
 * Columns exceed 2 GB: only the ADDRESS is taken from the int-sized view, the size travels as PinotDataBuffer#size() (a long).
 *
 * NOT compiled in this repository (no JDK in the build image): written against the reference's API by reading, see INTEGRATION.md.
 */
package org.apache.pinot.gpu;

import java.io.File;
import java.nio.ByteBuffer;
import java.util.ArrayList;
import java.util.List;
import org.apache.pinot.segment.local.segment.store.SegmentLocalFSDirectory;
import org.apache.pinot.segment.spi.ColumnMetadata;
import org.apache.pinot.segment.spi.ImmutableSegment;
import org.apache.pinot.segment.spi.index.StandardIndexes;
import org.apache.pinot.segment.spi.index.metadata.SegmentMetadataImpl;
import org.apache.pinot.segment.spi.index.startree.AggregationFunctionColumnPair;
import org.apache.pinot.segment.spi.index.startree.StarTreeV2Metadata;
import org.apache.pinot.segment.spi.memory.PinotDataBuffer;
import org.apache.pinot.segment.spi.store.SegmentDirectory;
import org.apache.pinot.spi.data.FieldSpec.DataType;
import org.apache.pinot.spi.utils.ReadMode;

public final class GpuBuffers {
  private GpuBuffers() {
  }

  /** Native address of a direct ByteBuffer (a serialized upsert snapshot, a NativeQuery record). */
  public static long address(ByteBuffer direct) {
    return PinotGpu.directBufferAddress(direct);
  }

  /** Native address of byte 0 of an index buffer (0 for null / empty buffers). */
  public static long address(PinotDataBuffer buffer) {
    if (buffer == null || buffer.size() == 0) {
      return 0;
    }
    return PinotGpu.directBufferAddress(buffer.toDirectByteBuffer(0, 1));
  }

  /**
   * A reader over the segment's index files.  ImmutableSegmentImpl keeps its SegmentDirectory private
   * (pinot-segment-local/.../indexsegment/immutable/ImmutableSegmentImpl.java:70), so the directory is opened once more, read-only and
   * mmap'ed (the pages are the ones the segment already maps; SegmentLocalFSDirectory.java:78-81).  Closed by the caller after the
   * registration: pg_segment_add_column has copied the bytes into HBM by then.
   */
  public static SegmentDirectory.Reader readerOf(ImmutableSegment segment)
      throws Exception {
    File indexDir = ((SegmentMetadataImpl) segment.getSegmentMetadata()).getIndexDir();
    return new SegmentLocalFSDirectory(indexDir, (SegmentMetadataImpl) segment.getSegmentMetadata(), ReadMode.mmap).createReader();
  }

  /** pg_data_type of the column's stored type (FieldSpec.DataType#getStoredType). */
  public static int storedType(ColumnMetadata md) {
    return storedType(md.getDataType().getStoredType());
  }

  public static int storedType(DataType stored) {
    switch (stored) {
      case INT:
        return 0;
      case LONG:
        return 1;
      case FLOAT:
        return 2;
      case DOUBLE:
        return 3;
      case STRING:
        return 4;
      case BYTES:
        return 5;
      default:
        throw new UnsupportedOperationException("stored type " + stored + " is outside the GPU path (BIG_DECIMAL, MAP, ...)");
    }
  }

  /**
   * Bytes per dictionary entry: 4 / 8 for numeric types; the padded entry length of a fixed-width STRING / BYTES dictionary
   * (BaseImmutableDictionary.java:45-58: numBytesPerValue = ColumnMetadata#getColumnMaxLength for padded dictionaries).
   */
  public static int dictionaryBytesPerValue(ColumnMetadata md) {
    if (!md.hasDictionary()) {
      return 0;
    }
    switch (md.getDataType().getStoredType()) {
      case INT:
      case FLOAT:
        return 4;
      case LONG:
      case DOUBLE:
        return 8;
      default:
        return md.getColumnMaxLength();
    }
  }

  /**
   * The dictionary values: fixed-width dictionaries are the `dictionary` index entry itself, sorted big-endian values with no header
   * (BaseImmutableDictionary.java:45-58 reads value i at i * numBytesPerValue); variable-length STRING / BYTES dictionaries
   * (VarLengthValueReader: magic + offsets) are refused — their columns keep the Java plan.
   */
  public static long dictionaryValuesAddress(PinotDataBuffer dictionary) {
    return address(dictionary);
  }

  public static long dictionaryValuesSize(PinotDataBuffer dictionary, ColumnMetadata md) {
    if (dictionary == null) {
      return 0;
    }
    // a variable-length STRING dictionary (VarLengthValueReader.isVarLengthValueBuffer: ".vl;" + version 1) travels whole: the library turns it
    // into the padded form at registration (pg_segment.cpp)
    if (dictionary.size() >= 20 && dictionary.getByte(0) == '.' && dictionary.getByte(1) == 'v' && dictionary.getByte(2) == 'l'
        && dictionary.getByte(3) == ';' && dictionary.getInt(4) == 1) {
      return dictionary.size();
    }
    long expected = (long) md.getCardinality() * dictionaryBytesPerValue(md);
    if (dictionary.size() < expected) {
      throw new UnsupportedOperationException("dictionary of " + md.getColumnName() + " is neither fixed-width nor a variable-length value buffer");
    }
    return expected;
  }

  /**
   * One PinotGpu.segmentAddStarTree per IndexSegment#getStarTrees() entry, in order (StarTreeLoaderUtils.java:55-86): the tree
   * (entry "<i>.inverted" of the star-tree index map, OffHeapStarTree.java:44-76), the dimensions' fixed-bit forward indexes over the
   * parent's dictionaries, and one raw forward index per function-column pair (AggregationFunctionColumnPair#toColumnName; stored type
   * = ValueAggregatorFactory#getAggregatedValueType: LONG for COUNT, DOUBLE for SUM / MIN / MAX, BYTES for DISTINCTCOUNTHLL / AVG /
   * MINMAXRANGE).
   */
  public static void registerStarTrees(long handle, ImmutableSegment segment, SegmentDirectory.Reader reader)
      throws Exception {
    List<StarTreeV2Metadata> trees = ((SegmentMetadataImpl) segment.getSegmentMetadata()).getStarTreeV2MetadataList();
    if (trees == null || !reader.hasStarTreeIndex()) {
      return;
    }
    for (int i = 0; i < trees.size(); i++) {
      StarTreeV2Metadata md = trees.get(i);
      SegmentDirectory.Reader tree = reader.getStarTreeIndexReader(i);
      PinotDataBuffer nodes = tree.getIndexFor(String.valueOf(i), StandardIndexes.inverted());
      List<String> dims = md.getDimensionsSplitOrder();
      long[] dimAddrSize = new long[2 * dims.size()];
      for (int d = 0; d < dims.size(); d++) {
        PinotDataBuffer fwd = tree.getIndexFor(dims.get(d), StandardIndexes.forward());
        dimAddrSize[2 * d] = address(fwd);
        dimAddrSize[2 * d + 1] = fwd.size();
      }
      List<AggregationFunctionColumnPair> pairs = new ArrayList<>(md.getFunctionColumnPairs());
      int[] functions = new int[pairs.size()];
      int[] types = new int[pairs.size()];
      String[] columns = new String[pairs.size()];
      long[] pairAddrSize = new long[2 * pairs.size()];
      for (int p = 0; p < pairs.size(); p++) {
        AggregationFunctionColumnPair pair = pairs.get(p);
        functions[p] = NativeQuery.functionCode(pair.getFunctionType());   // pg_agg_function; -1: a pair the GPU path never reads
        types[p] = storedType(org.apache.pinot.segment.local.aggregator.ValueAggregatorFactory
            .getAggregatedValueType(pair.getFunctionType()).getStoredType());
        columns[p] = pair.getColumn();
        PinotDataBuffer fwd = tree.getIndexFor(pair.toColumnName(), StandardIndexes.forward());
        pairAddrSize[2 * p] = address(fwd);
        pairAddrSize[2 * p + 1] = fwd.size();
      }
      PinotGpu.segmentAddStarTree(handle, md.getNumDocs(), md.getMaxLeafRecords(), dims.toArray(new String[0]), dimAddrSize, functions,
          types, columns, pairAddrSize, address(nodes), nodes.size());
    }
  }
}
