/**
 * Native entry points of libpinot_gpu.so, 1:1 with include/pinot_gpu.h through integration/jni/pinot_gpu_jni.c.
 * Handles are native pointers carried as long; buffers are (address, size) pairs of PinotDataBuffer; every failing status
 * surfaces as RuntimeException (EarlyTerminationException for PG_ERR_CANCELLED) with pg_last_error() as its message.
 * NOT compiled in this repository (no JDK in the build image) — see INTEGRATION.md.
 */
package org.apache.pinot.gpu;

import java.nio.ByteBuffer;

public final class PinotGpu {
  static {
    System.loadLibrary("pinot_gpu_jni");   // links libpinot_gpu.so
  }

  private PinotGpu() {
  }

  // pg_status values the Java side branches on
  public static final int PG_OK = 0;
  public static final int PG_ERR_UNSUPPORTED = -2;
  // pg_result_kind
  public static final int RESULT_LONG = 0, RESULT_DOUBLE = 1, RESULT_AVG_PAIR = 2, RESULT_MINMAX_PAIR = 3, RESULT_DICTID_SET = 4,
      RESULT_HLL = 5, RESULT_VALUE_SET = 6;
  public static final int GROUP_KEY_DICT_IDS = 0, GROUP_KEY_LONG_VALUES = 1, GROUP_KEY_DOUBLE_VALUES = 2, GROUP_KEY_BYTES_VALUES = 3;

  public static native int abiVersion();
  public static native void init(int device);
  public static native int deviceCount();

  public static native long segmentCreate(String name, int totalDocs, int device);   // device < 0: the default device
  public static native void segmentAddColumn(long segment, String name, int dataType, int fwdEncoding, boolean hasDictionary,
      int cardinality, int bitsPerValue, boolean sorted, int dictBytesPerValue, int totalNumberOfEntries, long fwdAddr, long fwdSize,
      long dictAddr, long dictSize, long invAddr, long invSize);   // totalNumberOfEntries: multi-value columns (fwdEncoding 3), else 0
  public static native void segmentSetNullVector(long segment, String column, long addr, long size);
  public static native void segmentSetQueryableDocIds(long segment, long addr, long size);
  public static native void segmentSetRangeIndex(long segment, String column, long addr, long size);   // the `range_index` entry (version 2)
  public static native long directBufferAddress(ByteBuffer direct);   // GetDirectBufferAddress: GpuBuffers.address
  public static native void segmentAddStarTree(long segment, int numDocs, int maxLeafRecords, String[] dimensions,
      long[] dimAddrSize, int[] pairFunctions, int[] pairTypes, String[] pairColumns, long[] pairAddrSize, long treeAddr,
      long treeSize);
  public static native long segmentDeviceBytes(long segment);
  public static native void segmentDestroy(long segment);

  public static native long queryParse(ByteBuffer directRecord, int size);   // NativeQuery record -> pg_query
  public static native void queryFree(long query);
  public static native int querySupported(long segment, long query);          // PG_OK or PG_ERR_UNSUPPORTED
  public static native long cancelCreate();
  public static native void cancelRequest(long token);
  public static native void cancelReset(long token);
  public static native void cancelDestroy(long token);
  public static native long queryExec(long segment, long query, long cancelToken);

  public static native int resultNumGroups(long result);
  public static native int resultKindOf(long result, int aggregation);
  public static native int resultGroupKeyType(long result, int groupByColumn);
  public static native void resultGroupDictIds(long result, int groupByColumn, int[] out);
  public static native void resultGroupValuesLong(long result, int groupByColumn, long[] out);
  public static native void resultGroupValuesDouble(long result, int groupByColumn, double[] out);
  public static native long resultGroupValuesBytesSize(long result, int groupByColumn);                 // raw STRING / BYTES keys: total bytes,
  public static native void resultGroupValuesBytes(long result, int groupByColumn, long[] offsets, byte[] out);   // offsets (groups + 1), values
  public static native void resultDoubles(long result, int aggregation, int component, double[] out);
  public static native void resultLongs(long result, int aggregation, int component, long[] out);
  public static native void resultSetSizes(long result, int aggregation, int[] out);
  public static native void resultSetDictIds(long result, int aggregation, int[] out);
  /** DISTINCTCOUNT over a raw column (RESULT_VALUE_SET): the groups' values, concatenated — Long for INT / LONG, Double for FLOAT / DOUBLE columns. */
  public static native void resultSetValuesLong(long result, int aggregation, long[] out);
  public static native void resultSetValuesDouble(long result, int aggregation, double[] out);
  public static native void resultHllRegisters(long result, int aggregation, byte[] out);
  // enableNullHandling: out[g] = 1 where group g's result of the aggregation / key of the group-by column is NULL
  public static native void resultAggNulls(long result, int aggregation, byte[] out);
  public static native void resultGroupKeyNulls(long result, int groupByColumn, byte[] out);
  // the (merged) intermediate results as DataTableImplV4 bytes: DataTableFactory.getDataTable(ByteBuffer.wrap(bytes)) reads them
  public static native long resultDataTableV4Size(long result);
  public static native void resultDataTableV4(long result, byte[] out);
  public static native void resultStats(long result, long[] out5);
  public static native void resultFree(long result);

  // GroupByCombineOperator in the library: results executed with QUERY_FLAG_KEEP_DEVICE_TABLE over segments that share their
  // dictionaries merge element-wise in HBM; UnsupportedOperationException -> merge by values (IndexedTable) as usual
  public static final int QUERY_FLAG_SKIP_STAR_TREE = 0x2, QUERY_FLAG_KEEP_DEVICE_TABLE = 0x4, QUERY_FLAG_APPROX_FILTER_STATS = 0x8,
      QUERY_FLAG_EXACT_FILTER_STATS = 0x10, QUERY_FLAG_FINAL_DISTINCT = 0x20, QUERY_FLAG_NULL_HANDLING = 0x40;
  public static final int COMM_UNIQUE_ID_BYTES = 128;
  public static native void resultMerge(long dst, long src);                 // same device
  public static native void resultAllReduce(long result, long comm);         // collective over the communicator's ranks (RCCL)
  public static native void commGetUniqueId(byte[] out128);                  // rank 0, one process per GPU
  public static native long commInitRank(int device, int worldSize, int rank, byte[] uniqueId128);
  public static native void commInitAll(int[] devices, long[] outComms);     // one process, N GPUs: outComms[i] for devices[i]
  public static native int commWorldSize(long comm);
  public static native void commDestroy(long comm);

  public static native long filterExec(long segment, long query);
  public static native long docIdSetCardinality(long set);
  public static native void docIdSetCopyWords(long set, long[] out);
  public static native void docIdSetCopyDocIds(long set, int[] out);
  public static native void docIdSetFree(long set);
}
