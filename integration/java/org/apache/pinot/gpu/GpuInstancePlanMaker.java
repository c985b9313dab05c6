/**
 * The plug-in seam: ServerQueryExecutorV1Impl instantiates the plan maker by class name
 * (pinot-core/.../query/executor/ServerQueryExecutorV1Impl.java:116-123, key pinot.server.query.executor.plan.maker.class,
 * pinot-spi/.../utils/CommonConstants.java:724-725).  Aggregation / group-by queries over immutable segments whose shape
 * libpinot_gpu.so takes (pg_query_supported) get a plan node that yields GpuGroupByOperator; everything else keeps
 * InstancePlanMakerImplV2's plan (core/plan/maker/InstancePlanMakerImplV2.java:275-294).
 *
 *   pinot.server.query.executor.plan.maker.class=org.apache.pinot.gpu.GpuInstancePlanMaker
 *   pinot.server.gpu.devices=0,1,2,3,4,5,6,7      # segments are spread round-robin over these GPUs (segment -> GPU map)
 *   pinot.server.gpu.library.merge=true           # RCCL communicators, one per GPU (pg_comm_init_all): GpuGroupByCombineOperator folds the
 *                                                 # per-segment tables with PinotGpu.resultMerge (same GPU) / resultAllReduce (across GPUs) before
 *                                                 # one decode — segments with dictionaries of their own included (the library re-keys their tables
 *                                                 # by value); UnsupportedOperationException (hashed key spaces, trimming, distinct-count sets over
 *                                                 # different dictionaries) means: merge by values in IndexedTable, as GroupByCombineOperator does
 *   pinot.server.gpu.library.merge.sets=2         # communicator SETS: that many cross-GPU merges proceed at once (a set admits one merge at a
 *                                                 # time: pg_comm.cpp); merged-query throughput is no longer 1 / merge latency per server
 */
package org.apache.pinot.gpu;

import org.apache.pinot.core.plan.PlanNode;
import org.apache.pinot.core.plan.maker.InstancePlanMakerImplV2;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.core.query.request.context.utils.QueryContextUtils;
import org.apache.pinot.segment.spi.ImmutableSegment;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.SegmentContext;
import org.apache.pinot.spi.env.PinotConfiguration;

public class GpuInstancePlanMaker extends InstancePlanMakerImplV2 {
  private GpuSegmentRegistry _registry;
  private int[] _devices;
  // null unless pinot.server.gpu.library.merge: the pool of communicator sets — set[i] is the communicator of _devices[i].  A cross-GPU merge
  // takes a whole set for its duration (every rank of ONE merge must use ONE set, and a set is inside one merge at a time)
  private java.util.concurrent.BlockingQueue<long[]> _commSets;
  private static volatile GpuInstancePlanMaker _current;   // the server's plan maker once the library merge is configured

  /** The plan maker GpuGroupByCombineOperator takes its communicators from; null when the library merge is not configured. */
  public static GpuInstancePlanMaker current() {
    return _current;
  }

  public int numDevices() {
    return _devices.length;
  }

  @Override
  public void init(PinotConfiguration config) {
    super.init(config);
    String[] devices = config.getProperty("pinot.server.gpu.devices", "0").split(",");
    int[] ordinals = new int[devices.length];
    for (int i = 0; i < devices.length; i++) {
      ordinals[i] = Integer.parseInt(devices[i].trim());
    }
    PinotGpu.init(ordinals[0]);
    _registry = new GpuSegmentRegistry(ordinals);
    _devices = ordinals;
    if (Boolean.parseBoolean(config.getProperty("pinot.server.gpu.library.merge", "false")) && ordinals.length > 1) {
      int sets = Math.max(1, Integer.parseInt(config.getProperty("pinot.server.gpu.library.merge.sets", "2").trim()));
      java.util.concurrent.BlockingQueue<long[]> pool = new java.util.concurrent.ArrayBlockingQueue<>(sets);
      for (int k = 0; k < sets; k++) {
        long[] comms = new long[ordinals.length];
        PinotGpu.commInitAll(ordinals, comms);   // RuntimeException when librccl cannot be loaded: the server then fails fast at start-up
        pool.add(comms);
      }
      _commSets = pool;
      _current = this;
    }
  }

  /**
   * Takes a communicator set out of the pool for ONE cross-GPU merge (blocks while every set is inside a merge); the caller hands it back with
   * releaseCommunicators once every rank's PinotGpu.resultAllReduce has returned.  set[i] belongs to the i-th configured device (indexOfDevice).
   */
  public long[] acquireCommunicators() throws InterruptedException {
    return _commSets.take();
  }

  public void releaseCommunicators(long[] set) {
    _commSets.add(set);
  }

  /** Position of `device` in pinot.server.gpu.devices (= index into a communicator set), or -1. */
  public int indexOfDevice(int device) {
    for (int i = 0; i < _devices.length; i++) {
      if (_devices[i] == device) {
        return i;
      }
    }
    return -1;
  }

  @Override
  public PlanNode makeSegmentPlanNode(SegmentContext segmentContext, QueryContext queryContext) {
    IndexSegment segment = segmentContext.getIndexSegment();
    // (enableNullHandling travels in the query record: three-valued filters, null-skipping aggregations and null group keys are answered by
    // the library — results and keys come back with NULL flags, GpuGroupByOperator#blockOf; it refuses nulls in multi-value columns)
    if (segment instanceof ImmutableSegment && QueryContextUtils.isAggregationQuery(queryContext)) {
      long handle = _registry.handleFor((ImmutableSegment) segment, segmentContext);   // pins the columns in HBM on first use; 0: Java plan only
      if (handle != 0) {
        // with the library merge configured the group-by tables stay in HBM for GpuGroupByCombineOperator (PinotGpu.resultMerge /
        // resultAllReduce); without the combine patch the operators decode them one by one as before
        // (a null-handling result is joined on the host from its null partitions: no device table to keep — those merge by values in Java)
        boolean keep = _commSets != null && queryContext.getGroupByExpressions() != null && !queryContext.isNullHandlingEnabled();
        NativeQuery nativeQuery = NativeQuery.from(queryContext, keep ? PinotGpu.QUERY_FLAG_KEEP_DEVICE_TABLE : 0);
        if (nativeQuery != null) {
          if (PinotGpu.querySupported(handle, nativeQuery.address()) == PinotGpu.PG_OK) {
            return () -> new GpuGroupByOperator(segment, queryContext, handle, nativeQuery,
                () -> super.makeSegmentPlanNode(segmentContext, queryContext).run(), _registry.deviceOf(handle));
          }
          nativeQuery.close();
        }
      }
    }
    return super.makeSegmentPlanNode(segmentContext, queryContext);   // PG_ERR_UNSUPPORTED: the default plan
  }
}
