/**
 * The plug-in seam: ServerQueryExecutorV1Impl instantiates the plan maker by class name
 * (pinot-core/.../query/executor/ServerQueryExecutorV1Impl.java:116-123, key pinot.server.query.executor.plan.maker.class,
 * pinot-spi/.../utils/CommonConstants.java:724-725).  Aggregation / group-by queries over immutable segments whose shape
 * libpinot_gpu.so takes (pg_query_supported) get a plan node that yields GpuGroupByOperator; everything else keeps
 * InstancePlanMakerImplV2's plan (core/plan/maker/InstancePlanMakerImplV2.java:275-294).
 *
 *   pinot.server.query.executor.plan.maker.class=org.apache.pinot.gpu.GpuInstancePlanMaker
 *   pinot.server.gpu.devices=0,1,2,3,4,5,6,7      # segments are spread round-robin over these GPUs (segment -> GPU map)
 *   pinot.server.gpu.library.merge=true           # one RCCL communicator per GPU (pg_comm_init_all) for tables whose segments share
 *                                                 # their dictionaries: GpuGroupByCombineOperator folds the per-segment tables with
 *                                                 # PinotGpu.resultMerge (same GPU) / resultAllReduce (across GPUs) before one decode;
 *                                                 # UnsupportedOperationException (different dictionaries, hashed key spaces, trimming)
 *                                                 # means: merge by values in IndexedTable, as GroupByCombineOperator always does
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
  private long[] _comms;   // null unless pinot.server.gpu.library.merge: _comms[i] is the communicator of _devices[i]
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
      long[] comms = new long[ordinals.length];
      PinotGpu.commInitAll(ordinals, comms);   // RuntimeException when librccl cannot be loaded: the server then fails fast at start-up
      _comms = comms;
      _current = this;
    }
  }

  /** The RCCL communicator of `device` for PinotGpu.resultAllReduce, or 0 when the library merge is not configured. */
  public long communicatorOf(int device) {
    for (int i = 0; _comms != null && i < _devices.length; i++) {
      if (_devices[i] == device) {
        return _comms[i];
      }
    }
    return 0;
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
        boolean keep = _comms != null && queryContext.getGroupByExpressions() != null && !queryContext.isNullHandlingEnabled();
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
