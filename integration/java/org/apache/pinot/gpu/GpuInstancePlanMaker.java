/**
 * The plug-in seam: ServerQueryExecutorV1Impl instantiates the plan maker by class name
 * (pinot-core/.../query/executor/ServerQueryExecutorV1Impl.java:116-123, key pinot.server.query.executor.plan.maker.class,
 * pinot-spi/.../utils/CommonConstants.java:724-725).  Aggregation / group-by queries over immutable segments whose shape
 * libpinot_gpu.so takes (pg_query_supported) get a plan node that yields GpuGroupByOperator; everything else keeps
 * InstancePlanMakerImplV2's plan (core/plan/maker/InstancePlanMakerImplV2.java:275-294).
 *
 *   pinot.server.query.executor.plan.maker.class=org.apache.pinot.gpu.GpuInstancePlanMaker
 *   pinot.server.gpu.devices=0,1,2,3,4,5,6,7      # segments are spread round-robin over these GPUs (segment -> GPU map)
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
  }

  @Override
  public PlanNode makeSegmentPlanNode(SegmentContext segmentContext, QueryContext queryContext) {
    IndexSegment segment = segmentContext.getIndexSegment();
    if (segment instanceof ImmutableSegment && QueryContextUtils.isAggregationQuery(queryContext)
        && !queryContext.isNullHandlingEnabled()) {
      long handle = _registry.handleFor((ImmutableSegment) segment, segmentContext);   // pins the columns in HBM on first use; 0: Java plan only
      if (handle != 0) {
        NativeQuery nativeQuery = NativeQuery.from(queryContext);
        if (nativeQuery != null) {
          if (PinotGpu.querySupported(handle, nativeQuery.address()) == PinotGpu.PG_OK) {
            return () -> new GpuGroupByOperator(segment, queryContext, handle, nativeQuery,
                () -> super.makeSegmentPlanNode(segmentContext, queryContext).run());
          }
          nativeQuery.close();
        }
      }
    }
    return super.makeSegmentPlanNode(segmentContext, queryContext);   // PG_ERR_UNSUPPORTED: the default plan
  }
}
