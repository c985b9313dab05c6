/**
 * A group-by results block whose DataTable was serialized by the library (pg_result_data_table_v4): the instance response is built from
 * DataTableImplV4 bytes directly — no Object[] per group, no IndexedTable (pinot-core/.../operator/blocks/results/GroupByResultsBlock.java
 * :186-236 builds the same table row by row from boxed records; pinot-common/.../common/datatable/DataTableImplV4.java:118-200 is the reader
 * the broker runs on it).  For the case where nothing is left to do on the server: every segment's table folded in the library
 * (GpuGroupByCombineOperator) and no server-side trim (no ORDER BY, or fewer groups than the trim size) — otherwise the combine operator
 * keeps the IndexedTable route.  The rows carry INTERMEDIATE results (AvgPair, HyperLogLog, value sets ...) in the result's group order.
 */
package org.apache.pinot.gpu;

import java.io.IOException;
import java.nio.ByteBuffer;
import org.apache.pinot.common.datatable.DataTable;
import org.apache.pinot.common.datatable.DataTableFactory;
import org.apache.pinot.common.utils.DataSchema;
import org.apache.pinot.core.operator.blocks.results.GroupByResultsBlock;
import org.apache.pinot.core.query.request.context.QueryContext;

public class GpuDataTableResultsBlock extends GroupByResultsBlock {
  private final byte[] _dataTableBytes;
  private final int _numGroups;

  /** `result`: a native result handle (kept by the caller); its table is serialized here, once. */
  public GpuDataTableResultsBlock(DataSchema dataSchema, QueryContext queryContext, long result) {
    super(dataSchema, queryContext);
    _numGroups = PinotGpu.resultNumGroups(result);
    _dataTableBytes = new byte[(int) PinotGpu.resultDataTableV4Size(result)];
    PinotGpu.resultDataTableV4(result, _dataTableBytes);
  }

  @Override
  public int getNumRows() {
    return _numGroups;
  }

  @Override
  public DataTable getDataTable()
      throws IOException {
    return DataTableFactory.getDataTable(ByteBuffer.wrap(_dataTableBytes));
  }
}
