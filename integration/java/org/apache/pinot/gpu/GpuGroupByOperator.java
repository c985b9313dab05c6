/**
 * GroupByOperator / AggregationOperator of the accelerated path (pinot-core/.../operator/query/GroupByOperator.java:100-140,
 * AggregationOperator.java): one pg_query_exec per segment, then the result is re-shaped into the blocks the unchanged
 * combine operator consumes.  Extends BaseOperator so that interruption checks and tracing wrap it
 * (core/operator/BaseOperator.java:36-53); a scheduler-side kill reaches the running kernel sequence through the cancel token.
 */
package org.apache.pinot.gpu;

import java.util.ArrayList;
import java.util.Collections;
import java.util.Iterator;
import java.util.List;
import java.util.function.Supplier;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.utils.DataSchema;
import org.apache.pinot.core.common.Operator;
import org.apache.pinot.core.operator.BaseOperator;
import org.apache.pinot.core.operator.ExecutionStatistics;
import org.apache.pinot.core.operator.blocks.results.AggregationResultsBlock;
import org.apache.pinot.core.operator.blocks.results.BaseResultsBlock;
import org.apache.pinot.core.operator.blocks.results.GroupByResultsBlock;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.aggregation.groupby.AggregationGroupByResult;
import org.apache.pinot.core.query.aggregation.groupby.DoubleGroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.GroupByResultHolder;
import org.apache.pinot.core.query.aggregation.groupby.GroupKeyGenerator;
import org.apache.pinot.core.query.aggregation.groupby.ObjectGroupByResultHolder;
import org.apache.pinot.core.query.request.context.QueryContext;
import org.apache.pinot.segment.local.customobject.AvgPair;
import org.apache.pinot.segment.local.customobject.MinMaxRangePair;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.index.reader.Dictionary;

public class GpuGroupByOperator extends BaseOperator<BaseResultsBlock> {
  private static final String EXPLAIN_NAME = "GPU_GROUP_BY";

  private final IndexSegment _segment;
  private final QueryContext _queryContext;
  private final long _segmentHandle;
  private final NativeQuery _nativeQuery;
  private final Supplier<Operator> _fallback;   // the default plan of this segment (run-time PG_ERR_UNSUPPORTED: hash bucket overflow)
  private final long[] _stats = new long[5];
  private final int _device;
  private long _adopted;
  private boolean _folded;
  private boolean _refused;
  private boolean _tableKept;   // the result in hand was executed with QUERY_FLAG_KEEP_DEVICE_TABLE: the library can fold it (pg_result_merge)

  public GpuGroupByOperator(IndexSegment segment, QueryContext queryContext, long segmentHandle, NativeQuery nativeQuery,
      Supplier<Operator> fallback) {
    this(segment, queryContext, segmentHandle, nativeQuery, fallback, 0);
  }

  public GpuGroupByOperator(IndexSegment segment, QueryContext queryContext, long segmentHandle, NativeQuery nativeQuery,
      Supplier<Operator> fallback, int device) {
    _device = device;
    _segment = segment;
    _queryContext = queryContext;
    _segmentHandle = segmentHandle;
    _nativeQuery = nativeQuery;
    _fallback = fallback;
  }

  @Override
  protected BaseResultsBlock getNextBlock() {
    if (_folded) {   // this segment's table went into another operator's (GpuGroupByCombineOperator): nothing of its own to add
      List<ExpressionContext> groupBy = _queryContext.getGroupByExpressions();
      return new GroupByResultsBlock(
          dataSchema(groupBy, _queryContext.getAggregationFunctions(), GpuResultObjects.keyTypes(_segment, groupBy)), _queryContext);
    }
    long result = _adopted;
    _adopted = 0;
    if (result == 0 && !_refused) {
      result = execute();
    }
    return _refused ? (BaseResultsBlock) _fallback.get().nextBlock() : blockOf(result);
  }

  /**
   * One pg_query_exec over this operator's segment; the caller owns the result handle (0: refused at run time).  GpuGroupByCombineOperator calls this on
   * operators whose query record carries QUERY_FLAG_KEEP_DEVICE_TABLE, folds the tables in the library and decodes one of them with
   * blockOf; the statistics of THIS segment are read here, before any merge, for getExecutionStatistics.
   */
  long execute() {
    long cancel = PinotGpu.cancelCreate();
    GpuCancellation.register(Thread.currentThread(), cancel);   // the query killer calls PinotGpu.cancelRequest(token) when it interrupts
    try {
      long result;
      try {
        result = PinotGpu.queryExec(_segmentHandle, _nativeQuery.address(), cancel);   // EarlyTerminationException when cancelled
        _tableKept = _nativeQuery.keepsDeviceTable();
      } catch (UnsupportedOperationException keepRefused) {
        // QUERY_FLAG_KEEP_DEVICE_TABLE is refused at run time for tables that do not merge element-wise (hashed key spaces, key spaces
        // beyond numGroupsLimit, multi-value tables beyond the limit: pg_exec.hip) — shapes the GPU answers perfectly well WITHOUT the flag.
        // Run the same query again without it; the result then merges by values in the unchanged combine loop (ADVICE r4).
        if (!_nativeQuery.keepsDeviceTable()) {
          throw keepRefused;
        }
        try (NativeQuery plain = NativeQuery.from(_queryContext, 0)) {
          if (plain == null) {
            throw keepRefused;
          }
          result = PinotGpu.queryExec(_segmentHandle, plain.address(), cancel);
          _tableKept = false;
        }
      }
      PinotGpu.resultStats(result, _stats);
      return result;
    } catch (UnsupportedOperationException e) {   // run-time PG_ERR_UNSUPPORTED: getNextBlock answers with the segment's default plan
      _refused = true;
      return 0;
    } finally {
      GpuCancellation.unregister(Thread.currentThread());
      PinotGpu.cancelDestroy(cancel);
      _nativeQuery.close();
    }
  }

  /** Whether the result execute() returned last kept its dense table in HBM (only such results enter pg_result_merge / _all_reduce). */
  boolean tableKept() {
    return _tableKept;
  }

  /** Frees a result adopted for a getNextBlock that never came (the combine stopped early: time-out, cancellation, another segment's failure). */
  void releaseAdopted() {
    if (_adopted != 0) {
      PinotGpu.resultFree(_adopted);
      _adopted = 0;
    }
  }

  /** A result executed earlier (execute) becomes what the next getNextBlock decodes: the combine operator's by-values route. */
  void adopt(long result) {
    _adopted = result;
  }

  /** This segment's table was folded into another operator's result (pg_result_merge / pg_result_all_reduce). */
  void adoptFolded() {
    _folded = true;
  }

  int device() {
    return _device;
  }

  /** Decodes `result` (this segment's, or a library merge over segments sharing its dictionaries) into a results block and frees it. */
  BaseResultsBlock blockOf(long result) {
    try {
      long[] stats = new long[5];
      PinotGpu.resultStats(result, stats);
      AggregationFunction[] functions = _queryContext.getAggregationFunctions();
      int numGroups = PinotGpu.resultNumGroups(result);
      List<ExpressionContext> groupBy = _queryContext.getGroupByExpressions();
      if (groupBy == null) {   // AggregationOperator: one intermediate result per function
        List<Object> results = new ArrayList<>(functions.length);
        for (int a = 0; a < functions.length; a++) {
          // AggregationFunction#extractAggregationResult's type (INTEGRATION.md §4 table): COUNT / COUNTMV hand a Long to
          // CountAggregationFunction#merge(Long, Long) and AggregationResultsBlock's `(long) result` (:165-166) — a Double there is a
          // ClassCastException (VERDICT r3); every other RESULT_* kind already is the function's intermediate object
          // enableNullHandling: SUM / MIN / MAX / AVG / MINMAXRANGE over no value extract null (SumAggregationFunction.java:215-222)
          results.add(isNull(result, a, 1)[0] ? null : intermediates(result, a, 1, functions[a], false)[0]);
        }
        return new AggregationResultsBlock(functions, results, _queryContext);
      }
      // group keys: dictIds decoded exactly as DictionaryBasedGroupKeyGenerator.getKeys does (:578-606), or the raw values of
      // no-dictionary group-by columns (NoDictionarySingleColumnGroupKeyGenerator.java:241-265,
      // NoDictionaryMultiColumnGroupKeyGenerator.java:60-130): LONG values for INT / LONG, DOUBLE values for FLOAT / DOUBLE
      Object[][] keys = new Object[numGroups][groupBy.size()];
      for (int j = 0; j < groupBy.size(); j++) {
        String column = groupBy.get(j).getIdentifier();
        if (PinotGpu.resultGroupKeyType(result, j) == PinotGpu.GROUP_KEY_LONG_VALUES) {
          long[] values = new long[numGroups];
          PinotGpu.resultGroupValuesLong(result, j, values);
          boolean isInt = _segment.getDataSource(column).getDataSourceMetadata().getDataType().getStoredType().name().equals("INT");
          for (int g = 0; g < numGroups; g++) {
            keys[g][j] = isInt ? (Object) (int) values[g] : (Object) values[g];
          }
        } else if (PinotGpu.resultGroupKeyType(result, j) == PinotGpu.GROUP_KEY_DOUBLE_VALUES) {
          double[] values = new double[numGroups];
          PinotGpu.resultGroupValuesDouble(result, j, values);
          boolean isFloat = _segment.getDataSource(column).getDataSourceMetadata().getDataType().getStoredType().name().equals("FLOAT");
          for (int g = 0; g < numGroups; g++) {
            keys[g][j] = isFloat ? (Object) (float) values[g] : (Object) values[g];
          }
        } else if (PinotGpu.resultGroupKeyType(result, j) == PinotGpu.GROUP_KEY_BYTES_VALUES) {
          // a raw STRING / BYTES column (NoDictionarySingleColumnGroupKeyGenerator.java:132-140): String keys, ByteArray keys for BYTES
          long[] offsets = new long[numGroups + 1];
          byte[] bytes = new byte[(int) PinotGpu.resultGroupValuesBytesSize(result, j)];
          PinotGpu.resultGroupValuesBytes(result, j, offsets, bytes);
          boolean isString = _segment.getDataSource(column).getDataSourceMetadata().getDataType().getStoredType().name().equals("STRING");
          for (int g = 0; g < numGroups; g++) {
            int from = (int) offsets[g], to = (int) offsets[g + 1];
            keys[g][j] = isString ? (Object) new String(bytes, from, to - from, java.nio.charset.StandardCharsets.UTF_8)
                : (Object) new org.apache.pinot.spi.utils.ByteArray(java.util.Arrays.copyOfRange(bytes, from, to));
          }
        } else {
          int[] dictIds = new int[numGroups];
          PinotGpu.resultGroupDictIds(result, j, dictIds);
          Dictionary dictionary = _segment.getDataSource(column).getDictionary();
          for (int g = 0; g < numGroups; g++) {
            keys[g][j] = dictionary.getInternal(dictIds[g]);
          }
        }
      }
      if (_queryContext.isNullHandlingEnabled()) {   // a null is a group key of its own (the value-based key generators hold a null key)
        for (int j = 0; j < groupBy.size(); j++) {
          byte[] keyNulls = new byte[numGroups];
          PinotGpu.resultGroupKeyNulls(result, j, keyNulls);
          for (int g = 0; g < numGroups; g++) {
            if (keyNulls[g] != 0) {
              keys[g][j] = null;
            }
          }
        }
      }
      GroupByResultHolder[] holders = new GroupByResultHolder[functions.length];
      for (int a = 0; a < functions.length; a++) {
        Object[] values = intermediates(result, a, numGroups, functions[a], true);
        boolean[] nulls = isNull(result, a, numGroups);
        // under null handling SUM / MIN / MAX keep Double OBJECTS in an ObjectGroupByResultHolder, null where no value was seen
        // (SumAggregationFunction.java:80-84,180-215); COUNT stays in its DoubleGroupByResultHolder
        boolean nullable = _queryContext.isNullHandlingEnabled() && PinotGpu.resultKindOf(result, a) == PinotGpu.RESULT_DOUBLE;
        for (int g = 0; g < numGroups; g++) {
          if (nulls[g]) {
            values[g] = null;
          }
        }
        boolean asDouble = !nullable && values.length > 0 && values[0] instanceof Double;
        GroupByResultHolder holder = asDouble ? new DoubleGroupByResultHolder(Math.max(numGroups, 1), Math.max(numGroups, 1), 0.0)
            : new ObjectGroupByResultHolder(Math.max(numGroups, 1), Math.max(numGroups, 1));
        holder.ensureCapacity(Math.max(numGroups, 1));
        for (int g = 0; g < numGroups; g++) {
          if (asDouble) {
            holder.setValueForKey(g, ((Double) values[g]).doubleValue());
          } else {
            holder.setValueForKey(g, values[g]);
          }
        }
        holders[a] = holder;
      }
      GroupByResultsBlock block = new GroupByResultsBlock(dataSchema(groupBy, functions, GpuResultObjects.keyTypes(_segment, groupBy)),
          new AggregationGroupByResult(new ArrayGroupKeyGenerator(keys), functions, holders), _queryContext);
      block.setNumGroupsLimitReached(stats[4] != 0);
      return block;
    } finally {
      PinotGpu.resultFree(result);
    }
  }

  /**
   * Intermediate result of aggregation `a` for every group.  Two consumers, two typings (INTEGRATION.md §4):
   *   forHolder = true   the VALUE the function's GroupByResultHolder keeps — what AggregationFunction#extractGroupByResult reads back
   *                      (COUNT: a double in a DoubleGroupByResultHolder, CountAggregationFunction.java:79-81,183-185)
   *   forHolder = false  the OBJECT AggregationFunction#extractAggregationResult returns, i.e. getIntermediateResultColumnType
   *                      (COUNT: a Long, CountAggregationFunction.java:178-180,193-195)
   */
  /** enableNullHandling: which of aggregation `a`'s results are NULL (none without the query option). */
  private boolean[] isNull(long result, int a, int n) {
    boolean[] out = new boolean[n];
    if (_queryContext.isNullHandlingEnabled()) {
      byte[] flags = new byte[n];
      PinotGpu.resultAggNulls(result, a, flags);
      for (int g = 0; g < n; g++) {
        out[g] = flags[g] != 0;
      }
    }
    return out;
  }

  private Object[] intermediates(long result, int a, int n, AggregationFunction function, boolean forHolder) {
    Object[] out = new Object[n];
    switch (PinotGpu.resultKindOf(result, a)) {
      case PinotGpu.RESULT_LONG: {
        long[] v = new long[n];
        PinotGpu.resultLongs(result, a, 0, v);
        for (int g = 0; g < n; g++) {
          out[g] = forHolder ? (Object) Double.valueOf((double) v[g]) : (Object) Long.valueOf(v[g]);
        }
        return out;
      }
      case PinotGpu.RESULT_DOUBLE: {
        double[] v = new double[n];
        PinotGpu.resultDoubles(result, a, 0, v);
        for (int g = 0; g < n; g++) {
          out[g] = v[g];
        }
        return out;
      }
      case PinotGpu.RESULT_AVG_PAIR: {
        double[] s = new double[n];
        long[] c = new long[n];
        PinotGpu.resultDoubles(result, a, 0, s);
        PinotGpu.resultLongs(result, a, 0, c);
        for (int g = 0; g < n; g++) {
          out[g] = new AvgPair(s[g], c[g]);
        }
        return out;
      }
      case PinotGpu.RESULT_MINMAX_PAIR: {
        double[] lo = new double[n];
        double[] hi = new double[n];
        PinotGpu.resultDoubles(result, a, 0, lo);
        PinotGpu.resultDoubles(result, a, 1, hi);
        for (int g = 0; g < n; g++) {
          out[g] = new MinMaxRangePair(lo[g], hi[g]);
        }
        return out;
      }
      case PinotGpu.RESULT_DICTID_SET:
        return GpuResultObjects.valueSets(result, a, n, _segment, function);     // dictIds -> typed value Set (BaseDistinctAggregate...:671-694)
      case PinotGpu.RESULT_VALUE_SET:
        return GpuResultObjects.rawValueSets(result, a, n, _segment, function);  // a raw column's values -> typed value Set (:325-380)
      default:
        return GpuResultObjects.hyperLogLogs(result, a, n, function);            // register bytes -> com.clearspring HyperLogLog (RegisterSet)
    }
  }

  private static DataSchema dataSchema(List<ExpressionContext> groupBy, AggregationFunction[] functions, DataSchema.ColumnDataType[] keyTypes) {   // GroupByOperator.java:74-97
    int n = groupBy.size() + functions.length;
    String[] names = new String[n];
    DataSchema.ColumnDataType[] types = new DataSchema.ColumnDataType[n];
    for (int i = 0; i < groupBy.size(); i++) {
      names[i] = groupBy.get(i).toString();
      types[i] = keyTypes[i];
    }
    for (int i = 0; i < functions.length; i++) {
      names[groupBy.size() + i] = functions[i].getResultColumnName();
      types[groupBy.size() + i] = functions[i].getIntermediateResultColumnType();
    }
    return new DataSchema(names, types);
  }

  /** GroupKeyGenerator over already materialised keys: only getGroupKeys() / getNumKeys() are used downstream
   *  (AggregationGroupByResult.java:35-56, GroupByCombineOperator.java:132-147). */
  private static final class ArrayGroupKeyGenerator implements GroupKeyGenerator {
    private final Object[][] _keys;

    ArrayGroupKeyGenerator(Object[][] keys) {
      _keys = keys;
    }

    @Override
    public int getGlobalGroupKeyUpperBound() {
      return _keys.length;
    }

    @Override
    public void generateKeysForBlock(org.apache.pinot.core.operator.blocks.ValueBlock valueBlock, int[] groupKeys) {
      throw new UnsupportedOperationException();
    }

    @Override
    public void generateKeysForBlock(org.apache.pinot.core.operator.blocks.ValueBlock valueBlock, int[][] groupKeys) {
      throw new UnsupportedOperationException();
    }

    @Override
    public int getCurrentGroupKeyUpperBound() {
      return _keys.length;
    }

    @Override
    public Iterator<GroupKey> getGroupKeys() {
      return new Iterator<GroupKey>() {
        private int _next = 0;

        @Override
        public boolean hasNext() {
          return _next < _keys.length;
        }

        @Override
        public GroupKey next() {
          GroupKey k = new GroupKey();
          k._groupId = _next;
          k._keys = _keys[_next++];
          return k;
        }
      };
    }

    @Override
    public int getNumKeys() {
      return _keys.length;
    }
  }

  @Override
  public List<Operator> getChildOperators() {
    return Collections.emptyList();
  }

  @Override
  public String toExplainString() {
    return EXPLAIN_NAME;
  }

  @Override
  public IndexSegment getIndexSegment() {
    return _segment;
  }

  @Override
  public ExecutionStatistics getExecutionStatistics() {
    return new ExecutionStatistics(_stats[0], _stats[1], _stats[2], _stats[3]);
  }
}
