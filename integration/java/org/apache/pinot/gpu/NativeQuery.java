/**
 * QueryContext -> the flat little-endian record integration/jni/pinot_gpu_shim.h describes (the native side turns it into a
 * pg_query).  Literals stay strings exactly as the Predicate objects hold them; the native planner parses them against the
 * column's stored type as PredicateEvaluatorProvider does (pinot-core/.../predicate/PredicateEvaluatorProvider.java:45-96).
 * Returns null for query shapes the record cannot express (transform expressions, MV / text / JSON / regexp predicates, other
 * aggregation functions): the plan maker then keeps the default plan.
 */
package org.apache.pinot.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.charset.StandardCharsets;
import java.util.List;
import org.apache.commons.lang3.tuple.Pair;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.request.context.FilterContext;
import org.apache.pinot.common.request.context.FunctionContext;
import org.apache.pinot.common.request.context.OrderByExpressionContext;
import org.apache.pinot.common.request.context.predicate.EqPredicate;
import org.apache.pinot.common.request.context.predicate.InPredicate;
import org.apache.pinot.common.request.context.predicate.NotEqPredicate;
import org.apache.pinot.common.request.context.predicate.NotInPredicate;
import org.apache.pinot.common.request.context.predicate.Predicate;
import org.apache.pinot.common.request.context.predicate.RangePredicate;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.core.query.aggregation.function.DistinctCountHLLAggregationFunction;
import org.apache.pinot.core.query.request.context.QueryContext;

public final class NativeQuery implements AutoCloseable {
  private static final int MAGIC = 0x32514750;   // "PGQ2": + ORDER BY block, LIMIT, minSegmentGroupTrimSize
  // pg_filter_type / pg_predicate_type / pg_agg_function
  private static final int F_AND = 0, F_OR = 1, F_NOT = 2, F_PREDICATE = 3, F_TRUE = 4, F_FALSE = 5;
  private static final int P_EQ = 0, P_NOT_EQ = 1, P_IN = 2, P_NOT_IN = 3, P_RANGE = 4, P_IS_NULL = 5, P_IS_NOT_NULL = 6;
  private static final int FLAG_SKIP_STAR_TREE = 0x2;
  // PG_QUERY_FLAG_NULL_HANDLING: three-valued filters, null-skipping aggregations, null group keys (pg_query_supported refuses nulls in
  // multi-value columns)
  private static final int FLAG_NULL_HANDLING = 0x40;

  private static final int FLAG_KEEP_DEVICE_TABLE = 0x4;   // PG_QUERY_FLAG_KEEP_DEVICE_TABLE

  private final long _address;
  private final int _flags;

  private NativeQuery(long address, int flags) {
    _address = address;
    _flags = flags;
  }

  /** The record asked for the dense accumulator table to stay in HBM with the result (GpuGroupByCombineOperator's library merge). */
  public boolean keepsDeviceTable() {
    return (_flags & FLAG_KEEP_DEVICE_TABLE) != 0;
  }

  public long address() {
    return _address;
  }

  @Override
  public void close() {
    PinotGpu.queryFree(_address);
  }

  public static NativeQuery from(QueryContext q) {
    return from(q, 0);
  }

  /** `extraFlags`: PinotGpu.QUERY_FLAG_* ORed into the record (GpuGroupByCombineOperator asks for QUERY_FLAG_KEEP_DEVICE_TABLE). */
  public static NativeQuery from(QueryContext q, int extraFlags) {
    ByteBuffer b = ByteBuffer.allocateDirect(estimate(q)).order(ByteOrder.LITTLE_ENDIAN);
    List<ExpressionContext> groupBy = q.getGroupByExpressions();
    AggregationFunction[] aggs = q.getAggregationFunctions();
    if (aggs == null || aggs.length == 0) {
      return null;
    }
    // Segment-level group trim (GroupByOperator.java:120-133): the ORDER BY expressions as TableResizer resolves them (TableResizer.java:
    // 129-161) — a group-by expression or an aggregation of the query.  Anything else (post-aggregations, literals, filtered aggregations)
    // leaves the block empty: the segment's groups then come back untrimmed, which only keeps groups the broker would drop anyway.
    int[] orderBy = groupBy == null ? null : orderBy(q, groupBy);
    b.putInt(MAGIC).putInt((q.isSkipStarTree() ? FLAG_SKIP_STAR_TREE : 0) | (q.isNullHandlingEnabled() ? FLAG_NULL_HANDLING : 0) | extraFlags).putInt(q.getNumGroupsLimit())
        .putInt(q.getMaxInitialResultHolderCapacity()).putInt(groupBy == null ? 0 : groupBy.size()).putInt(aggs.length)
        .putInt(q.getFilter() == null ? 0 : 1).putInt(orderBy == null ? 0 : orderBy.length / 4);
    b.putInt(q.getLimit()).putInt(orderBy == null ? -1 : q.getMinSegmentGroupTrimSize());
    if (groupBy != null) {
      for (ExpressionContext e : groupBy) {
        if (e.getType() != ExpressionContext.Type.IDENTIFIER) {
          return null;
        }
        putString(b, e.getIdentifier());
      }
    }
    for (AggregationFunction f : aggs) {
      int fn = function(f);
      List<ExpressionContext> args = f.getInputExpressions();
      if (fn < 0 || args.size() > 1 || (args.size() == 1 && args.get(0).getType() != ExpressionContext.Type.IDENTIFIER)) {
        return null;
      }
      b.putInt(fn).putInt(0);   // log2m 0: DEFAULT_HYPERLOGLOG_LOG2M (a literal second argument is not expressible here)
      putString(b, args.isEmpty() ? "*" : args.get(0).getIdentifier());
    }
    if (orderBy != null) {
      for (int v : orderBy) {
        b.putInt(v);
      }
    }
    if (q.getFilter() != null && !putFilter(b, q.getFilter())) {
      return null;
    }
    return new NativeQuery(PinotGpu.queryParse(b, b.position()), extraFlags);
  }

  /** {kind, index, ascending, nullsLast} per ORDER BY expression (pg_order_by), or null when the query has none or one the record cannot carry. */
  private static int[] orderBy(QueryContext q, List<ExpressionContext> groupBy) {
    List<OrderByExpressionContext> orderBy = q.getOrderByExpressions();
    if (orderBy == null || orderBy.isEmpty() || q.getMinSegmentGroupTrimSize() <= 0) {
      return null;
    }
    int[] out = new int[4 * orderBy.size()];
    for (int i = 0; i < orderBy.size(); i++) {
      OrderByExpressionContext o = orderBy.get(i);
      ExpressionContext e = o.getExpression();
      int kind;
      int index = groupBy.indexOf(e);
      if (index >= 0) {
        kind = 0;   // PG_ORDER_BY_GROUP_KEY
      } else {
        FunctionContext f = e.getFunction();
        if (f == null || f.getType() != FunctionContext.Type.AGGREGATION) {
          return null;
        }
        Integer a = q.getFilteredAggregationsIndexMap().get(Pair.of(f, null));
        if (a == null) {
          return null;
        }
        kind = 1;   // PG_ORDER_BY_AGGREGATION
        index = a;
      }
      out[4 * i] = kind;
      out[4 * i + 1] = index;
      out[4 * i + 2] = o.isAsc() ? 1 : 0;
      out[4 * i + 3] = o.isNullsLast() ? 1 : 0;
    }
    return out;
  }

  /** pg_agg_function of a star-tree function-column pair's function type; -1 for functions the GPU path never reads. */
  static int functionCode(org.apache.pinot.segment.spi.AggregationFunctionType type) {
    switch (type) {
      case COUNT: return 0;
      case SUM: return 1;
      case MIN: return 2;
      case MAX: return 3;
      case AVG: return 4;
      case DISTINCTCOUNT: return 5;
      case DISTINCTCOUNTHLL: return 6;
      case MINMAXRANGE: return 7;
      default: return -1;
    }
  }

  private static int function(AggregationFunction f) {
    switch (f.getType()) {
      case COUNT: return 0;
      case SUM: return 1;
      case MIN: return 2;
      case MAX: return 3;
      case AVG: return 4;
      case DISTINCTCOUNT: return 5;
      case DISTINCTCOUNTHLL: return f instanceof DistinctCountHLLAggregationFunction ? 6 : -1;
      case MINMAXRANGE: return 7;
      // the multi-value forms (pg_agg_function 8..15): every entry of every matching doc
      case COUNTMV: return 8;
      case SUMMV: return 9;
      case MINMV: return 10;
      case MAXMV: return 11;
      case AVGMV: return 12;
      case MINMAXRANGEMV: return 13;
      case DISTINCTCOUNTMV: return 14;
      case DISTINCTCOUNTHLLMV: return 15;
      default: return -1;
    }
  }

  private static boolean putFilter(ByteBuffer b, FilterContext f) {
    switch (f.getType()) {
      case AND:
      case OR:
      case NOT:
        b.putInt(f.getType() == FilterContext.Type.AND ? F_AND : f.getType() == FilterContext.Type.OR ? F_OR : F_NOT);
        b.putInt(f.getChildren().size());
        for (FilterContext c : f.getChildren()) {
          if (!putFilter(b, c)) {
            return false;
          }
        }
        return true;
      case CONSTANT:
        b.putInt(f.isConstantTrue() ? F_TRUE : F_FALSE).putInt(0);
        return true;
      default:
        return putPredicate(b, f.getPredicate());
    }
  }

  private static boolean putPredicate(ByteBuffer b, Predicate p) {
    if (p.getLhs().getType() != ExpressionContext.Type.IDENTIFIER) {
      return false;
    }
    String column = p.getLhs().getIdentifier();
    b.putInt(F_PREDICATE).putInt(0);
    switch (p.getType()) {
      case EQ:
        header(b, P_EQ, 1, column);
        putString(b, ((EqPredicate) p).getValue());
        bounds(b, null, null, false, false);
        return true;
      case NOT_EQ:
        header(b, P_NOT_EQ, 1, column);
        putString(b, ((NotEqPredicate) p).getValue());
        bounds(b, null, null, false, false);
        return true;
      case IN:
      case NOT_IN: {
        List<String> values = p.getType() == Predicate.Type.IN ? ((InPredicate) p).getValues() : ((NotInPredicate) p).getValues();
        header(b, p.getType() == Predicate.Type.IN ? P_IN : P_NOT_IN, values.size(), column);
        for (String v : values) {
          putString(b, v);
        }
        bounds(b, null, null, false, false);
        return true;
      }
      case RANGE: {
        RangePredicate r = (RangePredicate) p;
        header(b, P_RANGE, 0, column);
        bounds(b, r.getLowerBound(), r.getUpperBound(), r.isLowerInclusive(), r.isUpperInclusive());   // "*" = UNBOUNDED on both sides
        return true;
      }
      case IS_NULL:
      case IS_NOT_NULL:
        header(b, p.getType() == Predicate.Type.IS_NULL ? P_IS_NULL : P_IS_NOT_NULL, 0, column);
        bounds(b, null, null, false, false);
        return true;
      default:
        return false;   // REGEXP_LIKE, TEXT_MATCH, JSON_MATCH, VECTOR_SIMILARITY ...: the default plan
    }
  }

  private static void header(ByteBuffer b, int predicateType, int nValues, String column) {
    b.putInt(predicateType).putInt(nValues);
    putString(b, column);
  }

  private static void bounds(ByteBuffer b, String lower, String upper, boolean lowerInclusive, boolean upperInclusive) {
    putString(b, lower);
    putString(b, upper);
    b.putInt(lowerInclusive ? 1 : 0).putInt(upperInclusive ? 1 : 0);
  }

  private static void putString(ByteBuffer b, String s) {
    if (s == null) {
      b.putInt(-1);
      return;
    }
    byte[] utf8 = s.getBytes(StandardCharsets.UTF_8);
    b.putInt(utf8.length).put(utf8);
    while ((b.position() & 3) != 0) {
      b.put((byte) 0);
    }
  }

  private static int estimate(QueryContext q) {
    return 4096 + 64 * q.toString().length();
  }
}
