/**
 * Native result arrays -> the intermediate-result OBJECTS the reference's combine / reduce stages expect:
 *
 *   DISTINCTCOUNT      per group the dictIds of the group's set (pg_result_set_sizes / _set_dict_ids) -> the typed value Set
 *                      BaseDistinctAggregateAggregationFunction#extractGroupByResult builds from its dictId bitmap
 *                      (pinot-core/.../aggregation/function/BaseDistinctAggregateAggregationFunction.java:306-345,646-665,760-806:
 *                      IntOpenHashSet / LongOpenHashSet / FloatOpenHashSet / DoubleOpenHashSet / ObjectOpenHashSet by stored type,
 *                      BYTES wrapped in ByteArray)
 *   DISTINCTCOUNTHLL   per group 2^log2m one-byte registers (pg_result_hll_registers) -> com.clearspring.analytics HyperLogLog over a
 *                      RegisterSet whose int[] packs six five-bit registers per word (register i -> word i / 6, shift 5 * (i % 6)),
 *                      exactly what ObjectSerDeUtils.HYPER_LOG_LOG_SER_DE#deserialize rebuilds (ObjectSerDeUtils.java:733-767)
 *   group key types    DataSchema column types of the group-by columns from the segment's column metadata (GroupByOperator.java:74-97)
 *
 * NOT compiled in this repository (no JDK in the build image): written against the reference's API by reading, see INTEGRATION.md.
 */
package org.apache.pinot.gpu;

import com.clearspring.analytics.stream.cardinality.HyperLogLog;
import com.clearspring.analytics.stream.cardinality.RegisterSet;
import it.unimi.dsi.fastutil.doubles.DoubleOpenHashSet;
import it.unimi.dsi.fastutil.floats.FloatOpenHashSet;
import it.unimi.dsi.fastutil.ints.IntOpenHashSet;
import it.unimi.dsi.fastutil.longs.LongOpenHashSet;
import it.unimi.dsi.fastutil.objects.ObjectOpenHashSet;
import java.util.List;
import java.util.Set;
import org.apache.pinot.common.request.context.ExpressionContext;
import org.apache.pinot.common.utils.DataSchema;
import org.apache.pinot.core.query.aggregation.function.AggregationFunction;
import org.apache.pinot.segment.spi.IndexSegment;
import org.apache.pinot.segment.spi.index.reader.Dictionary;
import org.apache.pinot.spi.data.FieldSpec.DataType;
import org.apache.pinot.spi.utils.ByteArray;

public final class GpuResultObjects {
  private GpuResultObjects() {
  }

  /** One value Set per group (empty sets for groups without a value: cannot happen for a group that exists). */
  @SuppressWarnings({"rawtypes", "unchecked"})
  public static Object[] valueSets(long result, int aggregation, int numGroups, IndexSegment segment, AggregationFunction function) {
    String column = ((ExpressionContext) function.getInputExpressions().get(0)).getIdentifier();
    Dictionary dictionary = segment.getDataSource(column).getDictionary();
    DataType stored = dictionary.getValueType();
    int[] sizes = new int[numGroups];
    PinotGpu.resultSetSizes(result, aggregation, sizes);
    long total = 0;
    for (int s : sizes) {
      total += s;
    }
    if (total > Integer.MAX_VALUE - 8) {
      throw new UnsupportedOperationException("DISTINCTCOUNT state of " + total + " dictIds exceeds one Java array");
    }
    int[] dictIds = new int[(int) total];
    PinotGpu.resultSetDictIds(result, aggregation, dictIds);
    Object[] out = new Object[numGroups];
    int at = 0;
    for (int g = 0; g < numGroups; g++) {
      int n = sizes[g];
      Set set;
      switch (stored) {
        case INT: {
          IntOpenHashSet s = new IntOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(dictionary.getIntValue(dictIds[at + i]));
          }
          set = s;
          break;
        }
        case LONG: {
          LongOpenHashSet s = new LongOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(dictionary.getLongValue(dictIds[at + i]));
          }
          set = s;
          break;
        }
        case FLOAT: {
          FloatOpenHashSet s = new FloatOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(dictionary.getFloatValue(dictIds[at + i]));
          }
          set = s;
          break;
        }
        case DOUBLE: {
          DoubleOpenHashSet s = new DoubleOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(dictionary.getDoubleValue(dictIds[at + i]));
          }
          set = s;
          break;
        }
        case STRING: {
          ObjectOpenHashSet<String> s = new ObjectOpenHashSet<>(n);
          for (int i = 0; i < n; i++) {
            s.add(dictionary.getStringValue(dictIds[at + i]));
          }
          set = s;
          break;
        }
        case BYTES: {
          ObjectOpenHashSet<ByteArray> s = new ObjectOpenHashSet<>(n);
          for (int i = 0; i < n; i++) {
            s.add(new ByteArray(dictionary.getBytesValue(dictIds[at + i])));
          }
          set = s;
          break;
        }
        default:
          throw new IllegalStateException("Illegal data type for DISTINCT_AGGREGATE aggregation function: " + stored);
      }
      out[g] = set;
      at += n;
    }
    return out;
  }

  /** DISTINCTCOUNT over a raw (no-dictionary) INT / LONG / FLOAT / DOUBLE column (pg_result_kind PG_RESULT_VALUE_SET): the library hands the
   *  groups' VALUES over — the typed open-hash sets BaseDistinctAggregateAggregationFunction.java:325-380 keeps for such a column. */
  @SuppressWarnings("rawtypes")
  public static Object[] rawValueSets(long result, int aggregation, int numGroups, IndexSegment segment, AggregationFunction function) {
    String column = ((ExpressionContext) function.getInputExpressions().get(0)).getIdentifier();
    DataType stored = segment.getDataSource(column).getDataSourceMetadata().getDataType().getStoredType();
    int[] sizes = new int[numGroups];
    PinotGpu.resultSetSizes(result, aggregation, sizes);
    long total = 0;
    for (int s : sizes) {
      total += s;
    }
    if (total > Integer.MAX_VALUE - 8) {
      throw new UnsupportedOperationException("DISTINCTCOUNT state of " + total + " values exceeds one Java array");
    }
    boolean floating = stored == DataType.FLOAT || stored == DataType.DOUBLE;
    long[] longs = floating ? null : new long[(int) total];
    double[] doubles = floating ? new double[(int) total] : null;
    if (floating) {
      PinotGpu.resultSetValuesDouble(result, aggregation, doubles);
    } else {
      PinotGpu.resultSetValuesLong(result, aggregation, longs);
    }
    Object[] out = new Object[numGroups];
    int at = 0;
    for (int g = 0; g < numGroups; g++) {
      int n = sizes[g];
      Set set;
      switch (stored) {
        case INT: {
          IntOpenHashSet s = new IntOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add((int) longs[at + i]);
          }
          set = s;
          break;
        }
        case LONG: {
          LongOpenHashSet s = new LongOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(longs[at + i]);
          }
          set = s;
          break;
        }
        case FLOAT: {
          FloatOpenHashSet s = new FloatOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add((float) doubles[at + i]);   // widened exactly by the library
          }
          set = s;
          break;
        }
        case DOUBLE: {
          DoubleOpenHashSet s = new DoubleOpenHashSet(n);
          for (int i = 0; i < n; i++) {
            s.add(doubles[at + i]);
          }
          set = s;
          break;
        }
        default:
          throw new IllegalStateException("Illegal data type for a raw DISTINCT_AGGREGATE value set: " + stored);
      }
      out[g] = set;
      at += n;
    }
    return out;
  }

  /** One HyperLogLog per group from its 2^log2m register bytes. */
  public static Object[] hyperLogLogs(long result, int aggregation, int numGroups, AggregationFunction function) {
    int log2m = ((org.apache.pinot.core.query.aggregation.function.DistinctCountHLLAggregationFunction) function).getLog2m();
    int m = 1 << log2m;
    long total = (long) numGroups * m;
    if (total > Integer.MAX_VALUE - 8) {
      throw new UnsupportedOperationException("HyperLogLog state of " + total + " registers exceeds one Java array");
    }
    byte[] registers = new byte[(int) total];
    PinotGpu.resultHllRegisters(result, aggregation, registers);
    Object[] out = new Object[numGroups];
    int words = (m + 5) / 6;   // RegisterSet.getSizeForCount: six five-bit registers per int
    for (int g = 0; g < numGroups; g++) {
      int[] bits = new int[words];
      int base = g * m;
      for (int i = 0; i < m; i++) {
        bits[i / 6] |= (registers[base + i] & 0x1F) << (5 * (i % 6));
      }
      out[g] = new HyperLogLog(log2m, new RegisterSet(m, bits));
    }
    return out;
  }

  /** DataSchema column types of the group-by columns (GroupByOperator.java:74-97 takes them from the expressions' result metadata). */
  public static DataSchema.ColumnDataType[] keyTypes(IndexSegment segment, List<ExpressionContext> groupBy) {
    DataSchema.ColumnDataType[] types = new DataSchema.ColumnDataType[groupBy.size()];
    for (int i = 0; i < types.length; i++) {
      DataType dataType = segment.getDataSource(groupBy.get(i).getIdentifier()).getDataSourceMetadata().getDataType();
      types[i] = DataSchema.ColumnDataType.fromDataTypeSV(dataType);
    }
    return types;
  }
}
