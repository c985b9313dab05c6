"""Independent model of the segment-level group trim for the tests (GroupByOperator.java:120-133, GroupByUtils.getTableCapacity
:45-57, TableResizer.java:88-128,327-351,406-445): given the UNTRIMMED rows of a group-by result, which trimmed results are valid.
The reference keeps a heap of trimSize records: groups strictly before the cut must all survive, groups strictly after it must all
go, and any choice among the groups tied with the last survivor is valid."""
import math


def final_value(function, v):
    """AggregationFunction#extractFinalResult of the intermediate `v` as executor.ResultsBlock.columns presents it."""
    function = {"COUNTMV": "COUNT", "SUMMV": "SUM", "MINMV": "MIN", "MAXMV": "MAX", "AVGMV": "AVG", "MINMAXRANGEMV": "MINMAXRANGE",
                "DISTINCTCOUNTMV": "DISTINCTCOUNT"}.get(function, function)   # the multi-value forms extend the single-value functions
    if function in ("COUNT", "SUM", "MIN", "MAX"):
        return v
    if function == "DISTINCTCOUNT":      # the set's size (an Integer)
        return len(v)
    if function == "DISTINCTCOUNTHLL":   # HyperLogLog#cardinality (a Long)
        from pinot_amd.executor import hll_cardinality
        return hll_cardinality(v)
    if function == "AVG":
        s, c = v
        return -math.inf if c == 0 else s / c
    if function == "MINMAXRANGE":
        lo, hi = v
        return hi - lo
    raise ValueError(function)


def order_tuple(qc, key, row):
    """The group's order-by values, descending expressions negated so that tuples compare ascending (numbers only).  Under null handling a
    null key / a null final result (None) sorts by the expression's isNullsLast, whatever its direction (TableResizer.java:98-116): every
    value is a pair (-1 null first | 0 value | 1 null last, value)."""
    out = []
    nulls_last = list(getattr(qc, "order_by_nulls_last", [])) + [None] * len(qc.order_by)
    for i, (kind, index, asc) in enumerate(qc.resolved_order_by()):
        raw = key[index] if kind == 0 else row[index]
        if raw is None:
            last = asc if nulls_last[i] is None else nulls_last[i]   # OrderByExpressionContext#isNullsLast
            out.append((1 if last else -1, 0))
            continue
        v = raw if kind == 0 else final_value(qc.aggregations[index].function, raw)
        out.append((0, v if asc else -v))
    return tuple(out)


def trim_size(qc):
    return max(5 * qc.limit, qc.min_segment_group_trim_size)


def assert_valid_trim(qc, full_rows, trimmed_rows):
    k = trim_size(qc)
    if len(full_rows) <= k or qc.min_segment_group_trim_size <= 0 or not qc.order_by:
        assert trimmed_rows == full_rows
        return
    assert len(trimmed_rows) == k, (len(trimmed_rows), k)
    order = {key: order_tuple(qc, key, row) for key, row in full_rows.items()}
    cut = sorted(order.values())[k - 1]
    for key, row in trimmed_rows.items():
        assert full_rows[key] == row, key                       # a survivor keeps its values
        assert order[key] <= cut, (key, order[key], cut)        # nothing from behind the cut
    must = {key for key, t in order.items() if t < cut}          # everything before the cut survives
    assert must <= set(trimmed_rows), sorted(must - set(trimmed_rows))[:5]
