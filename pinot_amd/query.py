"""QueryContext for the hot path + a small SQL front end so tests read like the reference's (`getOperator(sql)`).

Mirrors the part of the reference a segment operator sees:
  QueryContext        pinot-core/.../query/request/context/QueryContext.java
  FilterContext       pinot-common/.../request/context/FilterContext.java
  Predicate & co      pinot-common/.../request/context/predicate/{Eq,NotEq,In,NotIn,Range}Predicate.java
SQL → QueryContext follows CalciteSqlParser.compileToPinotQuery + RequestContextUtils.getFilter for the shapes the
inner-segment tests use: nested AND/OR are flattened (CalciteSqlParser.compileAndExpression), comparisons become RANGE
predicates with "*" for an unbounded side (RequestContextUtils.java, RangePredicate.UNBOUNDED), BETWEEN is an inclusive
RANGE.  The broker-side QueryOptimizer (range merging etc.) is *not* applied, exactly as in BaseQueriesTest.getOperator
(pinot-core/src/test/.../queries/BaseQueriesTest.java:100-105).
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

from . import capi

UNBOUNDED = "*"


@dataclass
class Predicate:
    type: str                      # EQ / NOT_EQ / IN / NOT_IN / RANGE
    column: str
    values: List[str] = field(default_factory=list)
    lower: str = UNBOUNDED
    upper: str = UNBOUNDED
    lower_inclusive: bool = False
    upper_inclusive: bool = False


@dataclass
class FilterContext:
    type: str                      # AND / OR / NOT / PREDICATE / CONSTANT_TRUE / CONSTANT_FALSE
    children: List["FilterContext"] = field(default_factory=list)
    predicate: Optional[Predicate] = None

    @staticmethod
    def and_(children):
        return FilterContext("AND", list(children))

    @staticmethod
    def or_(children):
        return FilterContext("OR", list(children))

    @staticmethod
    def not_(child):
        return FilterContext("NOT", [child])

    @staticmethod
    def pred(p: Predicate):
        return FilterContext("PREDICATE", [], p)


def eq(col, v):
    return FilterContext.pred(Predicate("EQ", col, [str(v)]))


def neq(col, v):
    return FilterContext.pred(Predicate("NOT_EQ", col, [str(v)]))


def in_(col, vs):
    return FilterContext.pred(Predicate("IN", col, [str(v) for v in vs]))


def not_in(col, vs):
    return FilterContext.pred(Predicate("NOT_IN", col, [str(v) for v in vs]))


def range_(col, lower=UNBOUNDED, upper=UNBOUNDED, lower_inclusive=True, upper_inclusive=True):
    return FilterContext.pred(Predicate("RANGE", col, [], str(lower), str(upper),
                                        bool(lower_inclusive) and str(lower) != UNBOUNDED,
                                        bool(upper_inclusive) and str(upper) != UNBOUNDED))


@dataclass
class AggregationSpec:
    function: str                  # COUNT / SUM / MIN / MAX / AVG / DISTINCTCOUNT / DISTINCTCOUNTHLL / MINMAXRANGE
    column: Optional[str] = None   # None for COUNT(*)
    log2m: int = 0


@dataclass
class QueryContext:
    table: str = "testTable"
    filter: Optional[FilterContext] = None
    group_by: List[str] = field(default_factory=list)
    aggregations: List[AggregationSpec] = field(default_factory=list)
    select_columns: List[str] = field(default_factory=list)   # plain identifiers in the select list
    order_by: List[Tuple[str, bool]] = field(default_factory=list)  # (expression text, ascending)
    order_by_nulls_last: List[Optional[bool]] = field(default_factory=list)   # parallel: NULLS LAST / NULLS FIRST, None = the default (OrderByExpressionContext#isNullsLast: as ascending)
    limit: int = 10
    num_groups_limit: int = 0
    max_initial_result_holder_capacity: int = 0
    flags: int = 0
    has_group_by: bool = False
    min_segment_group_trim_size: int = -1   # InstancePlanMakerImplV2.DEFAULT_MIN_SEGMENT_GROUP_TRIM_SIZE: the segment's groups are not trimmed

    def resolved_order_by(self) -> "Optional[List[Tuple[int, int, bool]]]":
        """(kind, index, ascending) per ORDER BY expression as TableResizer resolves them (TableResizer.java:129-161): a group-by expression
        or an aggregation of the select list; None when some expression is neither (post-aggregations, literals: not carried by the ABI)."""
        out = []
        norm = lambda t: re.sub(r"\s+", "", t).upper()   # noqa: E731
        aggs = [norm(f"{a.function}({a.column or '*'}{',' + str(a.log2m) if a.log2m else ''})") for a in self.aggregations]
        for text, asc in self.order_by:
            if text in self.group_by:
                out.append((capi.ORDER_BY_GROUP_KEY, self.group_by.index(text), asc))
            elif norm(text) in aggs:
                out.append((capi.ORDER_BY_AGGREGATION, aggs.index(norm(text)), asc))
            else:
                return None
        return out


# ----------------------------------------------------------------------------------------------------------------------
# SQL subset parser
# ----------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"\s*(?:(?P<num>-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?)|(?P<str>'(?:[^']|'')*')|(?P<id>[A-Za-z_][A-Za-z_0-9.$]*)"
                    r"|(?P<op><=|>=|<>|!=|=|<|>|\(|\)|,|\*))")


class SqlError(ValueError):
    pass


def _tokenize(sql: str) -> List[Tuple[str, str]]:
    pos = 0
    out = []
    sql = sql.strip().rstrip(";")
    while pos < len(sql):
        m = _TOKEN.match(sql, pos)
        if not m or m.end() == pos:
            raise SqlError(f"cannot tokenize at: {sql[pos:pos + 20]!r}")
        pos = m.end()
        if m.group("num") is not None:
            out.append(("num", m.group("num")))
        elif m.group("str") is not None:
            out.append(("str", m.group("str")[1:-1].replace("''", "'")))
        elif m.group("id") is not None:
            out.append(("id", m.group("id")))
        else:
            out.append(("op", m.group("op")))
    return out


class _Parser:
    def __init__(self, sql: str):
        self.toks = _tokenize(sql)
        self.i = 0

    def peek(self, k=0):
        return self.toks[self.i + k] if self.i + k < len(self.toks) else ("eof", "")

    def kw(self, word: str, k=0) -> bool:
        t = self.peek(k)
        return t[0] == "id" and t[1].upper() == word

    def take(self):
        t = self.peek()
        self.i += 1
        return t

    def expect_kw(self, word: str):
        if not self.kw(word):
            raise SqlError(f"expected {word}, got {self.peek()}")
        self.i += 1

    def expect_op(self, op: str):
        t = self.take()
        if t != ("op", op):
            raise SqlError(f"expected {op!r}, got {t}")

    def literal(self) -> str:
        t = self.take()
        if t[0] not in ("num", "str"):
            raise SqlError(f"expected literal, got {t}")
        return t[1]

    # --- filter -------------------------------------------------------------------------------------------------
    def or_expr(self) -> FilterContext:
        parts = [self.and_expr()]
        while self.kw("OR"):
            self.i += 1
            parts.append(self.and_expr())
        if len(parts) == 1:
            return parts[0]
        flat = []
        for p in parts:
            flat.extend(p.children if p.type == "OR" else [p])
        return FilterContext.or_(flat)

    def and_expr(self) -> FilterContext:
        parts = [self.not_expr()]
        while self.kw("AND"):
            self.i += 1
            parts.append(self.not_expr())
        if len(parts) == 1:
            return parts[0]
        flat = []
        for p in parts:
            flat.extend(p.children if p.type == "AND" else [p])
        return FilterContext.and_(flat)

    def not_expr(self) -> FilterContext:
        if self.kw("NOT"):
            self.i += 1
            return FilterContext.not_(self.not_expr())
        return self.primary()

    def primary(self) -> FilterContext:
        if self.peek() == ("op", "("):
            self.i += 1
            e = self.or_expr()
            self.expect_op(")")
            return e
        t = self.take()
        if t[0] != "id":
            raise SqlError(f"expected column, got {t}")
        col = t[1]
        if self.kw("BETWEEN"):
            self.i += 1
            lo = self.literal()
            self.expect_kw("AND")
            hi = self.literal()
            return FilterContext.pred(Predicate("RANGE", col, [], lo, hi, True, True))
        if self.kw("IS"):   # IS [NOT] NULL → Predicate.Type.IS_NULL / IS_NOT_NULL (IsNullPredicate, IsNotNullPredicate)
            self.i += 1
            not_null = self.kw("NOT")
            if not_null:
                self.i += 1
            self.expect_kw("NULL")
            return FilterContext.pred(Predicate("IS_NOT_NULL" if not_null else "IS_NULL", col, []))
        negate = False
        if self.kw("NOT"):
            self.i += 1
            negate = True
            if self.kw("BETWEEN"):   # Calcite: NOT BETWEEN → NOT(RANGE)
                self.i += 1
                lo = self.literal()
                self.expect_kw("AND")
                hi = self.literal()
                return FilterContext.not_(FilterContext.pred(Predicate("RANGE", col, [], lo, hi, True, True)))
        if self.kw("IN"):
            self.i += 1
            self.expect_op("(")
            vals = [self.literal()]
            while self.peek() == ("op", ","):
                self.i += 1
                vals.append(self.literal())
            self.expect_op(")")
            return FilterContext.pred(Predicate("NOT_IN" if negate else "IN", col, vals))
        if negate:
            raise SqlError("NOT must be followed by IN here")
        op = self.take()
        if op[0] != "op":
            raise SqlError(f"expected comparison, got {op}")
        v = self.literal()
        o = op[1]
        if o == "=":
            return FilterContext.pred(Predicate("EQ", col, [v]))
        if o in ("!=", "<>"):
            return FilterContext.pred(Predicate("NOT_EQ", col, [v]))
        if o == ">":
            return FilterContext.pred(Predicate("RANGE", col, [], v, UNBOUNDED, False, False))
        if o == ">=":
            return FilterContext.pred(Predicate("RANGE", col, [], v, UNBOUNDED, True, False))
        if o == "<":
            return FilterContext.pred(Predicate("RANGE", col, [], UNBOUNDED, v, False, False))
        if o == "<=":
            return FilterContext.pred(Predicate("RANGE", col, [], UNBOUNDED, v, False, True))
        raise SqlError(f"unsupported operator {o}")

    # --- select -------------------------------------------------------------------------------------------------
    def select_item(self, q: QueryContext):
        t = self.take()
        if t[0] != "id":
            raise SqlError(f"bad select item {t}")
        if self.peek() == ("op", "("):
            fn = t[1].upper()
            self.i += 1
            if self.peek() == ("op", "*"):
                self.i += 1
                col = None
            else:
                c = self.take()
                if c[0] != "id":
                    raise SqlError(f"bad aggregation argument {c}")
                col = c[1]
            log2m = 0
            if self.peek() == ("op", ","):
                self.i += 1
                log2m = int(self.literal())
            self.expect_op(")")
            if fn not in capi.AGG_FUNCTIONS:
                raise SqlError(f"unsupported aggregation function {fn}")
            q.aggregations.append(AggregationSpec(fn, col, log2m))
        else:
            q.select_columns.append(t[1])
        if self.kw("AS"):
            self.i += 2

    def parse(self) -> QueryContext:
        q = QueryContext()
        self.expect_kw("SELECT")
        self.select_item(q)
        while self.peek() == ("op", ","):
            self.i += 1
            self.select_item(q)
        self.expect_kw("FROM")
        q.table = self.take()[1]
        if self.kw("WHERE"):
            self.i += 1
            q.filter = self.or_expr()
        if self.kw("GROUP"):
            self.i += 1
            self.expect_kw("BY")
            q.has_group_by = True
            q.group_by.append(self.take()[1])
            while self.peek() == ("op", ","):
                self.i += 1
                q.group_by.append(self.take()[1])
        if self.kw("ORDER"):
            self.i += 1
            self.expect_kw("BY")
            while True:
                t = self.take()
                text = t[1]
                if self.peek() == ("op", "("):
                    depth = 0
                    while True:
                        u = self.take()
                        text += u[1]
                        if u == ("op", "("):
                            depth += 1
                        if u == ("op", ")"):
                            depth -= 1
                            if depth == 0:
                                break
                asc = True
                if self.kw("DESC"):
                    self.i += 1
                    asc = False
                elif self.kw("ASC"):
                    self.i += 1
                nulls_last = None
                if self.kw("NULLS"):
                    self.i += 1
                    if self.kw("LAST"):
                        nulls_last = True
                    elif self.kw("FIRST"):
                        nulls_last = False
                    else:
                        raise SqlError("NULLS FIRST or NULLS LAST expected")
                    self.i += 1
                q.order_by.append((text, asc))
                q.order_by_nulls_last.append(nulls_last)
                if self.peek() == ("op", ","):
                    self.i += 1
                    continue
                break
        if self.kw("LIMIT"):
            self.i += 1
            q.limit = int(self.literal())
        if self.peek()[0] != "eof":
            raise SqlError(f"trailing tokens: {self.toks[self.i:]}")
        return q


def parse_sql(sql: str) -> QueryContext:
    return _Parser(sql).parse()


# ----------------------------------------------------------------------------------------------------------------------
# QueryContext → C structs (keeps every backing object alive on the returned holder)
# ----------------------------------------------------------------------------------------------------------------------
_FILTER_TYPES = {"AND": capi.FILTER_AND, "OR": capi.FILTER_OR, "NOT": capi.FILTER_NOT,
                 "PREDICATE": capi.FILTER_PREDICATE, "CONSTANT_TRUE": capi.FILTER_CONSTANT_TRUE,
                 "CONSTANT_FALSE": capi.FILTER_CONSTANT_FALSE}
_PRED_TYPES = {"EQ": capi.PRED_EQ, "NOT_EQ": capi.PRED_NOT_EQ, "IN": capi.PRED_IN, "NOT_IN": capi.PRED_NOT_IN,
               "RANGE": capi.PRED_RANGE, "IS_NULL": capi.PRED_IS_NULL, "IS_NOT_NULL": capi.PRED_IS_NOT_NULL}


class CQuery:
    def __init__(self, q: QueryContext):
        self._keep = []
        self.query = capi.PgQuery()
        if q.filter is not None:
            root = capi.PgFilterNode()
            self._fill(root, q.filter)
            self._keep.append(root)
            self.query.filter = C.pointer(root)
        else:
            self.query.filter = None
        ng = len(q.group_by)
        self.query.n_group_by = ng
        if ng:
            arr = (C.c_char_p * ng)(*[g.encode() for g in q.group_by])
            self._keep.append(arr)
            self.query.group_by_columns = arr
        na = len(q.aggregations)
        self.query.n_aggregations = na
        if na:
            aggs = (capi.PgAggSpec * na)()
            for i, a in enumerate(q.aggregations):
                aggs[i].function = capi.AGG_FUNCTIONS[a.function]
                aggs[i].log2m = a.log2m
                aggs[i].column = a.column.encode() if a.column else None
            self._keep.append(aggs)
            self.query.aggregations = aggs
        self.query.num_groups_limit = q.num_groups_limit
        self.query.max_initial_result_holder_capacity = q.max_initial_result_holder_capacity
        self.query.flags = q.flags
        # segment-level group trim (GroupByOperator.java:120-133): ORDER BY + LIMIT + minSegmentGroupTrimSize travel with group-by queries
        ob = q.resolved_order_by() if (ng and q.order_by) else None
        self.query.limit = q.limit
        self.query.min_segment_group_trim_size = q.min_segment_group_trim_size
        if ob:
            arr = (capi.PgOrderBy * len(ob))()
            for i, (kind, index, asc) in enumerate(ob):
                nl = q.order_by_nulls_last[i] if i < len(q.order_by_nulls_last) else None
                arr[i].kind, arr[i].index, arr[i].ascending, arr[i].nulls_last = kind, index, int(asc), int(asc if nl is None else nl)   # NULLS LAST for ASC is the default
            self._keep.append(arr)
            self.query.order_by = arr
            self.query.n_order_by = len(ob)

    def _fill(self, node: capi.PgFilterNode, f: FilterContext):
        node.type = _FILTER_TYPES[f.type]
        n = len(f.children)
        node.n_children = n
        if n:
            arr = (capi.PgFilterNode * n)()
            self._keep.append(arr)
            for i, ch in enumerate(f.children):
                self._fill(arr[i], ch)
            node.children = arr
        if f.type == "PREDICATE":
            p = f.predicate
            node.predicate_type = _PRED_TYPES[p.type]
            node.column = p.column.encode()
            node.n_values = len(p.values)
            if p.values:
                vals = (C.c_char_p * len(p.values))(*[v.encode() for v in p.values])
                self._keep.append(vals)
                node.values = vals
            node.lower = p.lower.encode()
            node.upper = p.upper.encode()
            node.lower_inclusive = int(p.lower_inclusive)
            node.upper_inclusive = int(p.upper_inclusive)

    def ptr(self):
        return C.byref(self.query)

    def filter_ptr(self):
        return self.query.filter
