"""Seeded random queries over one segment that holds every column kind on the path, plus a brute-force numpy evaluation of the
filter tree.  Used by tests/test_fuzz.py: oracle vs numpy on the CPU, HIP path vs oracle on the GPU (differential testing in the
spirit of the reference's BaseStarTreeV2Test / QueriesTestUtils comparisons)."""
import numpy as np

from pinot_amd import formats
from pinot_amd.query import AggregationSpec, FilterContext, Predicate, QueryContext, UNBOUNDED
from pinot_amd.segment import build_segment

N_DOCS = 120_000


def fuzz_segment(n=N_DOCS, seed=2024):
    rng = np.random.default_rng(seed)
    st_vals = np.array([f"k{v:02d}" for v in range(30)], dtype=object)
    data = {
        "ci": rng.integers(0, 8, n).astype(np.int32),
        "cj": rng.integers(0, 4, n).astype(np.int32),
        "g1": (rng.integers(0, 100, n) * 3 + 5).astype(np.int32),
        "g2": rng.integers(-25, 25, n).astype(np.int32),
        "st": st_vals[rng.integers(0, 30, n)],
        "so": np.sort(rng.integers(0, 200, n)).astype(np.int32),
        "u": rng.integers(0, 60_000, n).astype(np.int32),
        "lg": (rng.integers(-250, 250, n).astype(np.int64) * 16_777_259),
        "r": rng.integers(0, 1_000_000, n).astype(np.int32),
        "rl": (rng.integers(-1500, 1500, n).astype(np.int64) * 8_388_617),
        "rk": rng.integers(-100, 100, n).astype(np.int32),
        "rf": (rng.integers(-4000, 4000, n) * 0.25).astype(np.float32),
        "rd": rng.integers(-(1 << 28), 1 << 28, n) * 0.125,
        "m": rng.integers(0, 1 << 20, n).astype(np.int32),
    }
    schema = {"ci": "INT", "cj": "INT", "g1": "INT", "g2": "INT", "st": "STRING", "so": "INT", "u": "INT", "lg": "LONG",
              "r": "INT", "rl": "LONG", "rk": "INT", "rf": "FLOAT", "rd": "DOUBLE", "m": "INT"}
    host = build_segment("fuzz_0", data, schema, inverted_index_columns=["ci", "cj", "st"],
                         no_dictionary_columns=["r", "rl", "rk", "rf", "rd", "m"])
    nulls = {"ci": np.flatnonzero(rng.random(n) < 0.05), "r": np.arange(n // 3, n // 3 + n // 10)}
    for c, ids in nulls.items():
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(ids), dtype=np.uint8)
    return host, data, nulls


DICT_GROUP = ["ci", "cj", "g1", "g2", "st", "so", "lg"]
FILTER_COLS = ["ci", "cj", "g1", "g2", "st", "so", "u", "lg", "r", "rl", "rk", "rf", "rd", "m"]
NUMERIC_AGG = ["m", "r", "rl", "rk", "rf", "rd", "g1", "g2", "lg", "u", "so"]


def _lit(v):
    if isinstance(v, (np.floating, float)):
        return repr(float(v))
    if isinstance(v, (np.integer, int)):
        return str(int(v))
    return str(v)


class Gen:
    def __init__(self, data, seed):
        self.data = data
        self.rng = np.random.default_rng(seed)

    def value_near(self, col):
        """a literal: mostly a value that occurs, sometimes one that does not"""
        v = self.data[col]
        x = v[self.rng.integers(0, len(v))]
        if self.rng.random() < 0.2 and not isinstance(x, str):
            x = x + (1 if v.dtype.kind in "iu" else 0.0625)
        elif self.rng.random() < 0.1 and isinstance(x, str):
            x = x + "z"
        return x

    def predicate(self):
        col = FILTER_COLS[self.rng.integers(0, len(FILTER_COLS))]
        kind = self.rng.choice(["EQ", "NOT_EQ", "IN", "NOT_IN", "RANGE", "RANGE", "RANGE", "NULL"])
        if kind == "NULL":
            col = ["ci", "r", "g1"][self.rng.integers(0, 3)]
            return FilterContext.pred(Predicate("IS_NULL" if self.rng.random() < 0.5 else "IS_NOT_NULL", col, []))
        if kind in ("EQ", "NOT_EQ"):
            return FilterContext.pred(Predicate(kind, col, [_lit(self.value_near(col))]))
        if kind in ("IN", "NOT_IN"):
            k = int(self.rng.integers(1, 6))
            return FilterContext.pred(Predicate(kind, col, [_lit(self.value_near(col)) for _ in range(k)]))
        a, b = self.value_near(col), self.value_near(col)
        if b < a:
            a, b = b, a
        shape = self.rng.integers(0, 4)
        lo, hi = (_lit(a), _lit(b)) if shape == 0 else (_lit(a), UNBOUNDED) if shape == 1 else (UNBOUNDED, _lit(b)) if shape == 2 else (_lit(a), _lit(b))
        return FilterContext.pred(Predicate("RANGE", col, [], lo, hi,
                                            lo != UNBOUNDED and bool(self.rng.integers(0, 2)),
                                            hi != UNBOUNDED and bool(self.rng.integers(0, 2))))

    def tree(self, depth):
        if depth == 0 or self.rng.random() < 0.35:
            return self.predicate()
        kind = self.rng.choice(["AND", "AND", "OR", "NOT"])
        if kind == "NOT":
            return FilterContext.not_(self.tree(depth - 1))
        kids = []
        for _ in range(int(self.rng.integers(2, 5))):
            k = self.tree(depth - 1)
            kids.extend(k.children if k.type == kind else [k])   # the SQL front end flattens nested AND / OR
        return FilterContext(kind, kids)

    def query(self):
        q = QueryContext(table="fuzz")
        if self.rng.random() < 0.85:
            q.filter = self.tree(int(self.rng.integers(0, 4)))
        x = self.rng.random()
        if x < 0.30:
            q.group_by = []
        elif x < 0.75:
            k = int(self.rng.integers(1, 4))
            q.group_by = list(self.rng.choice(DICT_GROUP, size=k, replace=False))
        elif x < 0.88:
            q.group_by = ["u"] + list(self.rng.choice(DICT_GROUP, size=int(self.rng.integers(0, 3)), replace=False))
            self.rng.shuffle(q.group_by)
        else:
            q.group_by = [["rk", "rl"][self.rng.integers(0, 2)]]
        q.has_group_by = bool(q.group_by)
        n_aggs = int(self.rng.integers(1, 5))
        for _ in range(n_aggs):
            fn = self.rng.choice(["COUNT", "SUM", "SUM", "MIN", "MAX", "AVG", "MINMAXRANGE", "DISTINCTCOUNT", "DISTINCTCOUNTHLL"])
            if fn == "COUNT":
                q.aggregations.append(AggregationSpec("COUNT", None))
            elif fn == "DISTINCTCOUNT":
                q.aggregations.append(AggregationSpec(fn, str(self.rng.choice(["g2", "ci", "st", "g1", "u"]))))
            elif fn == "DISTINCTCOUNTHLL":
                q.aggregations.append(AggregationSpec(fn, str(self.rng.choice(["u", "g1", "r", "lg"])), int(self.rng.choice([0, 0, 6, 10]))))
            else:
                q.aggregations.append(AggregationSpec(fn, str(self.rng.choice(NUMERIC_AGG))))
        if q.group_by and self.rng.random() < 0.25:
            q.num_groups_limit = int(self.rng.choice([3, 40, 700, 5000]))
        q.limit = 1_000_000
        return q


    def raw_key_query(self):
        """query() with GROUP BY over at least one no-dictionary column: a FLOAT / DOUBLE key alone, or raw and dictionary columns mixed."""
        q = self.query()
        raw = ["rk", "rl", "rf", "rd"]
        if self.rng.random() < 0.3:
            q.group_by = [["rf", "rd"][self.rng.integers(0, 2)]]
        else:
            k_raw = int(self.rng.integers(1, 3))
            k_dict = int(self.rng.integers(0, 3)) if k_raw > 1 else int(self.rng.integers(1, 3))
            q.group_by = list(self.rng.choice(raw, size=k_raw, replace=False)) + list(self.rng.choice(DICT_GROUP, size=k_dict, replace=False))
            self.rng.shuffle(q.group_by)
        q.has_group_by = True
        q.aggregations = [a for a in q.aggregations if a.function not in ("DISTINCTCOUNT", "DISTINCTCOUNTHLL")] or [AggregationSpec("COUNT", None)]
        q.num_groups_limit = int(self.rng.choice([0, 0, 40, 700, 5000]))
        return q


def clone(q: QueryContext) -> QueryContext:
    c = QueryContext(table=q.table, filter=q.filter, group_by=list(q.group_by), aggregations=list(q.aggregations), limit=q.limit,
                     num_groups_limit=q.num_groups_limit, has_group_by=q.has_group_by)
    return c


def describe(q: QueryContext) -> str:
    def f(t):
        if t.type == "PREDICATE":
            p = t.predicate
            if p.type == "RANGE":
                return f"{p.column} {'[' if p.lower_inclusive else '('}{p.lower},{p.upper}{']' if p.upper_inclusive else ')'}"
            return f"{p.column} {p.type} {p.values}"
        return t.type + "(" + ", ".join(f(c) for c in t.children) + ")"
    aggs = ", ".join(f"{a.function}({a.column or '*'}{',' + str(a.log2m) if a.log2m else ''})" for a in q.aggregations)
    return f"SELECT {aggs} WHERE {f(q.filter) if q.filter else '-'} GROUP BY {q.group_by} numGroupsLimit={q.num_groups_limit}"


# ---- brute force ---------------------------------------------------------------------------------------------------------------
def _typed(col_values, s):
    if col_values.dtype.kind in "iu":
        return int(s)
    if col_values.dtype.kind == "f":
        return np.float32(s) if col_values.dtype == np.float32 else float(s)
    return s


def eval_filter(t: FilterContext, data, nulls, n):
    if t.type == "AND":
        m = np.ones(n, bool)
        for c in t.children:
            m &= eval_filter(c, data, nulls, n)
        return m
    if t.type == "OR":
        m = np.zeros(n, bool)
        for c in t.children:
            m |= eval_filter(c, data, nulls, n)
        return m
    if t.type == "NOT":
        return ~eval_filter(t.children[0], data, nulls, n)
    p = t.predicate
    v = data[p.column]
    if p.type in ("IS_NULL", "IS_NOT_NULL"):
        m = np.zeros(n, bool)
        if p.column in nulls:
            m[nulls[p.column]] = True
        elif p.type == "IS_NULL":
            return m
        return m if p.type == "IS_NULL" else (~m if p.column in nulls else np.ones(n, bool))
    if v.dtype == object:
        v = v.astype(str)
    if p.type in ("EQ", "NOT_EQ", "IN", "NOT_IN"):
        vals = [_typed(data[p.column], s) for s in p.values]
        m = np.isin(v, np.array(vals) if v.dtype.kind == "U" else np.array(vals, dtype=v.dtype))   # (a fixed-width cast would truncate 'k25z' to 'k25')
        return m if p.type in ("EQ", "IN") else ~m
    m = np.ones(n, bool)
    if p.lower != UNBOUNDED:
        lo = _typed(data[p.column], p.lower)
        m &= (v >= lo) if p.lower_inclusive else (v > lo)
    if p.upper != UNBOUNDED:
        hi = _typed(data[p.column], p.upper)
        m &= (v <= hi) if p.upper_inclusive else (v < hi)
    return m
