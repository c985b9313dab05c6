"""Cross-GPU group-by merge: the one exchange step of the path.

Segments shard one per GPU (one process per GPU); each rank runs the segment query on its own segment and the ranks then
merge their intermediate group tables — what GroupByCombineOperator.processSegments/mergeResults + IndexedTable#upsert do
across worker threads in the reference (pinot-core/.../operator/combine/GroupByCombineOperator.java:102-165,191-222;
merge functions: SumAggregationFunction.java:223-233, MaxAggregationFunction.java:237-251, MinAggregationFunction,
CountAggregationFunction, AvgAggregationFunction#merge).

Keys must be *values*, not dictIds, because dictionaries are per segment (GroupByCombineOperator.java:135-144).  Two forms:

  * `DenseGroupTable` + `all_reduce_tables()` — when every segment shares the key space (identical dictionaries, e.g. the
    synthetic gpuBench table, or a caller-supplied global key order): the intermediates are dense `[n_rows, G]` float64 /
    int64 arrays laid out by raw key Σ dictId_j·Π card_<j, and the merge is ONE collective: the SUM-like, MAX and MIN rows
    are packed into one buffer, all-gathered over RCCL/xGMI (`backend="nccl"`) or gloo on CPU, and reduced locally per row
    class.  The payload is a few KB–MB, i.e. latency-bound: one small collective beats three all-reduces.
  * `gather_merge()` — the general case: every rank's (key values → intermediates) rows are gathered to rank 0 with
    `gather_object` and upserted with `executor.GroupByCombineOperator` (IndexedTable semantics).  DISTINCTCOUNT sets are
    variable length and always take this form.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from .executor import GroupByCombineOperator, ResultsBlock, merge_intermediate

_SUM_LIKE = ("COUNT", "SUM")


@dataclass
class DenseGroupTable:
    """Dense intermediate table of one segment: row r of `sum_rows` / `max_rows` / `min_rows` per aggregation component."""
    functions: List[str]                 # aggregation function per query aggregation
    cards: List[int]                     # cardinality of each group-by column (shared key space)
    sum_rows: np.ndarray                 # [n_sum, G] float64 — COUNT, SUM, AVG.sum, AVG.count, and the presence row
    max_rows: np.ndarray                 # [n_max, G] float64 — MAX, MINMAXRANGE.max
    min_rows: np.ndarray                 # [n_min, G] float64 — MIN, MINMAXRANGE.min
    layout: List[tuple]                  # per aggregation: list of (kind, row) components

    @property
    def n_groups(self) -> int:
        g = 1
        for c in self.cards:
            g *= c
        return g


def _layout(functions: Sequence[str]):
    layout, n_sum, n_max, n_min = [], 1, 0, 0    # sum row 0 = presence (number of segments that saw the group)
    for f in functions:
        if f in _SUM_LIKE:
            layout.append([("sum", n_sum)]); n_sum += 1
        elif f == "MAX":
            layout.append([("max", n_max)]); n_max += 1
        elif f == "MIN":
            layout.append([("min", n_min)]); n_min += 1
        elif f == "AVG":
            layout.append([("sum", n_sum), ("sum", n_sum + 1)]); n_sum += 2
        elif f == "MINMAXRANGE":
            layout.append([("min", n_min), ("max", n_max)]); n_min += 1; n_max += 1
        else:
            raise ValueError(f"{f} has no fixed-shape intermediate: use gather_merge()")
    return layout, n_sum, n_max, n_min


def dense_from_block(block: ResultsBlock, cards: Sequence[int]) -> DenseGroupTable:
    """Scatters a segment's ResultsBlock into the dense layout (raw key = Σ dictId_j · Π card_<j, column 0 least
    significant — DictionaryBasedGroupKeyGenerator.java:312-323)."""
    functions = [a.function for a in block.query.aggregations]
    layout, n_sum, n_max, n_min = _layout(functions)
    g = 1
    for c in cards:
        g *= c
    sum_rows = np.zeros((n_sum, g), dtype=np.float64)
    max_rows = np.full((n_max, g), -np.inf, dtype=np.float64)
    min_rows = np.full((n_min, g), np.inf, dtype=np.float64)
    ids = block.group_dict_ids
    key = np.zeros(ids.shape[1] if ids is not None and ids.size else (0 if cards else 1), dtype=np.int64)
    if not cards:
        key = np.zeros(1, dtype=np.int64)
    mult = 1
    for j, c in enumerate(cards):
        key += ids[j].astype(np.int64) * mult
        mult *= c
    sum_rows[0, key] = 1.0
    rows = {"sum": sum_rows, "max": max_rows, "min": min_rows}
    for a, comps in enumerate(layout):
        arr = block.arrays[a]
        for ci, (kind, r) in enumerate(comps):
            rows[kind][r, key] = arr[1 + ci].astype(np.float64, copy=False)
    return DenseGroupTable(functions, list(cards), sum_rows, max_rows, min_rows, layout)


def all_reduce_tables(t: DenseGroupTable, device=None) -> DenseGroupTable:
    """In-place merge across the default process group with ONE collective: the SUM / MAX / MIN rows are packed into one
    buffer, all-gathered (the payload is KBs, i.e. latency-bound: one small collective beats three), and every rank reduces
    the gathered copies locally with the row class's merge function (SumAggregationFunction#merge `+`,
    MaxAggregationFunction#merge max, MinAggregationFunction#merge min).  `device` = torch device holding the buffers
    during the collective ("cuda:N" for RCCL over xGMI, None/cpu for gloo)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    world = dist.get_world_size()
    n_sum, n_max, n_min = t.sum_rows.shape[0], t.max_rows.shape[0], t.min_rows.shape[0]
    packed = np.concatenate([t.sum_rows, t.max_rows, t.min_rows], axis=0)
    x = torch.from_numpy(packed)
    if device is not None:
        x = x.to(device)
    x = x.reshape(-1)
    gathered = torch.empty(world * x.numel(), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(gathered, x)          # flat buffers: the one form both RCCL and gloo accept
    g = gathered.cpu().numpy().reshape(world, packed.shape[0], packed.shape[1])   # one D2H; the KB-sized reduce runs on the host
    if n_sum:
        t.sum_rows[...] = g[:, :n_sum].sum(axis=0)
    if n_max:
        t.max_rows[...] = g[:, n_sum:n_sum + n_max].max(axis=0)
    if n_min:
        t.min_rows[...] = g[:, n_sum + n_max:].min(axis=0)
    return t


def rows_from_dense(t: DenseGroupTable, dict_values: Sequence[Sequence]) -> Dict[tuple, list]:
    """Dense table → {decoded key tuple: intermediates} for the groups at least one segment produced."""
    present = np.flatnonzero(t.sum_rows[0] > 0)
    out: Dict[tuple, list] = {}
    rows = {"sum": t.sum_rows, "max": t.max_rows, "min": t.min_rows}
    for k in present.tolist():
        key, rem = [], k
        for j, c in enumerate(t.cards):
            key.append(dict_values[j][rem % c])
            rem //= c
        vals = []
        for f, comps in zip(t.functions, t.layout):
            comp = [float(rows[kind][r, k]) for kind, r in comps]
            if f == "COUNT":
                vals.append(int(comp[0]))
            elif f == "AVG":
                vals.append((comp[0], int(comp[1])))
            elif f == "MINMAXRANGE":
                vals.append((comp[0], comp[1]))
            else:
                vals.append(comp[0])
        out[tuple(key)] = vals
    return out


def gather_merge(block: ResultsBlock, dst: int = 0) -> Optional[Dict[tuple, list]]:
    """General merge: gather every rank's decoded rows on `dst` and upsert them (IndexedTable semantics).  Returns the
    merged table on `dst`, None elsewhere."""
    import torch.distributed as dist
    rows = block.rows()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(rows, gathered, dst=dst)
    if dist.get_rank() != dst:
        return None
    fns = [a.function for a in block.query.aggregations]
    table: Dict[tuple, list] = {}
    for part in gathered:
        for key, vals in part.items():
            cur = table.get(key)
            table[key] = list(vals) if cur is None else [merge_intermediate(f, x, y) for f, x, y in zip(fns, cur, vals)]
    return table
