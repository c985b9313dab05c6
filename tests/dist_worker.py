"""Worker of tests/test_distributed.py: one process per (virtual) GPU, gloo backend, oracle as the segment executor."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from pinot_amd import distributed as pd, synth
    from pinot_amd.executor import GroupByCombineOperator, NativeSegment
    from tests.oracle_binding import load_oracle

    docs, out_path = int(sys.argv[1]), sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    api = load_oracle()
    failures = []
    queries = [synth.QUERY_CFG3, synth.QUERY_NORTH_STAR,
               "SELECT g2, COUNT(*), MIN(r_int), MAX(m), AVG(m), MINMAXRANGE(r_int) FROM gpuBench WHERE c_inv1 = 3 GROUP BY g2",
               "SELECT COUNT(*), SUM(m), MAX(m) FROM gpuBench WHERE r_int < 1000"]
    # every rank can rebuild every segment (pure function of the seed), so rank 0 can check against the host merge
    mine = NativeSegment(api, synth.generate_segment(docs + 17 * rank, segment_index=rank, columns=synth.CFG3_COLUMNS))
    for q in queries:
        block = mine.execute(q)
        cards = [synth.GPU_BENCH[g].range for g in block.query.group_by]
        dense = pd.all_reduce_tables(pd.dense_from_block(block, cards))
        dicts = [mine.host.columns[g].dict_values for g in block.query.group_by]
        merged_dense = pd.rows_from_dense(dense, dicts)
        merged_gather = pd.gather_merge(block)
        if rank == 0:
            blocks = []
            for r in range(world):
                seg = mine if r == 0 else NativeSegment(api, synth.generate_segment(docs + 17 * r, segment_index=r, columns=synth.CFG3_COLUMNS))
                blocks.append(seg.execute(q))
            expect = GroupByCombineOperator(blocks).merge()
            if merged_dense != expect:
                failures.append(("dense", q))
            if merged_gather != expect:
                failures.append(("gather", q))
    # bench.py's own N > 1 parity check (multi_gpu_validation), with the oracle as the per-rank executor and no library communicator:
    # the by-value merge of the ranks' unmerged rows must accept the right table and reject a wrong one
    import types
    import bench
    args = types.SimpleNamespace(docs=docs + 17 * rank)
    q = synth.QUERY_CFG3
    merged_rows = pd.gather_merge(mine.execute(q))          # rank 0 holds the merged rows
    merged_rows = merged_rows if rank == 0 else {}
    v = bench.validate_multi_gpu(api, args, mine, None, q, list(synth.CFG3_COLUMNS), None, types.SimpleNamespace(rows=lambda: merged_rows), rank, world, 0)
    wrong = dict(list(merged_rows.items())[1:])
    w = bench.validate_multi_gpu(api, args, mine, None, q, list(synth.CFG3_COLUMNS), None, types.SimpleNamespace(rows=lambda: wrong), rank, world, 0)
    if rank == 0:
        if v.get("merged_equals_elementwise_merge") is not True or v.get("library_merge") is not False:
            failures.append(("bench validation accepts the merged table", str(v)))
        if w.get("merged_equals_elementwise_merge") is not False:
            failures.append(("bench validation rejects a wrong table", str(w)))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"failures": failures, "world": world}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
