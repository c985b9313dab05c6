#!/usr/bin/env python3
"""Headline benchmark: rows scanned/sec of the filter → group-by hot path on 1 B-row synthetic segments (BASELINE.json).

A "step" is one pass of the hot path over one batch of synthetic input = one `pg_query_exec` of the config-3 query
    SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)
                                            AND r_int BETWEEN 250000 AND 749999 GROUP BY g1
over one 1 B-row segment per GPU, columns already resident in HBM (2 inverted-index predicates + 1 raw-INT range scan,
group by a 100-value dictionary column, SUM/MAX of a raw INT metric), followed — when N > 1 — by the cross-GPU group-by
merge inside the library (pg_result_all_reduce: one grouped RCCL all-reduce over the dense accumulator table the kernels left
in HBM; segments share dictionaries).  Segments shard one per GPU (`scaling: weak`), no data-path collective other than that
merge.  `--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (HBM-bound: algorithmic bytes per launch ÷ HIP-event
kernel time ÷ 8 TB/s) and `cpu_baseline` (the C restatement of the reference algorithm, oracle/, one thread per segment
as the reference runs it, on a bounded prefix sample of the same segment).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CFG3_BYTES_PER_ROW = 9.625     # postings 6/8 + r_int 4 + g1 7/8 + m 4
NORTH_STAR_BYTES_PER_ROW = 10.375
CFG3_DICT_BYTES_PER_ROW = 6.625   # config 3 in Pinot's default encoding: postings 6/8 + r_int_d 20/8 + g1 7/8 + m_d 20/8


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=int(os.environ.get("PG_BENCH_DOCS", "1000000000")),
                    help="rows per segment (BASELINE config 3: 1e9)")
    ap.add_argument("--no-validation", action="store_true", help="N > 1: skip the parity checks of the merged table (multi_gpu_validation)")
    ap.add_argument("--query", choices=["cfg3", "northstar", "cfg2", "cfg5", "cfg3_dict", "cfg3_sparse"], default="cfg3",
                    help="cfg5: BASELINE config 5 — 4-dim GROUP BY (12 800 groups) + DISTINCTCOUNTHLL, flat segment timed like the others, "
                         "plus the star-tree route's latency (its cost does not depend on the parent segment's size)")
    ap.add_argument("--cpu-sample-docs", type=int, default=100_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in their own runs) that measure roofline.traffic")
    ap.add_argument("--no-full-check", action="store_true",
                    help="skip the one full-size oracle run that checks the timed GPU result (about 4 s of CPU per 1e9 rows)")
    ap.add_argument("--no-variants", action="store_true", help="skip the north-star (2-key) variant of the default run")
    ap.add_argument("--no-concurrency", action="store_true", help="skip the concurrent-callers block of the default run")
    ap.add_argument("--require-library-merge", action="store_true",
                    help="N > 1: exit non-zero when the library's own RCCL communicator cannot be created on every rank, instead of "
                         "falling back to the torch collective + host reduce")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not launched by torchrun: spawn the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus and rank == 0:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher decides the rank count")

    from pinot_amd import capi, distributed as pd, synth
    from pinot_amd.executor import Comm, NativeSegment
    from pinot_amd.query import parse_sql
    from pinot_amd.segment import HostSegment

    api = capi.gpu_api()
    api.call("init", local_rank)

    # the library's own RCCL communicator for the data-path merge: rank 0's unique id travels over the torch process group
    comm = None
    merge_kind = "none (1 segment)"
    merge_diag = None
    if world > 1:
        comm_error = ""
        try:
            uid = torch.zeros(capi.COMM_UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.frombuffer(bytearray(Comm.unique_id(api)), dtype=torch.uint8).cuda()
            dist.broadcast(uid, src=0)
            comm = Comm.init_rank(api, local_rank, world, rank, bytes(uid.cpu().numpy().tobytes()))
            ok = torch.ones(1, device="cuda")
        except Exception as e:   # noqa: BLE001 — an unusable RCCL setup must not sink the run: fall back to the torch collective
            log(f"library RCCL communicator unavailable on rank {rank}: {e}")
            comm_error = f"{type(e).__name__}: {e}"
            ok = torch.zeros(1, device="cuda")
        # which ranks hold a library communicator (all of them, or the run falls back / stops): on record in config.merge_diagnostics
        flags = [torch.zeros(1, device="cuda") for _ in range(world)]
        dist.all_gather(flags, ok)
        errors = [None] * world
        dist.all_gather_object(errors, comm_error)
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            rccl_version = "unknown"
        merge_diag = {"ranks_with_library_communicator": [r for r in range(world) if flags[r].item() >= 1],
                      "errors": {str(r): e for r, e in enumerate(errors) if e}, "rccl_version": rccl_version,
                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
        if rank == 0:
            log(f"library communicator on ranks {merge_diag['ranks_with_library_communicator']} of {world}, RCCL {rccl_version}")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() < 1:
            if comm is not None:
                comm.destroy()
            comm = None
            if args.require_library_merge:
                if rank == 0:
                    log(f"--require-library-merge: no library communicator on every rank: {merge_diag}")
                dist.destroy_process_group()
                raise SystemExit(3)
        merge_kind = "pg_result_all_reduce (RCCL inside libpinot_gpu)" if comm is not None else \
            "torch.distributed all_gather_into_tensor (nccl backend) + host reduce"

    sql = {"cfg3": synth.QUERY_CFG3, "northstar": synth.QUERY_NORTH_STAR, "cfg2": synth.QUERY_CFG2, "cfg5": synth.QUERY_CFG5,
           "cfg3_dict": synth.QUERY_CFG3_DICT, "cfg3_sparse": synth.QUERY_CFG3_SPARSE}[args.query]
    # cfg 5 flat = h1..h4 (4+4+4+3 bits) + u (20 bits) = 4.375 B/row
    bytes_per_row = {"cfg3": CFG3_BYTES_PER_ROW, "northstar": NORTH_STAR_BYTES_PER_ROW, "cfg2": 4.0, "cfg5": 4.375,
                     "cfg3_dict": CFG3_DICT_BYTES_PER_ROW, "cfg3_sparse": CFG3_DICT_BYTES_PER_ROW}[args.query]
    needed = {"cfg3": ["c_inv1", "c_inv2", "r_int", "g1", "m"], "northstar": ["c_inv1", "c_inv2", "r_int", "g1", "g2", "m"],
              "cfg2": ["r_int"], "cfg5": list(synth.CFG5_COLUMNS), "cfg3_dict": ["c_inv1", "c_inv2", "r_int_d", "g1", "m_d"],
              "cfg3_sparse": ["c_inv1", "c_inv2", "r_int_s", "g1", "m_s"]}[args.query]

    # ---- build this rank's segment (segment index = rank) and pin it in HBM, one column at a time ----------------------
    t0 = time.time()
    seg = NativeSegment(api, HostSegment(f"gpuBench_{rank}", args.docs))
    for name in needed:
        # N ranks generate at once on one host: each takes its share of the cores
        one = synth.generate_segment(args.docs, segment_index=rank, columns=[name], threads=max(1, (os.cpu_count() or 8) // max(world, 1)) if world > 1 else 0)
        seg.add_column(one.columns[name], keep_host_buffers=False)
        del one
    log(f"segment of {args.docs} docs generated + uploaded in {time.time() - t0:.1f}s, "
        f"{seg.device_bytes() / 1e9:.2f} GB in HBM")

    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_PROFILE
    cards = [synth.GPU_BENCH[g].range for g in qc.group_by]
    kernel_ms = []

    def run_query(q, times):
        """One pass of the hot path: segment query on this GPU + cross-GPU merge of the group table (GroupByCombineOperator over
        xGMI: identical dictionaries ⇒ the dense accumulator tables merge element-wise, in HBM, in one grouped RCCL launch)."""
        if comm is not None:
            nr = seg.execute_native(q, keep_device_table=True)
            times.append(nr.stats().device_ms_aggregate)     # this rank's kernel, before the merge
            nr.all_reduce(comm)
            block = nr.block()
            nr.free()
            return block, block.stats
        block = seg.execute(q)
        st = block.stats
        times.append(st.device_ms_aggregate)
        if world > 1:   # fallback merge: one all-gather of the packed dense rows + host reduce
            crd = [synth.GPU_BENCH[g].range for g in q.group_by]
            dense = pd.dense_from_block(block, crd)
            pd.all_reduce_tables(dense, device=torch.device("cuda", local_rank))
            return dense, st
        return block, st

    def step():
        return run_query(qc, kernel_ms)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dense, st = step()
    kernel_ms.clear()
    lat = []
    sync()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        dense, st = step()
        lat.append((time.perf_counter() - t1) * 1e3)
    sync()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_rows = float(args.docs) * world * args.steps
    value = total_rows / elapsed
    avg_kernel_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    alg_bytes = bytes_per_row * args.docs
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9 if avg_kernel_ms > 0 else 0.0
    lib_alg = st.algorithmic_bytes
    # HBM bytes per launch of the timed kernel, measured NOW: this same command re-run under rocprofv3 with FETCH_SIZE and
    # WRITE_SIZE in passes of their own (kernel-trace only), corrected as MI355X_MICROARCH.md §HBM prescribes; null when
    # rocprofv3 is not available
    traffic, traffic_detail = None, None
    if rank == 0 and world == 1 and not args.no_traffic:
        traffic, traffic_detail = measure_traffic(args.query, args.docs, st.kernel.decode() or "pg_generic_query_l")
    out = {
        "metric": {"cfg3": "rows scanned/sec, 1B-row segment filter+groupby (3 predicates, SUM/MAX GROUP BY g1)",
                   "northstar": "rows scanned/sec, segment filter+groupby (3 predicates, SUM GROUP BY g1, g2)",
                   "cfg2": "rows scanned/sec, segment range-predicate COUNT(*)",
                   "cfg5": "rows scanned/sec, 4-dim GROUP BY (12 800 groups) + DISTINCTCOUNTHLL, flat segment",
                   "cfg3_dict": "rows scanned/sec, config 3 over dictionary-encoded scan and value columns (Pinot's default encoding)",
                   "cfg3_sparse": "rows scanned/sec, config 3 over dictionary-encoded columns whose dictionaries are not arithmetic"}[args.query],
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "p50_query_latency_ms": statistics.median(lat),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": f"{world} segment(s) x {args.docs} rows, one per GPU; {sql}",
                   "query": args.query, "rows_per_segment": args.docs, "parallelism": f"segment-per-gpu x{world}",
                   "merge": merge_kind, "merge_diagnostics": merge_diag,
                   "matched_docs_per_segment": int(st.num_docs_scanned),
                   "entries_scanned_in_filter": int(st.num_entries_scanned_in_filter)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_detail": traffic_detail,
                     "kernel": st.kernel.decode() or "pg_generic_query_l", "kernel_ms": avg_kernel_ms,
                     "algorithmic_bytes_per_launch": alg_bytes, "library_accounted_bytes": int(lib_alg)},
    }

    if args.query == "cfg3" and not args.no_variants:
        # the north-star's own wording of the target — 2-key GROUP BY g1, g2 with SUM only — on the same segment (+ g2), same
        # timing discipline; reported next to the headline so both readings of BASELINE.json are on record
        one = synth.generate_segment(args.docs, segment_index=rank, columns=["g2"])
        seg.add_column(one.columns["g2"], keep_host_buffers=False)
        del one
        qn = parse_sql(synth.QUERY_NORTH_STAR)
        qn.flags |= capi.QUERY_FLAG_PROFILE
        cards_n = [synth.GPU_BENCH[g].range for g in qn.group_by]

        def step_n():
            t = []
            run_query(qn, t)
            return t[0]

        for _ in range(args.warmup):
            step_n()
        sync()
        t_n = time.perf_counter()
        kms = [step_n() for _ in range(args.steps)]
        sync()
        el_n = time.perf_counter() - t_n
        if world > 1:
            tt = torch.tensor([el_n], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_n = float(tt.item())
        k_n = sum(kms) / len(kms)
        out["north_star_variant"] = {
            "query": synth.QUERY_NORTH_STAR, "value": float(args.docs) * world * args.steps / el_n, "unit": "rows/s",
            "ms_per_step": el_n / args.steps * 1e3, "kernel_ms": k_n,
            "roofline_frac": NORTH_STAR_BYTES_PER_ROW * args.docs / (k_n * 1e-3) / 1e9 / HBM_PEAK_GBS if k_n > 0 else 0.0,
            "algorithmic_bytes_per_launch": NORTH_STAR_BYTES_PER_ROW * args.docs}

    if world > 1 and not args.no_validation:
        # N > 1: the line must carry its own parity flags — every rank takes part, rank 0 reports
        v = validate_multi_gpu(api, args, seg, qc, sql, needed, comm, dense, rank, world, local_rank)
        if rank == 0:
            out["multi_gpu_validation"] = v
    if args.query == "cfg3" and rank == 0 and world == 1 and not args.no_variants:
        out["merge_world_of_one"] = world_of_one_merge(api, seg, qc, local_rank)
    if args.query == "cfg3" and not args.no_variants and rank == 0 and world == 1 and not args.no_concurrency:
        t_blocks = time.time()
        out["concurrency"] = concurrency_block(api, args, bg_seg=seg, bg_sql=synth.QUERY_CFG3)
        log(f"concurrency block {time.time() - t_blocks:.1f}s")
    if args.query == "cfg3" and not args.no_variants and rank == 0 and world == 1:
        # BASELINE configs 2 and 5 next to the headline (same timing discipline, their own segments): every default run carries them
        headline_rows = dense.rows() if hasattr(dense, "rows") else None
        seg.destroy()
        seg = None
        t_blocks = time.time()
        out["cfg2"] = extra_block(api, args, "cfg2", min(args.docs, 100_000_000), 4.0, ["r_int"], synth.QUERY_CFG2)
        log(f"cfg2 block {time.time() - t_blocks:.1f}s")
        t_blocks = time.time()
        # config 2 in Pinot's DEFAULT encoding: the range predicate as a dictId interval over the 20-bit stream of r_int_d (2.5 B/row); no PMC
        # passes for this block (the kernel streams the column once: pg_dictrange_fo, pg_kernels_scan.hip)
        import copy
        a2 = copy.copy(args)
        a2.no_traffic = True
        out["cfg2_dict"] = extra_block(api, a2, "cfg2_dict", min(args.docs, 100_000_000), 2.5, ["r_int_d"],
                                       "SELECT COUNT(*) FROM gpuBench WHERE r_int_d BETWEEN 250000 AND 749999")
        log(f"cfg2_dict block {time.time() - t_blocks:.1f}s")
        t_blocks = time.time()
        # config 3 in Pinot's DEFAULT encoding (DictionaryIndexConfig.java:32): the same docs, r_int and m as 20-bit dictId streams — the range is
        # a dictId interval, SUM / MAX read dictionary.get(dictId).  Identity dictionaries (every value of the range occurs at this size), so
        # the rows must equal the headline's; then the same with dictionaries that are not arithmetic (values gathered)
        out["cfg3_dict"] = extra_block(api, args, "cfg3_dict", args.docs, CFG3_DICT_BYTES_PER_ROW, ["c_inv1", "c_inv2", "r_int_d", "g1", "m_d"],
                                       synth.QUERY_CFG3_DICT, same_rows_as=headline_rows)
        log(f"cfg3_dict block {time.time() - t_blocks:.1f}s")
        t_blocks = time.time()
        out["cfg3_sparse_dictionaries"] = extra_block(api, args, "cfg3_sparse", args.docs, CFG3_DICT_BYTES_PER_ROW,
                                                      ["c_inv1", "c_inv2", "r_int_s", "g1", "m_s"], synth.QUERY_CFG3_SPARSE)
        log(f"cfg3_sparse block {time.time() - t_blocks:.1f}s")
        t_blocks = time.time()
        out["cfg5_flat"] = extra_block(api, args, "cfg5", args.docs, 4.375, list(synth.CFG5_COLUMNS), synth.QUERY_CFG5)
        log(f"cfg5_flat block {time.time() - t_blocks:.1f}s")
        t_blocks = time.time()
        out["cfg5_star_tree"] = star_tree_leg(api, args, parent_docs=400_000)
        log(f"cfg5_star_tree block {time.time() - t_blocks:.1f}s")
    if args.query == "cfg5" and rank == 0 and not args.no_variants:   # (not in the PMC child runs: they only need the flat query's kernels)
        out["star_tree_route"] = star_tree_leg(api, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.query != "cfg5":
        out["cpu_baseline"] = cpu_baseline(args, sql, dense, seg)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_cfg5(args, sql)
    if rank == 0:
        print(json.dumps(out), flush=True)
        # the JSON line is the LAST thing on stdout: librccl prints a version banner from a destructor at exit (seen behind the line in round 5)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
    if seg is not None:
        seg.destroy()
    if world > 1:
        dist.destroy_process_group()


def validate_multi_gpu(api, args, seg, qc, sql, columns, comm, merged, rank, world, local_rank):
    """Parity of an N-GPU step, checked inside the run that is timed (no timing here):
      merged_equals_elementwise_merge   the table the timed step returned (pg_result_all_reduce over RCCL, or the fallback) against an
                                        independent merge of the ranks' UNMERGED results: every rank's rows gathered by torch.distributed
                                        (gather_object) and upserted by value on rank 0 (GroupByCombineOperator: IndexedTable semantics)
      merged_equals_oracle_on_prefix    the same pipeline (segment query per rank + library merge) on the first `sample` rows of every
                                        rank's segment against the CPU oracle's per-segment blocks combined by value
    Both are False when any rank disagrees; the flags travel in the JSON line (multi_gpu_validation)."""
    import torch.distributed as dist
    from pinot_amd import synth
    from pinot_amd.executor import GroupByCombineOperator, NativeSegment
    from pinot_amd.query import parse_sql
    from tests.oracle_binding import load_oracle
    out = {}
    # ---- (1) the timed result against the by-value merge of the unmerged per-rank results ------------------------------------------------
    own = seg.execute(parse_sql(sql))
    rows_all = [None] * world   # (all_gather_object: the collective this script already relies on under the nccl backend)
    dist.all_gather_object(rows_all, own.rows())
    merged_rows = merged.rows() if hasattr(merged, "rows") else None
    # (the torch fallback merge hands back a DenseGroupTable, not rows: the flag is then null and `library_merge` false)
    if rank == 0:
        fns = [a.function for a in own.query.aggregations]
        expect = {}
        from pinot_amd.executor import merge_intermediate
        for rows in rows_all:
            for k, vals in rows.items():
                if k in expect:
                    expect[k] = [merge_intermediate(f, a, b) for f, a, b in zip(fns, expect[k], vals)]
                else:
                    expect[k] = list(vals)
        out["merged_equals_elementwise_merge"] = None if merged_rows is None else bool(merged_rows == expect)
        out["groups"] = len(expect)
    # ---- (2) the pipeline on a prefix of every segment against the oracle ---------------------------------------------------------------------
    sample = min(args.docs, 10_000_000)
    host = synth.generate_segment(sample, segment_index=rank, columns=columns,
                                  threads=max(1, (os.cpu_count() or 8) // max(world, 1)))
    gp = NativeSegment(api, host)
    q = parse_sql(sql)
    if comm is not None:
        nr = gp.execute_native(q, keep_device_table=True)
        nr.all_reduce(comm)
        got = nr.block().rows()
        nr.free()
    else:
        got = None
    ora = NativeSegment(load_oracle(), host)
    ob = ora.execute(sql)
    oracle_rows, got_all = [None] * world, [None] * world
    dist.all_gather_object(oracle_rows, ob.rows())
    dist.all_gather_object(got_all, got)
    if rank == 0:
        fns = [a.function for a in ob.query.aggregations]
        from pinot_amd.executor import merge_intermediate
        expect = {}
        for rows in oracle_rows:
            for k, vals in rows.items():
                expect[k] = [merge_intermediate(f, a, b) for f, a, b in zip(fns, expect[k], vals)] if k in expect else list(vals)
        out["merged_equals_oracle_on_prefix"] = bool(comm is not None and all(g == expect for g in got_all))   # EVERY rank holds the merged table
        out["oracle_prefix_rows_per_segment"] = sample
        out["library_merge"] = comm is not None
    gp.destroy()
    ora.destroy()
    return out


def world_of_one_merge(api, seg, qc, device):
    """Latency of the library merge itself with nothing to exchange: pg_result_all_reduce over a communicator of ONE rank (the grouped
    RCCL launch, the signature check on the host, the commit and the reassembly of the rows), on the headline query's result — what a
    rank pays per query for the collective path before any xGMI transfer.  The result must equal the unmerged one."""
    from pinot_amd.executor import Comm
    try:
        comm = Comm.init_rank(api, device, 1, 0, Comm.unique_id(api))
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    ms = []
    same = True
    plain = seg.execute(qc).rows()
    for i in range(12):
        nr = seg.execute_native(qc, keep_device_table=True)
        t = time.perf_counter()
        nr.all_reduce(comm)
        ms.append((time.perf_counter() - t) * 1e3)
        if i == 0:
            same = nr.block().rows() == plain
        nr.free()
    comm.destroy()
    ms = ms[2:]
    return {"p50_ms": statistics.median(ms), "min_ms": min(ms), "max_ms": max(ms), "equals_unmerged_result": bool(same),
            "what": "pg_result_all_reduce, communicator of 1 rank, config 3 result (dense table of 100 groups x 2 accumulators + counts)"}


def concurrency_block(api, args, bg_seg=None, bg_sql=None, threads=(1, 4, 16, 64), seconds=1.0):
    """Many callers at once — the reference runs one worker thread per segment and query (BaseCombineOperator.java:97-142), the
    boundary gives every (thread, device) pair its own stream: N NATIVE host threads (pinot_amd/csrc/synth/pg_callers.cpp: no
    interpreter in the loop) call pg_query_exec on one segment for `seconds`; QPS, p50 and p99 per thread count for BASELINE config 2
    (10^8 rows, a 62 us kernel), the star-tree route of config 5 (launch-bound) and config 3 at 10^8 rows (a 0.15 ms kernel); then 16
    callers of each again while ONE more thread loops the headline's 10^9-row query (`bg_seg`) in the background."""
    import ctypes as C
    import numpy as np
    from pinot_amd import capi, startree, synth
    from pinot_amd.executor import NativeSegment
    from pinot_amd.query import CQuery, parse_sql
    lib = synth.synth_lib()
    if lib is None or not hasattr(lib, "pgs_concurrent_callers"):
        return {"skipped": "libpinot_synth.so lacks pgs_concurrent_callers"}
    fn = lib.pgs_concurrent_callers
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int64,
                   C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    exec_addr = C.cast(api.f("query_exec"), C.c_void_p).value
    free_addr = C.cast(api.f("result_free"), C.c_void_p).value
    docs = min(args.docs, 100_000_000)
    parent = synth.generate_segment(400_000, segment_index=0, columns=list(synth.CFG5_COLUMNS), native=False)
    startree.add_star_tree(parent, ["h1", "h2", "h3", "h4"], [("COUNT", "*"), ("DISTINCTCOUNTHLL", "u")], max_leaf_records=10000)
    q5 = parse_sql(synth.QUERY_CFG5)
    q5.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    workloads = [("cfg2_1e8", NativeSegment(api, synth.generate_segment(docs, columns=["r_int"])), parse_sql(synth.QUERY_CFG2)),
                 ("cfg5_star_tree", NativeSegment(api, parent), q5),
                 ("cfg3_1e8", NativeSegment(api, synth.generate_segment(docs, columns=synth.CFG3_COLUMNS)), parse_sql(synth.QUERY_CFG3))]
    cap = 1 << 20
    lat = np.empty(cap, dtype=np.float32)
    bg_cq = CQuery(parse_sql(bg_sql)) if bg_seg is not None else None

    def run(seg, cq, n, background):
        n_done, bg_done, err = C.c_int64(), C.c_int64(), C.c_int32()
        window = fn(exec_addr, free_addr, seg.handle, C.cast(cq.ptr(), C.c_void_p), n, 0.25, seconds, lat.ctypes.data, cap, C.byref(n_done),
                    bg_seg.handle if background else None, C.cast(bg_cq.ptr(), C.c_void_p) if background else None, C.byref(bg_done), C.byref(err))
        if window < 0:
            return {"threads": n, "error": int(err.value)}
        v = np.sort(lat[:min(n_done.value, cap)])
        r = {"threads": n, "qps": n_done.value / window, "p50_ms": float(v[len(v) // 2]) if len(v) else None,
             "p99_ms": float(v[min(len(v) - 1, int(len(v) * 0.99))]) if len(v) else None, "calls": int(n_done.value)}
        if background:
            r["background_scans_per_s"] = bg_done.value / window
        return r

    out = {"how": f"native caller threads, {seconds:.1f} s per point after 0.25 s warm-up, one GPU, one segment per workload; "
                  "latency = pg_query_exec call to return", "workloads": {}}
    for name, seg, qc in workloads:
        cq = CQuery(qc)
        points = [run(seg, cq, n, False) for n in threads]
        w = {"query": " ".join(str(x) for x in (qc.group_by, [a.function for a in qc.aggregations])), "points": points}
        one = next((p for p in points if p.get("threads") == 1 and "qps" in p), None)
        sixteen = next((p for p in points if p.get("threads") == 16 and "qps" in p), None)
        if one and sixteen:
            w["qps_16_over_1"] = sixteen["qps"] / one["qps"]
        if bg_seg is not None:
            w["with_1e9_row_scan_in_background"] = run(seg, cq, 16, True)
        out["workloads"][name] = w
    for _, seg, _ in workloads:
        seg.destroy()
    return out


def abi_call_latency(api, seg, qc, warmup, steps):
    """p50 of the C-ABI call alone — pg_query_exec entry to return, the result handle freed outside the timed region — which is what the
    reference-side caller (the JNI stub of INTEGRATION.md) pays; seg.execute() below it also walks the result into Python objects."""
    from pinot_amd.query import CQuery
    cq = CQuery(qc)
    exec_fn, free_fn = api.f("query_exec"), api.f("result_free")
    ptr = cq.ptr()
    lat = []
    for i in range(warmup + steps):
        h = C.c_void_p()
        t = time.perf_counter()
        rc = exec_fn(seg.handle, ptr, C.byref(h))
        dt = (time.perf_counter() - t) * 1e3
        if rc != 0:
            raise RuntimeError(f"pg_query_exec failed with {rc}")
        free_fn(h)
        if i >= warmup:
            lat.append(dt)
    return statistics.median(lat)


def extra_block(api, args, query, docs, bytes_per_row, columns, sql, same_rows_as=None):
    """One more BASELINE configuration inside the default run: its own segment of `docs` rows pinned in HBM, `steps` timed executions
    after `warmup`, HIP-event kernel time → roofline fraction, HBM traffic from the PMC passes (child runs), and an oracle equality
    flag — the whole segment for config 2 (0.3 s of CPU), a 10 M-row prefix for config 5 (its oracle needs minutes at 1 B rows) plus
    a full-size invariant (the groups' counts add up to the segment's rows)."""
    from pinot_amd import capi, synth
    from pinot_amd.executor import NativeSegment
    from pinot_amd.query import parse_sql
    from pinot_amd.segment import HostSegment
    from tests.oracle_binding import load_oracle
    t0 = time.time()
    seg = NativeSegment(api, HostSegment(f"gpuBench_{query}", docs))
    for name in columns:
        one = synth.generate_segment(docs, segment_index=0, columns=[name])
        seg.add_column(one.columns[name], keep_host_buffers=False)
        del one
    steps = max(5, args.steps // 2)
    kms, lat = [], []
    block = None
    q_plain = parse_sql(sql)                # the query as a caller issues it (no event records): the latency
    for i in range(args.warmup + steps):
        t = time.perf_counter()
        block = seg.execute(q_plain)
        if i >= args.warmup:
            lat.append((time.perf_counter() - t) * 1e3)
    # ... then with the library's HIP events around the kernels: the kernel time.  (In this order since round 6: the block's segment has just
    # been generated, and the first launches after it ran up to 25 % slower than the steady state — profiles/r06_u_cfg3_dict_kernel_stats.txt:
    # min 1.10 ms, max 1.53 ms over one run's 25 launches — which a block of five timed launches right behind it mostly measured.)
    q = parse_sql(sql)
    q.flags |= capi.QUERY_FLAG_PROFILE
    for i in range(args.warmup + steps):
        block = seg.execute(q)
        if i >= args.warmup:
            kms.append(block.stats.device_ms_aggregate)
    k = sum(kms) / len(kms)
    kernel = block.stats.kernel.decode()
    res = {"query": sql, "rows": docs, "steps": steps, "kernel": kernel, "kernel_ms": k, "ms_per_step": sum(lat) / len(lat),
           "p50_query_latency_ms": statistics.median(lat), "p50_abi_call_ms": abi_call_latency(api, seg, parse_sql(sql), args.warmup, max(steps, 20)), "value": docs / (sum(lat) / len(lat) * 1e-3), "unit": "rows/s",
           "algorithmic_bytes_per_launch": bytes_per_row * docs,
           "roofline_frac": bytes_per_row * docs / (k * 1e-3) / 1e9 / HBM_PEAK_GBS if k > 0 else 0.0}
    # oracle equality
    sample = docs if query == "cfg2" else min(docs, 10_000_000)
    host = synth.generate_segment(sample, segment_index=0, columns=columns)
    ora = NativeSegment(load_oracle(), host)
    ob = ora.execute(sql)
    if sample == docs:
        res["gpu_equals_oracle_at_full_size"] = block.rows() == ob.rows() and block.stats.num_docs_scanned == ob.stats.num_docs_scanned
    else:
        gp = NativeSegment(api, host)
        gb = gp.execute(sql)
        res["gpu_equals_oracle_on_sample"] = gb.rows() == ob.rows()
        res["oracle_sample_rows"] = sample
        gp.destroy()
        rows = block.rows()
        if query == "cfg5":
            res["full_size_invariant"] = {"sum_of_group_counts": int(sum(v[0] for v in rows.values())), "rows": docs, "groups": len(rows)}
        if same_rows_as is not None:   # the same docs under another encoding: the headline's rows (checked against the oracle at full size)
            res["rows_equal_the_raw_column_query_at_full_size"] = bool(rows == same_rows_as)
    ora.destroy()
    del host
    seg.destroy()
    if not args.no_traffic:
        traffic, detail = measure_traffic(query, docs, kernel)
        res["traffic"] = traffic
        res["traffic_over_algorithmic"] = traffic / (bytes_per_row * docs) if traffic else None
        res["traffic_detail"] = detail
    res["wall_s"] = time.time() - t0
    return res


def star_tree_leg(api, args, parent_docs=2_000_000):
    """BASELINE config 5 as stated: the same query answered from a star-tree over (h1, h2, h3, h4) holding count__* and
    distinctCountHLL__u.  The route reads the star-tree's pre-aggregated docs only, so its latency is the same for a 2 M-doc and a
    1 B-doc parent: measured on a 2 M-doc parent (the host-side tree builder is test tooling, not sized for 1 B rows)."""
    from pinot_amd import capi, startree, synth
    from pinot_amd.executor import NativeSegment
    from pinot_amd.query import parse_sql
    parent = synth.generate_segment(parent_docs, segment_index=0, columns=list(synth.CFG5_COLUMNS), native=False)
    startree.add_star_tree(parent, ["h1", "h2", "h3", "h4"], [("COUNT", "*"), ("DISTINCTCOUNTHLL", "u")], max_leaf_records=10000)
    seg = NativeSegment(api, parent)
    def timed(flags):
        q = parse_sql(synth.QUERY_CFG5)
        q.flags |= capi.QUERY_FLAG_PROFILE | flags
        lat, dev, lib = [], [], []
        b = None
        for i in range(args.warmup + args.steps):
            t = time.perf_counter()
            b = seg.execute(q)
            if i >= args.warmup:
                lat.append((time.perf_counter() - t) * 1e3)
                dev.append(b.stats.device_ms_total)
                lib.append(b.stats.host_ms_total)      # pg_query_exec wall time (planning, launches, device, copy back, assembly)
        return b, statistics.median(lat), statistics.median(lib), statistics.median(dev)

    # the query as BASELINE states it returns DISTINCTCOUNTHLL values: with PG_QUERY_FLAG_FINAL_DISTINCT the registers stay in HBM and the
    # cardinalities come back (the headline latency); the intermediate form (registers, what a cross-segment merge needs) is timed beside it
    bf, lat_f, lib_f, dev_f = timed(capi.QUERY_FLAG_FINAL_DISTINCT)
    b, lat_i, lib_i, dev_i = timed(0)
    q_abi = parse_sql(synth.QUERY_CFG5)        # without the profiling events: the call as the JNI stub issues it
    q_abi.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    abi_f = abi_call_latency(api, seg, q_abi, args.warmup, max(args.steps, 50))
    abi_i = abi_call_latency(api, seg, parse_sql(synth.QUERY_CFG5), args.warmup, max(args.steps, 50))
    from pinot_amd.executor import hll_cardinality
    inter, final = b.rows(), bf.rows()   # (rows() assembles a dict per call: once each)
    final_ok = len(final) == len(inter) and all(final[k] == [v[0], hll_cardinality(v[1])] for k, v in inter.items())
    out = {"star_tree_docs": int(parent.star_trees[0].num_docs), "groups": len(inter), "star_tree_index": int(b.stats.star_tree_index),
           "p50_query_latency_ms": lat_f, "p50_abi_call_ms": abi_f, "library_ms": lib_f, "device_ms": dev_f,
           "latency_how": "p50_query_latency_ms: Python binding, call + result walked into Python objects, profiling events on; "
                          "p50_abi_call_ms: pg_query_exec entry to return through ctypes, no events (the JNI caller's cost)",
           "final_values_equal_cardinality_of_registers": bool(final_ok),
           "intermediate_registers": {"p50_query_latency_ms": lat_i, "p50_abi_call_ms": abi_i, "library_ms": lib_i, "device_ms": dev_i},
           "kernel": b.stats.kernel.decode(),
           "docs_scanned": int(b.stats.num_docs_scanned), "parent_docs": parent.total_docs}
    seg.destroy()
    return out


def cpu_baseline_cfg5(args, sql):
    from pinot_amd import capi, synth
    from pinot_amd.executor import NativeSegment
    from tests.oracle_binding import load_oracle
    sample = min(args.docs, 20_000_000)
    host = synth.generate_segment(sample, segment_index=0, columns=list(synth.CFG5_COLUMNS))
    ora = NativeSegment(load_oracle(), host)
    times = []
    for _ in range(3):
        t = time.perf_counter()
        block = ora.execute(sql)
        times.append(time.perf_counter() - t)
    gpu = NativeSegment(capi.gpu_api(), host)
    gb = gpu.execute(sql)
    assert gb.rows() == block.rows(), "GPU result on the CPU sample differs from the oracle"
    gpu.destroy()
    ora.destroy()
    med = statistics.median(times)
    return {"value": sample / med, "unit": "rows/s", "cores": 1, "kind": "port", "seconds_per_run": med, "gpu_equals_oracle_on_sample": True,
            "sample": f"first {sample} docs of segment 0, flat config-5 query, median of 3 runs; C restatement of the reference operators (oracle/)"}


def measure_traffic(query, docs, kernel):
    """HBM bytes per launch of `kernel`: two child runs of this script under `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE
    costs 3 of the 4 TCC slots and WRITE_SIZE 2, so each gets its own pass; no other trace domain is enabled).  On gfx950 FETCH_SIZE
    reports half the bytes of wide coalesced reads (TCC_EA0_RDREQ x 64 B for 128-byte requests): doubled; WRITE_SIZE as reported.
    Both are in KiB."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pg_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "x", "--", sys.executable, os.path.abspath(__file__),
               "--docs", str(docs), "--query", query, "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-variants",
               "--no-traffic"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            db = sqlite3.connect(dbs[0])
            if kernel.endswith("_group_by"):   # a pipeline of kernels (radix / hash group-by): all of them, per query execution
                rows = list(db.execute("select sum(value) from counters_collection where counter_name = ? and (kernel_name like 'pg_radix%' "
                                       "or kernel_name like 'pg_p2%' or kernel_name like 'pg_oct%' or kernel_name like 'pg_hash%' or kernel_name like 'pg_fast_%_f' "
                                       "or kernel_name like 'pg_generic_query_f')", (counter,)))
                per[counter] = float(rows[0][0]) / 7.0    # 2 warm-up + 5 timed executions in the child run
            else:
                rows = list(db.execute(
                    "select avg(v) from (select dispatch_id, sum(value) as v from counters_collection where kernel_name = ? and counter_name = ? "
                    "group by dispatch_id)", (kernel, counter)))
                per[counter] = float(rows[0][0])
        except Exception as e:   # noqa: BLE001
            return None, f"{counter} pass failed: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    read_bytes = per["FETCH_SIZE"] * 1024.0 * 2.0
    write_bytes = per["WRITE_SIZE"] * 1024.0
    return read_bytes + write_bytes, {"kernel": kernel, "FETCH_SIZE_KiB": per["FETCH_SIZE"], "WRITE_SIZE_KiB": per["WRITE_SIZE"],
                                      "read_bytes": read_bytes, "write_bytes": write_bytes,
                                      "how": "rocprofv3 --pmc <counter> --kernel-trace, one pass per counter, 5 timed launches each; FETCH_SIZE x 2 (gfx950)"}


def cpu_baseline(args, sql, gpu_block, gpu_seg):
    """CPU leg: the C restatement of the reference algorithm (oracle/, kind "port"), 1 thread per segment exactly like
    the reference's combine operator, timed on a prefix sample of the same segment; the sample is also run through the GPU
    library and compared bit for bit.  Unless --no-full-check, ONE more oracle run over the WHOLE segment checks the timed GPU
    result at full size (group keys, SUM / MAX values, ExecutionStatistics)."""
    from pinot_amd import distributed as pd, synth
    from pinot_amd.executor import NativeSegment
    from tests.oracle_binding import load_oracle
    sample = min(args.docs, args.cpu_sample_docs)
    needed = sorted({c for c in synth.CFG3_COLUMNS} | ({"r_int_d", "m_d"} if args.query == "cfg3_dict" else set()) | ({"r_int_s", "m_s"} if args.query == "cfg3_sparse" else set()))
    host = synth.generate_segment(sample, segment_index=0, columns=needed)
    ora = NativeSegment(load_oracle(), host)
    times = []
    block = None
    for _ in range(5):
        t = time.perf_counter()
        block = ora.execute(sql)
        times.append(time.perf_counter() - t)
    med = statistics.median(times)
    full_checked = False
    if sample == args.docs:   # the sample is the whole segment: check the timed GPU result right away
        assert gpu_block.rows() == block.rows(), "GPU result differs from the oracle"
        full_checked = True
    # parity at sample scale in every run: the same prefix segment through the GPU library must equal the oracle bit for bit
    # (group keys, SUM / MAX values, ExecutionStatistics)
    from pinot_amd import capi
    gpu_prefix = NativeSegment(capi.gpu_api(), host)
    gb = gpu_prefix.execute(sql)
    assert gb.rows() == block.rows(), "GPU result on the CPU sample differs from the oracle"
    assert (gb.stats.num_docs_scanned, gb.stats.num_entries_scanned_in_filter) == \
        (block.stats.num_docs_scanned, block.stats.num_entries_scanned_in_filter), "ExecutionStatistics differ from the oracle"
    gpu_prefix.destroy()
    ora.destroy()
    full_seconds = None
    if not full_checked and not args.no_full_check:
        del host
        full = synth.generate_segment(args.docs, segment_index=0, columns=needed)
        ora_full = NativeSegment(load_oracle(), full)
        t = time.perf_counter()
        fb = ora_full.execute(sql)
        full_seconds = time.perf_counter() - t
        assert gpu_block.rows() == fb.rows(), "GPU result at full size differs from the oracle"
        assert (gpu_block.stats.num_docs_scanned, gpu_block.stats.num_entries_scanned_in_filter) == \
            (fb.stats.num_docs_scanned, fb.stats.num_entries_scanned_in_filter), "full-size ExecutionStatistics differ from the oracle"
        ora_full.destroy()
        del full
        full_checked = True
    return {"value": sample / med, "unit": "rows/s", "cores": 1, "kind": "port", "gpu_equals_oracle_at_full_size": full_checked,
            "full_size_oracle_seconds": full_seconds,
            "sample": f"first {sample} docs of segment 0, same query, median of 5 runs; C restatement of the reference "
                      f"operators (oracle/), not the JVM", "seconds_per_run": med,
            "gpu_equals_oracle_on_sample": True}


if __name__ == "__main__":
    main()
