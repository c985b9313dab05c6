"""bench.py's contract on a small segment: the one JSON line with `roofline` and `cpu_baseline`, and the same script started by
`python -m torch.distributed.run` the way the driver starts the multi-GPU runs (one rank here: the box has one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline"]


def last_json_line(out):
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[-1])


def test_bench_line_on_a_small_segment():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "3000000", "--steps", "3", "--warmup", "1", "--no-traffic",
                          "--cpu-sample-docs", "1000000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json_line(out)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "rows/s" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert c["kind"] == "port" and c["cores"] == 1 and c["gpu_equals_oracle_on_sample"] is True and c["value"] > 0
    # BASELINE configs 2 and 5 ride in every default run: kernel time, roofline fraction and an oracle equality flag each
    for key in ("cfg2", "cfg2_dict", "cfg5_flat"):
        b = d[key]
        assert b["kernel_ms"] > 0 and 0 < b["roofline_frac"] < 1 and b["rows"] == 3000000
        assert b.get("gpu_equals_oracle_at_full_size", b.get("gpu_equals_oracle_on_sample")) is True
    # 3 M docs are under 16 offers per HLL register (12 800 groups x 256): the partition pipeline; the 10^9-doc run takes the pruned-offer
    # passes (pg_kernels_oct.hip; tests/test_gpu_full_size.py asserts that name)
    assert d["cfg5_flat"]["kernel"] == "pg_part_group_by"
    st = d["cfg5_star_tree"]
    assert st["groups"] == 12800 and st["star_tree_index"] == 0 and st["device_ms"] > 0
    # the library merge with nothing to exchange (communicator of one rank): its latency is on record, its result the unmerged one
    m = d["merge_world_of_one"]
    assert m.get("error") is None and m["p50_ms"] > 0 and m["equals_unmerged_result"] is True


def test_bench_under_torchrun_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--docs", "1000000", "--steps", "2",
                          "--warmup", "1", "--no-traffic", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = last_json_line(out)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and "roofline" in d
