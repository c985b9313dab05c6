"""N > 1 path on CPU: world_size-2 gloo run of the cross-GPU group-by merge (pinot_amd/distributed.py) with the oracle
standing in for the per-GPU segment executor; rank 0 checks both merge forms against the host-side
GroupByCombineOperator over the same segments."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_group_by_merge(tmp_path, world):
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "70001", str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["world"] == world and res["failures"] == []


def test_dense_layout_single_process():
    """dense_from_block / rows_from_dense round trip without a process group."""
    from pinot_amd import distributed as pd, synth
    from pinot_amd.executor import NativeSegment
    from tests.oracle_binding import load_oracle
    seg = NativeSegment(load_oracle(), synth.generate_segment(30_011, columns=synth.CFG3_COLUMNS))
    for q in (synth.QUERY_NORTH_STAR, "SELECT g1, COUNT(*), AVG(m), MIN(m) FROM gpuBench WHERE c_inv2 = 1 GROUP BY g1"):
        b = seg.execute(q)
        cards = [synth.GPU_BENCH[g].range for g in b.query.group_by]
        dense = pd.all_reduce_tables(pd.dense_from_block(b, cards))
        assert pd.rows_from_dense(dense, [seg.host.columns[g].dict_values for g in b.query.group_by]) == b.rows()
