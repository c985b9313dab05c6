"""pg_result_all_reduce with world 2 and 8 on a ONE-GPU box: N ranks (threads) of one process, all tables on device 0, the
collectives served by the test double tests/fake_rccl (selected with PG_RCCL_LIBRARY, which the library reads when it first opens
RCCL — hence a process of its own per world size).  What runs is the product's own control flow (pinot_amd/csrc/pg_comm.cpp: probe,
refusals decided on reduced values, grouped table launch, all-gather + OR of dictId sets); the double only moves and reduces bytes,
and turns the two ways real RCCL hangs into counted errors.  The real-RCCL twin of this test is test_all_reduce_two_devices
(tests/test_gpu_multi.py), which waits for a box with two GPUs."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_all_reduce_ranks_on_one_device(world):
    if not os.path.exists(FAKE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fake_rccl")])
    env = dict(os.environ, PG_RCCL_LIBRARY=FAKE, FAKE_RCCL_TIMEOUT_MS="30000")
    p = subprocess.run([sys.executable, os.path.join(HERE, "fake_rccl_worker.py"), str(world)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["world"] == world
    assert line["merged_queries"] == 7 + 3 + 1 + 1 and line["refusal_cases"] == 3
    # per-rank dictionaries (overlapping / disjoint / nested) x four group-by shapes merged by value, + once with heads that pg_result_merge re-keyed
    assert line["value_keyed_merges"] == 3 * 4 + 1
    assert line["concurrent_merges"] == 4 * 3   # four communicator sets, four merges at once, three times each
    assert line["lonely_ranks"] == 0, "a rank entered a collective alone (real RCCL would have hung)"
    assert line["mismatched_collectives"] == 0, "the ranks enqueued different collectives in one launch"
    # per merged query: the probe (2 collectives) + the table launch; per refusal: the probe only
    assert line["collectives"] >= world * (2 * (12 + 3) + 12)
