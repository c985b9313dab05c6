"""Malformed index buffers at the boundary (SURVEY §5 "Race detection / sanitizers"): pg_segment_add_column / _add_star_tree / _set_null_vector /
_set_queryable_doc_ids / _set_range_index parse Roaring, RangeBitmap, chunk-header, dictionary, multi-value and star-tree bytes handed over by the caller, and
pg_decompress.hip runs LZ4 / Snappy decoders over them on the device.  Thousands of truncated and bit-flipped copies of valid buffers
(tests/malformed_worker.py) must come back as PG_OK / PG_ERR_INVALID_ARGUMENT / PG_ERR_UNSUPPORTED — never a crash, a hang or a device
fault — and a column that was accepted must answer queries (or refuse).  The same worker runs under AddressSanitizer +
UndefinedBehaviorSanitizer against libpinot_gpu_san.so in tools/sanitize.sh (profiles/r05_sanitizers.txt)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
def test_malformed_buffers_never_crash():
    p = subprocess.run([sys.executable, os.path.join(HERE, "malformed_worker.py")], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    t = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert t["registrations"] > 6000 and t["invalid"] > 2500 and t["ok"] > 1000 and t["queries"] > 3000, t   # both outcomes occur: the fuzz reaches the parsers and the kernels
