import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_api():
    """CPU oracle (oracle/): test infrastructure only."""
    from tests.oracle_binding import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def gpu_api():
    import torch  # noqa: F401  (initialises the ROCm runtime the same way bench.py does)
    from pinot_amd import capi
    api = capi.gpu_api()
    api.call("init", int(os.environ.get("LOCAL_RANK", "0")))
    return api


@pytest.fixture(scope="session")
def sv_data():
    z = np.load(os.path.join(ROOT, "tests", "golden", "test_data_sv.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture
def gpu_knobs(monkeypatch, gpu_api):
    """Sets PG_* knobs for one test: the library reads its environment once (pg_init), so a changed variable takes effect through
    pg_options_reload; the previous environment is restored (and re-read) afterwards."""
    def set_knobs(**kv):
        for k, v in kv.items():
            if v is None:            # flags are "variable present": None removes it
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        gpu_api.call("options_reload")
    yield set_knobs
    monkeypatch.undo()
    gpu_api.call("options_reload")
