"""The collective test double (tests/fake_rccl) checked on its own, on host buffers: N threads of one process rendezvous, the
reductions are the NCCL ones, a grouped launch is one rendezvous, and the two situations in which real RCCL hangs — a rank alone in a
collective, ranks enqueueing different collectives — come back as errors on EVERY rank.  The product path over it runs on the GPU
(tests/test_gpu_fake_rccl.py)."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
INT64, UINT8 = 4, 1
SUM, MAX, MIN = 0, 2, 3


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def fake():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fake_rccl")])
    os.environ["FAKE_RCCL_HOST_BUFFERS"] = "1"
    os.environ["FAKE_RCCL_TIMEOUT_MS"] = "400"
    lib = C.CDLL(LIB)
    lib.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    for f in ("fake_rccl_lonely_ranks", "fake_rccl_mismatched_collectives", "fake_rccl_collectives"):
        getattr(lib, f).restype = C.c_int64
    return lib


def _init_all(lib, n):
    comms = (C.c_void_p * n)()
    devs = (C.c_int * n)(*([0] * n))
    assert lib.ncclCommInitAll(comms, n, devs) == 0
    return [C.c_void_p(comms[i]) for i in range(n)]


def _run(n, body):
    out = [None] * n

    def work(r):
        out[r] = body(r)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    return out


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_grouped_launch_reduces_like_nccl(fake, world):
    comms = _init_all(fake, world)
    rng = np.random.default_rng(world)
    sums = [rng.integers(-2**62, 2**62, 1000).astype(np.int64) for _ in range(world)]     # wraps like the hardware's
    mins = [rng.integers(-2**63, 2**63 - 1, 77).astype(np.int64) for _ in range(world)]
    regs = [rng.integers(0, 64, 4096).astype(np.uint8) for _ in range(world)]
    sets = [rng.integers(0, 2**32, 33, dtype=np.uint64).astype(np.uint32) for _ in range(world)]
    got = {}

    def body(r):
        s, m, x, g = sums[r].copy(), np.empty(77, np.int64), np.empty(77, np.int64), np.empty(4096, np.uint8)
        gathered = np.empty(33 * world, np.uint32)
        assert fake.ncclGroupStart() == 0
        assert fake.ncclAllReduce(s.ctypes.data, s.ctypes.data, 1000, INT64, SUM, comms[r], None) == 0        # in place
        assert fake.ncclAllReduce(mins[r].ctypes.data, m.ctypes.data, 77, INT64, MIN, comms[r], None) == 0
        assert fake.ncclAllReduce(mins[r].ctypes.data, x.ctypes.data, 77, INT64, MAX, comms[r], None) == 0
        assert fake.ncclAllReduce(regs[r].ctypes.data, g.ctypes.data, 4096, UINT8, MAX, comms[r], None) == 0
        assert fake.ncclAllGather(sets[r].ctypes.data, gathered.ctypes.data, 33 * 4, UINT8, comms[r], None) == 0
        rc = fake.ncclGroupEnd()
        got[r] = (s, m, x, g, gathered)
        return rc
    assert _run(world, body) == [0] * world
    with np.errstate(over="ignore"):
        want_sum = np.sum(np.stack(sums).astype(np.uint64), axis=0, dtype=np.uint64).astype(np.int64)
    for r in range(world):
        s, m, x, g, gathered = got[r]
        assert np.array_equal(s, want_sum)
        assert np.array_equal(m, np.min(np.stack(mins), axis=0))
        assert np.array_equal(x, np.max(np.stack(mins), axis=0))
        assert np.array_equal(g, np.max(np.stack(regs), axis=0))
        assert np.array_equal(gathered, np.concatenate(sets))
    for c in comms:
        fake.ncclCommDestroy(c)
    assert fake.fake_rccl_lonely_ranks() == 0 and fake.fake_rccl_mismatched_collectives() == 0


def test_init_rank_is_collective_and_reuses_nothing(fake):
    uid = UniqueId()
    assert fake.ncclGetUniqueId(C.byref(uid)) == 0
    world = 3
    vals = [np.array([r + 1], np.int64) for r in range(world)]

    def body(r):
        c = C.c_void_p()
        assert fake.ncclCommInitRank(C.byref(c), world, uid, r) == 0
        for _ in range(5):   # back-to-back launches reuse the staging slots
            assert fake.ncclAllReduce(vals[r].ctypes.data, vals[r].ctypes.data, 1, INT64, SUM, c, None) == 0
        fake.ncclCommDestroy(c)
        return int(vals[r][0])
    assert _run(world, body) == [6 * 3**4] * world


def test_different_collectives_fail_on_every_rank(fake):
    comms = _init_all(fake, 2)
    before = fake.fake_rccl_mismatched_collectives()
    bufs = [np.zeros(16, np.int64) for _ in range(2)]

    def body(r):
        return fake.ncclAllReduce(bufs[r].ctypes.data, bufs[r].ctypes.data, 8 + 8 * r, INT64, SUM, comms[r], None)
    assert _run(2, body) == [4, 4]
    assert fake.fake_rccl_mismatched_collectives() == before + 1

    def again(r):   # the communicator survives: the next, matching, launch works
        bufs[r][:] = r + 1
        return fake.ncclAllReduce(bufs[r].ctypes.data, bufs[r].ctypes.data, 16, INT64, MAX, comms[r], None)
    assert _run(2, again) == [0, 0]
    assert bufs[0][0] == bufs[1][0] == 2


def test_rank_alone_times_out_instead_of_hanging(fake):
    comms = _init_all(fake, 2)
    before = fake.fake_rccl_lonely_ranks()
    buf = np.zeros(4, np.int64)
    assert fake.ncclAllReduce(buf.ctypes.data, buf.ctypes.data, 4, INT64, SUM, comms[0], None) == 2   # after FAKE_RCCL_TIMEOUT_MS
    assert fake.fake_rccl_lonely_ranks() == before + 1
    # the world is poisoned: the late rank does not wait for anybody
    assert fake.ncclAllReduce(buf.ctypes.data, buf.ctypes.data, 4, INT64, SUM, comms[1], None) == 2
