"""Loads the CPU oracle (oracle/_build/liboracle.so) behind the same ctypes wrapper the product library uses.
Test infrastructure only — nothing under pinot_amd/ imports this."""
import ctypes as C
import os
import subprocess

from pinot_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.environ.get("PO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "_build", "liboracle.so")   # PO_ORACLE_LIB: the sanitizer build (tools/sanitize.sh)

_api = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load_oracle() -> capi.NativeApi:
    global _api
    if _api is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if (not os.path.exists(ORACLE_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
            build_oracle()
        _api = capi.NativeApi(ORACLE_LIB, "po_")
        lib = _api.lib
        lib.po_hll_cardinality_from_registers.restype = C.c_int64
        lib.po_hll_cardinality_from_registers.argtypes = [C.c_void_p, C.c_int32]
        lib.po_hll_registers_for_values.restype = None
        lib.po_hll_registers_for_values.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        lib.po_read_fixed_bit.restype = C.c_int32
        lib.po_read_fixed_bit.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.po_read_var_bytes.restype = C.c_int32
        lib.po_read_var_bytes.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_int32]
        lib.po_read_fixed_bit_block.restype = None
        lib.po_read_fixed_bit_block.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    return _api
