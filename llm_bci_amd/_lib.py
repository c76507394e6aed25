"""ctypes binding of libnbci.so (the C-ABI declared in include/nbci.h).

The library is built in-tree (llm_bci_amd/csrc/libnbci.so) by `__graft_entry__.build()` or
`make -C llm_bci_amd/csrc`. There is no CPU fallback: if the library is missing, every op
raises `NbciUnavailable` loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnbci.so")

NBCI_F32, NBCI_BF16 = 0, 1
ACT = {"identity": 0, None: 0, "none": 0, "softsign": 1, "gelu": 2, "relu": 3, "tanh": 4}


class NbciUnavailable(RuntimeError):
    pass


class NbciError(RuntimeError):
    pass


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("kmajor", C.c_int32), ("rpb", C.c_int32),
                ("gstride", C.c_int64), ("zs1", C.c_int64), ("zs2", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("in_dtype", C.c_int32),
                ("A", Operand), ("B", Operand),
                ("C", C.c_void_p), ("C2", C.c_void_p), ("ldc", C.c_int64),
                ("czs1", C.c_int64), ("czs2", C.c_int64), ("c_dtype", C.c_int32),
                ("batch", C.c_int32), ("zdiv", C.c_int32), ("splitk", C.c_int32),
                ("alpha", C.c_float), ("beta", C.c_float), ("bias", C.c_void_p),
                ("act", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32),
                ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("residual_rows", C.c_void_p), ("residual_first", C.c_int32),
                ("gate", C.c_void_p), ("ldg", C.c_int64), ("gate_act", C.c_int32)]


_lib = None


def lib():
    """Load libnbci.so once; raise NbciUnavailable (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NbciUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C llm_bci_amd/csrc` (needs hipcc). There is no CPU fallback for the product path.")
    l = C.CDLL(LIB_PATH)
    l.nbci_version.restype = C.c_int
    l.nbci_last_error.restype = C.c_char_p
    l.nbci_gemm.restype = C.c_int
    l.nbci_gemm.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    _bind_rest(l)
    _lib = l
    return l


def _bind_rest(l):
    """argtypes for the remaining entry points (filled in as they are added)."""
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = restype
        fn.argtypes = argtypes


_SIGNATURES = {}


def check(status, what=""):
    if status != 0:
        msg = lib().nbci_last_error().decode("utf-8", "replace")
        raise NbciError(f"{what} failed with status {status}: {msg}")


def exported_symbols():
    """Names declared in include/nbci.h (used by the CPU test that checks the .so exports them)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "nbci.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbci_[a-z0-9_]+)\s*\(", text)))
