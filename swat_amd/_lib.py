"""ctypes binding of libswat_hip.so (include/swat_hip.h).  Fails loudly when the library is absent."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libswat_hip.so")
_lib = None

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

HEADER = os.path.join(os.path.dirname(_HERE), "include", "swat_hip.h")
TUNING_HEADER = os.path.join(os.path.dirname(_HERE), "include", "swat_hip_tuning.h")  # measurement / tuning: not the boundary
_CTYPES = {"int": c_int, "int64_t": c_int64, "float": c_float, "swat_stream_t": c_void_p}


def _parse_header():
    """Prototypes straight from include/swat_hip.h, so the binding cannot drift from the ABI."""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read() + open(TUNING_HEADER).read(), flags=re.S)
    protos, restypes = {}, {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(swat_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret = ret.strip()
        argtypes = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            if "*" in a:
                argtypes.append(c_void_p)
            else:
                argtypes.append(_CTYPES[a.replace("const", "").split()[0]])
        protos[name] = argtypes
        restypes[name] = (ctypes.c_char_p if "char" in ret else c_void_p if "swat_stream_t" in ret
                          else c_int64 if "int64_t" in ret else c_int)
    return protos, restypes


_PROTOS, _RESTYPES = _parse_header()


class SwatHipError(RuntimeError):
    pass


def build():
    """Compile the HIP sources in swat_amd/csrc for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh")])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwatHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(swat_amd has no CPU or PyTorch fallback)")
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            fn = getattr(h, name)  # AttributeError if the .so is stale: loud by design
            fn.argtypes = argtypes
            fn.restype = _RESTYPES[name]
        _lib = h
    return _lib


def exported_symbols():
    return list(_PROTOS)


def check(rc, what):
    if rc != 0:
        raise SwatHipError(f"{what} failed ({rc}): {lib().swat_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SwatHipError("swat_amd ops need GPU tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise SwatHipError("swat_amd ops need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
