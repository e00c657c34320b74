"""ctypes binding of libswat_hip.so (include/swat_hip.h).  Fails loudly when the library is absent."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SWAT_HIP_LIB: an alternate build of the same ABI, for A/B measurements — tools/micro/_ab/ — without touching the in-tree library)
LIB_PATH = os.environ.get("SWAT_HIP_LIB") or os.path.join(_HERE, "lib", "libswat_hip.so")
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


class PersistentForwardGaveUp(SwatHipError):
    pass


def persistent_forward_failed():
    """After a device synchronisation: did a persistent forward (csrc/vit_mega.hip) give up on a team hand-off since the last
    check?  If so its outputs are void: read and clear the status, switch this process to the op-by-op forward for good and
    return the 1-based number of the first failed launch; else 0."""
    import warnings
    h = lib()
    failed = h.swat_vit_mega_poll()
    if not failed:
        return 0
    st = h.swat_vit_mega_status()  # synchronises, clears the device word and the host flag
    h.swat_vit_set_mega(0)
    warnings.warn(f"vit_fwd_mega_kernel launch {failed} gave up on a team hand-off (status {st:#x}: a member workgroup was not "
                  "resident); the work since then is re-run and this process keeps to the op-by-op forward from here on")
    return failed


def retry_op_by_op(fn):
    """Inference entry points (validate, feature extraction, the zero-shot filter) end on a host synchronisation: check the
    persistent forward's status there and, on a give-up, run the whole call again on the op-by-op forward instead of returning
    features computed from unpublished activations."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        out = fn(*a, **k)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            if persistent_forward_failed():
                out = fn(*a, **k)
                torch.cuda.synchronize()
                if lib().swat_vit_mega_poll():
                    raise PersistentForwardGaveUp("the persistent forward gave up although it is switched off")
        return out
    return wrapped


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
