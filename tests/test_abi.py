"""CPU: the C-ABI library loads and exports every symbol include/swat_hip.h and include/swat_hip_tuning.h declare (no
compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("swat_hip.h", "swat_hip_tuning.h")):
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in headers)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swat_[a-z0-9_]+)\s*\(", text)))


def test_tuning_switches_are_not_in_the_public_header():
    public, tuning = _declared(("swat_hip.h",)), _declared(("swat_hip_tuning.h",))
    assert not [n for n in public if n.startswith("swat_gemm_set_") or n.startswith("swat_gemm_prof_")]
    assert {"swat_gemm_set_exp", "swat_gemm_set_tile", "swat_gemm_prof_begin", "swat_vit_set_split_k"} <= set(tuning)


def test_library_exports_every_declared_symbol():
    from swat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 4
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/swat_hip.h but not exported"
    assert h.swat_version() >= 1


def test_python_binding_covers_header():
    from swat_amd import _lib
    assert sorted(_lib.exported_symbols()) == _declared()


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    from swat_amd import _lib
    from swat_amd.retrieval import ops
    with pytest.raises(_lib.SwatHipError):
        ops.similarity_segmented(torch.zeros(4, 64), torch.zeros(1, 64))
