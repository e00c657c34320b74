"""Runs inside a python started with LD_PRELOAD=libclang_rt.asan (tests/test_sanitizer_host.py): drives the host side of the
ASan + UBSan build of the extension (tools/build_asan.sh) through every path reachable without a GPU.  Pointers handed to the
compute entry points are fake device addresses: host code must never dereference them — it plans, checks, and fails at the first
HIP call (no device here) with an error string.  Any sanitizer report aborts the process (exit != 0)."""
import ctypes, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swat_amd import _lib

h = ctypes.CDLL(sys.argv[1])
for name, argtypes in _lib._PROTOS.items():
    fn = getattr(h, name)
    fn.argtypes = argtypes
    fn.restype = _lib._RESTYPES[name]

FAKE = 0x7F0000000000  # never dereferenced by host code
P = lambda k=0: ctypes.c_void_p(FAKE + 4096 * k)
calls = fails = 0


def run(name, *args, expect_fail=True):
    global calls, fails
    rc = getattr(h, name)(*args)
    calls += 1
    if isinstance(rc, int) and rc != 0:
        fails += 1
        msg = h.swat_last_error()
        assert msg is not None and len(msg) > 0, f"{name}: failed without an error string"
    return rc


print("version", h.swat_version())
# ---- pure host arithmetic: workspace / packed-image sizes over a grid of shapes ----
for B in (1, 2, 31, 32, 64, 257):
    for T, H in ((50, 12), (197, 12), (257, 16), (77, 8)):
        assert h.swat_attn_bwd_tiled_workspace_bytes(B, T, H) >= B * T * H * 4
for n in (1, 7, 200, 1000, 4096):
    for k in (1, 10, 500):
        assert h.swat_merge_topk_workspace_bytes(n, 8, k) > 0
# ---- setters: accepted and rejected values ----
for v in (-1, 0, 14, 43, 50, 51, 80, 99, 1000):
    run("swat_gemm_set_tile", v)
for v in (0, 1, 2, 3, -1):
    run("swat_vit_set_mega", v)
for v in (0, 3, 4, 5, -1):
    run("swat_vit_set_split_k", v)
for v in (0, 256, 4096, 5000, -3):
    run("swat_adamw_set_blocks", v)
run("swat_gemm_set_tile", -1)
# ---- argument checks of the compute entry points (null pointers, bad shapes) ----
N = None
run("swat_gemm_bf16", 7, P(), P(1), P(2), 64, 64, 64, 64, 64, 64, 1, 1, 0, 0, 0, 0, 0, 0, N, N, N, 0, 0, N, 0, 1.0, N)
run("swat_gemm_bf16", 0, P(), P(1), P(2), 64, 64, 64, 63, 64, 64, 1, 1, 0, 0, 0, 0, 0, 0, N, N, N, 0, 0, N, 0, 1.0, N)
run("swat_gemm_bf16", 0, P(), P(1), N, 64, 64, 64, 64, 64, 64, 1, 1, 0, 0, 0, 0, 0, 0, N, N, N, 0, 0, N, 0, 1.0, N)
run("swat_attn_fwd_tiled", N, N, N, 1, 100, 64, 1, 0, N)
run("swat_attn_fwd_tiled", P(), P(1), N, 2, 400, 128, 2, 0, N)
run("swat_attn_bwd_tiled", P(), P(1), P(2), P(3), P(4), 2, 100, 128, 2, N, N)
run("swat_attn_fwd_small", P(), P(1), 2, 65, 128, 2, N)
assert h.swat_dense_shadow_bytes(1000, 512) == 1024 * 512 * 2 and h.swat_dense_shadow_bytes(1000, 640) == 1000 * 640 * 2 and h.swat_dense_shadow_bytes(1000, 100) == 0
run("swat_dense_build_shadow", N, 10, 512, P(), P(1), N)
run("swat_dense_build_shadow", P(), 10, 100, P(1), P(2), N)
run("swat_dense_build_shadow", P(), 1 << 20, 512, P(1), P(2), N)
for nq in (1, 200, 1000):
    wb = h.swat_dense_workspace_bytes(nq)
    assert wb > 0 and h.swat_dense_packed_bytes(nq, 512) > 0
    run("swat_sim_candidates_shadow", P(), nq, P(1), 1 << 20, 512, P(2), 0, P(3), 8192, P(4), P(5), wb, N)
    run("swat_sim_candidates_shadow", P(), nq, P(1), 1 << 20, 512, P(2), 0, P(3), 8192, P(4), P(5), wb - 1, N)
    run("swat_sim_candidates_packed", P(), nq, P(1), 1 << 20, 768, P(2), 0, P(3), 8192, P(4), P(5), P(6), wb, N)
run("swat_candidates_select", P(), 8192, P(1), 200, 1024, P(2), P(3), P(4), P(5), N)
run("swat_candidates_select", P(), 9000, P(1), 200, 1024, P(2), P(3), P(4), N, N)
run("swat_candidates_topk_proof", P(), 1024, P(1), 200, 500, P(2), P(3), P(4), P(1), 8192, 1024, P(5), P(6), 500, 0.004, P(7), P(8), P(9), 1, N)
run("swat_candidates_topk_proof", P(), 1024, P(1), 200, 500, P(2), P(3), P(4), N, 8192, 1024, P(5), P(6), 500, 0.004, P(7), P(8), P(9), 0, N)
run("swat_candidates_topk", P(), 8192, P(1), 200, 9000, P(2), P(3), P(4), N)
run("swat_cast_bf16", P(), P(1), 7, N)
run("swat_widen_bf16", P(), P(1), 7, N)
# ---- dispatch / tile planning over the step's shapes with fake pointers: fails at the first HIP call, cleanly ----
for M in (200, 1600, 1568, 6304, 8224, 51200):
    for (n, k) in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (1024, 4096), (4096, 1024), (1000, 1024)):
        for layout in (0, 1, 2):
            m_, n_, k_ = (M, n, k) if layout != 2 else (n, k, (M + 63) // 64 * 64)
            lda = k_ if layout != 2 else m_
            ldb = k_ if layout == 0 else n_
            for out_f32 in (0, 1):
                run("swat_gemm_bf16", layout, P(), P(1), P(2), m_, n_, k_, lda, ldb, (n_ + 3) // 4 * 4, 1, 1, 0, 0, 0, 0, 0, 0,
                    P(3), N, N, 0, 0, N, out_f32, 1.0, N)
for cfg in (14, 20, 30, 33, 43, 50, 51, 53, 54, 55, 56, 58, 59, 66, 80):
    run("swat_gemm_set_tile", cfg)
    run("swat_gemm_bf16", 0, P(), P(1), P(2), 8224, 3072, 1024, 1024, 1024, 3072, 1, 1, 0, 0, 0, 0, 0, 0, P(3), N, N, 0, 1, P(4), 0, 1.0, N)
    run("swat_gemm_bf16", 2, P(), P(1), P(2), 768, 3072, 1600, 768, 3072, 3072, 1, 1, 0, 0, 0, 0, 0, 0, N, N, N, 0, 0, N, 1, 1.0, N)
run("swat_gemm_set_tile", -1)
for B, T, H in ((32, 257, 16), (32, 197, 12), (3, 77, 8), (2, 320, 2), (1, 65, 1)):
    d = 64 * H
    run("swat_attn_fwd_tiled", P(), P(1), P(2), B, T, d, H, 0, N)
    run("swat_attn_bwd_tiled", P(), P(1), P(2), P(3), P(4), B, T, d, H, P(5), N)
run("swat_adamw", P(), P(1), P(2), P(3), P(4), 1 << 20, (ctypes.c_int64 * 4)(0, 100, 200, 300), 3, P(5), 0.9, 0.999, 1e-8, 0.01, P(6), 1.0, N)
run("swat_adamw", P(), P(1), P(2), P(3), P(4), 1 << 20, (ctypes.c_int64 * 4)(0, 100, 200, 300), 9, P(5), 0.9, 0.999, 1e-8, 0.01, P(6), 1.0, N)

# ---- the JPEG host parser and entropy decoder on hostile bytes (web-retrieved files reach them in every loader worker) ----
import io
import numpy as np
from PIL import Image

def _jpeg_calls(blob):
    a = np.frombuffer(bytes(blob), np.uint8)
    W, H = ctypes.c_int32(), ctypes.c_int32()
    dims = (ctypes.c_int32 * 8)()
    run("swat_jpeg_info", a.ctypes.data_as(ctypes.c_void_p), a.size, ctypes.cast(ctypes.byref(W), ctypes.c_void_p), ctypes.cast(ctypes.byref(H), ctypes.c_void_p))
    rc = run("swat_jpeg_entropy_decode", a.ctypes.data_as(ctypes.c_void_p), a.size, ctypes.cast(dims, ctypes.c_void_p), N, 0)
    if rc == 0 and dims[0] * dims[1] <= 1 << 20:
        need = sum(dims[5] * (dims[3] if c == 0 else 1) * dims[6] * (dims[4] if c == 0 else 1) * 64 for c in range(dims[2]))
        out = np.zeros(need, np.int16)
        run("swat_jpeg_entropy_decode", a.ctypes.data_as(ctypes.c_void_p), a.size, ctypes.cast(dims, ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), out.size)

_rng = np.random.default_rng(9)
_seg = lambda m, payload: bytes([0xFF, m]) + (len(payload) + 2).to_bytes(2, "big") + payload
for sub, rst in ((0, 0), (2, 0), (1, 3)):
    b = io.BytesIO()
    kw = {"restart_marker_blocks": rst} if rst else {}
    Image.fromarray((_rng.random((40, 56, 3)) * 255).astype(np.uint8)).save(b, "JPEG", quality=80, subsampling=sub, **kw)
    good = b.getvalue()
    _jpeg_calls(good)
    for counts in ([255] + [0] * 15, [3] + [0] * 15, [0, 5] + [0] * 14, [0] * 15 + [255]):   # tables that are not prefix codes
        _jpeg_calls(b"\xff\xd8" + _seg(0xC4, bytes([0x10]) + bytes(counts) + bytes(range(min(sum(counts), 256)))) + good[2:])
    for cut in range(0, len(good), 7):        # truncated at every 7th byte
        _jpeg_calls(good[:cut])
    sos = good.find(b"\xff\xda")
    for _ in range(300):                      # random byte damage: headers (tables, frame, scan header) and entropy-coded data
        bad = bytearray(good)
        for _k in range(int(_rng.integers(1, 4))):
            pos = int(_rng.integers(2, sos + 14)) if _rng.random() < 0.7 else int(_rng.integers(sos, len(good)))
            bad[pos] = int(_rng.integers(0, 256))
        _jpeg_calls(bad)
print(f"{calls} calls, {fails} returned an error with a message; no sanitizer report")
