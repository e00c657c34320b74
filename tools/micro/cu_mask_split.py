"""Dev experiment: phase 3 of the step (weight gradients beside AdamW) with the two streams on DISJOINT sets of CUs
(hipExtStreamCreateWithCUMask).  AdamW (88 M parameters, wide form) and a stand-in for the blocks' weight gradients (48 TN products
of the ViT-B/32 shapes) alone on k CUs, then both at once on complementary masks, against both at once on unmasked streams."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from swat_amd import _lib, ops
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
def masked_stream(bits):   # bits: iterable of CU bit indices
    words = (ctypes.c_uint32 * 8)()
    for b in bits: words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(st.value)
n = 85_000_000 // 4 * 4
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); sh = torch.empty(n, dtype=torch.bfloat16, device=dev)
lrs = torch.tensor([1e-4, 1e-6, 1e-4], device=dev); bc = torch.tensor([0.1, 0.001], device=dev)
segs = [0, 102400, n - 4]
per = n // 12 // 4 * 4
def adamw_blocks():   # twelve per-block launches, as the step issues them
    for b in range(12):
        sl = slice(b * per, (b + 1) * per)
        ops.adamw_step(p[sl], g[sl], m[sl], v[sl], sh[sl], [0], lrs[1:2], bc)
M = 1568
X = {k: torch.randn(M, k, device=dev).to(torch.bfloat16) for k in (768, 2304, 3072)}
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]   # dW [N_out, N_in] = dY^T [N_out, M] x X [M, N_in]
outs = [torch.empty(a, b, device=dev) for a, b in shapes]
def wgrads():
    for _ in range(12):
        for (a, b), o in zip(shapes, outs):
            ops.gemm(ops.TN, X[a], X[b], a, b, M, a, b, o, b)
def timed(fn_by_stream, reps=5):
    res = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        ends = []
        for st, fn in fn_by_stream:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                fn()
                e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
        torch.cuda.synchronize()
        res.append(max(e0.elapsed_time(e) for e in ends))
    res.sort(); return res[len(res) // 2] * 1e3
try:
    wgrads(); adamw_blocks(); torch.cuda.synchronize()
except Exception as ex:
    print("setup failed:", ex); raise
plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
print(f"unmasked: AdamW alone {timed([(plain_a, adamw_blocks)]):.0f} us, weight gradients alone {timed([(plain_b, wgrads)]):.0f} us, both at once {timed([(plain_a, adamw_blocks), (plain_b, wgrads)]):.0f} us")
for k in (32, 64, 96, 128):
    sa = masked_stream(range(0, k)); sb = masked_stream(range(k, 256))
    ta = timed([(sa, adamw_blocks)]); tb = timed([(sb, wgrads)]); tc = timed([(sa, adamw_blocks), (sb, wgrads)])
    print(f"AdamW on {k} CUs (mask bits 0..{k - 1}) {ta:.0f} us ({n * 30 / ta / 1e6:.2f} TB/s); weight gradients on the other {256 - k}: {tb:.0f} us; both at once {tc:.0f} us")
