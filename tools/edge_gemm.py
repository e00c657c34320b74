"""Dev tool: random product shapes through the GEMM dispatcher (whatever tile it picks) against fp32 torch, over poisoned memory."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swat_amd import ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
random.seed(3)
bad = 0
shapes = [(51200, 768, 3072), (51200, 2304, 768), (40000, 1000, 512), (33000, 256, 4096), (32768, 1024, 64), (70000, 512, 512)]
for _ in range(70):
    shapes.append((random.choice([1, 7, 50, 199, 1600, 6304, 8224, 12800, 33333]), 8 * random.randint(1, 400), 8 * random.randint(1, 300)))
for M, N, K in shapes:
    for lay in ("nt", "nn", "tn"):
        poison = torch.full((32 << 20,), -1, dtype=torch.int8, device=dev); del poison
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        if lay == "tn":
            m, n, k = N, (K + 3) // 4 * 4, (M + 7) // 8 * 8   # dW[m, n] = A[k, m]^T B[k, n]
            if m * n > 64_000_000: continue
            A = torch.randn(k, m, device=dev, generator=g).to(BF); B = (0.05 * torch.randn(k, n, device=dev, generator=g)).to(BF)
            C = torch.full((m, n), float("nan"), device=dev)
            ops.gemm(ops.TN, A, B, m, n, k, m, n, C, n)
            ref = A.float().t() @ B.float()
        else:
            if M * N > 200_000_000: continue
            A = torch.randn(M, K, device=dev, generator=g).to(BF); W = (0.05 * torch.randn(N, K, device=dev, generator=g)).to(BF)
            ldc = (N + 3) // 4 * 4
            C = torch.full((M, ldc), float("nan"), device=dev)
            if lay == "nt": ops.gemm(ops.NT, A, W, M, N, K, K, K, C, ldc)
            else: ops.gemm(ops.NN, A, W.t().contiguous(), M, N, K, K, N, C, ldc)
            ref = A.float() @ W.float().t(); C = C[:, :N]; k = K
        err = (C - ref).abs().max().item()
        tol = 2e-3 * (k ** 0.5) * 0.05 * 4 + 1e-3
        if not (err <= tol):
            bad += 1
            print("BAD", lay, M, N, K, err, tol, flush=True)
print("bad:", bad, "of", len(shapes) * 3)
