#!/usr/bin/env python3
"""Reference point only: the vendor library's fp32 GEMM through torch (TF32 off) at the benchmark shapes."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
for n in (4096, 2048):
    a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
    at = a.t().contiguous(); bt = b.t().contiguous()
    for x, y in ((a, b), (a, bt.t()), (at.t(), b)):
        for _ in range(4):
            c = x @ y
    torch.cuda.synchronize()
