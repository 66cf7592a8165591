"""Time the NT GEMM at a few shapes with whichever library ST5_HIP_LIB points at (ablation variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
out = []
for (M, N, K) in ((8192, 3072, 768), (3992, 3072, 768), (8192, 768, 3072), (3992, 768, 768), (4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16))
    out.append(f"{t*1e6:7.1f}us")
print(os.path.basename(os.environ.get("ST5_HIP_LIB", "default")), " ".join(out))
