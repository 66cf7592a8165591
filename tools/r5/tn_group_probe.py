"""What would ONE launch over a layer's four weight-gradient GEMMs (no split-K, no slabs, no reduce) cost?  Emulated with a single
TN GEMM of the same tile count and reduction length: dW [768 x 9216] = dY^T [K x 768] . X [K x 9216] (9216 = 2304 + 768 + 3072 +
3072 columns: 432 tiles of 128^2, over the tile-count threshold of split-K) against the four separate launches with their
split-K slabs and batched reduction, as the step issues them."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
L = hip.lib()
fl = hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32
for K in (8192, 3992, 2504):
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    ops = []
    for (M, N) in shapes:
        A = torch.randn(K, M, device=dev).to(bf); B = torch.randn(K, N, device=dev).to(bf)
        C = torch.zeros(M, N, device=dev); asum = torch.zeros(M, device=dev)
        ops.append((hip.operand(A, M), hip.operand(B, N), hip.operand(C, N), M, N, asum, (A, B, C)))

    def four(defer):
        for (oa, ob, oc, M, N, asum, _) in ops:
            hip.gemm(oa, ob, oc, M, N, K, hip.BF16, flags=fl | (hip.DEFERRABLE if defer else 0), beta=1.0, asum=asum)
        if defer:
            hip.check(L.st5_gemm_flush_splitk(hip.stream()), "flush")
    t_sep = timeit(lambda: four(False))
    hip.check(L.st5_gemm_defer_splitk(1, hip.stream()), "defer")
    t_def = timeit(lambda: four(True))
    hip.check(L.st5_gemm_defer_splitk(0, hip.stream()), "defer")
    A = torch.randn(K, 768, device=dev).to(bf); B = torch.randn(K, 9216, device=dev).to(bf)
    C = torch.zeros(768, 9216, device=dev); asum = torch.zeros(768, device=dev)
    t_one = timeit(lambda: hip.gemm(hip.operand(A, 768), hip.operand(B, 9216), hip.operand(C, 9216), 768, 9216, K, hip.BF16, flags=fl, beta=1.0, asum=asum))
    f = 2 * K * 768 * 9216
    print(f"K={K:5d}: four launches, split-K + reduce each {t_sep*1e6:7.1f} us | deferred batched reduce {t_def*1e6:7.1f} us | "
          f"one 432-tile launch, no split {t_one*1e6:7.1f} us ({f/t_one/1e12:.0f} TF)", flush=True)
