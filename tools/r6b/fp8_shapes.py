"""MX-fp8 NT GEMM at the Large shapes (B = 32: M = 15968 speech / 8192 text rows), per block tile: bf16 (st5_gemm's own choice), fp8 on
128 x 128 tiles, fp8 on the phased 256 x 256 kernel (round 6), the activation quantiser beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
bf = torch.bfloat16
L = hip.lib()
for (M, N, K) in ((15968, 1024, 1024), (15968, 3072, 1024), (15968, 4096, 1024), (15968, 1024, 4096), (8192, 1024, 1024), (8192, 3072, 1024),
                  (8192, 4096, 1024), (8192, 1024, 4096), (10016, 1024, 1024), (10016, 4096, 1024), (32768, 1024, 4096), (32768, 4096, 1024)):
    A = torch.randn(M, K, device=dev).to(bf); B = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    oA, oB, oC = hip.operand(A, K), hip.operand(B, K), hip.operand(C, N)
    Aq, As = hip.quant_mxfp8(A); Bq, Bs = hip.quant_mxfp8(B)
    f = 2.0 * M * N * K
    t_bf = timeit(lambda: hip.gemm(oA, oB, oC, M, N, K, hip.BF16))
    line = f"M={M:6d} N={N:5d} K={K:5d} | bf16 {t_bf*1e6:7.1f} us {f/t_bf/1e12:5.0f} TF"
    for mode, name in ((1, "fp8 128^2"), (2, "fp8 256^2 phased")):
        L.st5_gemm_set_mx8_tile(mode)
        t = timeit(lambda: hip.gemm_mxfp8(Aq, As, Bq, Bs, oC, M, N, K))
        line += f" | {name} {t*1e6:7.1f} us {f/t/1e12:5.0f} TF"
    L.st5_gemm_set_mx8_tile(0)
    tq = timeit(lambda: hip.quant_mxfp8(A))
    print(line + f" | quantise A {tq*1e6:6.1f} us", flush=True)
