"""MX-fp8 NT GEMM vs the bf16 NT GEMM on SpeechT5-Large's Linear shapes (text micro-batch 16 x 512, speech 8 x 499), plain epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
for (M, N, K) in ((8192, 4096, 1024), (8192, 1024, 4096), (8192, 3072, 1024), (8192, 1024, 1024), (3992, 4096, 1024), (3992, 1024, 4096), (3992, 3072, 1024)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t16 = timeit(lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16))
    Aq, As = hip.quant_mxfp8(A); Bq, Bs = hip.quant_mxfp8(B)
    t8 = timeit(lambda: hip.gemm_mxfp8(Aq, As, Bq, Bs, hip.operand(C, N), M, N, K))
    tq = timeit(lambda: hip.quant_mxfp8(A))
    f = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d} | bf16 {t16*1e6:6.1f} us {f/t16/1e12:5.0f} TF | mxfp8 {t8*1e6:6.1f} us {f/t8/1e12:5.0f} TF | quantise A {tq*1e6:5.1f} us "
          f"({(M*K*3+M*K/32)/tq/1e9:5.0f} GB/s) | fp8 + quant {f/(t8+tq)/1e12:5.0f} TF", flush=True)
