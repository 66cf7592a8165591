"""Wall time of the NT GEMM kernels on the model's shapes with the model's epilogue classes (HIP events, 20 launches each).
ST5_HIP_LIB=<other .so> for an A/B against another build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
bf = torch.bfloat16
def case(M, N, K, epi, reps=20):
    A = torch.randn(M, K, device=dev).to(bf); B = torch.randn(N, K, device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf); P = torch.empty(M, N, device=dev, dtype=bf); R = torch.randn(M, N, device=dev).to(bf)
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi == "fc1": kw = dict(bias=bias, act=hip.ACT_GELU, Cpre=hip.operand(P, N))
    elif epi == "bias": kw = dict(bias=bias)
    elif epi == "drop_res": kw = dict(bias=bias, dropout_p=0.1, seed=1234, R=hip.operand(R, N))
    elif epi == "dact": kw = dict(P=hip.operand(R, N), act=hip.ACT_GELU, flags=hip.DACT)
    elif epi == "res": kw = dict(R=hip.operand(R, N))
    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16, **kw)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6
cases = ((8192, 3072, 768, "plain"), (8192, 3072, 768, "fc1"), (8192, 3072, 768, "dact"), (8192, 768, 3072, "plain"), (8192, 768, 3072, "drop_res"),
         (8192, 768, 768, "plain"), (8192, 768, 768, "drop_res"), (8192, 768, 768, "res"), (8192, 2304, 768, "bias"), (3992, 768, 768, "drop_res"),
         (3992, 3072, 768, "fc1"), (3992, 3072, 768, "dact"), (3992, 768, 3072, "drop_res"), (2504, 768, 768, "drop_res"), (2504, 768, 3072, "drop_res"))
for c in cases:
    us, tf = case(*c)
    print(f"CASE {c[0]:5d} {c[1]:4d} {c[2]:4d} {c[3]:8s} {us:7.1f} us {tf:6.0f} TFLOP/s", flush=True)
