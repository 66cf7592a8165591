"""Per-phase cycle attribution (s_memtime probes, library built with tools/build_timing_lib.sh) + wall time of the 128x128 NT
kernel on the model's shapes with the model's epilogue classes.  ST5_HIP_LIB=speecht5_amd/libspeecht5_hip_timing.so python tools/gemm_phase.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
L.st5_gemm_timing.restype = ctypes.c_int
L.st5_gemm_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
names = ["prologue", "wait+barrier", "issue+lds+mfma", "post barrier", "epilogue"]
bf = torch.bfloat16
def case(M, N, K, epi):
    A = torch.randn(M, K, device=dev).to(bf); B = torch.randn(N, K, device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf); P = torch.empty(M, N, device=dev, dtype=bf); R = torch.randn(M, N, device=dev).to(bf)
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi == "fc1": kw = dict(bias=bias, act=hip.ACT_GELU, Cpre=hip.operand(P, N))
    elif epi == "bias": kw = dict(bias=bias)
    elif epi == "drop_res": kw = dict(bias=bias, dropout_p=0.1, seed=1234, R=hip.operand(R, N))
    elif epi == "dact": kw = dict(P=hip.operand(R, N), act=hip.ACT_GELU, flags=hip.DACT)
    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize(); L.st5_gemm_timing(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); L.st5_gemm_timing(buf, 1)
    us = e0.elapsed_time(e1) * 100
    n = buf[7]; tot = sum(buf[i] for i in range(5))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M:5d} N={N:4d} K={K:4d} {epi:8s}: {us:6.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s  tiles={tiles:4d}  cycles/wave={tot / n:7.0f}  " +
          "  ".join(f"{nm} {buf[i] / n:6.0f} ({100 * buf[i] / tot:4.1f}%)" for i, nm in enumerate(names)), flush=True)
for M, N, K, epi in ((8192, 3072, 768, "plain"), (8192, 3072, 768, "fc1"), (8192, 3072, 768, "dact"), (8192, 768, 3072, "plain"), (8192, 768, 3072, "drop_res"),
                     (8192, 768, 768, "plain"), (8192, 768, 768, "drop_res"), (8192, 2304, 768, "bias"), (3992, 768, 768, "drop_res"), (3992, 3072, 768, "fc1"),
                     (2504, 768, 768, "drop_res"), (4096, 4096, 4096, "plain")):
    case(M, N, K, epi)
