"""Diagnostic (GPU): LDS canary blocks (own stream) next to each GEMM kernel family (noise stream): do words of the canaries'
LDS change under them?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
import ctypes
D = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libst5_diag.so'))   # tools/diag/build.sh
err = torch.zeros(1, dtype=torch.int32, device=dev)
main, noise = torch.cuda.Stream(), torch.cuda.Stream()
bf = torch.bfloat16
def T(*s): return torch.randn(*s, device=dev, dtype=bf)
a, b, c = T(8192, 768), T(3072, 768), torch.empty(8192, 3072, device=dev, dtype=bf)
a2, b2, c2 = T(65536, 1024), T(512, 1024), torch.empty(65536, 512, device=dev, dtype=bf)
at, bt, ct = T(8192, 768), T(8192, 3072), torch.empty(768, 3072, device=dev, dtype=torch.float32)
def noise_fn(kind):
    if kind == "nt128": hip.gemm(hip.operand(a, 768), hip.operand(b, 768), hip.operand(c, 3072), 8192, 3072, 768, hip.BF16)
    elif kind == "nt256": hip.gemm(hip.operand(a2, 1024), hip.operand(b2, 1024), hip.operand(c2, 512), 65536, 512, 1024, hip.BF16)
    elif kind == "tn": hip.gemm(hip.operand(at, 768), hip.operand(bt, 3072), hip.operand(ct, 3072), 768, 3072, 8192, hip.BF16, flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
    elif kind == "general": L.st5_gemm_set_glds(0); hip.gemm(hip.operand(a, 768), hip.operand(b, 768), hip.operand(c, 3072), 8192, 3072, 768, hip.BF16); L.st5_gemm_set_glds(1)
    elif kind == "blas": torch.matmul(a, b.t(), out=c)
for kind in ("none", "blas", "general", "nt256", "tn", "nt128"):
    for lds_bytes in (4096, 16384, 32768):
        err.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(noise):
            for _ in range(60): noise_fn(kind)
        with torch.cuda.stream(main):
            for _ in range(20):
                hip.check(D.st5_debug_lds_canary(err.data_ptr(), 512, lds_bytes, 200, hip.stream()), "canary")
        torch.cuda.synchronize()
        print(f"noise={kind:8s} canary LDS {lds_bytes:6d} B: corrupted word reads = {int(err.item())}", flush=True)
