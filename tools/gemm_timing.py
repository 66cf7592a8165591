"""Per-phase cycle attribution of gemm_nt_glds_kernel (library built with -DGEMM_TIMING)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
L.st5_gemm_timing.restype = ctypes.c_int
L.st5_gemm_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
names = ["prologue (addr + first issue)", "k-loop: vmcnt wait + barrier", "k-loop: issue + lds reads + mfma", "post-loop barrier", "epilogue"]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # 1 = 128^2 kernel, 2 = 256^2 kernel
L.st5_gemm_set_nt_tile(mode)
kstep = 64 if mode == 1 else 32
for (M, N, K) in ((8192, 3072, 768), (3992, 3072, 768), (3992, 2304, 768), (8192, 768, 3072), (3992, 768, 768), (3992, 768, 3072), (4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16)
    f(); torch.cuda.synchronize(); L.st5_gemm_timing(buf, 1)
    f(); torch.cuda.synchronize(); L.st5_gemm_timing(buf, 1)
    n = buf[7]; tot = sum(buf[i] for i in range(5))
    print(f"M={M} N={N} K={K}: waves={n}  cycles/wave={tot/n:.0f}  (k-steps={K//kstep}, ideal mfma cycles/wave={K//64*(512 if mode == 1 else 1024)}, 2 waves/SIMD)")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} {buf[i]/n:9.0f} cycles/wave  {100*buf[i]/tot:5.1f}%")
