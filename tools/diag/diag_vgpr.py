import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
import ctypes
D = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libst5_diag.so'))   # tools/diag/build.sh
hist = torch.zeros(64, dtype=torch.int32, device=dev)
main, noise = torch.cuda.Stream(), torch.cuda.Stream()
bf = torch.bfloat16
def T(*s): return torch.randn(*s, device=dev, dtype=bf)
a, b, c = T(8192, 768), T(3072, 768), torch.empty(8192, 3072, device=dev, dtype=bf)
at, bt, ct = T(8192, 768), T(8192, 3072), torch.empty(768, 3072, device=dev, dtype=torch.float32)
def noise_fn(kind):
    if kind == "nt128": hip.gemm(hip.operand(a, 768), hip.operand(b, 768), hip.operand(c, 3072), 8192, 3072, 768, hip.BF16)
    elif kind == "tn": hip.gemm(hip.operand(at, 768), hip.operand(bt, 3072), hip.operand(ct, 3072), 768, 3072, 8192, hip.BF16, flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
    elif kind == "general": L.st5_gemm_set_glds(0); hip.gemm(hip.operand(a, 768), hip.operand(b, 768), hip.operand(c, 3072), 8192, 3072, 768, hip.BF16); L.st5_gemm_set_glds(1)
    elif kind == "blas": torch.matmul(a, b.t(), out=c)
for kind in ("none", "blas", "nt128", "general", "tn"):
    hist.zero_(); torch.cuda.synchronize()
    for rep in range(6):
        with torch.cuda.stream(noise):
            for _ in range(60): noise_fn(kind)
        with torch.cuda.stream(main):
            for _ in range(10):
                hip.check(D.st5_debug_vgpr_canary(hist.data_ptr(), 1024, 2000, hip.stream()), "canary")
        torch.cuda.synchronize()
    h = hist.cpu().tolist()
    print(f"noise={kind:8s}: corrupted register values per lane: total {sum(h)}; lanes {[i for i, v in enumerate(h) if v]}", flush=True)
