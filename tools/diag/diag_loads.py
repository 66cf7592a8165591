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
def pattern(n):
    i = torch.arange(n, dtype=torch.int64, device=dev)
    return ((i * 2654435761) & 0xFFFFFFFF).to(torch.int64).to(torch.int32) if False else ((i * 2654435761) % (1 << 32)).to(torch.uint32) if hasattr(torch, "uint32") else None
def make(n):
    i = torch.arange(n, dtype=torch.int64, device=dev)
    v = (i * 2654435761) % (1 << 32)
    v = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
    return v
big, small = make(64 * 1024 * 1024), make(512)
torch.cuda.synchronize()
def noise_fn(kind):
    if kind == "nt128": hip.gemm(hip.operand(a, 768), hip.operand(b, 768), hip.operand(c, 3072), 8192, 3072, 768, hip.BF16)
    elif kind == "nt256": hip.gemm(hip.operand(a2, 1024), hip.operand(b2, 1024), hip.operand(c2, 512), 65536, 512, 1024, hip.BF16)
    elif kind == "blas": torch.matmul(a, b.t(), out=c)
for kind in ("none", "blas", "nt256", "nt128"):
    for name, buf, blocks, passes in (("256 MB stream", big, 2048, 2), ("2 KB hot", small, 1024, 4000)):
        tot = 0
        for rep in range(5):
            err.zero_(); torch.cuda.synchronize()
            with torch.cuda.stream(noise):
                for _ in range(60): noise_fn(kind)
            with torch.cuda.stream(main):
                hip.check(D.st5_debug_load_check(buf.data_ptr(), buf.numel(), err.data_ptr(), blocks, passes, hip.stream()), "check")
            torch.cuda.synchronize()
            tot += int(err.item())
        print(f"noise={kind:6s} {name:14s}: wrong words over 5 runs = {tot}", flush=True)
