"""conv layer 0 + GroupNorm + GELU (st5_conv0_gn_gelu_fwd / _bwd) at the bench shape: time per call and HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit
dev = torch.device("cuda:0"); L = hip.lib()
B, S, C, k, st = 8, 160000, 512, 10, 5
Lo = (S - k) // st + 1
wav = torch.randn(B, S, device=dev); w = torch.randn(C, k, device=dev) * 0.3
g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev) * 0.1
out = torch.empty(B, Lo, C, device=dev, dtype=torch.bfloat16); stats = torch.empty(B, C, 2, device=dev)
ws = torch.empty(L.st5_conv0_ws_bytes(B, S, C, k, st), dtype=torch.uint8, device=dev)
dy = torch.randn_like(out); dw = torch.zeros(C, k, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
tf = timeit(lambda: hip.check(L.st5_conv0_gn_gelu_fwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), stats.data_ptr(),
                                                      ws.data_ptr(), B, S, C, k, st, 1e-5, hip.BF16, hip.stream()), "fwd"))
tb = timeit(lambda: hip.check(L.st5_conv0_gn_gelu_bwd(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), stats.data_ptr(), dy.data_ptr(),
                                                      dw.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), B, S, C, k, st, 1.0, hip.BF16,
                                                      hip.stream()), "bwd"))
mb = (out.numel() * 2 + wav.numel() * 4) / 1e6
print(f"conv0 fwd {tf*1e6:.1f} us ({mb/tf/1e6:.2f} TB/s of {mb:.0f} MB)  bwd {tb*1e6:.1f} us ({mb/tb/1e6:.2f} TB/s)")
