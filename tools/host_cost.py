"""Host-side cost of the launch path pieces (microseconds per call) on this box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip, functional as Fn
dev = torch.device("cuda:0")
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
L = hip.lib()
a = torch.randn(256, 64, device=dev).to(torch.bfloat16); b = torch.randn(128, 64, device=dev).to(torch.bfloat16); c = torch.empty(256, 128, device=dev, dtype=torch.bfloat16)
print("st5_version (ctypes no-op call)      %.2f us" % t(lambda: L.st5_version()))
print("hip.stream()                          %.2f us" % t(lambda: hip.stream()))
print("torch.empty(256,128)                  %.2f us" % t(lambda: torch.empty(256, 128, device=dev, dtype=torch.bfloat16)))
print("hip.operand x3                        %.2f us" % t(lambda: (hip.operand(a, 64), hip.operand(b, 64), hip.operand(c, 128))))
print("hip.gemm tiny (wrapper + launch)      %.2f us" % t(lambda: hip.gemm(hip.operand(a, 64), hip.operand(b, 64), hip.operand(c, 128), 256, 128, 64, hip.BF16)))
x = torch.randn(256, 768, device=dev).to(torch.bfloat16)
print("st5_act_fwd tiny (ctypes launch)      %.2f us" % t(lambda: L.st5_act_fwd(x.data_ptr(), x.data_ptr(), x.numel(), 2, hip.BF16, hip.stream())))
print("torch add tiny                        %.2f us" % t(lambda: x + x))
w = torch.randn(768, 768, device=dev, requires_grad=True); bb = torch.zeros(768, device=dev, requires_grad=True)
Fn.set_compute_dtype(torch.bfloat16)
xr = x.clone().requires_grad_(True)
print("Fn.linear fwd (autograd apply)        %.2f us" % t(lambda: Fn.linear(xr, w, bb), 500))
def fb():
    y = Fn.linear(xr, w, bb); y.backward(x)
print("Fn.linear fwd+bwd                     %.2f us" % t(fb, 300))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): Fn.linear(xr, w, bb)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
