"""Run a few st5_gemm launches (for rocprofv3 --pmc / --kernel-trace passes).  SHAPE=M,N,K FORM=NT|TN"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
M, N, K = (int(x) for x in os.environ.get("SHAPE", "8192,3072,768").split(","))
form = os.environ.get("FORM", "NT")
dt = torch.bfloat16
if os.environ.get("ST5_NT_TILE"):
    hip.lib().st5_gemm_set_nt_tile(int(os.environ["ST5_NT_TILE"]))
if form == "NT":
    A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt); C = torch.empty(M, N, device=dev, dtype=dt)
    f = lambda: hip.gemm(hip.operand(A, K), hip.operand(B, K), hip.operand(C, N), M, N, K, hip.BF16)
else:
    A = torch.randn(K, M, device=dev).to(dt); B = torch.randn(K, N, device=dev).to(dt); C = torch.empty(M, N, device=dev)
    f = lambda: hip.gemm(hip.operand(A, M), hip.operand(B, N), hip.operand(C, N), M, N, K, hip.BF16,
                         flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
for _ in range(5):
    f()
torch.cuda.synchronize()
