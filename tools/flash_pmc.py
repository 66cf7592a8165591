"""Run the fused attention kernels a few times (for rocprofv3 --pmc / --kernel-trace passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
B, H, T, rel = 16, 12, 512, int(os.environ.get("REL", "1"))
p = float(os.environ.get("PDROP", "0.1"))
d = H * 64
qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
pe = torch.randn(320, 64, device=dev).to(torch.bfloat16) if rel else None
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, T, device=dev)
kpm = torch.zeros(B, T, dtype=torch.uint8, device=dev)
dqkv = torch.empty_like(qkv); do = torch.randn_like(o); dvec = torch.empty(B * H * T, device=dev)
L = hip.lib()
qrow = L.st5_flash_attn_qp_row(320)     # second-generation layout: 8 | nb | 8
qp = torch.randn(B * H, T, qrow, device=dev).to(torch.bfloat16) if rel else None
dqp = torch.empty(B * H, T, 320, dtype=torch.bfloat16, device=dev) if rel else None
for _ in range(int(os.environ.get("ITERS", "3"))):
    # (with the table workspace: the second-generation forward; without it a bias call would run the first generation)
    L.st5_flash_attn_fwd_qp(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                            lse.data_ptr(), hip.ptr(pe), kpm.data_ptr(), B, H, T, T, 64, 320 if rel else 0, 160 if rel else 0, 0, T,
                            0.125, p, 5, hip.ptr(qp), hip.BF16, hip.stream())
    L.st5_flash_attn_bwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                         do.data_ptr(), d, dqkv.data_ptr(), 3 * d, dqkv.data_ptr() + d * 2, 3 * d, dqkv.data_ptr() + 4 * d, 3 * d,
                         lse.data_ptr(), dvec.data_ptr(), hip.ptr(pe), hip.ptr(qp), hip.ptr(dqp), kpm.data_ptr(), B, H, T, T, 64,
                         320 if rel else 0, 160 if rel else 0, 0, T, 0.125, p, 5, hip.BF16, hip.stream())
torch.cuda.synchronize()
