"""Per-segment cycle attribution of flash_bwd_dkv_kernel<true> (library built with -DFLASH_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
B, H, T = 16, 12, 512
d = H * 64
rel = int(os.environ.get("REL", "1"))
qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
pe = torch.randn(320, 64, device=dev).to(torch.bfloat16) if rel else None
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B * H, T, device=dev)
kpm = torch.zeros(B, T, dtype=torch.uint8, device=dev)
dqkv = torch.empty_like(qkv); do = torch.randn_like(o); dvec = torch.zeros(B * H * T, device=dev)
qp = torch.randn(B * H, T, 320, device=dev).to(torch.bfloat16) if rel else None
dqp = torch.empty(B * H, T, 320, dtype=torch.bfloat16, device=dev) if rel else None
L = hip.lib()
L.st5_flash_attn_fwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                     lse.data_ptr(), hip.ptr(pe), kpm.data_ptr(), B, H, T, T, 64, 320 if rel else 0, 160 if rel else 0, 0, T, 0.125, 0.1, 5, hip.BF16, hip.stream())
lse.fill_(3.0)   # the timing build overwrote part of it
L.st5_flash_attn_bwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                     do.data_ptr(), d, dqkv.data_ptr(), 3 * d, dqkv.data_ptr() + d * 2, 3 * d, dqkv.data_ptr() + 4 * d, 3 * d,
                     lse.data_ptr(), dvec.data_ptr(), hip.ptr(pe), hip.ptr(qp), hip.ptr(dqp), kpm.data_ptr(), B, H, T, T, 64,
                     320 if rel else 0, 160 if rel else 0, 0, T, 0.125, 0.1, 5, hip.BF16, hip.stream())
torch.cuda.synchronize()
t = dvec.view(B * H, 4, 4, 32)[:, :, :, 8:16]
names = ["top: q-tile prefetch issue", "bias prefetch + frag reads + S/dP mfma (x2 subs)", "element-wise (x2)", "dV/dK mfma (x2)", "st.store", "barrier", "-", "-"]
m = t.mean(dim=(0, 1, 2)); tot = m.sum()
for n, v in zip(names, m.tolist()):
    print(f"{n:50s} {v/8:9.0f} cycles/q-tile  {100*v/tot:5.1f}%")
print("per-wave loop total", tot.item())
