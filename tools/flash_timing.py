"""Per-segment cycle attribution of flash_fwd_kernel (library must be built with -DFLASH_TIMING)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
B, H, T = 16, 12, 512
d = H * 64
qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
pe = torch.randn(320, 64, device=dev).to(torch.bfloat16)
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev); lse = torch.zeros(B * H, T, device=dev)
kpm = torch.zeros(B, T, dtype=torch.uint8, device=dev)
L = hip.lib()
for _ in range(3):
    L.st5_flash_attn_fwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                         lse.data_ptr(), hip.ptr(pe), kpm.data_ptr(), B, H, T, T, 64, 320, 160, 0, T, 0.125, 0.1, 5, hip.BF16, hip.stream())
torch.cuda.synchronize()
t = lse.view(B * H, 4, 4, 32)[:, :, :, 8:20]      # [bh, qblock, wave, seg]
names = ["K+bias lds reads", "QK mfma + V issue", "scores", "probs", "alpha + PV mfma", "vmcnt + st.store", "barrier", "top: prefetch issue", "prologue: q frags + kv0 issue", "table build", "kv0 store + barrier", "epilogue"]
m = t.mean(dim=(0, 1, 2)); tot = m.sum()
for n, v in zip(names, m.tolist()):
    print(f"{n:22s} {v/8:9.0f} cycles/tile  {100*v/tot:5.1f}%")
print("per-wave total", tot.item(), "memtime ticks; per (qblock,wave) totals:")
print((t[..., :8].sum(-1).mean(0) / 8).int())
