import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import functional as Fn, hip
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
for (B, H, T, rel) in ((8, 12, 499, 1), (16, 12, 512, 1), (8, 12, 313, 0)):
    d = H * 64
    qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
    pe = torch.randn(320, 64, device=dev).to(torch.bfloat16) if rel else None
    o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H, T, device=dev)
    L = hip.lib()
    f = lambda: L.st5_flash_attn_fwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                                     lse.data_ptr(), hip.ptr(pe), 0, B, H, T, T, 64, 320 if rel else 0, 160 if rel else 0, 0, (T + 7) // 8 * 8,
                                     0.125, 0.1, 5, hip.BF16, hip.stream())
    t = timeit(f)
    fl = 4.0 * B * H * T * T * 64 + (2.0 * B * H * T * 320 * 64 if rel else 0)
    u = lambda: Fn._attn_fwd((qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d), B, H, T, T, 64, pe, 160 if rel else 0, None, False, 0.1, 5)
    tu = timeit(u)
    kpm = torch.zeros(B, T, dtype=torch.uint8, device=dev)
    dqkv = torch.empty_like(qkv); do = torch.randn_like(o); dvec = torch.empty(B * H * T, device=dev)
    qp = torch.randn(B * H, T, 320, device=dev).to(torch.bfloat16) if rel else None
    dqp = torch.empty(B * H, T, 320, dtype=torch.bfloat16, device=dev) if rel else None
    fk = lambda: L.st5_flash_attn_fwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                                      lse.data_ptr(), hip.ptr(pe), kpm.data_ptr(), B, H, T, T, 64, 320 if rel else 0, 160 if rel else 0, 0, (T + 7) // 8 * 8,
                                      0.125, 0.1, 5, hip.BF16, hip.stream())
    tk = timeit(fk)
    fb = lambda: L.st5_flash_attn_bwd(qkv.data_ptr(), 3 * d, qkv.data_ptr() + d * 2, 3 * d, qkv.data_ptr() + 4 * d, 3 * d, o.data_ptr(), d,
                                      do.data_ptr(), d, dqkv.data_ptr(), 3 * d, dqkv.data_ptr() + d * 2, 3 * d, dqkv.data_ptr() + 4 * d, 3 * d,
                                      lse.data_ptr(), dvec.data_ptr(), hip.ptr(pe), hip.ptr(qp), hip.ptr(dqp), kpm.data_ptr(), B, H, T, T, 64,
                                      320 if rel else 0, 160 if rel else 0, 0, (T + 7) // 8 * 8, 0.125, 0.1, 5, hip.BF16, hip.stream())
    tb = timeit(fb)
    print(f"   with kpm: fwd {tk*1e6:.0f} us, bwd (prep+dq+dkv) {tb*1e6:.0f} us")
    print(f"B={B} H={H} T={T} rel={rel}: flash {t*1e6:.0f} us ({fl/t/1e12:.0f} TF) | unfused {tu*1e6:.0f} us")
