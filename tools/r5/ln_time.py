"""Device time of st5_layernorm_fwd / _bwd at the step's two shapes (bf16, 768 columns), back-to-back calls, per block cap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
C = 768
for cap in (256,):
    hip.check(L.st5_layernorm_set_max_blocks(cap), "cap")
    for rows in (8192, 3992):
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16); dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x); dx = torch.empty_like(x)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, C), dev)
        def fwd():
            hip.check(L.st5_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C, 1e-5, hip.BF16, hip.stream()), "fwd")
        def bwd():
            hip.check(L.st5_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                          ws.data_ptr(), rows, C, 0, 0.0, 0, hip.BF16, hip.stream()), "bwd")
        for nm, fn, nbytes in (("fwd", fwd, rows * C * 4), ("bwd", bwd, rows * C * 6)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            print(f"LN cap {cap:5d} rows {rows:5d} {nm}: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
