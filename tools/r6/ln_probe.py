"""Where do the 20.8 us of st5_layernorm_bwd at 8192 x 768 (37.7 MB: 1.8 TB/s) go?  With / without the dgamma / dbeta partials, by block cap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
dev = torch.device("cuda:0")
L = hip.lib()
C = 768
DEFER = int(os.environ.get("LN_DEFER", "0"))
hip.check(L.st5_layernorm_defer(DEFER, hip.stream()), "defer")
ONE = os.environ.get("LN_ONE")
for cap in ((int(ONE.split()[0]),) if ONE else (256, 384, 512, 1024)):
    hip.check(L.st5_layernorm_set_max_blocks(cap), "cap")
    for rows in ((int(ONE.split()[1]),) if ONE else (8192, 3992)):
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16); dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        dx = torch.empty_like(x); dxd = torch.empty_like(x)
        g = torch.ones(C, device=dev)
        mean, rstd = torch.zeros(rows, device=dev), torch.ones(rows, device=dev)
        dgs = [torch.zeros(C, device=dev) for _ in range(64)]; dbs = [torch.zeros(C, device=dev) for _ in range(64)]
        ctr = [0]
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, C), dev)
        def run(pg, drop):
            ctr[0] += 1
            dg, db = dgs[ctr[0] % 64], dbs[ctr[0] % 64]
            hip.check(L.st5_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                          dg.data_ptr() if pg else 0, db.data_ptr() if pg else 0, ws.data_ptr(), rows, C,
                                          dxd.data_ptr() if drop else 0, 0.1 if drop else 0.0, 7 if drop else 0, hip.BF16, hip.stream()), "bwd")
        for pg, drop in (((bool(int(ONE.split()[2])), bool(int(ONE.split()[3]))),) if ONE else ((True, False), (False, False), (True, True))):
            for _ in range(3): run(pg, drop)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run(pg, drop)
            e1.record(); torch.cuda.synchronize()
            hip.check(L.st5_layernorm_flush(hip.stream()), "flush")
            us = e0.elapsed_time(e1) / 50 * 1e3
            nbytes = rows * C * (8 if drop else 6)
            print(f"LN bwd cap {cap:4d} rows {rows:5d} partials {int(pg)} dropped-copy {int(drop)}: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
