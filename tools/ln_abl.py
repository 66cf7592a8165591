"""LayerNorm backward: block-count A/B (st5_layernorm_set_max_blocks) at the model's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speecht5_amd import hip
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
L = hip.lib()
cols = 768
g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
for rows in (2504, 3992, 8192):
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16); dy = torch.randn_like(x); dx = torch.empty_like(x); y = torch.empty_like(x)
    mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
    dg = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
    tf = timeit(lambda: L.st5_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, cols, 1e-5, hip.BF16, hip.stream()))
    line = f"rows {rows}: fwd {tf*1e6:5.1f} us |"
    for nb in (128, 256, 512, 1024, 2048):
        L.st5_layernorm_set_max_blocks(nb)
        ws = torch.empty(L.st5_layernorm_bwd_ws_bytes(rows, cols), dtype=torch.uint8, device=dev)
        t = timeit(lambda: L.st5_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                               ws.data_ptr(), rows, cols, None, 0.0, 0, hip.BF16, hip.stream()))
        line += f" bwd@{nb}: {t*1e6:5.1f} us"
    print(line, flush=True)
L.st5_layernorm_set_max_blocks(256)
