"""Device time of the fused Adam step at the benched size (154.4 M parameters, two gradient buffers, bf16 mirror): 20 back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speecht5_amd import hip
n = 154392384
dev = torch.device("cuda:0")
p, m, v = (torch.randn(n, device=dev) * 0.02 for _ in range(3))
v.abs_()
g, g2 = torch.randn(n, device=dev) * 1e-3, torch.randn(n, device=dev) * 1e-3
w = torch.empty(n, dtype=torch.bfloat16, device=dev)
gn = torch.ones(1, device=dev)
L = hip.lib()
def step():
    hip.check(L.st5_adam_step_pair(p.data_ptr(), g.data_ptr(), g2.data_ptr(), 1, m.data_ptr(), v.data_ptr(), n, 2e-4, 0.9, 0.98, 1e-6, 0.01, 5,
                                   gn.data_ptr(), 5.0, 0.5, w.data_ptr(), 0, hip.stream()), "adam")
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"ADAM {os.path.basename(os.environ.get('ST5_HIP_LIB', 'default'))}: {ms:.4f} ms per step, {n * 42 / ms / 1e9:.2f} TB/s of 42 B/parameter")
