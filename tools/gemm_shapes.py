"""Per-shape GEMM time breakdown of one bench step (HIP events around every st5_gemm launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speecht5_amd import functional as Fn, hip
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample

dev = torch.device("cuda:0")
args, task, model, crit = bench.build(dev, torch.bfloat16)
ddp = FlatGradDataParallel(model)
opt = FusedAdam(ddp)
speech = speech_pretrain_sample(B=8, device=dev)
text = text_pretrain_sample(B=16, T=512, vocab=len(task.dicts["text"]), mask_idx=task.dicts["text"].index("<mask>"), device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
micro = {"speech": [speech], "text": [text], "both": [speech, text]}[which]
def step(i):
    ddp.zero_grad()
    for s in micro:
        task.train_step(s, model, crit, None, i, sync=False)
    ddp.finish(); opt.step(0.5)
for i in range(2): step(i)
torch.cuda.synchronize()
hip.profiler.reset(); hip.profiler.enabled = True
import time
t0 = time.perf_counter(); step(2); torch.cuda.synchronize(); dt = time.perf_counter() - t0
hip.profiler.enabled = False
rows = sorted(hip.profiler.by_shape().items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for _, v in rows)
print(f"step {dt*1e3:.1f} ms, gemm total {tot*1e3:.1f} ms")
for k, (n, f, t) in rows[:45]:
    print(f"{k[0]:8s} M={k[1]:6d} N={k[2]:5d} K={k[3]:6d} b={k[4]:3d}  x{n:3d}  {t*1e3:8.3f} ms  {f/t/1e12:7.1f} TF")
