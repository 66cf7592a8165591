"""Attribute device-to-device copies / torch element-wise kernels of one bench step to Python call sites."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
args, task, model, crit = bench.build(dev, torch.bfloat16)
ddp = FlatGradDataParallel(model); opt = FusedAdam(ddp)
speech = speech_pretrain_sample(B=8, device=dev)
text = text_pretrain_sample(B=16, T=512, vocab=len(task.dicts["text"]), mask_idx=task.dicts["text"].index("<mask>"), device=dev)
def step(i):
    ddp.zero_grad()
    for s in (speech, text):
        task.train_step(s, model, crit, None, i, sync=False)
    ddp.finish(); opt.step(0.5)
for i in range(2): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2); torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    n = ev.name
    if not (n.startswith("Memcpy") or n.startswith("Memset") or "hipMemcpy" in n or "hipMemset" in n):
        continue
    chain = []
    par = ev.cpu_parent
    while par is not None and len(chain) < 5:
        chain.append(par.name[:50]); par = par.cpu_parent
    key = (n[:40], " <- ".join(chain))
    agg[key][0] += 1; agg[key][1] += getattr(ev, "self_device_time_total", 0) or ev.cpu_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
for (name, chain), (n, t) in rows[:40]:
    print(f"x{n:4d} {t:8.0f} us {name:40s} {chain}")
