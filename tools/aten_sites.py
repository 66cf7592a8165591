"""Which Python lines launch the non-st5 GPU kernels of a step (ATen element-wise / copies / fills / reductions)?
torch.profiler with stacks over ONE eager step: device time and launch count per (innermost speecht5_amd frame, aten op)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample

dev = torch.device("cuda:0")
args, task, model, crit = bench.build(dev, torch.bfloat16)
ddp = FlatGradDataParallel(model); opt = FusedAdam(ddp)
speech = speech_pretrain_sample(B=8, device=dev)
text = text_pretrain_sample(B=16, T=512, vocab=len(task.dicts["text"]), mask_idx=task.dicts["text"].index("<mask>"), device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
micro = {"speech": [speech], "text": [text], "both": [speech, text]}[which]


def step(i):
    ddp.zero_grad()
    ddp.accumulate(micro, lambda s: task.train_step(s, model, crit, None, i, sync=False))
    ddp.finish(); opt.step(0.5)


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, set()])
for e in prof.events():
    if not e.kernels:
        continue
    site = "(autograd / no python frame)"
    for fr in e.stack:
        if "speecht5_amd/" in fr or "bench.py" in fr:
            site = fr.split("speecht5_amd/")[-1] if "speecht5_amd/" in fr else fr
            break
    key = (site, e.name)
    agg[key][0] += len(e.kernels)
    agg[key][1] += sum(k.duration for k in e.kernels)
    for k in e.kernels:
        agg[key][2].add(k.name[:60])
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{sum(v[0] for _, v in rows)} kernels launched by torch ops, {sum(v[1] for _, v in rows)/1e3:.2f} ms device time")
for (site, op), (n, t, names) in rows[:90]:
    print(f"{t:9.1f} us  x{n:4d}  {op:34s} {site[:110]}")
