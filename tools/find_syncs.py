"""Lists host<->device synchronisation points inside one bench step (torch sync-debug mode) and the CPU enqueue time."""
import os, sys, time, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample
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
for i in range(3): step(i)
torch.cuda.synchronize()
# CPU enqueue time vs wall
t0 = time.perf_counter()
for i in range(3): step(3 + i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/3:.1f} ms/step, wall {1e3*(t2-t0)/3:.1f} ms/step")
torch.cuda.set_sync_debug_mode("warn")
sites = collections.Counter()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step(10)
torch.cuda.set_sync_debug_mode("default")
for x in w:
    if "synchroniz" in str(x.message):
        sites[f"{os.path.basename(x.filename)}:{x.lineno}"] += 1
for k, v in sites.most_common():
    print(v, k)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(11); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
