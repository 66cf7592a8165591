"""How much of a bench step is host-side launch overhead?  Measures the enqueue time of steps (no sync inside) against
the synchronized wall time."""
import os, sys, time
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
n = 6
t0 = time.perf_counter()
enq = []
for i in range(n):
    t = time.perf_counter(); step(i); enq.append(time.perf_counter() - t)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("enqueue ms per step:", [round(e * 1e3, 1) for e in enq])
print(f"mean enqueue {t_enq/n*1e3:.1f} ms/step, synchronized wall {t_all/n*1e3:.1f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(0); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(45)
