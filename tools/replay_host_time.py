"""Host-side cost of one graph replay: time of StepGraph._pre_replay() (seed slots, staged host inputs, hyper-parameters) and of
graph.replay() itself (does hipGraphLaunch return before the GPU is done?), against the GPU time of the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speecht5_amd import functional as Fn
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.graph import StepGraph
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample

dev = torch.device("cuda:0")
args, task, model, crit = bench.build(dev, torch.bfloat16)
ddp = FlatGradDataParallel(model); opt = FusedAdam(ddp)
speech = speech_pretrain_sample(B=8, device=dev)
text = text_pretrain_sample(B=16, T=512, vocab=len(task.dicts["text"]), mask_idx=task.dicts["text"].index("<mask>"), device=dev)
n = [0]


def one_update():
    ddp.zero_grad()
    ddp.accumulate([speech, text], lambda s: task.train_step(s, model, crit, None, n[0], sync=False))
    ddp.finish(); opt.step(0.5)


def advance():
    n[0] += 1
    model.set_num_updates(n[0])


sg = StepGraph(one_update, opt=opt, model=model, device=dev, on_step=advance)
for _ in range(2):
    advance(); one_update()
sg.record(); sg.record(); sg.capture()
t_pre, t_launch = [], []
with torch.cuda.stream(sg.stream):
    for i in range(12):
        t0 = time.perf_counter()
        if sg._pending:
            sg._pending = False
        else:
            sg._pre_replay()
        t1 = time.perf_counter()
        sg.graph.replay()
        t2 = time.perf_counter()
        opt.t += 1
        t_pre.append(t1 - t0); t_launch.append(t2 - t1)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
print("pre_replay ms:", " ".join(f"{x*1e3:.2f}" for x in t_pre))
print("graph.replay() call ms:", " ".join(f"{x*1e3:.2f}" for x in t_launch))
print(f"final synchronize waited {1e3*(t4-t3):.1f} ms")
