"""Wall time of the phases of one update (eager enqueue, events on the main stream): speech forward / backward, text forward /
backward, finish (gradient sync + split-K flush + join of the weight-gradient stream), optimizer.  Run with
ST5_WGRAD_STREAM=0 to see the phases with the weight-gradient GEMMs inline."""
import os, sys
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
names = ["zero_grad", "speech fwd", "speech bwd", "text fwd", "text bwd", "finish", "adam"]
acc = {n: 0.0 for n in names}


def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e


def step(i, rec):
    model.train(); model.set_num_updates(i)
    marks = [ev()]
    ddp.zero_grad(); marks.append(ev())
    with ddp.no_sync():
        loss, ss, _ = crit(model, speech); marks.append(ev())
        (loss / ss).backward(); marks.append(ev())
    loss, ss, _ = crit(model, text); marks.append(ev())
    (loss / ss).backward(); marks.append(ev())
    ddp.finish(); marks.append(ev())
    opt.step(0.5); marks.append(ev())
    if rec:
        torch.cuda.synchronize()
        for n, a, b in zip(names, marks[:-1], marks[1:]):
            acc[n] += a.elapsed_time(b)


for i in range(3):
    step(i, False)
N = 5
for i in range(N):
    step(3 + i, True)
tot = sum(acc.values()) / N
print(f"ST5_WGRAD_STREAM={os.environ.get('ST5_WGRAD_STREAM', '1')}: {tot:.2f} ms per update (eager, main-stream events)")
for n in names:
    print(f"  {n:12s} {acc[n]/N:7.2f} ms")
