"""Feasibility: forward passes (model + criterion) of the speech and the text micro-batch one after the other on one stream vs
side by side on two streams (timing only: the shared scratch workspaces are not per-stream yet, values of (b) are not checked)."""
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
model.train(); model.set_num_updates(1)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()


def seq():
    l1 = crit(model, speech)[0]
    l2 = crit(model, text)[0]
    return l1, l2


def par():
    s2.wait_stream(main)
    l1 = crit(model, speech)[0]
    with torch.cuda.stream(s2):
        l2 = crit(model, text)[0]
    main.wait_stream(s2)
    return l1, l2


from speecht5_amd import functional as Fn
from speecht5_amd.graph import StepGraph
keep = []


def seq_fn():
    keep.clear()
    keep.extend(seq())


def par_fn():
    keep.clear()
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    l1 = crit(model, speech)[0]
    with torch.cuda.stream(s2):
        l2 = crit(model, text)[0]
    cur.wait_stream(s2)
    keep.extend((l1, l2))


for fn, name in ((seq_fn, "sequential"), (par_fn, "two streams")):
    sg = StepGraph(fn, opt=None, model=model, device=dev)
    sg.record(); sg.record(); sg.capture()
    with torch.cuda.stream(sg.stream):
        for _ in range(3):
            sg.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sg.replay()
        e1.record()
        torch.cuda.synchronize()
    print(f"{name:12s} (graph replay): {e0.elapsed_time(e1)/10:.2f} ms per forward pair", flush=True)
