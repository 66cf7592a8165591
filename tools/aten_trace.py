"""Every ATen op a training step dispatches on GPU tensors, by Python call site (TorchDispatchMode + traceback; backward runs on
the calling thread so the mode sees it).  Views / metadata ops are skipped.  Answers: which lines still launch torch kernels?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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


VIEW = ("view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
        "t.default", "_unsafe_view", "unbind", "split", "chunk", "narrow", "empty", "size", "stride", "is_", "_local_scalar", "lift_fresh",
        "unfold", "diagonal", "record_stream", "_reshape_alias", "new_empty", "set_", "resize_", "_has_", "item")
log = collections.Counter(); elems = collections.Counter()


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW):
            return out
        ts = [a for a in list(args) + ([out] if isinstance(out, torch.Tensor) else list(out) if isinstance(out, (tuple, list)) else [])
              if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "speecht5_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                site = f"{fr.filename.split('speecht5_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
        log[(site, name)] += 1
        elems[(site, name)] += max((t.numel() for t in ts), default=0)
        return out


for i in range(2):
    step(i)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Trace():
    step(2)
torch.cuda.synchronize()
tot = sum(log.values())
print(f"{tot} non-view ATen ops on GPU tensors in one step ({which})")
bysite = collections.Counter()
for (site, name), n in log.items():
    bysite[site] += n
print("---- by site")
for site, n in bysite.most_common(70):
    ops = ", ".join(f"{nm.replace('aten.', '')}x{c}" for (s, nm), c in sorted(log.items(), key=lambda kv: -kv[1]) if s == site)
    print(f"{n:4d}  {site:60s} {ops[:150]}")
