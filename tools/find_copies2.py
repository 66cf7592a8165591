"""Count real tensor copies (contiguous/float/to/clone/cat/add...) per Python call site during one bench step."""
import os, sys, collections, traceback
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
for i in range(2): step(i)
torch.cuda.synchronize()
log = collections.Counter(); byt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "speecht5_amd/" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{fr.filename[-40:]}:{fr.lineno}"
    return "?"
def wrap(obj, name, is_copy):
    orig = getattr(obj, name)
    def f(*a, **k):
        out = orig(*a, **k)
        try:
            if is_copy(a, out):
                s = (name, site()); log[s] += 1; byt[s] += out.numel() * out.element_size()
        except Exception:
            pass
        return out
    setattr(obj, name, f)
T = torch.Tensor
diff = lambda a, out: isinstance(out, T) and out.is_cuda and out.numel() > 4096 and out.data_ptr() != a[0].data_ptr()
for nm in ("contiguous", "float", "to", "clone", "bfloat16", "long", "int", "type_as", "__add__", "__mul__", "__sub__", "__truediv__", "__radd__", "__rmul__", "add", "mul", "masked_fill", "transpose"):
    if nm == "transpose": continue
    wrap(T, nm, diff)
wrap(torch, "cat", lambda a, out: out.is_cuda and out.numel() > 4096)
wrap(torch, "zeros", lambda a, out: out.is_cuda and out.numel() > 4096)
wrap(torch, "zeros_like", lambda a, out: out.is_cuda and out.numel() > 4096)
step(2); torch.cuda.synchronize()
for s, n in sorted(log.items(), key=lambda kv: -byt[kv[0]])[:50]:
    print(f"{byt[s]/1e6:9.1f} MB  x{n:4d}  {s[0]:12s} {s[1]}")
