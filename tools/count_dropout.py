"""Count separate dropout-kernel launches in one bench step (after the LayerNorm-backward fusion)."""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from speecht5_amd import functional as Fn
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample
dev = torch.device("cuda:0")
args, task, model, crit = bench.build(dev, torch.bfloat16)
ddp = FlatGradDataParallel(model); opt = FusedAdam(ddp, lr=2e-4)
Fn.manual_seed(1)
speech = speech_pretrain_sample(B=8, seconds=10.0, device=dev, seed=1)
text = text_pretrain_sample(B=16, T=512, vocab=len(task.dicts["text"]), mask_idx=task.dicts["text"].index("<mask>"), device=dev, seed=2)
n = [0]; orig = Fn._dropout
def counted(x, p, seed):
    n[0] += 1
    return orig(x, p, seed)
Fn._dropout = counted
for i in range(2):
    n[0] = 0
    ddp.zero_grad()
    for s in (speech, text):
        task.train_step(s, model, crit, None, i, sync=False)
    ddp.finish(); opt.step(grad_scale=0.5)
torch.cuda.synchronize()
print("separate dropout launches per step:", n[0], " tags left:", len(Fn._drop_tags), " grads left:", len(Fn._drop_grads))
