"""Do the padding gaps of the flat gradient buffers stay zero?  (They are inside the range the gradient-norm kernel sums.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn
cuda = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "side_by_side"
_, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=False, micro=mode, layerdrop=0.05, prefetch_host=False, wgrad_stream=False)
ddp = upd.ddp
names = {id(p): n for n, p in model.named_parameters()}
gaps, prev = [], 0
for p, o in zip(ddp.params, ddp.offsets):
    if o != prev:
        gaps.append((prev, o, names[id(p)]))
    prev = o + p.numel()
Fn._S.force_static = True
for it in range(3):
    upd.advance()
    with torch.cuda.stream(upd.stream):
        # the update without its optimizer step: gradients stay in the buffers
        ddp.zero_grad()
        if mode == "in_turn":
            ddp.accumulate(upd.micro, lambda s: upd.task.train_step(s, upd.model, upd.crit, None, upd.n, sync=False))
        else:
            ddp.accumulate_overlapped(upd.micro, upd._fwd, backward="side_by_side")
        ddp.finish() if mode == "in_turn" else None
    torch.cuda.synchronize()
    for buf, nm in ((ddp.flat, "flat"), (ddp.flat2, "flat2")):
        if buf is None:
            continue
        for a, b, nxt in gaps:
            g = buf[a:b]
            if bool((g != 0).any()) or not bool(torch.isfinite(g).all()):
                print(f"iter {it} {nm}: gap [{a},{b}) before {nxt}: nonzero {int((g != 0).sum())} values {g[g != 0][:6].tolist()}", flush=True)
    ddp.flat.zero_()
    if ddp.flat2 is not None:
        ddp.flat2.zero_()
    ddp._pair_pending = False
print("gap check done", mode)
