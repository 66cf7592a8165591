"""How long is each micro-batch's chain on its own?  Replayed updates of (speech only), (text only), (both in turn), (both side by
side): the side-by-side update cannot be shorter than the longer chain + the optimizer tail."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
def run(which, micro):
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 8, 0, graph=True, micro=micro)
    if which is not None:
        upd.micro = [upd.micro[which]]
    upd.prepare_graph()
    for _ in range(3):
        upd.update()
    upd.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        upd.update()
    upd.finish(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    upd.close()
    del upd, model
    torch.cuda.empty_cache()
    return dt * 1e3
for name, which, micro in (("speech only", 0, "in_turn"), ("text only", 1, "in_turn"), ("both, in turn", None, "in_turn"), ("both, side by side", None, "side_by_side")):
    print(f"{name:22s} {run(which, micro):7.2f} ms per update", flush=True)
