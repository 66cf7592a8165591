"""Which parameters differ between the one-rank in-turn update and the phased several-rank form (one-rank RCCL group), with the
weight-gradient groups on?  python tools/r5/group_diff.py [n_updates]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NCCL_ALGO"] = "Ring"
os.environ["ST5_EAGER_PHASED"] = "1"
import bench  # noqa: E402
from speecht5_amd import functional as Fn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def run(exchange, force):
    if force:
        os.environ["ST5_DDP_FORCE_COLLECTIVES"] = "1"
    else:
        os.environ.pop("ST5_DDP_FORCE_COLLECTIVES", None)
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 8, 0, graph=False, micro="in_turn", layerdrop=0.05, exchange=exchange)
    Fn._S.force_static = True
    for _ in range(n):
        upd.eager_update()
    torch.cuda.synchronize()
    out = {k: v.detach().clone() for k, v in model.named_parameters()}
    info = dict(phased=upd.phased, split=upd.split)
    Fn._S.force_static = False
    upd.close()
    Fn.bf16_mirror.__init__()
    Fn.weight_cache.clear()
    Fn.set_layer_boundary_hook(None)
    Fn.set_compute_dtype(torch.float32)
    return out, info


import torch.distributed as dist
a, ia = run("phased", False)
os.environ["ST5_DDP_FORCE_COLLECTIVES"] = "1"
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29700 + os.getpid() % 200}", rank=0, world_size=1, device_id=dev)
b, ib = run("phased", True)
print(ia, ib)
bad = [(k, float((a[k] - b[k]).abs().max()), tuple(a[k].shape)) for k in a if not torch.equal(a[k], b[k])]
print(len(bad), "of", len(a), "parameters differ")
for k, d, s in bad[:60]:
    print(f"  {d:.3e}  {s}  {k}")
