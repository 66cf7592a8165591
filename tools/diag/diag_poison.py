"""Diagnostic (GPU): does any kernel of the full-size update consume uninitialised memory?  One eager update with every
torch.empty / hipMalloc'ed buffer poisoned (0xFF bytes = NaN) vs the same update unpoisoned."""
import os, sys
os.environ["ST5_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn
from tests.util import poisoned_allocations
cuda = torch.device("cuda:0")

def run(poison, static, micro="in_turn", n=1):
    upd = None
    try:
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=False, micro=micro, layerdrop=0.0, prefetch_host=False, wgrad_stream=False)
        upd.opt.clip = 0.0
        Fn._S.force_static = static
        ctx = poisoned_allocations() if poison else __import__("contextlib").nullcontext()
        with ctx:
            for _ in range(n):
                upd.eager_update()
            torch.cuda.synchronize()
        return {k: p.detach().float().clone() for k, p in model.named_parameters()}
    finally:
        Fn._S.force_static = False
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__(); Fn.weight_cache.clear(); Fn.set_layer_boundary_hook(None); Fn.set_compute_dtype(torch.float32)

for static in (True, False):
    for micro in ("in_turn", "in_turn_2buf"):
        clean = run(False, static, micro)
        dirty = run(True, static, micro)
        bad = [(k, int((~torch.isfinite(dirty[k])).sum()), float((dirty[k] - clean[k]).abs().nan_to_num(0).max())) for k in clean
               if not torch.equal(dirty[k], clean[k])]
        print(f"== static={static} micro={micro}: {len(bad)} of {len(clean)} parameters differ under poisoning", flush=True)
        for k, nn, d in bad[:40]:
            print(f"   {k:70s} non-finite {nn:9d}  max diff {d:.2e}", flush=True)
