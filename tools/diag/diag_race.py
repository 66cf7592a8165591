"""Diagnostic (GPU): run-to-run determinism of the replayed full-size update under different amounts of concurrency."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn
cuda = torch.device("cuda:0")

def run(graph, micro, n, ld=0.0, serial=False):
    upd = None
    if serial:
        os.environ["ST5_SERIAL_MICRO"] = "1"
    try:
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=graph, micro=micro, layerdrop=ld, prefetch_host=False, wgrad_stream=False)
        upd.opt.clip = 0.0
        if graph:
            upd.prepare_graph()
            for _ in range(n - 2):
                upd.update()
            upd.finish()
        else:
            Fn._S.force_static = True
            for _ in range(n):
                upd.eager_update()
        return upd.state()[0]
    finally:
        os.environ.pop("ST5_SERIAL_MICRO", None)
        Fn._S.force_static = False
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__(); Fn.weight_cache.clear(); Fn.set_layer_boundary_hook(None); Fn.set_compute_dtype(torch.float32)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ref = run(False, "in_turn_2buf", n)
ref1 = run(False, "in_turn", n)
print(f"eager in_turn (1 buffer) vs in_turn_2buf: {float((ref1 - ref).abs().max()):.3e}", flush=True)
for tag, kw, base, nn in (("eager in_turn", dict(graph=False, micro="in_turn"), ref1, n),
                          ("eager in_turn_2buf", dict(graph=False, micro="in_turn_2buf"), ref, n),
                          ("eager side_by_side", dict(graph=False, micro="side_by_side"), ref, n),
                          ("graph in_turn (1 stream, 1 buffer)", dict(graph=True, micro="in_turn"), ref1, n),
                          ("graph side_by_side", dict(graph=True, micro="side_by_side"), ref, n)):
    ds = [float((run(n=nn, **kw) - base).abs().max()) for _ in range(reps)]
    print(f"{tag:45s} n={nn} vs eager: " + " ".join(f"{d:.2e}" for d in ds), flush=True)
ref3 = run(False, "in_turn", 3)
ds = [float((run(True, "in_turn", 3) - ref3).abs().max()) for _ in range(reps)]
print(f"{'graph in_turn (1 stream, 1 buffer)':45s} n=3 vs eager: " + " ".join(f"{d:.2e}" for d in ds), flush=True)
