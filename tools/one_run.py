"""One full-size update sequence in a fresh process; prints a hash of the final parameters + Adam moments.
usage: one_run.py <eager|graph> <in_turn|in_turn_2buf|side_by_side> <n_updates> [layerdrop]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speecht5_amd import functional as Fn
cuda = torch.device("cuda:0")
graph, micro, n = sys.argv[1] == "graph", sys.argv[2], int(sys.argv[3])
ld = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
clip = os.environ.get("DIAG_CLIP")
_, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=graph, micro=micro, layerdrop=ld, prefetch_host=False, wgrad_stream=False)
if clip is not None:
    upd.opt.clip = float(clip)
if graph:
    upd.prepare_graph()
    for _ in range(n - 2):
        upd.update()
    upd.finish()
else:
    Fn._S.force_static = True
    for _ in range(n):
        upd.eager_update()
p, m, v, t = upd.state()
h = hashlib.sha1(p.cpu().numpy().tobytes() + m.cpu().numpy().tobytes() + v.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"RESULT {sys.argv[1]:5s} {micro:13s} n={n} ld={ld} finite={bool(torch.isfinite(p).all())} hash={h} psum={float(p.double().sum()):.10f}", flush=True)

out = os.environ.get("DIAG_DUMP")
if out:
    import json
    torch.cuda.synchronize()
    d = {k: hashlib.sha1(q.detach().float().cpu().numpy().tobytes()).hexdigest()[:10] for k, q in model.named_parameters()}
    json.dump(d, open(out, "w"))
