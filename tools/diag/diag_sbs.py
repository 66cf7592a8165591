"""Diagnostic (GPU): eager side-by-side backward vs in-turn backward (same two-buffer arithmetic) after n updates, under
switches that remove one source of concurrency at a time; reports which parameters differ (clipping off: a differing gradient
only moves its own parameter)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn
cuda = torch.device("cuda:0")

def run(micro, n, env=(), sync_between=False, ld=0.0):
    upd = None
    for k, v in env:
        os.environ[k] = v
    try:
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=False, micro=micro, layerdrop=ld, prefetch_host=False, wgrad_stream=False)
        upd.opt.clip = 0.0
        if os.environ.get("DIAG_NO_SPLITK_DEFER"):
            from speecht5_amd import hip
            hip.check(hip.lib().st5_gemm_defer_splitk(0, hip.stream()), "x")
        Fn._S.force_static = True
        for _ in range(n):
            upd.eager_update()
            if sync_between:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return {k: p.detach().float().clone() for k, p in model.named_parameters()}
    finally:
        for k, _ in env:
            os.environ.pop(k, None)
        Fn._S.force_static = False
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__(); Fn.weight_cache.clear(); Fn.set_layer_boundary_hook(None); Fn.set_compute_dtype(torch.float32)

def report(tag, a, b, show=12):
    groups = collections.OrderedDict()
    for k in a:
        d = float((a[k] - b[k]).abs().max())
        key = k if os.environ.get("DIAG_FULL") else ".".join(k.split(".")[:3])
        g = groups.setdefault(key, [0, 0, 0.0])
        g[0] += 1; g[1] += d > 0; g[2] = max(g[2], d)
    bad = {k: v for k, v in groups.items() if v[1]}
    print(f"== {tag}: {sum(v[1] for v in groups.values())} of {sum(v[0] for v in groups.values())} tensors differ", flush=True)
    for k, v in list(bad.items())[:show]:
        print(f"   {k:60s} {v[1]}/{v[0]}  max {v[2]:.2e}", flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["DIAG_FULL"] = "1"
ref = run("in_turn_2buf", n)
for rep in range(5):
    report(f"side_by_side n={n} rep {rep}", run("side_by_side", n), ref, show=12)
os.environ["DIAG_NO_SPLITK_DEFER"] = "1"
ref = run("in_turn_2buf", n)
for rep in range(4):
    report(f"no split-K deferral: side_by_side n={n} rep {rep}", run("side_by_side", n), ref, show=12)
del os.environ["DIAG_NO_SPLITK_DEFER"]
ref = run("in_turn_2buf", n, env=(("ST5_LN_DEFER", "0"),))
for rep in range(4):
    report(f"ST5_LN_DEFER=0: side_by_side n={n} rep {rep}", run("side_by_side", n, env=(("ST5_LN_DEFER", "0"),)), ref, show=12)
