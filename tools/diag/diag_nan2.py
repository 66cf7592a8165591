import sys, os
os.environ["ST5_POISON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn, hip
cuda = torch.device("cuda:0")
B, S, C, k, s = 8, 160000, 512, 10, 5
L = (S - k) // s + 1
def run(micro, n, own_stream=False):
    upd = None
    try:
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=False, micro=micro, layerdrop=0.0, prefetch_host=False, wgrad_stream=False)
        upd.opt.clip = 0.0
        Fn._S.force_static = True
        st = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            for _ in range(n):
                upd.eager_update()
        torch.cuda.synchronize()
        bad = sum(int((~torch.isfinite(p)).sum()) for p in model.parameters())
        # the workspace of the stream the speech micro-batch ran on: which conv0 partials are still NaN?
        key = (cuda.type, cuda.index, st.cuda_stream)
        ws = hip._ws.get(key)
        info = None
        if ws is not None:
            nch = (L + 255) // 256
            for tch in (128, 256, 512):
                pass
            part = ws.view(torch.float32)
            info = int(torch.isnan(part[: 8 * 1000 * 512 * 12 // 8]).sum())
        return bad, info
    finally:
        Fn._S.force_static = False
        if upd is not None:
            upd.close()
        Fn.bf16_mirror.__init__(); Fn.weight_cache.clear(); Fn.set_layer_boundary_hook(None); Fn.set_compute_dtype(torch.float32)
print("in_turn_2buf            ", run("in_turn_2buf", 1), flush=True)
for rep in range(4):
    print("side_by_side null stream", run("side_by_side", 1), flush=True)
for rep in range(4):
    print("side_by_side own stream ", run("side_by_side", 1, own_stream=True), flush=True)
