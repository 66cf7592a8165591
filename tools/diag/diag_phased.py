import os, sys, collections
os.environ["ST5_DDP_FORCE_COLLECTIVES"] = "1"
os.environ["ST5_EAGER_PHASED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import bench
from speecht5_amd import functional as Fn
cuda = torch.device("cuda:0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29811", rank=0, world_size=1, device_id=cuda)
def grads(exchange):
    upd = None
    try:
        _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=False, micro="in_turn", layerdrop=0.05, prefetch_host=False, exchange=exchange)
        Fn._S.force_static = True
        upd.advance()
        with torch.cuda.stream(upd.stream):
            if upd.phased:
                for fn, bt in zip(*upd.phase_fns()):
                    fn(); bt()
                upd.ddp.wait_reductions()
            else:
                upd.local_part(); upd.ddp.all_reduce_gradients(average=False)
        torch.cuda.synchronize()
        print("phased:", upd.phased, flush=True)
        names = {}
        for (n, p), o in zip([(n, p) for n, p in model.named_parameters()], range(10**6)):
            names[n] = p.grad.detach().float().clone()
        return names
    finally:
        Fn._S.force_static = False
        if upd is not None: upd.close()
        Fn.bf16_mirror.__init__(); Fn.weight_cache.clear(); Fn.set_layer_boundary_hook(None); Fn.set_compute_dtype(torch.float32)
a = grads("one_message"); b = grads("phased"); c = grads("one_message")
def rep(tag, x, y):
    bad = [(k, float((x[k] - y[k]).abs().max()), float(x[k].abs().max())) for k in x if not torch.equal(x[k], y[k])]
    print(f"{tag}: {len(bad)} of {len(x)} gradients differ", flush=True)
    bad.sort(key=lambda t: -t[1] / max(t[2], 1e-30))
    for k, d, m in bad[:24]:
        print(f"    {k:60s} max diff {d:.3e} of {m:.3e}", flush=True)
rep("one_message vs one_message", a, c)
rep("phased vs one_message", b, a)
