"""Which call sites of the timed update still launch torch (ATen) kernels?  One eager in-turn update of bench.make_update's
object under torch.profiler with Python stacks; every device kernel that is not one of the library's is attributed to the
innermost frame inside speecht5_amd/ (or bench.py) and the launches are counted per (call site, kernel).
python tools/r5/torch_tail.py > gpurun_out/r5_torch_tail.txt"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speecht5_amd import functional as Fn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    _, _, model, upd = bench.make_update(dev, graph=False, micro="in_turn")
    Fn._S.force_static = True
    upd.eager_update()
    upd.eager_update()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        upd.eager_update()
        torch.cuda.synchronize()
    ev = prof.events()
    # CPU ops with stacks; device kernels linked through correlation ids
    per_site = collections.Counter()
    dur_site = collections.Counter()
    ours = 0
    for e in ev:
        if e.device_type.name != "CPU" or not e.kernels:
            continue
        name = e.name
        if not (name.startswith("aten::") or "Memcpy" in name or "Memset" in name):
            continue
        # only leaf ops: an aten op whose kernels are also reported by a child would be counted twice; events() gives
        # kernels on the op that launched them (the innermost), parents carry none
        site = None
        node = e
        top = e
        while node is not None and site is None:
            for fr in (node.stack or []):
                if "speecht5_amd/" in fr or "bench.py" in fr:
                    site = fr[fr.index("speecht5_amd/"):] if "speecht5_amd/" in fr else fr[fr.index("bench.py"):]
                    site = site[:110]
                    break
            top = node
            node = node.cpu_parent
        if site is None:        # autograd thread: no Python frames; name the graph node that ran the op
            node = e
            while node is not None:
                if "evaluate_function" in node.name or "Backward" in node.name:
                    site = node.name
                node = node.cpu_parent
            site = site or ("top: " + top.name)
        for k in e.kernels:
            per_site[(site, name)] += 1
            dur_site[(site, name)] += k.duration
    tot = sum(per_site.values())
    print(f"torch-launched device kernels in one eager update: {tot}, {sum(dur_site.values()) / 1e3:.3f} ms")
    by_site = collections.Counter()
    for (s, n), c in per_site.items():
        by_site[s] += c
    for s, c in by_site.most_common():
        ops = ", ".join(f"{n.replace('aten::', '')} x{per_site[(s2, n)]}" for (s2, n) in per_site if s2 == s)
        d = sum(v for (s2, n), v in dur_site.items() if s2 == s)
        print(f"{c:4d}  {d / 1e3:7.3f} ms  {s}  [{ops}]")


if __name__ == "__main__":
    main()
