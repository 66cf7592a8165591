"""Call sites of the ATen operators the timed update still issues on device tensors (each is a torch kernel launch or a copy):
one eager in-turn update of bench.make_update's object under a TorchDispatchMode that records, per operator call with a CUDA
tensor argument, the innermost Python frame inside speecht5_amd/ (forward ops and the backward of the library's own autograd
Functions run Python; the autograd engine's own accumulations have no Python frame and are listed as <engine>).
python tools/r5/torch_sites.py > gpurun_out/r5_torch_sites.txt"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speecht5_amd import functional as Fn  # noqa: E402

NO_KERNEL = ("view", "reshape", "as_strided", "detach", "alias", "empty", "t.default", "transpose", "permute", "expand", "slice", "select.int",
             "unsqueeze", "squeeze", "_unsafe_view", "split", "unbind", "is_", "size", "stride", "storage_offset", "numel", "dim",
             "_local_scalar_dense", "record_stream", "lift_fresh", "_to_copy_skip", "unfold", "narrow", "chunk", "contiguous", "set_", "resize_")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = str(func).replace("aten.", "")
        flat = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [t for t in a if isinstance(t, torch.Tensor)]
        if any(t.is_cuda for t in flat) and not any(name.startswith(k) or ("." + k) in name for k in NO_KERNEL):
            site = "<engine>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if "speecht5_amd/" in fr.filename and "torch_sites" not in fr.filename:
                    site = f"{fr.filename.split('speecht5_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
                if fr.filename.endswith("bench.py"):
                    site = f"bench.py:{fr.lineno} {fr.name}"
                    break
            shp = "x".join(str(d) for d in flat[0].shape) if flat else ""
            self.count[(site, name, shp)] += 1
        return func(*args, **kwargs)


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    _, _, model, upd = bench.make_update(dev, graph=False, micro="in_turn")
    Fn._S.force_static = True
    upd.eager_update()
    torch.cuda.synchronize()
    with Sites() as m:
        upd.eager_update()
        torch.cuda.synchronize()
    tot = sum(m.count.values())
    print(f"ATen operator calls on device tensors in one eager update (views and allocations excluded): {tot}")
    by_site = collections.defaultdict(list)
    for (site, name, shp), c in m.count.items():
        by_site[site].append((c, name, shp))
    for site, ops in sorted(by_site.items(), key=lambda kv: -sum(o[0] for o in kv[1])):
        n = sum(o[0] for o in ops)
        print(f"{n:4d}  {site}")
        for c, name, shp in sorted(ops, reverse=True):
            print(f"        {c:3d} x {name} [{shp}]")


if __name__ == "__main__":
    main()
