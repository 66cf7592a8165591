"""CPU, world_size 2 over gloo: the flat-gradient data-parallel layer gives every rank the mean gradient of the
concatenated batch (N-rank result == 1-rank result), reduces parameters the local micro-batch never touched
(zeros take part, as fairseq legacy_ddp + --find-unused-parameters), and launches buckets from the backward
triggers.  The model here is a plain-torch stand-in with the same module layout hooks; the DDP code is the
product's (speecht5_amd/ddp.py)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from speecht5_amd import functional as Fn


class Layer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc = nn.Linear(d, d)

    def forward(self, x):
        x = Fn.layer_boundary(x, self)
        return torch.tanh(self.fc(x))


class Toy(nn.Module):
    def __init__(self, d=8):
        super().__init__()
        self.inp = nn.Linear(d, d)
        self.layers = nn.ModuleList([Layer(d) for _ in range(3)])
        self.unused = nn.Linear(d, d)       # never touched by any rank
        self.rank1_only = nn.Linear(d, 1)   # touched by rank 1 only ("other modality")
        self.head = nn.Linear(d, 1)

    def forward(self, x, use_extra):
        x = self.inp(x)
        for l in self.layers:
            x = l(x)
        y = self.head(x)
        if use_extra:
            y = y + self.rank1_only(x)
        return y


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speecht5_amd.ddp import FlatGradDataParallel
    torch.manual_seed(0)
    model = Toy()
    launched = []
    ddp = FlatGradDataParallel(model, bucket_groups=[[model.head, model.rank1_only]] + [[l] for l in reversed(list(model.layers))])
    orig = ddp._bucket_ready
    ddp._bucket_ready = lambda bi: (launched.append(bi), orig(bi))[1]
    torch.manual_seed(1)
    X = torch.randn(8, 8)
    xb = X[rank * 4:(rank + 1) * 4]
    ddp.zero_grad()
    loss = model(xb, use_extra=(rank == 1)).pow(2).mean()
    loss.backward()
    triggered = list(launched)
    ddp.finish()
    ret[rank] = dict(flat=ddp.flat.clone(), triggered=triggered, nb=len(ddp.buckets),
                     grads={n: p.grad.clone() for n, p in model.named_parameters()})
    dist.destroy_process_group()


def test_two_rank_mean_equals_single_process():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    # single-process reference: mean over the two micro-batches of each micro-batch's loss gradient
    torch.manual_seed(0)
    model = Toy()
    torch.manual_seed(1)
    X = torch.randn(8, 8)
    tot = None
    for r in range(2):
        model.zero_grad()
        model(X[r * 4:(r + 1) * 4], use_extra=(r == 1)).pow(2).mean().backward()
        g = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
        tot = g if tot is None else {n: tot[n] + g[n] for n in g}
    ref = {n: v / 2 for n, v in tot.items()}
    for r in range(2):
        for n, v in ref.items():
            assert torch.allclose(ret[r]["grads"][n], v, atol=1e-6), (r, n)
        assert float(ret[r]["grads"]["unused.weight"].abs().sum()) == 0.0
    assert torch.equal(ret[0]["flat"], ret[1]["flat"])
    # the layer triggers fired during backward for the layer buckets whose input needed a gradient
    assert len(ret[0]["triggered"]) >= 2 and ret[0]["nb"] == 5
