"""Worker of tests/test_two_rank_gpu.py: one rank of a several-rank replayed update on a box with ONE GPU (the ranks share the
device and exchange over gloo -- bench.py's `shared` branch).  Every rank is given the SAME data and seeds (rank 0's), so the sum
over R ranks is R x one rank's gradient -- an exact power-of-two scaling for R = 2 -- and with grad_scale 1 / (micro-batches x R)
the update must equal the one-rank update BIT FOR BIT on every rank.  Writes a digest of (parameters, moments) per rank.
--own-data: rank r holds ITS OWN data and seeds (bench.make_update(rank=r)) and the learning rate is 0, so the
parameters stay at their initial values and update k's local gradient depends on the rank's data and seeds alone; the worker saves
a strided sample of the gradient buffer as the optimizer step receives it (after the exchange).  fp32 addition of TWO values is
commutative, so the two-rank buffer must equal g_rank0 + g_rank1 of two one-rank runs bit for bit -- a reduction that mixed the
ranks' buffers, ranges or phases would not."""
import argparse
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exchange", default="phased")
    ap.add_argument("--updates", type=int, default=4)
    ap.add_argument("--out", required=True)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--micro", default="in_turn")
    ap.add_argument("--payload", default="fp32")
    ap.add_argument("--own-data", action="store_true")
    ap.add_argument("--data-rank", type=int, default=None, help="one-rank reference run over rank R's data and seeds")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from speecht5_amd import functional as Fn
    # small cfg-2-shaped update: Base, speech 2 x 4 s + text 4 x 128 (every rank: rank 0's data and seeds)
    drank = a.data_rank if a.data_rank is not None else (rank if a.own_data else 0)
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 2, drank, graph=not a.no_graph, micro=a.micro, layerdrop=0.05,
                                         text_batch=4, text_len=128, seconds=4.0, exchange=a.exchange, exchange_payload=a.payload)
    info = {"rank": rank, "world": world, "phased": bool(upd.phased), "split": bool(upd.split), "data_rank": drank}
    grads = []
    p0 = upd.state()[0][::61].float().cpu()
    if a.own_data or a.data_rank is not None:
        upd.opt.lr = 0.0                      # parameters stay put: every update's gradient is a function of (data, seeds) only
        upd.opt.clip = 0.0
        real_step = upd.opt.step

        def step(*args, **kw):                # the eager optimizer step (one rank eager; several ranks: the eager tail of a replay)
            grads.append(upd.ddp.flat[::61].clone())
            return real_step(*args, **kw)
        upd.opt.step = step
    if a.no_graph:
        Fn._S.force_static = True
        for _ in range(a.updates):
            upd.eager_update()
        Fn._S.force_static = False
    else:
        upd.prepare_graph()
        for _ in range(a.updates - 2):
            upd.update()
        upd.finish()
    p, m, v, t = upd.state()
    h = hashlib.sha1()
    for x in (p, m, v):
        h.update(x.cpu().numpy().tobytes())
    info.update(digest=h.hexdigest(), t=int(t), pnorm=float(p.double().norm()), finite=bool(torch.isfinite(p).all()))
    torch.save({"p": p[::61].float().cpu(), "p0": p0}, f"{a.out}.rank{rank}.params.pt")
    if grads:
        info["grad_calls"] = len(grads)
        torch.save(grads[-1].cpu(), f"{a.out}.rank{rank}.grad.pt")
    json.dump(info, open(f"{a.out}.rank{rank}.json", "w"))
    upd.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
