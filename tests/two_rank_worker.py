"""Worker of tests/test_two_rank_gpu.py: one rank of a several-rank replayed update on a box with ONE GPU (the ranks share the
device and exchange over gloo -- bench.py's `shared` branch).  Every rank is given the SAME data and seeds (rank 0's), so the sum
over R ranks is R x one rank's gradient -- an exact power-of-two scaling for R = 2 -- and with grad_scale 1 / (micro-batches x R)
the update must equal the one-rank update BIT FOR BIT on every rank.  Writes a digest of (parameters, moments) per rank."""
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
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 2, 0, graph=not a.no_graph, micro="in_turn", layerdrop=0.05,
                                         text_batch=4, text_len=128, seconds=4.0, exchange=a.exchange)
    info = {"rank": rank, "world": world, "phased": bool(upd.phased), "split": bool(upd.split)}
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
    json.dump(info, open(f"{a.out}.rank{rank}.json", "w"))
    upd.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
