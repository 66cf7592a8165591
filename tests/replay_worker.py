"""Worker of tests/test_replay_long_gpu.py: ONE process replays N seeded updates of the benched workload (bench.make_update, full
size) in the given micro-batch mode and writes the trajectory of (parameters, first moment, second moment) integer checksums, computed
on the device behind every update without a host synchronisation.  Every random stream is re-seeded before each update, so update k
draws the same masks / dropout seeds / LayerDrop flags in every process and in every mode."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="side_by_side")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--sync", type=int, default=0, help="1: host synchronisation behind every update")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import bench
    from speecht5_amd import functional as Fn
    cuda = torch.device("cuda:0")

    def seed(k):
        Fn._S.seed, Fn._S.counter = 4242, 1 + 100000 * k
        np.random.seed(1000 + k)
        torch.manual_seed(1000 + k)

    seed(0)
    _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=True, micro=a.mode, layerdrop=0.05, prefetch_host=False)
    seed(0)
    upd.prepare_graph()
    cs = torch.zeros(a.n, 3, dtype=torch.int64, device=cuda)
    for k in range(1, a.n + 1):
        seed(k)
        upd.update()
        with torch.cuda.stream(upd.sg.stream):
            for j, x in enumerate((upd.opt.pflat, upd.opt.m, upd.opt.v)):
                cs[k - 1, j] = x.view(torch.int32).sum(dtype=torch.int64)
        if a.sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    finite = bool(torch.isfinite(upd.opt.pflat).all())
    json.dump({"mode": a.mode, "cs": cs.cpu().tolist(), "finite": finite}, open(a.out, "w"))
    upd.close()


if __name__ == "__main__":
    main()
