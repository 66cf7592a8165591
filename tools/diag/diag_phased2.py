import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from tests.test_bench_update_gpu import _run
cuda = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref = _run(cuda, True, "in_turn", n)
refe = _run(cuda, False, "in_turn", n)
print("one-rank graph vs one-rank eager:", float((ref[0] - refe[0]).abs().max()), flush=True)
os.environ["ST5_DDP_FORCE_COLLECTIVES"] = "1"
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29812", rank=0, world_size=1, device_id=cuda)
for graph in (True,):
    for exchange in ("one_message", "phased"):
        got = _run(cuda, graph, "in_turn", n, exchange=exchange)
        print(f"graph={graph} exchange={exchange:12s}: max param diff vs one-rank {float((ref[0] - got[0]).abs().max()):.3e}  t={got[3]}", flush=True)
