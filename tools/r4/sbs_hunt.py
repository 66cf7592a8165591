"""Hunt for the rare difference between the replayed side-by-side update and the replayed in-turn update (about 1 % of the replays,
tools/r4/sbs_rate.sh).  One update object per process (the library keeps per-optimizer state in module globals); every random
stream is re-seeded before each update, so update k sees the same draws in every process and a process can be compared with a
recorded trajectory update by update:
  sbs_hunt.py record <mode> <n> <clip> <file>            checksums of (p, m, v) after every update -> file
  sbs_hunt.py check  <mode> <n> <clip> <file> <dump>     compare; at the first mismatch write per-tensor digests -> dump, exit
  sbs_hunt.py dump   <mode> <k> <clip> <dump>            per-tensor digests after update k"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from speecht5_amd import functional as Fn

cuda = torch.device("cuda:0")
what, mode, N, clip = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])


def seed(k):
    Fn._S.seed, Fn._S.counter = 4242, 1 + 100000 * k
    np.random.seed(1000 + k)
    torch.manual_seed(1000 + k)


if os.environ.get("HUNT_FLASH_IMPL"):   # discriminator: 1 = first-generation attention kernels (bit-identical results, other LDS layout)
    from speecht5_amd import hip as _hip
    _hip.check(_hip.lib().st5_flash_attn_set_impl(int(os.environ["HUNT_FLASH_IMPL"])), "st5_flash_attn_set_impl")
seed(0)
_, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", 8, 0, graph=True, micro=mode, layerdrop=0.05, prefetch_host=False, wgrad_stream=False)
upd.opt.clip = clip
seed(0)
upd.prepare_graph()


def checksum():
    torch.cuda.synchronize()
    return [int(x.view(torch.int32).to(torch.int64).sum().item()) for x in (upd.opt.pflat, upd.opt.m, upd.opt.v)]


def digests():
    torch.cuda.synchronize()
    out = {}
    for p, o in zip(upd.ddp.params, upd.ddp.offsets):
        pass
    names = {id(p): n for n, p in model.named_parameters()}
    for p, o in zip(upd.ddp.params, upd.ddp.offsets):
        c = p.numel()
        out[names[id(p)]] = [hashlib.sha1(x[o:o + c].cpu().numpy().tobytes()).hexdigest()[:10] for x in (upd.opt.pflat, upd.opt.m)]
    return out


if what == "record":
    rows = []
    for k in range(1, N + 1):
        seed(k); upd.update()
        rows.append(checksum())
    json.dump(rows, open(sys.argv[5], "w"))
    print(f"HUNT recorded {N} updates ({mode}, clip {clip})", flush=True)
elif what == "check":
    ref = json.load(open(sys.argv[5]))
    for k in range(1, N + 1):
        seed(k); upd.update()
        if checksum() != ref[k - 1]:
            json.dump({"k": k, "digests": digests()}, open(sys.argv[6], "w"))
            print(f"HUNT MISMATCH at update {k} ({mode}, clip {clip})", flush=True)
            break
    else:
        print(f"HUNT clean: {N} updates ({mode}, clip {clip})", flush=True)
else:
    for k in range(1, N + 1):
        seed(k); upd.update()
    json.dump({"k": N, "digests": digests()}, open(sys.argv[5], "w"))
    print(f"HUNT dumped update {N} ({mode})", flush=True)
