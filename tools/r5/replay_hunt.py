"""Is the replayed update reproducible across processes, update by update -- and if not, where does the difference enter?
(the in-turn replay differed from a recorded trajectory at update 195 in every process that synchronised after
each update.)  One process = one trajectory:
  * (p, m, v) integer checksums after every update, computed ON THE DEVICE without a host synchronisation (fetched once at the end),
    so a trajectory can be taken with or without per-update synchronisation and the two kinds compared;
  * at the updates named by --dump-at, integer checksums of every module output and every module-output gradient of the captured
    step (forward / backward hooks record the static tensors of the capture pass, in execution order).  Buffers the graph's memory
    pool re-uses later in the step hold later values -- the same ones in every process of a deterministic step -- so the FIRST
    name (in execution order) that differs between two processes brackets the kernel that went wrong.
Every random stream is re-seeded before each update (as tools/r4/sbs_hunt.py did), so update k sees the same draws in every process.
  replay_hunt.py run  <out.json> [--mode in_turn] [--n 300] [--sync 0|1] [--sleep-ms 0] [--seed-off 0] [--dump-at 194,195,196]
  replay_hunt.py diff <a.json> <b.json>"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def diff(a, b):
    A, B = json.load(open(a)), json.load(open(b))
    n = min(len(A["cs"]), len(B["cs"]))
    first = next((k for k in range(n) if A["cs"][k] != B["cs"][k]), None)
    tag = f"{os.path.basename(a)} vs {os.path.basename(b)}"
    if first is None:
        print(f"DIFF {tag}: identical over {n} updates", flush=True)
    else:
        which = [w for w, x, y in zip("pmv", A["cs"][first], B["cs"][first]) if x != y]
        print(f"DIFF {tag}: first mismatch at update {first + 1} ({'/'.join(which)})", flush=True)
    if "stg" in A and "stg" in B:
        fs = next((k for k in range(n) if A["stg"][k] != B["stg"][k]), None)
        if fs is not None:
            cols = [j for j, (x, y) in enumerate(zip(A["stg"][fs], B["stg"][fs])) if x != y]
            print(f"  uploaded inputs first differ at update {fs + 1}, entries {cols} of {len(A['stg'][fs])} (last two = seed slots, {{lr, step}})", flush=True)
        else:
            print("  uploaded inputs (staged buffers, seed slots, lr/step) identical at every update", flush=True)
    for k in sorted(set(A["dumps"]) & set(B["dumps"]), key=int):
        da, db = A["dumps"][k], B["dumps"][k]
        bad = [i for i, (x, y) in enumerate(zip(da, db)) if x != y]
        if bad:
            names = A["names"]
            print(f"  update {k}: {len(bad)} of {len(da)} hooked tensors differ; first in execution order:", flush=True)
            for i in bad[:14]:
                print(f"    [{i}] {names[i]}", flush=True)
        else:
            print(f"  update {k}: all {len(da)} hooked tensors identical", flush=True)
    return first


def run(a):
    import numpy as np
    import torch
    import bench
    from speecht5_amd import functional as Fn
    cuda = torch.device("cuda:0")
    dump_at = set(int(x) for x in a.dump_at.split(",") if x)

    def seed(k):
        k += a.seed_off
        Fn._S.seed, Fn._S.counter = 4242, 1 + 100000 * k
        np.random.seed(1000 + k)
        torch.manual_seed(1000 + k)

    seed(0)
    _, _, model, upd = bench.make_update(cuda, torch.bfloat16, "base", a.batch, 0, graph=True, micro=a.mode, layerdrop=0.05,
                                         prefetch_host=False, wgrad_stream=bool(a.wgrad))
    upd.opt.clip = a.clip
    # ---- hooks: static tensors of the capture pass, in execution order -------------------------------------------------------
    live = {"on": True}
    order, tensors = [], {}
    names = {id(m): n for n, m in model.named_modules()}
    base_adv = upd.advance

    def advance():
        if live["on"]:
            order.clear(); tensors.clear()
        base_adv()
    upd.advance = advance

    def note(name, t):
        if live["on"] and isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
            key = f"{len(order):04d} {name} {tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"
            order.append(key); tensors[key] = t

    def flat(o):
        if isinstance(o, torch.Tensor):
            yield o
        elif isinstance(o, (list, tuple)):
            for x in o:
                yield from flat(x)
        elif isinstance(o, dict):
            for x in o.values():
                yield from flat(x)

    def fwd_hook(mod, inp, out):
        if not live["on"]:
            return
        nm = names.get(id(mod), "?")
        for j, t in enumerate(flat(out)):
            note(f"fwd {nm}#{j}", t)
            if t.requires_grad and t.is_floating_point():
                t.register_hook(lambda g, nm=nm, j=j: note(f"bwd d({nm}#{j})", g))
    if dump_at:      # (the hooks keep every activation of the captured step alive: only in the runs that dump)
        for m in model.modules():
            m.register_forward_hook(fwd_hook)
    seed(0)
    upd.prepare_graph()
    live["on"] = False
    keys = list(order)
    print(f"HUNT captured; {len(keys)} hooked tensors", flush=True)

    N = a.n
    st = upd.sg.stream
    cs = torch.zeros(N, 3, dtype=torch.int64, device=cuda)
    dumps = {}
    # what every replay uploads (device images of the staged host inputs, the dropout seed slots, {lr, step}): checksummed behind
    # EVERY update -- the same in every process for the same k, or the upload itself is the problem
    staged = [e[0] for e in upd.sg.staging.entries if e[0] is not None] + [upd.sg.slots.dev, upd.opt.hyper_dev]
    stg = torch.zeros(N, len(staged), dtype=torch.int64, device=cuda)
    junk = []

    def isum(t):
        if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
            return t.view(torch.int16).sum(dtype=torch.int64)
        if t.dtype == torch.float32:
            return t.view(torch.int32).sum(dtype=torch.int64)
        return t.to(torch.int64).sum()
    t0 = time.time()
    for k in range(1, N + 1):
        seed(k)
        upd.update()
        with torch.cuda.stream(st):
            for j, x in enumerate((upd.opt.pflat, upd.opt.m, upd.opt.v)):
                cs[k - 1, j] = x.view(torch.int32).sum(dtype=torch.int64)
            for j, x in enumerate(staged):
                stg[k - 1, j] = isum(x)
            if k in dump_at and keys:
                dumps[k] = torch.stack([isum(tensors[key]) for key in keys])
            for _ in range(a.extra_events):      # does the event move with the number of runtime signals consumed per update?
                ev = torch.cuda.Event(); ev.record(); junk.append(ev)
        if a.sync:
            torch.cuda.synchronize()
        if a.sleep_ms:
            torch.cuda.synchronize()
            time.sleep(a.sleep_ms * 1e-3)
    torch.cuda.synchronize()
    dt = time.time() - t0
    json.dump({"cs": cs.cpu().tolist(), "stg": stg.cpu().tolist(), "dumps": {str(k): v.cpu().tolist() for k, v in dumps.items()}, "names": keys,
               "args": vars(a), "seconds": dt}, open(a.out, "w"))
    print(f"HUNT ran {N} updates ({a.mode}, wgrad {a.wgrad}, sync {a.sync}, sleep {a.sleep_ms} ms, seed-off {a.seed_off}) in {dt:.1f} s -> {a.out}", flush=True)
    upd.close()


if __name__ == "__main__":
    if sys.argv[1] == "diff":
        diff(sys.argv[2], sys.argv[3])
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("what"); ap.add_argument("out")
        ap.add_argument("--mode", default="in_turn"); ap.add_argument("--n", type=int, default=300)
        ap.add_argument("--sync", type=int, default=0); ap.add_argument("--sleep-ms", type=float, default=0.0)
        ap.add_argument("--seed-off", type=int, default=0); ap.add_argument("--dump-at", default="")
        ap.add_argument("--clip", type=float, default=0.0); ap.add_argument("--batch", type=int, default=8)
        ap.add_argument("--extra-events", type=int, default=0)
        ap.add_argument("--wgrad", type=int, default=0, help="1: weight-gradient GEMMs on their own stream")
        run(ap.parse_args())
