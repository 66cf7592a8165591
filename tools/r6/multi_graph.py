"""Timing experiment (round 6): the update as THREE graphs -- speech micro-batch (forward + backward) on stream 1, text micro-batch on
stream 2, the optimizer tail on stream 1 behind both -- stitched by plain stream events OUTSIDE the graphs, against the shipped form
(one graph, the second stream forked and joined INSIDE it).  Every extra cross-stream edge inside a replayed graph measured 0.13-0.6 ms
this round (weight-gradient groups moved across streams, lazy transposes, phase joins); this asks what the two edges of the shipped
form cost.  Host inputs are frozen at their last staged values (timing only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speecht5_amd import functional as Fn
from speecht5_amd.graph import StepGraph

dev = torch.device("cuda:0")
N = int(os.environ.get("STEPS", 40))


def shipped():
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 8, 0, graph=True, micro="side_by_side")
    upd.prepare_graph()
    for _ in range(3):
        upd.update()
    upd.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        upd.update()
    upd.finish(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    upd.close()
    return dt * 1e3


def three_graphs():
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 8, 0, graph=False, micro="side_by_side")
    ddp, opt = upd.ddp, upd.opt
    S1, S2 = upd.stream, torch.cuda.Stream(device=dev)
    ddp.pair_streams(2)          # (creates the second gradient buffer)
    sg = StepGraph(None, opt=opt, model=model, device=dev, on_step=upd.advance, stream=S1, phases=[], between=[])

    def part(i):
        with ddp._grad_slot(i):
            loss = upd._fwd(upd.micro[i])
            loss.backward()
            ddp.flush_stream_deferred()

    def tail():
        ddp._pair_pending = True
        opt.step(grad_scale=0.5)

    def eager_step():
        S2.wait_stream(S1)
        with torch.cuda.stream(S1):
            part(0)
        with torch.cuda.stream(S2):
            part(1)
        S1.wait_stream(S2)
        with torch.cuda.stream(S1):
            tail()

    Fn._S.force_static = True
    for _ in range(2):           # two recorded steps (static buffers, seed slots)
        with torch.cuda.stream(S1):
            upd.advance(); opt.push_hyper(); sg.slots.begin_step()
        sg._enter("record")
        try:
            eager_step()
        finally:
            sg._exit()
        if not sg.slots.used:
            sg.slots.used = sg.slots.k
    torch.cuda.synchronize()
    with torch.cuda.stream(S1):
        sg._pre_replay()
    torch.cuda.synchronize()
    import gc; gc.collect()
    graphs = []
    sg._enter("capture")
    try:
        for st, fn in ((S1, lambda: part(0)), (S2, lambda: part(1)), (S1, tail)):
            Fn.weight_cache.clear()       # (a cached cast made inside one graph must not be read by another)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                fn()
            graphs.append((st, g))
    finally:
        sg._exit()
    Fn._S.force_static = False
    torch.cuda.synchronize()

    def replay():
        S2.wait_stream(S1)
        for st, g in graphs[:2]:
            with torch.cuda.stream(st):
                g.replay()
        S1.wait_stream(S2)
        with torch.cuda.stream(S1):
            graphs[2][1].replay()

    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    p = opt.pflat
    print("three graphs: parameters finite:", bool(torch.isfinite(p).all()), "norm", float(p.double().norm()))
    return dt * 1e3


def wgrad_chains(text_only=False):
    """The backward of a micro-batch cut into three phases (the bucket cuts of the phased exchange) = three graphs on its stream; the
    weight-gradient groups a phase queues are HELD and launched as a graph of their own on a third / fourth stream, ordered behind that
    phase by a stream event and free to run beside the next phase."""
    _, _, model, upd = bench.make_update(dev, torch.bfloat16, "base", 8, 0, graph=False, micro="side_by_side")
    ddp, opt = upd.ddp, upd.opt
    S = [upd.stream, torch.cuda.Stream(device=dev)]
    W = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    ddp.pair_streams(2)
    sg = StepGraph(None, opt=opt, model=model, device=dev, on_step=upd.advance, stream=S[0], phases=[], between=[])
    cuts_b = upd.cut_buckets()
    nph = len(cuts_b) + 1
    chained = [1] if text_only else [0, 1]
    state = {}
    alive = []

    def phase(i, k):                      # micro-batch i, backward phase k (k = 0: forward first)
        with ddp._grad_slot(i):
            if k == 0:
                if i in chained:
                    with ddp.cut_points(cuts_b) as cuts:
                        loss = upd._fwd(upd.micro[i])
                    state[i] = ddp.backward_phases(loss, cuts)
                    assert len(state[i]) == nph, len(state[i])
                else:
                    loss = upd._fwd(upd.micro[i])
                    loss.backward()
                    ddp.flush_stream_deferred()
                    return
            state[i][k][0]()
            ddp.flush_stream_deferred()   # (queued groups -> held; LayerNorm / split-K partials of this stream folded)

    def wphase(i):
        Fn.launch_held_wgrads(alive)
        from speecht5_amd import hip
        hip.check(hip.lib().st5_gemm_flush_splitk(hip.stream()), "st5_gemm_flush_splitk")

    def tail():
        ddp._pair_pending = True
        opt.step(grad_scale=0.5)

    # the schedule: (stream, function, waits-for list of indices) in host order
    sched = []
    for i in (0, 1):
        if i in chained:
            for k in range(nph):
                sched.append(("P", i, k))
                sched.append(("W", i, k))
        else:
            sched.append(("P", i, 0))

    def run(kind, i, k):
        if kind == "P":
            if i in chained:
                Fn.hold_wgrads(S[i])
            try:
                phase(i, k)
            finally:
                Fn.hold_wgrads(None) if False else None
        else:
            wphase(i)

    def eager_step():
        S[1].wait_stream(S[0])
        for kind, i, k in sched:
            if kind == "P":
                Fn.hold_wgrads(S[i]) if i in chained else None
                with torch.cuda.stream(S[i]):
                    phase(i, k)
            else:
                W[i].wait_stream(S[i])
                with torch.cuda.stream(W[i]):
                    wphase(i)
        Fn.hold_wgrads(None)
        for st in (S[1], W[0], W[1]):
            S[0].wait_stream(st)
        with torch.cuda.stream(S[0]):
            tail()
        del alive[:]

    Fn._S.force_static = True
    for _ in range(2):
        with torch.cuda.stream(S[0]):
            upd.advance(); opt.push_hyper(); sg.slots.begin_step()
        sg._enter("record")
        try:
            eager_step()
        finally:
            sg._exit()
        if not sg.slots.used:
            sg.slots.used = sg.slots.k
    torch.cuda.synchronize()
    with torch.cuda.stream(S[0]):
        sg._pre_replay()
    torch.cuda.synchronize()
    import gc; gc.collect()
    graphs = []
    sg._enter("capture")
    try:
        for kind, i, k in sched:
            st = S[i] if kind == "P" else W[i]
            Fn.weight_cache.clear()
            if kind == "P" and i in chained:
                Fn.hold_wgrads(S[i])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                phase(i, k) if kind == "P" else wphase(i)
            graphs.append((kind, i, st, g))
        Fn.hold_wgrads(None)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=S[0], capture_error_mode="thread_local"):
            tail()
        graphs.append(("T", 0, S[0], g))
    finally:
        sg._exit()
    Fn._S.force_static = False
    torch.cuda.synchronize()

    def replay():
        S[1].wait_stream(S[0])
        for kind, i, st, g in graphs[:-1]:
            if kind == "W":
                st.wait_stream(S[i])
            with torch.cuda.stream(st):
                g.replay()
        for st in (S[1], W[0], W[1]):
            S[0].wait_stream(st)
        with torch.cuda.stream(S[0]):
            graphs[-1][3].replay()

    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    p = opt.pflat
    print(f"weight-gradient chains ({'text only' if text_only else 'both micro-batches'}): {len(graphs)} graphs, parameters finite:",
          bool(torch.isfinite(p).all()), "norm", float(p.double().norm()))
    return dt * 1e3


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("both", "shipped"):
        print(f"shipped (one graph, fork / join inside): {shipped():.3f} ms per update", flush=True)
    if which in ("both", "three"):
        print(f"three graphs stitched by stream events:   {three_graphs():.3f} ms per update", flush=True)
    if which == "wtext":
        print(f"text micro-batch's weight gradients as a third chain: {wgrad_chains(True):.3f} ms per update", flush=True)
    if which == "wboth":
        print(f"both micro-batches' weight gradients as chains of their own: {wgrad_chains(False):.3f} ms per update", flush=True)
