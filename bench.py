#!/usr/bin/env python
"""SpeechT5-Base pre-training step benchmark on MI355X (BASELINE.json metric: audio-sec/s forward+backward).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

One step = what one optimizer update of the reference recipe does with --update-freq 2 (SURVEY.md 3.1-3.2,
8d cfg 2): forward+backward of ONE speech micro-batch (8 x 10 s synthetic 16 kHz clips, HuBERT-mask + NCE +
mel decoder branch) and ONE text micro-batch (16 x 512 tokens, BART infilling) -- side by side on two streams, replayed
as one HIP graph -- the gradient all-reduce over the ranks (RCCL: one message behind every replay; `--no-graph`: bucketed and
overlapped with an eagerly enqueued backward), global-norm clip and the fused Adam update.  bf16 compute, fp32
master weights / statistics; dropout active as in t5_transformer_base (0.1, attention 0.1, pre-net 0.5,
post-net 0.5); LayerDrop is set to 0 so that every step does the full work.  Inputs are resident in HBM.

value = audio seconds of the speech micro-batches processed per wall second by the whole job.
Extra objects: `roofline` for the dominant kernel (bf16 NT MFMA GEMM; algorithmic FLOPs of its launches /
their HIP-event time: one eagerly enqueued update right after the timed region when the timed steps are graph replays, which
carry no events) and `cpu_baseline` (the CPU oracle, fp32, host cores, one 4 s clip forward+backward, median of 5)."""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

# kernel arguments in device memory: ~3 us less launch latency per kernel on MI300-class parts, ~3000 launches per step
# (must be in the environment before the HIP runtime initialises)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA


def build(device, compute_dtype, arch="base"):
    from speecht5_amd import functional as Fn
    from speecht5_amd.criterions import SpeechT5Criterion
    from speecht5_amd.speecht5 import t5_transformer_base, t5_transformer_large
    from speecht5_amd.task import SpeechT5Task
    Fn.set_compute_dtype(compute_dtype)
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True,
                     share_input_output_embed=True, encoder_layerdrop=0.0, decoder_layerdrop=0.0)
    (t5_transformer_large if arch == "large" else t5_transformer_base)(args)
    if arch == "large":   # (the architecture function leaves the LayerDrop default of 0.05 in: not replayable, see graph.py)
        args.encoder_layerdrop = args.decoder_layerdrop = 0.0
    task = SpeechT5Task.synthetic(args)
    torch.manual_seed(1337)
    model = task.build_model(args).to(device)
    crit = SpeechT5Criterion(task, loss_weights=[10, 0.1], sync_logging=False)
    return args, task, model, crit


PMC_TRAFFIC_FILE = "r2_pmc_traffic.json"   # written by tools/pmc_traffic.sh


def cpu_baseline(model, args, seconds=4.0, runs=5):
    """The CPU oracle (oracle/speecht5_oracle.py: fp32 restatement of the reference path) on the host cores: one clip,
    speech_pretrain forward + loss + backward; median of `runs` timed iterations after 2 warm-ups."""
    from oracle import speecht5_oracle as O
    from speecht5_amd.synthetic import speech_pretrain_sample
    from types import SimpleNamespace
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**vars(args))

    def run(secs):
        s = speech_pretrain_sample(B=1, seconds=secs, device="cpu", seed=7)
        T = int(secs * 50) - 1
        mask = torch.zeros(1, T, dtype=torch.bool)
        mask[:, : int(0.6 * T)] = True
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.forward_speech_pretrain(sd, cfg, s, mask_indices=mask, mix_idx=torch.arange(0, T, 2)[: T // 2], gumbel_noise=None)
        loss, ss, _ = O.speech_pretrain_loss(out, s, cfg, loss_weights=(10, 0.1))
        (loss / ss).backward()
        return time.perf_counter() - t0

    run(1.0)
    run(1.0)
    ts = sorted(run(seconds) for _ in range(runs))
    t = ts[len(ts) // 2]
    return {"value": round(seconds / t, 4), "unit": "audio-sec/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 x {seconds:g} s clip, speech_pretrain fwd+bwd (fp32 CPU oracle): median of {runs} runs after 2 warm-ups, "
                      f"{t:.2f} s per run (min {ts[0]:.2f}, max {ts[-1]:.2f})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="speech clips (10 s) per GPU per step")
    ap.add_argument("--arch", default="base", choices=["base", "large"],
                    help="large = t5_transformer_large (24 + 6 layers, d = 1024, pre-LN), same two micro-batches, bf16 GEMMs: a side "
                         "measurement (BASELINE.json cfg 5 asks for fp8 GEMMs, which do not exist here); the headline is base")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step from Python instead of replaying a captured HIP graph")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("ST5_DDP_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")

    from speecht5_amd import functional as Fn, hip
    from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
    from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    args, task, model, crit = build(device, dtype, a.arch)
    # The update is captured into a HIP graph after the warm-up steps and REPLAYED in the timed region (fresh dropout
    # seeds / span masks / lr per replay: speecht5_amd/graph.py).  ST5_GRAPH=0 or --no-graph: eager enqueue (with the bucketed
    # all-reduces overlapped with the backward when there are several ranks).
    use_graph = not a.no_graph and os.environ.get("ST5_GRAPH", "1") == "1"
    overlap_fwd = use_graph and os.environ.get("ST5_OVERLAP_FWD", "1") == "1"   # (eager enqueue is host-bound: nothing to gain)
    # several ranks: the captured part is the local phase (both micro-batches, no collectives); ONE all-reduce of the flat
    # gradient buffer and the Adam step follow every replay eagerly (ddp.local_phase / all_reduce_gradients)
    split_update = use_graph and dist.is_initialized()
    assert overlap_fwd or not split_update, "replayed multi-rank update: needs the side-by-side micro-batches (ST5_OVERLAP_FWD=1)"
    wgrad_env = os.environ.get("ST5_WGRAD_STREAM")
    # replayed step: forward passes of the two micro-batches side by side, no weight-gradient stream (see ddp.py); eager step:
    # the weight-gradient stream hides ~3.5 ms of dW GEMMs behind the data-gradient chain
    if os.environ.get("ST5_NT_TILE"):   # A/B: 1 = 128x128 always, 2 = 256x256 always (default: per problem)
        hip.lib().st5_gemm_set_nt_tile(int(os.environ["ST5_NT_TILE"]))
    if os.environ.get("ST5_DEEP_RING"):   # A/B: "max_blocks,nbuf" of the 128x128 NT kernel's deep operand ring (nbuf 2 = off)
        hip.lib().st5_gemm_set_deep_ring(*[int(v) for v in os.environ["ST5_DEEP_RING"].split(",")])
    if os.environ.get("ST5_SPLITK_TARGET"):   # A/B: block count the weight-gradient split-K aims for
        hip.lib().st5_gemm_set_splitk_target(int(os.environ["ST5_SPLITK_TARGET"]))
    ddp = FlatGradDataParallel(model, wgrad_stream=(wgrad_env == "1") if wgrad_env is not None else not (use_graph and overlap_fwd))
    opt = FusedAdam(ddp, lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=5.0)
    Fn.manual_seed(1337 + rank)
    import numpy as np
    np.random.seed(1337 + rank)
    torch.manual_seed(1337 + rank)
    vocab = len(task.dicts["text"])
    speech = speech_pretrain_sample(B=a.batch, seconds=10.0, device=device, seed=1337 + rank)
    text = text_pretrain_sample(B=16, T=512, vocab=vocab, mask_idx=task.dicts["text"].index("<mask>"), device=device, seed=2337 + rank)
    micro = [speech, text]

    def local_part(i):   # (split_update) what the graph holds: gradients of this rank's two micro-batches, summed into ddp.flat
        ddp.zero_grad()
        with ddp.local_phase():
            ddp.accumulate_overlapped(micro, lambda s: task.forward_loss(s, model, crit, i))
        ddp.sum_gradient_buffers()

    def exchange_and_update():   # (split_update) eager tail: sum over ranks, then mean over ranks and micro-batches inside Adam
        ddp.all_reduce_gradients(average=False)
        opt.step(grad_scale=1.0 / (len(micro) * world))

    def step(i):
        if split_update:
            local_part(i)
            exchange_and_update()
            return
        ddp.zero_grad()
        # --update-freq 2: gradients of the first micro-batch only accumulate (no_sync); the bucket all-reduces are
        # launched from the backward of the LAST micro-batch, each bucket once, after its last local contribution
        if overlap_fwd:   # the two micro-batches side by side on two streams, forward and backward (two gradient buffers)
            ddp.accumulate_overlapped(micro, lambda s: task.forward_loss(s, model, crit, i))
        else:
            ddp.accumulate(micro, lambda s: task.train_step(s, model, crit, None, i, sync=False))
        ddp.finish()
        opt.step(grad_scale=1.0 / len(micro))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    counter = [0]

    def one_update():
        (local_part if split_update else step)(counter[0])

    def advance():   # host-side state a replayed step does not touch: the update counter behind the quantizer temperature etc.
        counter[0] += 1
        model.set_num_updates(counter[0])

    if use_graph:
        from speecht5_amd.graph import StepGraph
        sg = StepGraph(one_update, opt=opt, model=model, device=device, on_step=advance, prefetch_host=os.environ.get("ST5_PREFETCH_HOST", "1") == "1",
                       after_fn=exchange_and_update if split_update else None)
        # untimed: eager steps, the two recording steps, and ONE replay (the first launch of a graph uploads it to the device:
        # ~150 ms that belong to set-up, not to the steady state) -- W updates in all when W >= 4, else 4
        n_eager = max(a.warmup - 3, 1)
        for i in range(n_eager):
            step(i)
        counter[0] = n_eager - 1
        sg.record()
        sg.record()
        sg.capture()
        cap_stream = sg.stream

        def run():
            with torch.cuda.stream(cap_stream):
                sg.replay()
        run()
    else:
        for i in range(a.warmup):
            step(i)
        run = lambda: step(a.warmup)
    barrier()
    hip.profiler.reset()
    t0 = time.perf_counter()
    for i in range(a.steps):
        # eager mode: HIP events around every st5_gemm launch of the LAST timed step (recording them on all K steps costs
        # ~10 % of the step in host time: two events per launch, ~750 launches per step)
        hip.profiler.enabled = (not use_graph) and (i == a.steps - 1)
        run()
    barrier()
    dt = time.perf_counter() - t0
    if use_graph:
        sg.drain()   # (the helper thread preparing a step that will not run)
    hip.profiler.enabled = False
    if use_graph:
        # roofline leg of the graph mode: the replayed launches carry no events, so the SAME update is enqueued once more
        # eagerly, outside the timed region, with HIP events around every st5_gemm launch (same kernels, shapes, streams)
        hip.profiler.enabled = True
        step(a.warmup)
        torch.cuda.synchronize()
        hip.profiler.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    audio_seconds = a.batch * 10.0 * world * a.steps
    prof = hip.profiler.summary()
    key = "bf16_NT" if a.dtype == "bf16" else "f32_NT"
    n, flops, secs = prof.get(key, (0, 0.0, 0.0))
    peak = BF16_DENSE_PEAK_TFLOPS if a.dtype == "bf16" else 157.3
    # measured HBM bytes per launch of the same kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command
    # (tools/pmc_traffic.sh), summary committed under profiles/ (the counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    try:
        import json as _json
        here = os.path.dirname(os.path.abspath(__file__))
        pm = _json.load(open(os.path.join(here, "profiles", PMC_TRAFFIC_FILE)))
        # only a summary measured on THIS kernel source counts (the file records the hash of csrc/gemm.hip it was taken on)
        import hashlib
        cur = hashlib.sha1(open(os.path.join(here, "speecht5_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:12]
        if pm.get("gemm_hip_sha1") == cur and a.dtype == "bf16":
            cand = [v for kname, v in pm["kernels"].items() if "gemm_nt_glds_kernel" in kname]
            if cand:
                v = max(cand, key=lambda v: v["launches"])   # (the bf16 instantiation; the fp32 one serves the NCE head)
                traffic = v["hbm_corrected_bytes_per_launch"]
                traffic_src = f"profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, bytes/launch; gemm.hip {cur})"
    except Exception:
        traffic = None
    alg_bytes = hip.profiler.nt_bytes_per_launch() if hasattr(hip.profiler, "nt_bytes_per_launch") else None
    roof = {"bound": "mfma", "kernel": f"NT-form st5_gemm launches <{a.dtype}>: gemm_nt_glds_kernel (128x128 tiles: Linear / attention-projection "
                      "forward and data-gradient GEMMs) + gemm_nt256_kernel (256x256 tiles: the long conv feature-extractor GEMMs); "
                      "`traffic` is per launch of gemm_nt_glds_kernel",
            "note": ("launch durations: HIP events around every st5_gemm launch of ONE eagerly enqueued update after the timed region "
                     "(replayed launches carry no events), micro-batches side by side on two streams as in the timed steps"
                     if use_graph else
                     "launch durations are measured inside the step, i.e. beside the weight-gradient stream's kernels "
                     "(ST5_WGRAD_STREAM=0 gives the isolated rate, ~8 % higher)"),
            "achieved": round(flops / secs / 1e12, 2) if secs > 0 else None, "peak": peak, "unit": "TFLOP/s",
            "frac": round(flops / secs / 1e12 / peak, 4) if secs > 0 else None, "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes,
            "launches_per_step": n, "sampled_steps": 1, "avg_launch_us": round(secs / max(n, 1) * 1e6, 2),
            "all_variants": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None,
                                 "ms_per_step": round(v[2] * 1e3, 3)} for k, v in sorted(prof.items())}}
    # HBM-bound leg ("HBM GB/s (conv frontend)"): conv layer 0 + GroupNorm + GELU, same HIP-event method
    hbm = {}
    for name, (n_, b_, t_) in sorted(hip.profiler.region_summary().items()):
        hbm[name] = {"launches": n_, "algorithmic_MB": round(b_ / n_ / 1e6, 1), "achieved_GBps": round(b_ / t_ / 1e9, 1) if t_ > 0 else None,
                     "frac_of_8TBps": round(b_ / t_ / 8e12, 4) if t_ > 0 else None}
    roof["hbm_bound_kernels"] = hbm
    if rank == 0:
        out = {"metric": "audio-sec/s fwd+bwd SpeechT5-" + ("Large" if a.arch == "large" else "Base"), "value": round(audio_seconds / dt, 2), "unit": "audio-sec/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "SpeechT5-" + ("Large" if a.arch == "large" else "Base") + " pretrain step (speech 8x10s micro-batch + text 16x512 micro-batch, update-freq 2), "
                                      "fwd+bwd+allreduce+clip+Adam, per GPU", "arch": ("t5_transformer_large (24 enc + 6 dec, d=1024, pre-LN, layer-norm feature extractor)" if a.arch == "large"
                                   else "t5_transformer_base (12 enc + 6 dec, d=768)"),
                          "enqueue": ("hip-graph replay of the local phase + eager all-reduce (one message) + Adam" if split_update else
                                      "hip-graph replay" if use_graph else "eager"), "micro_batches": "forward and backward side by side on two streams, two gradient buffers" if overlap_fwd else "in turn",
                          "global_speech_batch": a.batch * world, "clip_seconds": 10, "parallelism": f"dp{world}",
                          "dropout": 0.1, "layerdrop": 0.0},
               "roofline": roof}
        if world == 1 and not a.no_cpu_baseline and a.arch == "base":
            out["cpu_baseline"] = cpu_baseline(model, args)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
