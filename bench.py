#!/usr/bin/env python
"""SpeechT5-Base pre-training step benchmark on MI355X (BASELINE.json metric: audio-sec/s forward+backward).

  python bench.py --gpus N --steps K --warmup W      (N>1: one rank per GPU; launched by torch.distributed.run, or -- from a
                                                      bare shell, WORLD_SIZE unset -- bench.py re-executes itself under it)

One step = what one optimizer update of the reference recipe does with --update-freq 2 (SURVEY.md 3.1-3.2,
8d cfg 2): forward+backward of ONE speech micro-batch (8 x 10 s synthetic 16 kHz clips, HuBERT-mask + NCE +
mel decoder branch) and ONE text micro-batch (16 x 512 tokens, BART infilling) -- side by side on two streams, replayed as one HIP
graph (`--micro in_turn`: one after the other on one stream, the reference trainer's order; the SAME bits either way)
-- the gradient all-reduce over the ranks (RCCL; default `--exchange phased`: the local phase is three graphs with both micro-batches inside
each, and each completed bucket range -- 234 / 227 / 156 MB -- is all-reduced underneath the next graph; `--exchange one_message`: one graph,
one message behind it; `--no-graph`: eagerly enqueued), global-norm clip and the fused Adam update.  bf16 compute, fp32
master weights / statistics; dropout and LayerDrop active as t5_transformer_base ships them (0.1, attention 0.1, pre-net 0.5,
post-net 0.5; encoder / decoder LayerDrop 0.05, models/speecht5.py:1397-1398 -- inside the replayed graph a dropped layer is
selected away on the device, i.e. it still runs: no work is skipped in the timed region).  Inputs are resident in HBM.

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
FP8_DENSE_PEAK_TFLOPS = 5000.0   # MI355X_MICROARCH.md: ~5 PFLOP/s dense fp8 (MX-scaled K = 64 / 128 forms)


def build(device, compute_dtype, arch="base", layerdrop=0.05):
    from speecht5_amd import functional as Fn
    from speecht5_amd.criterions import SpeechT5Criterion
    from speecht5_amd.speecht5 import t5_transformer_base, t5_transformer_large
    from speecht5_amd.task import SpeechT5Task
    Fn.set_compute_dtype(compute_dtype)
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True,
                     share_input_output_embed=True, encoder_layerdrop=layerdrop, decoder_layerdrop=layerdrop)
    (t5_transformer_large if arch == "large" else t5_transformer_base)(args)
    args.encoder_layerdrop = args.decoder_layerdrop = layerdrop
    task = SpeechT5Task.synthetic(args)
    torch.manual_seed(1337)
    model = task.build_model(args).to(device)
    crit = SpeechT5Criterion(task, loss_weights=[10, 0.1], sync_logging=False)
    return args, task, model, crit


PMC_TRAFFIC_FILE = "r6_pmc_traffic.json"   # written by tools/pmc_traffic.sh
KERNEL_STATS_FILE = "r6_bench_kernel_stats.csv"   # rocprofv3 --kernel-trace --stats of `bench.py --steps 13` (tools/finals.sh)
NT_KERNEL_NAME = "gemm_nt_glds_kernel"      # the dominant kernel of the update (most NT launches)


def kernel_source_hash():
    """sha1 of the GEMM kernels' sources (gemm.hip + common.h, where the fused epilogues' math lives): committed profiler summaries
    record it and are only trusted for the tree they were measured on."""
    import hashlib
    h = hashlib.sha1()
    for f in ("gemm.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "speecht5_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


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

    # thread sweep (on a 128-thread host all-threads is SLOWER than 8 for this one-clip workload -- the oracle's
    # GEMMs are small): one warm-up + one timed run per setting on a 2 s clip, then the median of `runs` at the best setting
    avail = torch.get_num_threads()
    cand = sorted({n for n in (8, 16, 32, 64, avail) if 1 <= n <= avail})
    sweep = {}
    for n in cand:
        torch.set_num_threads(n)
        run(1.0)
        sweep[n] = round(run(2.0), 3)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    run(1.0)
    ts = sorted(run(seconds) for _ in range(runs))
    torch.set_num_threads(avail)
    t = ts[len(ts) // 2]
    ref = None
    try:   # the verbatim reference stack timed beside this port on the build container's cores (oracle/time_reference_cpu.py)
        ref = json.load(open(os.path.join(ROOT, "profiles", "r6_cpu_reference_vs_port.json")))
    except Exception:
        pass
    out = {"value": round(seconds / t, 4), "unit": "audio-sec/s", "cores": best, "kind": "port",
           "sample": f"1 x {seconds:g} s clip, speech_pretrain fwd+bwd (fp32 CPU oracle): median of {runs} runs after warm-up at the best "
                     f"thread count of a sweep, {t:.2f} s per run (min {ts[0]:.2f}, max {ts[-1]:.2f})",
           "thread_sweep_s_per_2s_clip": {str(k): v for k, v in sweep.items()}, "host_threads": avail}
    if ref is not None:
        # the reference itself cannot travel to this box: its ratio to the port, measured where both exist, is read from the committed file
        rs, ps = ref.get("reference", {}).get("median_s"), ref.get("port", {}).get("median_s")
        out["reference_vs_port"] = {"source": "profiles/r6_cpu_reference_vs_port.json", "reference_s_per_run": rs, "port_s_per_run": ps,
                                    "cores": ref.get("cores"), "port_over_reference": round(ps / rs, 3) if rs and ps else None,
                                    "reference_estimate_audio_sec_per_s": round(seconds / t * ps / rs, 4) if rs and ps else None}
    return out


def conv0_device_time(device, B, reps=20):
    """ms per st5_conv0_gn_gelu_fwd / _bwd call at the benched shape (B x 160 000 samples -> 512 channels, bf16), back-to-back calls."""
    from speecht5_amd import hip
    L_ = hip.lib()
    S, C, k, stride = 160000, 512, 10, 5
    Lo = (S - k) // stride + 1
    wav = torch.randn(B, S, device=device)
    w = torch.randn(C, k, device=device) * 0.4
    g, b_ = torch.ones(C, device=device), torch.zeros(C, device=device)
    out = torch.empty(B, Lo, C, dtype=torch.bfloat16, device=device)
    dy = (torch.randn(B, Lo, C, device=device) * 0.1).to(torch.bfloat16)
    stats = torch.empty(B, C, 2, device=device)
    dw, dg, db = torch.zeros(C, k, device=device), torch.zeros(C, device=device), torch.zeros(C, device=device)
    ws = hip.workspace(L_.st5_conv0_ws_bytes(B, S, C, k, stride), device)
    mom = torch.empty(B, L_.st5_conv0_mom_count(k), dtype=torch.float64, device=device)   # (as functional.ConvFeatureExtractorFunction calls them)

    def fwd():
        hip.check(L_.st5_conv0_gn_gelu_fwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b_.data_ptr(), out.data_ptr(), stats.data_ptr(), mom.data_ptr(),
                                              ws.data_ptr(), B, S, C, k, stride, 1e-5, hip.BF16, hip.stream()), "conv0 fwd")

    def bwd():
        hip.check(L_.st5_conv0_gn_gelu_bwd_m(wav.data_ptr(), w.data_ptr(), g.data_ptr(), b_.data_ptr(), stats.data_ptr(), mom.data_ptr(), dy.data_ptr(),
                                              dw.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), B, S, C, k, stride, 1.0, hip.BF16, hip.stream()),
                  "conv0 bwd")
    res = {}
    for nm, fn in (("conv0_gn_gelu_fwd", fwd), ("conv0_gn_gelu_bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[nm] = round(e0.elapsed_time(e1) / reps, 4)
    return res


def box_info():
    """Which box this line was measured on (the pool's boxes differ by up to 20 % for the same commit, profiles/r4_knob_ab.txt): device
    name, CU count, and the clocks / power cap rocm-smi reports at the end of the run -- so a slow line is attributable."""
    import socket
    import subprocess
    info = {"host": socket.gethostname()}
    try:
        p = torch.cuda.get_device_properties(0)
        info.update(device=p.name, cus=p.multi_processor_count, hbm_gb=round(p.total_memory / 2 ** 30))
    except Exception:
        pass
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=15)
        d = json.loads(r.stdout)
        c0 = d.get("card0", {})
        keep = {k: v for k, v in c0.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "junction", "edge"))}
        info["rocm_smi_card0"] = dict(list(keep.items())[:12])
    except Exception as e:
        info["rocm_smi_card0"] = f"unavailable ({type(e).__name__})"
    return info


def make_update(device, dtype=torch.bfloat16, arch="base", batch=8, rank=0, graph=True, micro="side_by_side", layerdrop=0.05,
                wgrad_stream=None, prefetch_host=True, text_batch=16, text_len=512, seconds=10.0, exchange="phased", exchange_payload="fp32"):
    """The update bench.py times, as an object (speecht5_amd/update.py): model, criterion, the two synthetic micro-batches of
    BASELINE.json cfg 2, FlatGradDataParallel + FusedAdam with the recipe's hyper-parameters.  tests/test_bench_update_gpu.py
    builds its runs from this function too."""
    import numpy as np
    from speecht5_amd import functional as Fn
    from speecht5_amd.synthetic import speech_pretrain_sample, text_pretrain_sample
    from speecht5_amd.update import PretrainUpdate
    args, task, model, crit = build(device, dtype, arch, layerdrop)
    Fn.manual_seed(1337 + rank)
    np.random.seed(1337 + rank)
    torch.manual_seed(1337 + rank)
    vocab = len(task.dicts["text"])
    speech = speech_pretrain_sample(B=batch, seconds=seconds, device=device, seed=1337 + rank)
    text = text_pretrain_sample(B=text_batch, T=text_len, vocab=vocab, mask_idx=task.dicts["text"].index("<mask>"), device=device, seed=2337 + rank)
    batches = [text, speech] if os.environ.get("ST5_TEXT_FIRST") == "1" else [speech, text]    # (A/B: which micro-batch owns the first stream)
    upd = PretrainUpdate(task, model, crit, batches, lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=5.0,
                         graph=graph, micro=micro, wgrad_stream=wgrad_stream, prefetch_host=prefetch_host, device=device, exchange=exchange,
                         exchange_payload=exchange_payload)
    return args, task, model, upd


def respawn(a):
    """`python bench.py --gpus N` from a bare shell: run N ranks of this script under torch.distributed.run (one per GPU; when
    the box has fewer GPUs than ranks -- a functional check on a 1-GPU box -- the ranks share devices and talk over gloo)."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 400))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    try:
        import torch
        if torch.cuda.device_count() < a.gpus:
            # ranks SHARING a device (functional check, gloo): every process brings its update stream, the second micro-batch's stream and
            # gloo's private copy streams.  With ROCm's default of 4 hardware queues per process the streams are multiplexed onto the same
            # queues and the phased side-by-side form crawled (10 s per update, tools/r6/call9.sh / call10.sh; 0.3 s with 8 queues).  The
            # runtime reads the variable when it is loaded, so it has to be in the children's environment.  Ranks with a GPU each: default.
            env.setdefault("GPU_MAX_HW_QUEUES", "8")
    except Exception:
        pass
    sys.exit(subprocess.call(cmd, env=env))


def config3(a):
    """`--config 3`: one line in the same contract; `value` = audio-seconds of TTS fine-tuning targets per second of training."""
    assert a.gpus == 1, "--config 3 is a one-GPU measurement"
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import bench_cfg3
    r = bench_cfg3.run(steps=a.steps, warmup=a.warmup, graph=not a.no_graph)
    tts, voc = r["tts_finetune_step"], r["hifigan_forward"]
    nt = tts["gemm"]["by_variant"].get("bf16_NT", {})
    out = {"metric": "audio-sec/s fwd+bwd SpeechT5-Base TTS fine-tune", "value": tts["audio_sec_per_s"], "unit": "audio-sec/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": tts["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[2]: SpeechT5-Base TTS fine-tune step (32 texts x 100 tokens -> 32 x 600 mel frames, r = 2, "
                                  "guided attention, dropout 0.15), fwd+bwd+clip+Adam; + HiFi-GAN (HF SpeechT5HifiGan config) forward on the same 32 x 600 frames",
                      "enqueue": tts["enqueue"], "utterances_per_s": tts["utterances_per_s"]},
           "roofline": {"bound": "mfma", "kernel": "NT-form st5_gemm launches <bf16> of the TTS step (HIP events, one eagerly enqueued step)",
                        "achieved": nt.get("tflops"), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(nt["tflops"] / BF16_DENSE_PEAK_TFLOPS, 4) if nt.get("tflops") else None, "traffic": None,
                        "step": {"gemm_tflop_per_update": tts["gemm"]["gemm_tflop"], "tflops": tts["step_tflops"], "frac_of_peak": tts["step_frac_of_peak"]},
                        "all_variants": tts["gemm"]["by_variant"]},
           "vocoder": voc}
    emit(out)


_RESULT_FD = None
_AS_SCRIPT = False     # set by `python bench.py`: only then is the process's stdout descriptor re-pointed


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner from C when a
    process group comes up or goes down), so file descriptor 1 is pointed at stderr for the whole run and the result line goes to
    the saved original descriptor (emit)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_RESULT_FD, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: bf16 compute mode with the forward / data-gradient GEMMs of the large Linears on the block-scaled (MX) fp8 "
                         "MFMA kernel, weight gradients bf16 (BASELINE.json configs[4]: --arch large --dtype fp8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="speech clips (10 s) per GPU per step")
    ap.add_argument("--arch", default="base", choices=["base", "large"],
                    help="large = t5_transformer_large (24 + 6 layers, d = 1024, pre-LN), same two micro-batches: a side measurement "
                         "(BASELINE.json configs[4]; with --dtype fp8 the forward / data-gradient GEMMs run on the MX-fp8 kernel); the "
                         "headline is base")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step from Python instead of replaying a captured HIP graph")
    ap.add_argument("--layerdrop", type=float, default=0.05, help="encoder / decoder LayerDrop (t5_transformer_base: 0.05)")
    ap.add_argument("--micro", default="side_by_side", choices=["side_by_side", "in_turn_2buf", "in_turn"],
                    help="how the update's two micro-batches are enqueued (speecht5_amd/update.py): side_by_side = two streams (default, "
                         "~20 %% faster); in_turn = one stream, the reference trainer's order.  Bit-identical results (DESIGN.md 4c)")
    ap.add_argument("--exchange", default="phased", choices=["phased", "one_message"],
                    help="several ranks, graph replay: phased = the local phase as 3 graphs, each completed bucket range all-reduced "
                         "under the next graph; one_message = one graph, then one all-reduce of the whole gradient buffer")
    ap.add_argument("--exchange-payload", default="fp32", choices=["fp32", "bf16"],
                    help="several ranks, one-message exchange: what travels -- the fp32 gradient buffer (default, exact) or its bf16 rounding "
                         "(half the link bytes; local sums and Adam's moments stay fp32)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="BASELINE.json configs[]: 2 = the pre-training update (the headline metric, default); 3 = TTS fine-tuning step "
                         "(32 texts x 100 tokens -> 600 mel frames, guided attention, replayed) + full-size HiFi-GAN on the same "
                         "32 x 600 frames, one GPU")
    a = ap.parse_args()
    if _AS_SCRIPT and ("WORLD_SIZE" in os.environ or a.gpus == 1):   # (not in the parent that only re-executes itself under torch.distributed.run)
        quiet_stdout()
    if a.config == 3:
        return config3(a)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    assert ndev >= 1, "bench.py needs a GPU"
    shared = ndev < world          # functional check only: several ranks per device, gloo instead of RCCL
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1 or os.environ.get("ST5_DDP_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # The forms that run collectives UNDERNEATH the backward (phased replay -- the default --, eager bucketed exchange) pin RCCL to its ring
        # functions (no packed-fp32 VALU ops in them, unlike the tree / PreMulSum ones: tools/rccl_packed_ops.sh,
        # profiles/r3_rccl_packed_fp32.txt -- a round-3 precaution whose basis round 5 no longer believes to be a hardware hazard,
        # DESIGN.md 4c, kept because it is free there: 150-230 MB messages on a full mesh are ring territory anyway).  `--exchange
        # one_message` -- ONE all-reduce behind the local phase, nothing else on the chip -- leaves the algorithm to RCCL.
        if a.no_graph or (a.micro in ("in_turn", "side_by_side") and a.exchange == "phased"):
            os.environ.setdefault("NCCL_ALGO", "Ring")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from speecht5_amd import hip
    dtype = torch.float32 if a.dtype == "f32" else torch.bfloat16
    if a.dtype == "fp8":
        from speecht5_amd import functional as _Fn
        _Fn.set_fp8(True)
    # The update is captured into a HIP graph after the warm-up steps and REPLAYED in the timed region (fresh dropout
    # seeds / span masks / LayerDrop flags / lr per replay: speecht5_amd/graph.py).  ST5_GRAPH=0 or --no-graph: eager enqueue
    # (with the bucketed all-reduces overlapped with the backward when there are several ranks).
    use_graph = not a.no_graph and os.environ.get("ST5_GRAPH", "1") == "1"
    micro_mode = a.micro
    if os.environ.get("ST5_NT_TILE"):   # A/B: 1 = 128x128 always, 2 = 256x256 always (default: per problem)
        hip.lib().st5_gemm_set_nt_tile(int(os.environ["ST5_NT_TILE"]))
    if os.environ.get("ST5_DEEP_RING"):   # A/B: "max_blocks,nbuf" of the 128x128 NT kernel's deep operand ring (nbuf 2 = off)
        hip.lib().st5_gemm_set_deep_ring(*[int(v) for v in os.environ["ST5_DEEP_RING"].split(",")])
    if os.environ.get("ST5_NT_LONGK"):       # A/B: "tiles,nk" -- long-reduction NT problems of >= tiles 256x256 tiles on the phased kernel
        hip.lib().st5_gemm_set_nt_longk(*[int(v) for v in os.environ["ST5_NT_LONGK"].split(",")])
    if os.environ.get("ST5_MX8_HEAVY_NK"):   # A/B (fp8 mode): k-tiles an epilogue-heavy fp8 GEMM needs before it takes the phased 256x256 kernel
        hip.lib().st5_gemm_set_mx8_heavy_nk(int(os.environ["ST5_MX8_HEAVY_NK"]))
    if os.environ.get("ST5_MX8_TILE"):       # A/B (fp8 mode): 1 = 128x128 always, 2 = phased 256x256 always
        hip.lib().st5_gemm_set_mx8_tile(int(os.environ["ST5_MX8_TILE"]))
    if os.environ.get("ST5_TN_GROUP_TILE"):  # A/B: 1 = grouped weight gradients on 128x128 tiles always (default: phased 256x256 where the shapes allow)
        hip.lib().st5_gemm_set_tn_group_tile(int(os.environ["ST5_TN_GROUP_TILE"]))
    if os.environ.get("ST5_SPLITK_TARGET"):   # A/B: block count the weight-gradient split-K aims for
        hip.lib().st5_gemm_set_splitk_target(int(os.environ["ST5_SPLITK_TARGET"]))
    if os.environ.get("ST5_CONV0_MFMA"):  # A/B: 0 = VALU form of conv layer 0's forward apply pass (default 1: matrix cores)
        hip.lib().st5_conv0_set_mfma(int(os.environ["ST5_CONV0_MFMA"]))
    if os.environ.get("ST5_CONV0_GELU_TABLE"):  # A/B: 0 = conv layer 0's forward GELU as the polynomial (default 1: chord table in LDS)
        hip.lib().st5_conv0_set_gelu_table(int(os.environ["ST5_CONV0_GELU_TABLE"]))
    if os.environ.get("ST5_CONV0_FOLD"):  # A/B: 0 = conv layer 0's statistics and weight fragments in two launches (default 1: one)
        hip.lib().st5_conv0_set_fold(int(os.environ["ST5_CONV0_FOLD"]))
    if os.environ.get("ST5_TN_PHASED"):   # A/B: 1 / 2 = weight-gradient GEMMs on the phased 256x256 kernel (default 0 = the 128x128 kernel)
        hip.lib().st5_gemm_set_tn_phased(int(os.environ["ST5_TN_PHASED"]))
    if os.environ.get("ST5_NT_SLOTS"):   # A/B: 4 = two-stage operand ring of the 128x128 NT kernel (default 5 = five operand slots)
        hip.lib().st5_gemm_set_nt_slots(int(os.environ["ST5_NT_SLOTS"]))
    if os.environ.get("ST5_LN_MAX_BLOCKS"):   # A/B: block cap of the single-pass LayerNorm backward (default 256)
        hip.lib().st5_layernorm_set_max_blocks(int(os.environ["ST5_LN_MAX_BLOCKS"]))
    wgrad_env = os.environ.get("ST5_WGRAD_STREAM")
    args, task, model, upd = make_update(device, dtype, a.arch, a.batch, rank, graph=use_graph, micro=micro_mode, layerdrop=a.layerdrop,
                                         wgrad_stream=(wgrad_env == "1") if wgrad_env is not None else None,
                                         prefetch_host=os.environ.get("ST5_PREFETCH_HOST", "1") == "1", exchange=a.exchange,
                                         exchange_payload=a.exchange_payload)
    split_update = upd.split

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph:
        # untimed: eager updates, the two recording updates, and ONE replay (the first launch of a graph uploads it to the
        # device: ~150 ms that belong to set-up, not to the steady state) -- W updates in all when W >= 4, else 4
        for _ in range(max(a.warmup - 3, 1)):
            upd.eager_update()
        upd.prepare_graph()
        upd.update()
    else:
        for _ in range(a.warmup):
            upd.eager_update()
    barrier()
    hip.profiler.reset()
    t0 = time.perf_counter()
    for i in range(a.steps):
        # eager mode: HIP events around every st5_gemm launch of the LAST timed step (recording them on all K steps costs
        # ~10 % of the step in host time: two events per launch, ~750 launches per step)
        hip.profiler.enabled = (not use_graph) and (i == a.steps - 1)
        upd.update()
    barrier()
    dt = time.perf_counter() - t0
    upd.finish()
    hip.profiler.enabled = False
    if use_graph:
        # roofline leg of the graph mode: the replayed launches carry no events, so the SAME update is enqueued once more
        # eagerly, outside the timed region, with HIP events around every st5_gemm launch (same kernels, shapes, streams; the
        # fixed-shape / device-side-LayerDrop forms the graph is made of)
        # -- with the micro-batches IN TURN on one stream, whatever the timed mode: a launch's event pair then brackets that kernel
        # alone (side by side, the other stream's kernels share the chip during it and the rate would be the pair's, not the kernel's)
        from speecht5_amd import functional as Fn
        Fn._S.force_static = True
        timed_mode, upd.mode = upd.mode, "in_turn"
        upd.eager_update()           # (untimed: the in-turn form launches kernel instantiations the side-by-side steps never used --
        torch.cuda.synchronize()     #  their one-time set-up must not land inside an event pair of the sampled update)
        hip.profiler.enabled = True
        upd.eager_update()
        upd.mode = timed_mode
        Fn._S.force_static = False
        torch.cuda.synchronize()
        hip.profiler.enabled = False
    local_ms = None
    if use_graph and split_update and upd.sg is not None and (upd.sg.graphs is not None or upd.sg.graph is not None):
        # the captured local phase alone (zero_grad + both micro-batches; idempotent: it starts from cleared buffers), 5 replays
        torch.cuda.synchronize()
        with torch.cuda.stream(upd.sg.stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            upd.ddp.flat.zero_()
            if upd.ddp.flat2 is not None:
                upd.ddp.flat2.zero_()
            e0.record()
            for _ in range(5):
                if upd.sg.graphs is not None:       # (phased form: the phase graphs back to back, no exchange between them)
                    for g in upd.sg.graphs:
                        g.replay()
                    upd.ddp.flat.zero_()            # (each phase's sums start from cleared buffers, like the first)
                    if upd.ddp.flat2 is not None:
                        upd.ddp.flat2.zero_()
                    upd.ddp._reset_round()
                else:
                    upd.sg.graph.replay()
            e1.record()
        torch.cuda.synchronize()
        local_ms = round(e0.elapsed_time(e1) / 5, 3)
        upd.ddp.flat.zero_()
        if upd.ddp.flat2 is not None:
            upd.ddp.flat2.zero_()
        upd.ddp._pair_pending = False
        upd.ddp._grads_zeroed = True
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if os.environ.get("ST5_BENCH_SHAPES") and rank == 0:   # per-shape table of the sampled update's GEMM launches (stderr)
        rows = sorted(hip.profiler.by_shape().items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows)
        print(f"# GEMM launches of the sampled update by (variant, M, N, K, batch): {len(rows)} shapes, {tot * 1e3:.2f} ms", file=sys.stderr)
        for k, (n_, f_, t_) in rows[:60]:
            print(f"#  {k[0]:8s} M={k[1]:7d} N={k[2]:6d} K={k[3]:7d} b={k[4]:4d}  x{n_:3d}  {t_ * 1e3:7.3f} ms  {t_ / n_ * 1e6:7.1f} us/launch  {f_ / t_ / 1e12:6.1f} TFLOP/s", file=sys.stderr)
    audio_seconds = a.batch * 10.0 * world * a.steps
    prof = hip.profiler.summary()
    key = {"bf16": "bf16_NT", "f32": "f32_NT", "fp8": "fp8_NT"}[a.dtype]
    n, flops, secs = prof.get(key, (0, 0.0, 0.0))
    peak = {"bf16": BF16_DENSE_PEAK_TFLOPS, "f32": 157.3, "fp8": FP8_DENSE_PEAK_TFLOPS}[a.dtype]
    # measured HBM bytes per launch of the same kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command
    # (tools/pmc_traffic.sh), summary committed under profiles/ (the counters cannot be read from inside the process)
    traffic, traffic_src = None, None
    try:
        import json as _json
        here = os.path.dirname(os.path.abspath(__file__))
        pm = _json.load(open(os.path.join(here, "profiles", PMC_TRAFFIC_FILE)))
        # only a summary measured on THIS kernel source counts (the file records the hash of csrc/gemm.hip it was taken on)
        cur = kernel_source_hash()
        if pm.get("gemm_hip_sha1") == cur and a.dtype == "bf16":
            # every bf16 instantiation of the kernel (one per epilogue feature set), weighted by launches; the fp32 one serves the
            # NCE head only
            cand = [v for kname, v in pm["kernels"].items() if NT_KERNEL_NAME in kname and "IfLi" not in kname and "<float" not in kname]
            if cand:
                traffic = int(sum(v["hbm_corrected_bytes_per_launch"] * v["launches"] for v in cand) / sum(v["launches"] for v in cand))
                traffic_src = f"profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, bytes/launch; gemm.hip {cur})"
    except Exception:
        traffic = None
    alg_bytes = hip.profiler.nt_bytes_per_launch() if hasattr(hip.profiler, "nt_bytes_per_launch") else None
    # cross-check from the committed kernel trace of REPLAYED updates: the NT kernels' time per update =
    # sum of their TotalDurationNs / number of updates in the trace (= launches of the Adam kernel), against the same FLOPs
    rocprof_leg = None
    try:
        import csv as _csv, hashlib as _hl
        here = os.path.dirname(os.path.abspath(__file__))
        meta = json.load(open(os.path.join(here, "profiles", KERNEL_STATS_FILE.replace(".csv", ".meta.json"))))
        cur = kernel_source_hash()
        if meta.get("gemm_hip_sha1") == cur and a.dtype == "bf16" and a.arch == "base" and world == 1:
            rows = list(_csv.DictReader(open(os.path.join(here, "profiles", KERNEL_STATS_FILE))))
            upd_n = sum(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
            nt_ns = sum(float(r["TotalDurationNs"]) for r in rows
                        if "gemm_nt_glds_kernel" in r["Name"] or "gemm_nt256_kernel" in r["Name"] or "gemm_nt8p_kernel" in r["Name"]
                        or ("gemm_kernel" in r["Name"] and "Lb0ELb0" in r["Name"] and "DF16b" in r["Name"]))
            all_ns = sum(float(r["TotalDurationNs"]) for r in rows)
            if upd_n and nt_ns and flops > 0:
                rocprof_leg = {"source": f"profiles/{KERNEL_STATS_FILE} (rocprofv3 --kernel-trace --stats, {upd_n} updates, graph replays; gemm.hip {cur})",
                               "nt_ms_per_update": round(nt_ns / upd_n / 1e6, 3), "tflops": round(flops / (nt_ns / upd_n / 1e9) / 1e12, 1),
                               "frac": round(flops / (nt_ns / upd_n / 1e9) / 1e12 / peak, 4),
                               "all_kernels_ms_per_update": round(all_ns / upd_n / 1e6, 3)}
    except Exception:
        rocprof_leg = None
    roof = {"bound": "mfma", "kernel": f"NT-form st5_gemm launches <{a.dtype}> (Linear / attention-projection / conv forward and data-gradient GEMMs; "
                      f"`traffic` is per launch of {NT_KERNEL_NAME})",
            "note": ("launch durations: HIP events around every st5_gemm launch of ONE eagerly enqueued update after the timed region "
                     "(replayed launches carry no events) with the micro-batches in turn on one stream, so that every event pair brackets "
                     "one kernel alone"
                     if use_graph else
                     "launch durations are measured inside the step, i.e. beside the weight-gradient stream's kernels "
                     "(ST5_WGRAD_STREAM=0 gives the isolated rate, ~8 % higher)"),
            "achieved": round(flops / secs / 1e12, 2) if secs > 0 else None, "peak": peak, "unit": "TFLOP/s",
            "frac": round(flops / secs / 1e12 / peak, 4) if secs > 0 else None, "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes,
            "from_kernel_trace": rocprof_leg,
            "launches_per_step": n, "sampled_steps": 1, "avg_launch_us": round(secs / max(n, 1) * 1e6, 2),
            "all_variants": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None,
                                 "ms_per_step": round(v[2] * 1e3, 3)} for k, v in sorted(prof.items())}}
    # whole update against the MFMA peak: algorithmic FLOPs of every st5_gemm launch of the sampled update (fwd + bwd,
    # all variants; attention-core FLOPs are not GEMM launches and are left out) / the step time
    all_flops = sum(v[1] for v in prof.values())
    roof["step"] = {"gemm_tflop_per_update": round(all_flops / 1e12, 3), "ms": round(dt / a.steps * 1e3, 3),
                    "tflops": round(all_flops / (dt / a.steps) / 1e12, 1), "frac_of_peak": round(all_flops / (dt / a.steps) / 1e12 / peak, 4)}
    # HBM-bound leg ("HBM GB/s (conv frontend)"): conv layer 0 + GroupNorm + GELU, and the frontend as a unit
    # (7 conv layers: 324.8 MB per 10 s clip fwd + bwd in bf16), same HIP-event method
    hbm, rfl = {}, hip.profiler.region_flops()
    for name, (n_, b_, t_) in sorted(hip.profiler.region_summary().items()):
        hbm[name] = {"launches": n_, "algorithmic_MB": round(b_ / n_ / 1e6, 1), "ms": round(t_ / n_ * 1e3, 4), "achieved_GBps": round(b_ / t_ / 1e9, 1) if t_ > 0 else None,
                     "frac_of_8TBps": round(b_ / t_ / 8e12, 4) if t_ > 0 else None}
        if rfl.get(name):
            hbm[name]["tflops"] = round(rfl[name] / t_ / 1e12, 1) if t_ > 0 else None
    # conv layer 0 once more, as DEVICE time: the figures above bracket one eagerly enqueued call each (4 / 6 small launches whose host
    # gaps the events include); here 20 forward and 20 backward calls are enqueued back to back on the same shapes, events around the
    # whole train -- the queue runs ahead of the GPU, so the quotient is the kernels' own time per call, as inside a replayed graph
    if world == 1 and a.arch == "base" and "conv0_gn_gelu_fwd" in hbm:
        try:
            c0 = conv0_device_time(device, a.batch)
            for nm, ms in c0.items():
                mb = hbm[nm]["algorithmic_MB"]
                hbm[nm].update(device_ms=ms, device_GBps=round(mb / ms, 1), device_frac_of_8TBps=round(mb / ms / 8e3, 4))
        except Exception as e:      # (a measurement extra must never cost the line)
            hbm["conv0_device_time_error"] = repr(e)
    front = [hbm.pop(k) for k in ("conv_frontend_fwd", "conv_frontend_bwd") if k in hbm]
    if len(front) == 2:
        fb, ft = sum(f["algorithmic_MB"] for f in front), sum(f["ms"] for f in front)
        ffl = rfl.get("conv_frontend_fwd", 0.0) + rfl.get("conv_frontend_bwd", 0.0)
        roof["frontend"] = {"what": "conv feature extractor as a unit (7 layers, GroupNorm + GELU fused), forward + backward of the speech micro-batch; "
                                    "HIP events around the whole launch sequence of the sampled update",
                            "bytes": int(fb * 1e6), "ms": round(ft, 4), "GBps": round(fb / ft, 1), "frac_of_8TBps": round(fb / ft / 8e3, 4),
                            "tflops": round(ffl / (ft * 1e-3) / 1e12, 1), "fwd": front[0], "bwd": front[1]}
    roof["hbm_bound_kernels"] = hbm
    if rank == 0:
        nm = "Large" if a.arch == "large" else "Base"
        out = {"metric": f"audio-sec/s fwd+bwd SpeechT5-{nm}", "value": round(audio_seconds / dt, 2), "unit": "audio-sec/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"SpeechT5-{nm} pretrain step (speech {a.batch}x10s micro-batch + text 16x512 micro-batch, update-freq 2), "
                                      "fwd+bwd+allreduce+clip+Adam, per GPU", "arch": ("t5_transformer_large (24 enc + 6 dec, d=1024, pre-LN, layer-norm feature extractor)" if a.arch == "large"
                                   else "t5_transformer_base (12 enc + 6 dec, d=768)"),
                          "enqueue": ("hip-graph replay of the local phase in 3 graphs cut at bucket boundaries (both micro-batches inside each, "
                                      "on two streams), the bucket range each completes summed over the two gradient buffers and all-reduced "
                                      "(async, the group's stream) under the next graph, then Adam" if (getattr(upd, "phased", False) and micro_mode == "side_by_side") else
                                      "hip-graph replay of the local phase in 3 graphs cut at bucket boundaries, the bucket range each completes "
                                      "all-reduced (RCCL, async) under the next graph, then Adam" if getattr(upd, "phased", False) else
                                      "hip-graph replay of the local phase + eager all-reduce (one message) + Adam" if split_update else
                                      "hip-graph replay" if use_graph else "eager"),
                          "micro_batches": {"side_by_side": "forward and backward side by side on two streams, two gradient buffers",
                                            "in_turn_2buf": "in turn, two gradient buffers", "in_turn": "in turn"}[micro_mode],
                          "global_speech_batch": a.batch * world, "clip_seconds": 10, "parallelism": f"dp{world}",
                          "dropout": 0.1, "layerdrop": a.layerdrop,
                          "layerdrop_form": ("device-side select (every layer runs)" if use_graph else "host skip") if a.layerdrop > 0 else "off"},
               "roofline": roof}
        # what the ranks exchange per update and how (so that a scaling curve can be read): message sizes, payload,
        # the RCCL algorithm in force, and the time of the replayed local phase alone (ms_per_step minus it = exposed exchange + Adam)
        if split_update or world > 1:
            msgs = upd.ddp.exchange_plan(phased=bool(getattr(upd, "phased", False)), cuts=upd.cut_buckets() if getattr(upd, "phased", False) else None)
            if a.exchange_payload == "bf16" and split_update and not getattr(upd, "phased", False):
                msgs = [m // 2 for m in msgs]
            out["config"]["exchange"] = {"form": "phased" if getattr(upd, "phased", False) else ("one_message" if split_update else "bucketed, overlapped with an eager backward"),
                                         "backend": dist.get_backend() if dist.is_initialized() else None,
                                         "NCCL_ALGO": os.environ.get("NCCL_ALGO"),
                                         "payload": ("bf16 rounding of the fp32 gradient sum" if (a.exchange_payload == "bf16" and split_update and not getattr(upd, "phased", False))
                                                     else "fp32 gradients") + ", sum; mean over ranks and micro-batches inside Adam",
                                         "message_bytes": msgs, "local_phase_ms": local_ms}
        out["config"]["box"] = box_info()
        if shared:
            out["config"]["note"] = f"FUNCTIONAL CHECK ONLY: {world} ranks share {ndev} GPU(s), gradients exchanged over gloo"
        if world == 1 and not a.no_cpu_baseline and a.arch == "base":
            out["cpu_baseline"] = cpu_baseline(model, args)
        emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    _AS_SCRIPT = True
    main()
