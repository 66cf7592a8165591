"""BASELINE.json cfg 3 on one MI355X (`python bench.py --config 3` prints its line; this file run directly prints the raw dict):
(a) the TTS fine-tuning step of the README recipe (SpeechT5-Base, batch 32 texts of 100 tokens -> 600 mel frames, reduction
factor 2, guided-attention loss, dropout 0.15; forward + backward + clip + Adam, bf16), replayed as a HIP graph, and (b) the
full-size HiFi-GAN vocoder (HuggingFace SpeechT5HifiGan configuration) on the same 32 x 600 frames, forward only -- with its output
checked against the HuggingFace module on the CPU for a short clip (same random weights)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch

PEAK = 2500.0   # dense bf16 MFMA TFLOP/s (MI355X_MICROARCH.md)


def _gemm_summary(hip):
    prof = hip.profiler.summary()
    fl, t = sum(v[1] for v in prof.values()), sum(v[2] for v in prof.values())
    return {"launches": sum(v[0] for v in prof.values()), "gemm_tflop": round(fl / 1e12, 3), "gemm_ms": round(t * 1e3, 3),
            "tflops_in_gemms": round(fl / t / 1e12, 1) if t > 0 else None, "frac_of_peak_in_gemms": round(fl / t / 1e12 / PEAK, 4) if t > 0 else None,
            "by_variant": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None, "ms": round(v[2] * 1e3, 3)}
                           for k, v in sorted(prof.items())}}


def run(steps=20, warmup=5, graph=True, vocoder_reps=10, check_hf=True, tts=True):
    from speecht5_amd import functional as Fn
    dev = torch.device("cuda:0")
    Fn.set_compute_dtype(torch.bfloat16)
    B, Tt, L = 32, 100, 600
    audio_s = B * L * 256 / 16000.0
    stream = torch.cuda.Stream(device=dev)       # (never the NULL stream: DESIGN.md 4a)
    out = {"cfg": 3}
    if tts:
        out["tts_finetune_step"] = _tts(steps, warmup, graph, dev, stream, B, Tt, L, audio_s)
    out["hifigan_forward"] = _vocoder(vocoder_reps, check_hf, dev, stream, B, L, audio_s)
    Fn.set_compute_dtype(torch.float32)
    return out


def _tts(steps, warmup, graph, dev, stream, B, Tt, L, audio_s):
    from speecht5_amd import functional as Fn, hip
    from speecht5_amd.criterions import SpeechT5Criterion
    from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
    from speecht5_amd.graph import StepGraph
    from speecht5_amd.speecht5 import t5_transformer_base
    from speecht5_amd.synthetic import t2s_sample
    from speecht5_amd.task import SpeechT5Task
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=False, share_input_output_embed=True,
                     encoder_layerdrop=0.0, decoder_layerdrop=0.0, dropout=0.15, attention_dropout=0.15, activation_dropout=0.15)
    t5_transformer_base(args)
    task = SpeechT5Task.synthetic(args)
    task.t5_task = "t2s"
    torch.manual_seed(1337)
    model = task.build_model(args).to(dev)
    crit = SpeechT5Criterion(task, use_guided_attn_loss=True, guided_attn_loss_lambda=10.0, guided_attn_loss_sigma=0.4, bce_pos_weight=5.0,
                             sync_logging=False)
    ddp = FlatGradDataParallel(model)
    opt = FusedAdam(ddp, lr=1e-4, clip_norm=25.0, weight_decay=0.1)
    sample = t2s_sample(B=B, T_text=Tt, L=L, vocab=len(task.dicts["text"]), device=dev)
    n_upd = [0]

    def step():
        ddp.zero_grad()
        ddp.accumulate([sample], lambda s: task.train_step(s, model, crit, None, n_upd[0], sync=False))
        ddp.finish()
        opt.step(1.0)

    enqueue, sg = "eager", None
    with torch.cuda.stream(stream):
        for _ in range(max(warmup - 3, 1)):
            step()
    stream.synchronize()
    if graph:
        try:
            sg = StepGraph(step, opt=opt, model=model, device=dev, stream=stream, on_step=lambda: n_upd.__setitem__(0, n_upd[0] + 1))
            sg.record(); sg.record(); sg.capture()
            with torch.cuda.stream(stream):
                sg.replay()
            enqueue = "hip-graph replay"
        except Exception as e:      # the eager number is still a number; say why
            enqueue, sg = f"eager (capture failed: {type(e).__name__}: {str(e)[:120]})", None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            sg.replay() if sg is not None else step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if sg is not None:
        sg.drain()
    # roofline leg: one eagerly enqueued step with HIP events around every st5_gemm launch
    hip.profiler.reset()
    hip.profiler.enabled = True
    Fn._S.force_static = sg is not None
    with torch.cuda.stream(stream):
        step()
    torch.cuda.synchronize()
    Fn._S.force_static = False
    hip.profiler.enabled = False
    g = _gemm_summary(hip)
    res = {"tts_finetune_step": {"ms_per_step": round(dt * 1e3, 2), "utterances_per_s": round(B / dt, 1),
                                           "audio_sec_per_s": round(audio_s / dt, 1), "batch": B, "text_tokens": Tt, "mel_frames": L,
                                           "enqueue": enqueue, "dtype": "bf16",
                                           "work": "fwd + bwd + clip + Adam, guided-attention loss, dropout 0.15, reduction factor 2",
                                           "gemm": g, "step_tflops": round(g["gemm_tflop"] / dt, 1),
                                           "step_frac_of_peak": round(g["gemm_tflop"] / dt / PEAK, 4)}}
    if sg is not None:
        sg.step_fn = sg.on_step = None
        sg.graph = None
    ddp.close()
    return res["tts_finetune_step"]


def _vocoder(vocoder_reps, check_hf, dev, stream, B, L, audio_s):
    from speecht5_amd import functional as Fn, hip
    from speecht5_amd.hifigan import SpeechT5HifiGan
    out = {}
    torch.manual_seed(7)
    voc = SpeechT5HifiGan().to(dev).eval()
    for p in voc.parameters():
        torch.nn.init.normal_(p, std=0.02)
    mel = torch.randn(B, L, 80, device=dev) * 0.5 - 1
    with torch.cuda.stream(stream):
        for _ in range(3):
            wav = voc(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(vocoder_reps):
            wav = voc(mel)
    torch.cuda.synchronize()
    dv = (time.perf_counter() - t0) / vocoder_reps
    hip.profiler.reset()
    hip.profiler.enabled = True
    with torch.cuda.stream(stream):
        voc(mel)
    torch.cuda.synchronize()
    hip.profiler.enabled = False
    gv = _gemm_summary(hip)
    # per upsampling stage (channels 256 / 128 / 64 / 32 at 4 / 16 / 64 / 256 x the frame rate): the GEMM launches by output width
    stages = {}
    for (variant, M, N, K, batch), (n_, f_, t_) in hip.profiler.by_shape().items():
        st = stages.setdefault(f"N={N}", [0, 0.0, 0.0])
        st[0] += n_; st[1] += f_; st[2] += t_
    flops = 164e9 * B * L / 600.0           # 164 GFLOP per 600-frame utterance
    out["hifigan_forward"] = {"ms_per_batch": round(dv * 1e3, 2), "audio_sec_per_s": round(audio_s / dv, 1), "tflops": round(flops / dv / 1e12, 1),
                              "frac_of_peak": round(flops / dv / 1e12 / PEAK, 4),
                              "batch": B, "mel_frames": L, "samples_out": int(wav.shape[-1]), "dtype": "bf16", "gemm": gv,
                              "by_output_channels": {k: {"launches": v[0], "ms": round(v[2] * 1e3, 3), "tflops": round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None}
                                                     for k, v in sorted(stages.items(), key=lambda kv: -kv[1][2])}}
    if check_hf:
        try:
            from transformers import SpeechT5HifiGan as HF, SpeechT5HifiGanConfig
            hf = HF(SpeechT5HifiGanConfig()).eval()
            sd = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
            hf_sd = hf.state_dict()
            hf.load_state_dict({k: sd[k] for k in hf_sd if k in sd and sd[k].shape == hf_sd[k].shape}, strict=False)
            missing = [k for k in hf_sd if k not in sd or sd[k].shape != hf_sd[k].shape]
            m1 = mel[:1, :100].float().cpu()
            with torch.no_grad():
                ref = hf(m1)
            Fn.set_compute_dtype(torch.float32)
            got = voc(m1.to(dev)).float().cpu()
            err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
            out["hifigan_forward"]["check_vs_hf_cpu_fp32"] = {"frames": 100, "max_err_rel": err, "unmatched_params": missing[:4]}
        except Exception as e:   # the check is informative; the timing above stands on its own
            out["hifigan_forward"]["check_vs_hf_cpu_fp32"] = f"not run: {type(e).__name__}: {e}"
    return out["hifigan_forward"]


if __name__ == "__main__":
    print(json.dumps(run(steps=int(os.environ.get("STEPS", 20)), tts=os.environ.get("CFG3_VOCODER_ONLY") != "1",
                         check_hf=os.environ.get("CFG3_VOCODER_ONLY") != "1")))
