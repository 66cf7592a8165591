"""BASELINE.json cfg 3 on one MI355X: (a) the TTS fine-tuning step of the README recipe (SpeechT5-Base, batch 32 texts of 100
tokens -> 600 mel frames, reduction factor 2, guided-attention loss, dropout 0.15; forward + backward + clip + Adam, bf16) and
(b) the full-size HiFi-GAN vocoder (HuggingFace SpeechT5HifiGan configuration) on the same 32 x 600 frames, forward only --
with its output checked against the HuggingFace module on the CPU for a short clip (same random weights).
One JSON object on stdout (copied to profiles/ by hand)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from speecht5_amd import functional as Fn, hip
from speecht5_amd.criterions import SpeechT5Criterion
from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
from speecht5_amd.hifigan import SpeechT5HifiGan
from speecht5_amd.speecht5 import t5_transformer_base
from speecht5_amd.synthetic import t2s_sample
from speecht5_amd.task import SpeechT5Task

dev = torch.device("cuda:0")
steps, warm = int(os.environ.get("STEPS", 20)), 5
Fn.set_compute_dtype(torch.bfloat16)
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
B, Tt, L = 32, 100, 600
sample = t2s_sample(B=B, T_text=Tt, L=L, vocab=len(task.dicts["text"]), device=dev)


def step(i):
    ddp.zero_grad()
    ddp.accumulate([sample], lambda s: task.train_step(s, model, crit, None, i, sync=False))
    ddp.finish()
    opt.step(1.0)


for i in range(warm):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(warm + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
audio_s = B * L * 256 / 16000.0
out = {"cfg": 3, "tts_finetune_step": {"ms_per_step": round(dt * 1e3, 2), "utterances_per_s": round(B / dt, 1),
                                       "audio_sec_per_s": round(audio_s / dt, 1), "batch": B, "text_tokens": Tt, "mel_frames": L,
                                       "enqueue": "eager", "dtype": "bf16",
                                       "work": "fwd + bwd + clip + Adam, guided-attention loss, dropout 0.15, reduction factor 2"}}
ddp.close()

# ---- vocoder ----
torch.manual_seed(7)
voc = SpeechT5HifiGan().to(dev).eval()
for p in voc.parameters():
    torch.nn.init.normal_(p, std=0.02)
mel = torch.randn(B, L, 80, device=dev) * 0.5 - 1
for _ in range(3):
    wav = voc(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    wav = voc(mel)
torch.cuda.synchronize()
dv = (time.perf_counter() - t0) / n
flops = 164e9 * B * L / 600.0           # 164 GFLOP per 600-frame utterance
out["hifigan_forward"] = {"ms_per_batch": round(dv * 1e3, 2), "audio_sec_per_s": round(audio_s / dv, 1), "tflops": round(flops / dv / 1e12, 1),
                          "batch": B, "mel_frames": L, "samples_out": int(wav.shape[-1]), "dtype": "bf16"}
try:
    from transformers import SpeechT5HifiGan as HF, SpeechT5HifiGanConfig
    hf = HF(SpeechT5HifiGanConfig()).eval()
    sd = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
    hf_sd = hf.state_dict()
    # HF keeps weight_norm parametrisations off at inference in this version: same tensor names as ours when shapes agree
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
print(json.dumps(out))
