"""Parity AT THE cfg-3 SHAPE (BASELINE.json configs[2]; VERDICT r5 "next round" item 1).

`bench.py --config 3` times Base TTS fine-tuning (texts of 100 tokens -> 600 mel frames = 300 decoder frames at reduction factor 2,
guided-attention loss on the per-head cross-attention weights of EVERY decoder layer: models/modules/decoder.py:247-254 with
`alignment_layer = -1`, criterions/text_to_speech_loss.py:296-427); until this file the t2s parity and the 1e-3 mel bar of
BASELINE.json existed only on the 2-layer d = 128 golden model (`tests/test_model_gpu.py`).  Here, against the CPU oracle on the
model's own random-init weights:

  * `t5_transformer_base` (12 + 6 layers, d = 768, 12 heads), task t2s, `TexttoSpeechLoss(use_guided_attn_loss=True, lambda 10,
    sigma 0.4, bce_pos_weight 5)` -- the README recipe's criterion flags
  * batch 4 x 100 tokens -> 4 x 600 mel frames; item 3 PADDED (80 tokens, 500 frames): key-padding mask live in the encoder
    self-attention and in the unfused cross-attention that returns the alignment weights, the length trim / stop-label fix-up and the
    guided-attention masks with ragged (ilen, olen)
  * dropout 0 everywhere (incl. the always-on Tacotron pre-net dropout), BatchNorm of the post-net on batch statistics

fp32 parity mode through the C ABI: loss and every loss term 2e-4, EVERY parameter's gradient (relative Frobenius error 1e-3, cosine
0.99999); bf16 compute mode (what bench.py times): loss 3e-2 and every parameter's gradient at the cosine bars of
test_cfg2_shape_gpu.py.  `generate_speech` (models/speecht5.py:1188-1249) at Base size, KV-cached on the product side and
re-running the whole prefix on the oracle side: mel within 1e-3 relative (BASELINE.json's TTS bar; measured 3e-6), with the data-dependent stop step equal to the oracle's."""
from argparse import Namespace
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL, COS_TOL = 1e-3, 0.99999
from tests.util import BF16_POST_COS
BF16_COS, BF16_REL = 0.999, 5e-2
BF16_LOOSE = {"speech_decoder_postnet.feat_out.weight", "speech_decoder_postnet.feat_out.bias"}


def build(dev, dtype, seed=4245):
    from speecht5_amd import functional as Fn
    from speecht5_amd.speecht5 import t5_transformer_base
    from speecht5_amd.task import SpeechT5Task
    Fn.set_compute_dtype(dtype)
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=False,
                     share_input_output_embed=True, encoder_layerdrop=0.0, decoder_layerdrop=0.0)
    t5_transformer_base(args)
    for k, v in list(vars(args).items()):          # parity needs identical arithmetic: no dropout anywhere
        if "dropout" in k and isinstance(v, float):
            setattr(args, k, 0.0)
    task = SpeechT5Task.synthetic(args)
    task.t5_task = "t2s"
    torch.manual_seed(seed)
    model = task.build_model(args).to(dev)
    return args, task, model


def t2s_batch(vocab):
    """4 x 100 tokens -> 4 x 600 frames; item 3 padded the way TextToSpeechDataset.collater pads (data/text_to_speech_dataset.py:
    228-291): pad index 1 behind </s>, zeros in the mel target and the decoder input, stop label 1 on the last valid frame."""
    from speecht5_amd.synthetic import t2s_sample
    s = t2s_sample(B=4, T_text=100, L=600, vocab=vocab, device="cpu", seed=21)
    ni = s["net_input"]
    n, L3 = 80, 500
    ni["src_tokens"][3, n - 1] = 2
    ni["src_tokens"][3, n:] = 1
    ni["src_lengths"][3] = n
    s["src_lengths"][3] = n
    s["dec_target"][3, L3:] = 0.0
    s["dec_target_lengths"][3] = L3
    s["labels"][3] = 0.0
    s["labels"][3, L3 - 1:] = 1.0
    ni["prev_output_tokens"][3, L3 // 2:] = 0.0
    ni["tgt_lengths"][3] = L3 // 2
    s["ntokens"] = int(s["dec_target_lengths"].sum())
    return s


def off_the_kinks(sample, out, margin=1e-4, step=1e-2):
    """|prediction - target| has a kink where they are equal: one element whose difference changes sign between two fp32
    evaluations moves d(after) by 2 / sqrt(184 000) = 4.7e-3 of its norm (measured here: 3.2e-3 on a BatchNorm bias gradient from ONE
    such element, `tools/r6/postnet_probe.py`; the post-net kernels themselves agree with torch fp64 to 1e-6 on the same d(after)).
    The synthetic loss targets are therefore moved off the kinks of the oracle's own predictions (the decoder INPUT frames stay as
    they are, so the predictions do not move) -- a few tens of the 184 000 elements, by 0.01."""
    ys = sample["dec_target"]
    n = 0
    for _ in range(8):
        near = ((out["after"].detach() - ys).abs() < margin) | ((out["before"].detach() - ys).abs() < margin)
        if not bool(near.any()):
            break
        n += int(near.sum())
        ys[near] += step
    assert not bool((((out["after"].detach() - ys).abs() < margin) | ((out["before"].detach() - ys).abs() < margin)).any())
    return n


def oracle_run(model, args, sample):
    from oracle import speecht5_oracle as O
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**vars(args))
    with torch.no_grad():
        moved = off_the_kinks(sample, O.forward_t2s(sd, cfg, sample, training=True))
    print(f"loss targets moved off an L1 kink: {moved} of {sample['dec_target'].numel()}")
    out = O.forward_t2s(sd, cfg, sample, training=True)
    loss, l1, l2, bce = O.tacotron_loss(out["before"], out["after"], out["stop_logits"], sample, cfg.reduction_factor, bce_pos_weight=5.0)
    att = torch.cat([a[:, :2] for a in out["attn"]], dim=1)
    olens_in = torch.div(sample["dec_target_lengths"], cfg.reduction_factor, rounding_mode="floor")
    ga = O.guided_attention_loss(att, sample["src_lengths"], olens_in, sigma=0.4, alpha=10.0)
    (loss + ga).backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    from tests.util import TIED
    tied = [grads[a] for a in TIED if a in grads]
    if tied:
        total = sum(tied)
        for a in TIED:
            grads[a] = total
    return dict(loss=float((loss + ga).detach()), l1=float(l1.detach()), l2=float(l2.detach()), bce=float(bce.detach()), ga=float(ga.detach()),
                n_attn=len(out["attn"]), attn_shape=tuple(out["attn"][0].shape), grads=grads)


def product_run(model, task, dev, sample):
    from speecht5_amd.criterions import TexttoSpeechLoss
    from tests.util import to_dev
    model.train()
    crit = TexttoSpeechLoss(task, False, use_guided_attn_loss=True, guided_attn_loss_sigma=0.4, guided_attn_loss_lambda=10.0,
                            bce_pos_weight=5.0, sync_logging=False)
    for p in model.parameters():
        p.grad = None
    loss, ss, log = crit(model, to_dev(sample, dev))
    assert ss == 1
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    f = lambda v: float(torch.as_tensor(v).detach())
    return dict(loss=f(loss), l1=f(log["l1_loss"]), l2=f(log["l2_loss"]), bce=f(log["bce_loss"]), ga=f(log["enc_dec_attn_loss"]), grads=grads)


def compare_grads(got, ref, rnorm):
    rows = []
    zero_bad = []
    for n, g in got.items():
        r = ref.get(n)
        if r is None:
            continue
        r = r.double()
        if float(r.norm()) <= 1e-6 * rnorm:      # structurally ~zero (softmax-shift-invariant key bias; modules t2s does not touch)
            if float(g.norm()) > 1e-4 * rnorm:
                zero_bad.append((n, "expected ~0", float(g.norm())))
            continue
        rel = float((g - r).norm() / r.norm())
        cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
        rows.append((cos, rel, n, tuple(g.shape)))
    rows.sort()
    return rows, zero_bad


def test_cfg3_shape_tts_finetune_step_matches_oracle(cuda):
    from speecht5_amd import functional as Fn
    try:
        args, task, model = build(cuda, torch.float32)
        sample = t2s_batch(len(task.dicts["text"]))
        ref = oracle_run(model, args, sample)
        assert ref["n_attn"] == 6 and ref["attn_shape"] == (4, 12, 300, 100), (ref["n_attn"], ref["attn_shape"])
        got = product_run(model, task, cuda, sample)
        bad = []
        for k in ("loss", "l1", "l2", "bce", "ga"):
            if abs(got[k] - ref[k]) > 2e-4 * max(abs(ref[k]), 1e-3):
                bad.append((k, got[k], ref[k]))
        rnorm = sum(float(ref["grads"][n].double().pow(2).sum()) for n in got["grads"] if n in ref["grads"]) ** 0.5
        gnorm = sum(float(g.pow(2).sum()) for g in got["grads"].values()) ** 0.5
        if abs(gnorm - rnorm) > 5e-3 * rnorm:
            bad.append(("grad norm", gnorm, rnorm))
        rows, zero_bad = compare_grads(got["grads"], ref["grads"], rnorm)
        bad += zero_bad + [(n, rel, cos) for cos, rel, n, _ in rows if rel > REL_TOL or cos < COS_TOL]
        print(f"fp32 parity mode at the cfg-3 shape: loss {got['loss']:.6f} (oracle {ref['loss']:.6f}), guided attention {got['ga']:.6f} "
              f"({ref['ga']:.6f}), grad norm {gnorm:.6f} (oracle {rnorm:.6f}), {len(rows)} parameters compared; worst:", rows[:5])
        assert len(rows) > 240, len(rows)
        assert not bad, bad[:12]

        # ---- bf16 compute mode: the kernels `bench.py --config 3` times ----
        sdict = {k: v.detach().clone() for k, v in model.state_dict().items()}
        del model
        args2, task2, model2 = build(cuda, torch.bfloat16)
        torch.nn.Module.load_state_dict(model2, sdict, strict=True)
        got2 = product_run(model2, task2, cuda, sample)
        for k, tol in (("loss", 3e-2), ("l1", 3e-2), ("bce", 3e-2), ("ga", 3e-2)):
            assert abs(got2[k] - ref[k]) <= tol * max(abs(ref[k]), 1e-3), ("bf16", k, got2[k], ref[k])
        g2norm = sum(float(g.pow(2).sum()) for g in got2["grads"].values()) ** 0.5
        assert abs(g2norm - rnorm) <= 3e-2 * rnorm, ("bf16 grad norm", g2norm, rnorm)
        rows2, _ = compare_grads(got2["grads"], ref["grads"], rnorm)
        post = [w for w in rows2 if "speech_decoder_postnet.postnet" in w[2]]
        rest = [w for w in rows2 if "speech_decoder_postnet.postnet" not in w[2]]
        print("bf16 at the cfg-3 shape, worst outside the post-net:", rest[:8])
        print("bf16 at the cfg-3 shape, worst inside the post-net:", post[:4])
        bad2 = [w for w in rest if (w[0] < BF16_COS and w[2] not in BF16_LOOSE) or (w[1] > BF16_REL and len(w[3]) > 0 and w[2] not in BF16_LOOSE)] + \
               [w for w in rest if w[2] in BF16_LOOSE and w[0] < 0.997] + [w for w in post if w[0] < BF16_POST_COS]
        assert len(rows2) > 240
        assert not bad2, bad2[:10]
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()


@pytest.mark.parametrize("threshold,stop_bias", [(2.0, 0.0), (None, -0.35)])
def test_cfg3_generate_speech_mel_matches_oracle_at_base_size(cuda, threshold, stop_bias):
    """`T5TransformerModel.generate_speech` at Base size.  threshold = 2.0: no stop probability can reach it and the reference reads
    the same keyword for BOTH length ratios (speecht5.py:1191-1201), so the loop runs exactly T_in * 2 / r = 100 decoder steps = 200
    mel frames.  threshold absent: the defaults (0.5 / 0.0 / 20.0), the loop ends at the first step whose stop probability reaches
    0.5 -- data dependent; the stop projection's bias is lowered so that this is not step 1, and the step must agree with the oracle."""
    from oracle import speecht5_oracle as O
    from speecht5_amd import functional as Fn
    try:
        args, task, model = build(cuda, torch.float32, seed=4246)
        with torch.no_grad():
            model.speech_decoder_postnet.prob_out.bias.fill_(stop_bias)
        model.eval()
        sample = t2s_batch(len(task.dicts["text"]))
        ni = sample["net_input"]
        sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        cfg = SimpleNamespace(**vars(args))
        kw = {} if threshold is None else {"threshold": threshold}
        with torch.no_grad():
            ref = O.generate_speech(sd, cfg, ni["src_tokens"][:1], ni["spkembs"][:1], **kw)
            mel, probs, attn = model.generate_speech(src_tokens=ni["src_tokens"][:1].to(cuda), spkembs=ni["spkembs"][:1].to(cuda), **kw)
        torch.cuda.synchronize()
        print(f"generate_speech at Base size (threshold {threshold}): {tuple(mel.shape)} mel frames (oracle {tuple(ref.shape)}), "
              f"max |mel| {float(ref.abs().max()):.4f}, max err {float((mel.float().cpu() - ref).abs().max()) if mel.shape == ref.shape else float('nan'):.3e}")
        assert mel.shape == ref.shape, (mel.shape, ref.shape)
        if threshold is not None:
            assert mel.shape[0] == 200
        else:
            assert 2 < mel.shape[0] < 2000     # neither the first step nor the length limit
        from tests.util import close
        close(mel, ref, 1e-3, what="generated mel at Base size (TTS parity bar: 1e-3 relative)")
        assert attn.shape[:2] == (6, 12) and attn.shape[2] == mel.shape[0] // 2 and attn.shape[3] == 100, attn.shape
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()
