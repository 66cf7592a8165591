"""Parity AT THE BENCHED SHAPE (BASELINE.json configs[1]; VERDICT r2 "next round" item 1a).

`test_fullsize_gpu.py` pins the Base architecture on a 2 s clip (T = 99: one attention tile, no padding, no tails).  bench.py
times 10 s clips (T_enc = 499, T_dec = 313: several attention tiles with row / key tails, the 104-bucket relative-position
window sliding over the +-160 clip) next to 512-token text.  This file runs those shapes against the CPU oracle:

  * speech micro-batch 2 x 10 s, the second clip PADDED to 8 s (key-padding mask live in every encoder self-attention and
    decoder cross-attention tile, a shorter mel target with its own stop label, the padded rows inside the convolution halo)
  * text micro-batch 4 x 512 tokens, the last sentence padded to 400
  * Base (12 + 6 layers, d = 768), dropout 0, the same injected span mask / code-book time mix / zero Gumbel noise on both sides

fp32 parity mode through the C ABI: loss 2e-4, every loss term 5e-4, EVERY parameter's gradient (relative Frobenius error
1e-3, cosine 0.99999); bf16 compute mode (the kernels bench.py times): loss 3e-2 and every parameter's gradient at the cosine
bars of test_fullsize_gpu.py."""
from argparse import Namespace
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL, COS_TOL = 1e-3, 0.99999
from tests.util import BF16_POST_COS   # (one bar for the mel post-net's bf16 gradients; measured here: 0.9934)
BF16_COS, BF16_REL = 0.999, 5e-2      # (measured at this shape: everything outside the post-net >= 0.99948)
BF16_LOOSE = {"quantizer.vars", "speech_decoder_postnet.feat_out.weight", "speech_decoder_postnet.feat_out.bias"}


def build(dev, dtype, seed=4243):
    from speecht5_amd import functional as Fn
    from speecht5_amd.speecht5 import t5_transformer_base
    from speecht5_amd.task import SpeechT5Task
    Fn.set_compute_dtype(dtype)
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True,
                     share_input_output_embed=True, encoder_layerdrop=0.0, decoder_layerdrop=0.0)
    t5_transformer_base(args)
    for k, v in list(vars(args).items()):          # parity needs identical arithmetic: no dropout anywhere
        if "dropout" in k and isinstance(v, float):
            setattr(args, k, 0.0)
    task = SpeechT5Task.synthetic(args)
    torch.manual_seed(seed)
    model = task.build_model(args).to(dev)
    return args, task, model


def speech_batch():
    """2 x 10 s; clip 1 padded from 8 s on, the way the reference's collaters pad (zeros in the waveform and the mel target,
    stop label 1 from the last valid frame on: data/speech_to_speech_dataset.py / speech_dataset.py collater)."""
    from speecht5_amd.synthetic import speech_pretrain_sample
    s = speech_pretrain_sample(B=2, seconds=10.0, device="cpu", seed=11)
    ni = s["net_input"]
    valid = 128000
    ni["source"][1, valid:] = 0.0
    ni["padding_mask"][1, valid:] = True
    L1 = 1 + valid // 256                      # 501 mel frames -> 500 after the reduction-factor trim
    L1 -= L1 % 2
    s["dec_target"][1, L1:] = 0.0
    s["dec_target_lengths"][1] = L1
    s["labels"][1] = 0.0
    s["labels"][1, L1 - 1:] = 1.0
    ni["prev_output_tokens"][1, L1 // 2:] = 0.0
    ni["tgt_lengths"][1] = L1 // 2
    s["src_lengths"] = [160000, valid]
    T = 499
    Tv = 399                                    # conv_out_lengths(128000)
    mask = torch.zeros(2, T, dtype=torch.bool)
    g = torch.Generator().manual_seed(3)
    for b, lim in ((0, T), (1, Tv)):            # ~0.65 of the valid frames in 10-frame spans (mask_prob 0.8, overlaps allowed)
        for st in torch.randint(0, lim - 10, (int(0.08 * lim),), generator=g).tolist():
            mask[b, st:st + 10] = True
    mix_idx = torch.randperm(T, generator=g)[: T // 2].sort().values
    return s, mask, mix_idx


def text_batch(vocab, mask_idx):
    from speecht5_amd.synthetic import text_pretrain_sample
    s = text_pretrain_sample(B=4, T=512, vocab=vocab, mask_idx=mask_idx, device="cpu", seed=12)
    ni = s["net_input"]
    n = 400
    s["target"][3, n - 1] = 2
    s["target"][3, n:] = 1
    ni["src_tokens"][3, n - 1] = 2
    ni["src_tokens"][3, n:] = 1
    ni["src_lengths"][3] = n
    ni["prev_output_tokens"][3, n:] = 1
    s["ntokens"] = int(s["target"].ne(1).sum())
    g = torch.Generator().manual_seed(4)
    mix_idx = torch.randperm(512, generator=g)[:256].sort().values
    return s, mix_idx


def oracle_run(model, args, speech, mask, mix_s, text, mix_t):
    """CPU oracle on the model's weights: per-micro-batch losses + the summed gradient of (loss_s/ss_s + loss_t/ss_t)."""
    from oracle import speecht5_oracle as O
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**vars(args))
    noise = torch.zeros(1)
    ro = O.forward_speech_pretrain(sd, cfg, speech, mask_indices=mask, mix_idx=mix_s, gumbel_noise=noise)
    ls, ss, logs = O.speech_pretrain_loss(ro, speech, cfg, loss_weights=(10, 0.1))
    (ls / ss).backward()
    del ro
    to = O.forward_text_pretrain(sd, cfg, text, mix_idx=mix_t, gumbel_noise=noise)
    lt, st, logt = O.text_pretrain_loss(to, text, loss_weights=(0.1,))
    (lt / st).backward()
    grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    # the text embedding is ONE parameter under three names (tied input / output embeddings): the oracle's state-dict entries
    # are independent leaves, so the parameter's gradient is the sum over its aliases
    from tests.util import TIED
    tied = [grads[a] for a in TIED if a in grads]
    if tied:
        total = sum(tied)
        for a in TIED:
            grads[a] = total
    return dict(ls=float(ls.detach()), ss=ss, lt=float(lt.detach()), st=st, logs={k: float(torch.as_tensor(v).detach()) for k, v in logs.items() if v is not None},
                bart=float(logt["bart_loss"].detach()), grads=grads)


def product_run(model, task, dev, speech, mask, mix_s, text, mix_t):
    from speecht5_amd.criterions import SpeechPretrainCriterion, TextPretrainCriterion
    from tests.util import injected_randomness, to_dev
    model.train()
    noise = torch.zeros(1)
    cs = SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1], sync_logging=False)
    ct = TextPretrainCriterion(task, False, 1.0, loss_weights=[0.1], sync_logging=False)
    with injected_randomness(model, mask, mix_s, noise, 2.0):
        ls, ss, logs = cs(model, to_dev(speech, dev))
    (ls / ss).backward()
    with injected_randomness(model, None, mix_t, noise, 2.0):
        lt, st, logt = ct(model, to_dev(text, dev))
    (lt / st).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return dict(ls=float(ls.detach()), ss=int(ss), lt=float(lt.detach()), st=int(st), logs=logs, logt=logt, grads=grads)


def test_cfg2_shape_padded_speech_and_text_match_oracle(cuda):
    from speecht5_amd import functional as Fn
    try:
        args, task, model = build(cuda, torch.float32)
        vocab = len(task.dicts["text"])
        speech, mask, mix_s = speech_batch()
        text, mix_t = text_batch(vocab, task.dicts["text"].index("<mask>"))
        ref = oracle_run(model, args, speech, mask, mix_s, text, mix_t)
        got = product_run(model, task, cuda, speech, mask, mix_s, text, mix_t)
        bad = []
        if got["ss"] != ref["ss"] or got["st"] != ref["st"]:
            bad.append(("sample sizes", got["ss"], ref["ss"], got["st"], ref["st"]))
        for k in ("ls", "lt"):
            if abs(got[k] - ref[k]) > 2e-4 * abs(ref[k]):
                bad.append((k, got[k], ref[k]))
        for k, rk in (("loss_m_0", "loss_m"), ("dec_loss", "dec_loss"), ("l1_loss", "l1"), ("l2_loss", "l2"), ("bce_loss", "bce")):
            a_, b_ = float(torch.as_tensor(got["logs"][k]).detach()), ref["logs"][rk]
            if abs(a_ - b_) > 5e-4 * max(abs(b_), 1e-3):
                bad.append((k, a_, b_))
        rnorm = sum(float(ref["grads"][n].double().pow(2).sum()) for n in got["grads"] if n in ref["grads"]) ** 0.5   # (one alias per tied weight)
        gnorm = sum(float(g.pow(2).sum()) for g in got["grads"].values()) ** 0.5
        if abs(gnorm - rnorm) > 5e-3 * rnorm:
            bad.append(("grad norm", gnorm, rnorm))
        worst, n_cmp = [], 0
        for n, g in got["grads"].items():
            r = ref["grads"].get(n)
            if r is None:
                continue
            r = r.double()
            if float(r.norm()) <= 1e-6 * rnorm:      # structurally ~zero gradients (softmax-shift-invariant key bias, unused heads)
                if float(g.norm()) > 1e-4 * rnorm:
                    bad.append((n, "expected ~0", float(g.norm())))
                continue
            rel = float((g - r).norm() / r.norm())
            cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
            n_cmp += 1
            worst.append((cos, rel, n))
            if rel > REL_TOL or cos < COS_TOL:
                bad.append((n, rel, cos))
        worst.sort()
        print(f"fp32 parity mode at the cfg-2 shape: speech loss {got['ls']:.6f} (oracle {ref['ls']:.6f}), text loss {got['lt']:.6f} "
              f"(oracle {ref['lt']:.6f}), grad norm {gnorm:.6f} (oracle {rnorm:.6f}), {n_cmp} parameters compared; worst:", worst[:5])
        assert n_cmp > 250, n_cmp
        assert not bad, bad[:12]

        # ---- bf16 compute mode: the kernels bench.py times (LDS-DMA GEMMs, fused attention forward + backward with tails) ----
        sdict = {k: v.detach().clone() for k, v in model.state_dict().items()}
        del model
        args2, task2, model2 = build(cuda, torch.bfloat16)
        torch.nn.Module.load_state_dict(model2, sdict, strict=True)
        got2 = product_run(model2, task2, cuda, speech, mask, mix_s, text, mix_t)
        assert abs(got2["ls"] - ref["ls"]) <= 3e-2 * abs(ref["ls"]), (got2["ls"], ref["ls"])
        assert abs(got2["lt"] - ref["lt"]) <= 3e-2 * abs(ref["lt"]), (got2["lt"], ref["lt"])
        g2norm = sum(float(g.pow(2).sum()) for g in got2["grads"].values()) ** 0.5
        assert abs(g2norm - rnorm) <= 3e-2 * rnorm, ("bf16 grad norm", g2norm, rnorm)
        worst2 = []
        for n, g in got2["grads"].items():
            r = ref["grads"].get(n)
            if r is None or float(r.double().norm()) <= 1e-6 * rnorm:
                continue
            r = r.double()
            rel = float((g - r).norm() / r.norm())
            cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
            worst2.append((cos, rel, n, tuple(g.shape)))
        worst2.sort()
        post = [w for w in worst2 if "speech_decoder_postnet.postnet" in w[2]]
        rest = [w for w in worst2 if "speech_decoder_postnet.postnet" not in w[2]]
        print("bf16 at the cfg-2 shape, worst outside the post-net:", rest[:8])
        print("bf16 at the cfg-2 shape, worst inside the post-net:", post[:4])
        bad2 = [w for w in rest if (w[0] < BF16_COS and w[2] not in BF16_LOOSE) or (w[1] > BF16_REL and len(w[3]) > 0 and w[2] not in BF16_LOOSE)] + \
               [w for w in rest if w[2] in BF16_LOOSE and w[0] < 0.997] + [w for w in post if w[0] < BF16_POST_COS]
        assert len(worst2) > 250
        assert not bad2, bad2[:10]
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()
