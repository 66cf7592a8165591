"""Full-architecture parity: SpeechT5-Base (t5_transformer_base: 12 encoder + 6 decoder layers, d = 768, 12 heads of 64,
FFN 3072, the 7-layer conv front end, rel-pos keys +-160, code book, mel decoder with post-net) on one 2 s clip of the cfg-2
speech micro-batch, fp32 parity mode through the C ABI, against the CPU oracle on the same random-initialised weights, inputs
and random draws: loss, every loss term, gradient norm and a set of per-parameter gradients.  The tiny golden fixtures pin
the oracle to the reference; this test carries the comparison to the real layer shapes (head_dim 64, 3072-wide FFN, 512-channel
strided convolutions, 320-bucket position table).  bf16 compute mode (fused attention kernels) is checked for closeness."""
from argparse import Namespace
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

REL_TOL, COS_TOL = 1e-3, 0.99999   # measured on MI355X: relative Frobenius error < 5e-6, cosine 1.0 for all ten tensors


def _build(dev, dtype):
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
    torch.manual_seed(4242)
    model = task.build_model(args).to(dev)
    return args, task, model


def test_base_architecture_speech_pretrain_matches_oracle(cuda):
    from oracle import speecht5_oracle as O
    from speecht5_amd import functional as Fn
    from speecht5_amd.criterions import SpeechPretrainCriterion
    from speecht5_amd.synthetic import speech_pretrain_sample
    from tests.util import injected_randomness, to_dev
    try:
        args, task, model = _build(cuda, torch.float32)
        model.train()
        sample = speech_pretrain_sample(B=1, seconds=2.0, device="cpu", seed=7)
        T = int(2.0 * 50) - 1
        mask = torch.zeros(1, T, dtype=torch.bool)
        mask[:, 10:70] = True
        # the model mixes randperm(T)[:int(T * codebook_prob)] code-book positions (speecht5.py:866): hand both sides exactly that many
        mix_idx = torch.arange(0, T, 2)[: int(T * getattr(args, "codebook_prob", 0.5))]
        noise = torch.zeros(1)                                      # Gumbel noise replaced by zeros on both sides
        crit = SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1], sync_logging=False)
        with injected_randomness(model, mask, mix_idx, noise, 2.0):
            loss, ss, log = crit(model, to_dev(sample, cuda))
        (loss / ss).backward()
        torch.cuda.synchronize()
        got = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
        gnorm = sum(float(g.double().pow(2).sum()) for g in got.values()) ** 0.5

        sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        cfg = SimpleNamespace(**vars(args))
        ref = O.forward_speech_pretrain(sd, cfg, sample, mask_indices=mask, mix_idx=mix_idx, gumbel_noise=noise)
        rl, rs, rlog = O.speech_pretrain_loss(ref, sample, cfg, loss_weights=(10, 0.1))
        (rl / rs).backward()
        bad = []
        if ss != rs:
            bad.append(("sample_size", ss, rs))
        lv, rv = float(loss.detach()), float(rl.detach())
        if abs(lv - rv) > 2e-4 * abs(rv):
            bad.append(("loss", lv, rv))
        for k, rk in (("loss_m_0", "loss_m"), ("dec_loss", "dec_loss"), ("l1_loss", "l1"), ("l2_loss", "l2"), ("bce_loss", "bce")):
            a_, b_ = float(torch.as_tensor(log[k]).detach()), float(torch.as_tensor(rlog[rk]).detach())
            if abs(a_ - b_) > 5e-4 * max(abs(b_), 1e-3):
                bad.append((k, a_, b_))
        rnorm = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.requires_grad and v.grad is not None) ** 0.5
        if abs(gnorm - rnorm) > 5e-3 * rnorm:
            bad.append(("grad norm", gnorm, rnorm))
        names = ["encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.11.fc1.weight", "encoder.layers.5.self_attn_layer_norm.weight",
                 "encoder.pos_emb.pe_k.weight", "decoder.layers.0.encoder_attn.k_proj.weight", "decoder.layers.5.fc2.bias",
                 "speech_encoder_prenet.feature_extractor.conv_layers.3.0.weight", "speech_encoder_prenet.post_extract_proj.weight",
                 "speech_decoder_postnet.feat_out.weight", "hubert_layer.final_proj.weight"]
        checked = 0
        report = []
        for n in names:
            if n in got and n in sd and sd[n].grad is not None:
                r = sd[n].grad.double()
                g = got[n].double()
                rel = float((g - r).norm() / r.norm().clamp_min(1e-12))
                cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
                report.append((n, f"{rel:.2e}", round(cos, 7)))
                if rel > REL_TOL or cos < COS_TOL:
                    bad.append((n, rel, cos))
                checked += 1
        if checked < 8:
            bad.append(("parameters compared", checked, [n for n in names if n not in got or n not in sd]))
        print("per-parameter gradient agreement (relative Frobenius error, cosine):", report)
        assert not bad, (bad, report)
        fp32_loss = lv
        # bf16 compute mode on the same weights: fused attention / bf16 MFMA GEMMs, closeness only
        del model
        args2, task2, model2 = _build(cuda, torch.bfloat16)
        model2.train()
        crit2 = SpeechPretrainCriterion(task2, False, 1.0, 0.0, loss_weights=[10, 0.1], sync_logging=False)
        with injected_randomness(model2, mask, mix_idx, noise, 2.0):
            l2, s2, _ = crit2(model2, to_dev(sample, cuda))
        assert abs(float(l2.detach()) - fp32_loss) <= 3e-2 * abs(fp32_loss), (float(l2.detach()), fp32_loss)
    finally:
        Fn.set_compute_dtype(torch.float32)
        Fn.weight_cache.clear()
