"""Whole-model parity on the GPU: our T5TransformerModel (HIP kernels, fp32 parity mode) against the golden
outputs of the verbatim reference (tests/golden/*.pt) on identical weights, inputs and random draws, plus
the CPU oracle at the same point.  Tolerances: forward 2e-4 / gradients 2e-3 relative to the tensor scale
in fp32 (MFMA fp32 accumulation order differs from the CPU's); bf16 compute is sanity-checked at 6e-2."""
import pytest
import torch
import torch.nn.functional as F

from oracle import speecht5_oracle as O
from tests.util import check_grads, close, build_tiny, injected_randomness, load_golden, to_dev

pytestmark = pytest.mark.gpu


def _speech_pretrain(cuda, dtype):
    from speecht5_amd import functional as Fn
    _, fx = load_golden("tiny_speech_pretrain.pt")
    model, args = build_tiny(cuda, dtype)
    model.train()
    sample = to_dev(fx["sample"], cuda)
    with injected_randomness(model, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        net_output, dec = model(target_list=sample["target_list"], **sample["net_input"])
    before, after, logits, attn = dec
    out = dict(logit_m=net_output["logit_m_list"][0], logit_u=net_output["logit_u_list"][0], features_pen=net_output["features_pen"],
               prob_perplexity=net_output["prob_perplexity"], code_perplexity=net_output["code_perplexity"], num_vars=net_output["num_vars"],
               before=before, after=after, stop_logits=logits, attn=attn)
    return model, args, fx, sample, out


def test_speech_pretrain_fp32_matches_reference(cuda):
    model, args, fx, sample, out = _speech_pretrain(cuda, torch.float32)
    for k in ("logit_m", "logit_u", "features_pen", "prob_perplexity", "code_perplexity", "before", "after", "stop_logits", "attn"):
        close(out[k], fx["out"][k], 2e-4, what=k)
    cpu = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    # loss through the oracle's criterion arithmetic (fp32 torch on our outputs), gradient through our kernels
    loss, ss, _ = O.speech_pretrain_loss(out, sample_to(sample, out["before"].device), args, loss_weights=(10, 0.1))
    close(loss, fx["loss"], 2e-4, what="loss")
    (loss / ss).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)


def sample_to(sample, device):
    return to_dev(sample, device)


def test_speech_pretrain_bf16_is_close(cuda):
    model, args, fx, sample, out = _speech_pretrain(cuda, torch.bfloat16)
    for k in ("before", "after", "stop_logits", "attn", "features_pen"):
        close(out[k], fx["out"][k], 6e-2, what=k)
    loss, ss, _ = O.speech_pretrain_loss(out, sample, args, loss_weights=(10, 0.1))
    close(loss, fx["loss"], 5e-2, what="loss")
    (loss / ss).backward()
    torch.cuda.synchronize()
    g = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert len(g) > 100 and all(torch.isfinite(v).all() for v in g.values())
    from speecht5_amd import functional as Fn
    Fn.set_compute_dtype(torch.float32)


def test_text_pretrain_fp32_matches_reference(cuda):
    _, fx = load_golden("tiny_text_pretrain.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.train()
    sample = to_dev(fx["sample"], cuda)
    with injected_randomness(model, None, fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        (logits, _), codebook_out, _ = model(**sample["net_input"])
    out = dict(logits=logits, **{k: v for k, v in codebook_out.items()})
    loss, ss, _ = O.text_pretrain_loss(out, sample, loss_weights=(0.1,))
    close(loss, fx["loss"], 2e-4, what="loss")
    (loss / ss).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)


def test_s2t_fp32_matches_reference_and_greedy_ids_bit_exact(cuda):
    _, fx = load_golden("tiny_s2t.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.speech_encoder_prenet.mask_prob = 0.5
    model.train()
    sample = to_dev(fx["sample"], cuda)
    with injected_randomness(model, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        (logits, _), enc = model(**sample["net_input"])
    loss, ss, _ = O.s2t_loss(dict(logits=logits, encoder_out=enc), sample, args, ce_weight=0.5, ctc_weight=0.5,
                             label_smoothing=0.1, blank_idx=fx["blank_idx"])
    close(loss, fx["loss"], 2e-4, what="loss")
    (loss / ss).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)
    # inference: logits and greedy token ids (the reference's forward_encoder / forward_decoder API)
    model.eval()
    ni = sample["net_input"]
    with torch.no_grad():
        enc = model.forward_encoder(ni["source"], ni["padding_mask"])
        dec_logits, _ = model.forward_decoder(ni["prev_output_tokens"], enc, None)
        close(dec_logits, fx["eval_logits"], 2e-4, what="eval logits")
        B = ni["source"].shape[0]
        tokens = torch.full((B, 1), 2, dtype=torch.long, device=cuda)
        done = torch.zeros(B, dtype=torch.bool, device=cuda)
        inc = {}   # one incremental state for the whole decode (KV cache), as fairseq's SequenceGenerator keeps it
        for step in range(fx["max_len"]):
            out, _ = model.forward_decoder(tokens, enc, inc)
            lp = torch.log_softmax(out[:, -1].float(), -1)
            for f in fx["forbid"]:
                lp[:, f] = -float("inf")
            if step == fx["max_len"] - 1:
                lp[:, :2] = -float("inf")
                lp[:, 3:] = -float("inf")
            nxt = lp.argmax(-1)
            nxt = torch.where(done, torch.ones_like(nxt), nxt)
            tokens = torch.cat([tokens, nxt[:, None]], 1)
            done |= nxt.eq(2)
            if bool(done.all()):
                break
    assert tokens[:, 1:].cpu().tolist() == fx["greedy_tokens"].tolist()  # bit-exact token ids (BASELINE.json)
    # the KV-cache path reproduces the full (teacher-forced) forward position by position, and survives a beam re-order
    with torch.no_grad():
        prev = ni["prev_output_tokens"]
        full, _ = model.forward_decoder(prev, enc, None)
        inc = {}
        for t in range(prev.shape[1]):
            step_out, _ = model.forward_decoder(prev[:, : t + 1], enc, inc)
            close(step_out[:, 0], full[:, t], 2e-4, what=f"cached step {t}")
            if t == 2:   # swap the two hypotheses' caches and swap them back: a no-op overall
                order = torch.tensor([1, 0], device=cuda)
                model.decoder.reorder_incremental_state_scripting(inc, order)
                model.decoder.reorder_incremental_state_scripting(inc, order)


def test_t2s_fp32_matches_reference_and_generated_mel(cuda):
    _, fx = load_golden("tiny_t2s.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.use_codebook = False
    model.train()
    sample = to_dev(fx["sample"], cuda)
    before, after, logits, attn = model(**sample["net_input"])
    for got, ref, name in ((before, fx["out"]["before"], "before"), (after, fx["out"]["after"], "after"),
                           (logits, fx["out"]["stop_logits"], "stop_logits")):
        close(got, ref, 2e-4, what=name)
    for a, b in zip(attn, fx["out"]["attn"]):
        close(a, b, 2e-4, what="attn")
    loss, l1, _, bce = O.tacotron_loss(before, after, logits, sample, args.reduction_factor)
    att = torch.cat([a[:, :2] for a in attn], dim=1)
    # guided-attention loss evaluated on device so that its gradient flows into our attention kernels
    W = _guided_weights(fx["sample"]["src_lengths"], fx["sample"]["net_input"]["tgt_lengths"], att.shape, cuda)
    ga_dev = 10.0 * torch.mean((W[0].unsqueeze(1) * att).masked_select(W[1].unsqueeze(1)))
    close(ga_dev, fx["guided"], 2e-4, what="guided attention loss")
    close(loss + ga_dev, fx["loss"], 2e-4, what="loss")
    (loss + ga_dev).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)
    # the training forward updated the BatchNorm running statistics: restore the stored weights (as the
    # golden generator does) before synthesis
    m, _ = load_golden(None)
    torch.nn.Module.load_state_dict(model, m["state_dict"], strict=True)
    model.eval()
    n = int(fx["sample"]["net_input"]["src_lengths"][0])
    mel, _, _ = model.generate_speech(src_tokens=sample["net_input"]["src_tokens"][:1, :n], spkembs=sample["net_input"]["spkembs"][:1])
    close(mel, fx["generated_mel"], 1e-3, what="generated mel (TTS parity bar: 1e-3 relative)")


def _guided_weights(ilens, olens, shape, device):
    B, _, To, Ti = shape
    gm = torch.zeros(B, To, Ti)
    mk = torch.zeros(B, To, Ti, dtype=torch.bool)
    for b in range(B):
        il, ol = int(ilens[b]), int(olens[b])
        gx, gy = torch.meshgrid(torch.arange(ol).float(), torch.arange(il).float(), indexing="ij")
        gm[b, :ol, :il] = 1.0 - torch.exp(-((gy / il - gx / ol) ** 2) / (2 * 0.4 ** 2))
        mk[b, :ol, :il] = True
    return gm.to(device), mk.to(device)
