"""Generates tests/golden/tiny_large_speech_pretrain.pt: the VERBATIM reference (microsoft/SpeechT5 under /root/reference,
imported through oracle/ref_stubs.py) built with the `t5_transformer_large` architecture function
(SpeechT5/speecht5/models/speecht5.py:1402-1425: pre-LN encoder layers with `pos_bias = norm_k(pos_bias)`,
transformer_layer.py:90-111; `extractor_mode=layer_norm` conv front end, speech_encoder_prenet.py:290-347; pre-LN decoder with
a final LayerNorm, decoder.py) at tiny dimensions, run on a seeded speech_pretrain batch through the reference criterion.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).  The fixture is self-contained (args,
state dict, inputs, recorded random draws, outputs, loss, gradients).

    python oracle/make_golden_large.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402
from make_golden import OUT, Recorder, Task, build, grads_of, speech_batch, tiny_args  # noqa: E402


def main():
    ref = ref_stubs.load_reference_models()
    crit = ref_stubs.load_reference_criterions()
    # tiny dimensions; everything the Large architecture function decides is left to it (layer_norm_first,
    # decoder_normalize_before, extractor_mode=layer_norm, conv_bias default, dropouts 0)
    args = tiny_args(arch="t5_transformer_large", encoder_layers=3, decoder_layers=2)
    task = Task(vocab=30, n_units=20)
    model = build(ref, args, "t5_transformer_large", task, seed=23)
    assert args.layer_norm_first and args.decoder_normalize_before and args.extractor_mode == "layer_norm"
    model.train()
    sample = speech_batch(args, B=2, S=6400, n_units=20, pad_last=0, seed=9)
    c = crit.speech_pretrain.SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1])
    np.random.seed(6)
    torch.manual_seed(6)
    with Recorder(ref) as r:
        net_output, net_output_dec = model(target_list=sample["target_list"], **sample["net_input"])
    rec = dict(r.rec)
    np.random.seed(6)
    torch.manual_seed(6)
    with Recorder(ref):
        loss, sample_size, log = c(model, sample)
    (loss / sample_size).backward()
    before, after, logits, attn = net_output_dec
    n_mix = int(rec["randperm"].numel() * args.codebook_prob)
    fx = dict(
        args=vars(args), state_dict={k: v.detach().clone() for k, v in model.state_dict().items()},
        sample=sample,
        mask_indices=rec["mask_indices"], mix_idx=rec["randperm"][:n_mix], gumbel_noise=rec["gumbel_noise"], tau=rec["tau"],
        out=dict(logit_m=net_output["logit_m_list"][0].detach(), logit_u=net_output["logit_u_list"][0].detach(),
                 features_pen=net_output["features_pen"].detach(), prob_perplexity=net_output["prob_perplexity"].detach(),
                 before=before.detach(), after=after.detach(), stop_logits=logits.detach(), attn=attn.detach()),
        loss=loss.detach(), sample_size=sample_size, log={k: v for k, v in log.items() if isinstance(v, (int, float))},
        grads=grads_of(model))
    torch.save(fx, os.path.join(OUT, "tiny_large_speech_pretrain.pt"))
    print("large-style speech_pretrain: loss", float(loss), "sample_size", sample_size, "grads", len(fx["grads"]["norms"]),
          "norm_k grad", fx["grads"]["norms"].get("encoder.layers.0.norm_k.weight"))


if __name__ == "__main__":
    main()
