"""Golden fixture for the `extractor_mode=layer_norm` feature extractor (SpeechT5-Large recipe: conv bias + LayerNorm
after every conv, speech_encoder_prenet.py:290-354), produced by the VERBATIM reference module.

TEST INFRASTRUCTURE ONLY; needs /root/reference (build container).  Writes tests/golden/tiny_extractor_ln.pt:
conv layer spec, state dict, waveform, output [B,C,T], the top gradient used and the parameter gradients.

    python oracle/make_golden_extractor_ln.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    ref_stubs.load_reference_models()
    import importlib
    mod = importlib.import_module("speecht5.models.modules.speech_encoder_prenet")
    layers = [(32, 10, 5)] + [(32, 3, 2)] * 4 + [(32, 2, 2)] * 2
    torch.manual_seed(21)
    fx = mod.ConvFeatureExtractionModel(conv_layers=layers, dropout=0.0, mode="layer_norm", conv_bias=True)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for n, p in fx.named_parameters():   # non-trivial biases / LayerNorm affine parameters
            if n.endswith("bias") or ".2.1." in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    wav = torch.randn(2, 4000, generator=g)
    out = fx(wav)                                   # [B, C, T]
    top = torch.randn(out.shape, generator=g)
    out.backward(top)
    torch.save(dict(layers=layers, state_dict={k: v.detach().clone() for k, v in fx.state_dict().items()}, wav=wav,
                    out=out.detach(), top=top, grads={n: p.grad.clone() for n, p in fx.named_parameters()}),
               os.path.join(OUT, "tiny_extractor_ln.pt"))
    print("extractor layer_norm:", tuple(out.shape), "params", len(list(fx.parameters())))


if __name__ == "__main__":
    main()
