"""extractor_mode=layer_norm feature extractor (SpeechT5-Large recipe) against the fixture produced by the verbatim
reference module (oracle/make_golden_extractor_ln.py): the CPU oracle on CPU, the HIP path on the GPU."""
import os
import sys
from argparse import Namespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "tiny_extractor_ln.pt")


def test_oracle_extractor_layer_norm_matches_reference():
    from oracle import speecht5_oracle as O
    fx = torch.load(GOLD)
    sd = {"fe." + k: v.clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    cfg = Namespace(conv_feature_layers=repr(fx["layers"]), extractor_mode="layer_norm")
    out = O.conv_feature_extractor(sd, "fe.", fx["wav"], cfg)
    assert torch.allclose(out, fx["out"], rtol=1e-5, atol=1e-5)
    out.backward(fx["top"])
    for n, g in fx["grads"].items():
        got = sd["fe." + n].grad
        assert torch.allclose(got, g, rtol=2e-4, atol=2e-5), n


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_extractor_layer_norm_matches_reference(dtype):
    from speecht5_amd import functional as Fn
    from speecht5_amd.modules.speech_encoder_prenet import ConvFeatureExtractionModel
    dev = torch.device("cuda:0")
    fx = torch.load(GOLD)
    prev = Fn._S.dtype
    Fn.set_compute_dtype(dtype)
    try:
        m = ConvFeatureExtractionModel(fx["layers"], dropout=0.0, mode="layer_norm", conv_bias=True)
        missing, unexpected = m.load_state_dict(fx["state_dict"], strict=True)
        m = m.to(dev)
        out = m(fx["wav"].to(dev))                      # [B, T, C]
        ref = fx["out"].transpose(1, 2)
        tol = 2e-4 if dtype == torch.float32 else 6e-2
        err = (out.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < tol, f"forward rel err {err}"
        out.backward(fx["top"].transpose(1, 2).to(dev).to(out.dtype))
        gtol = 2e-3 if dtype == torch.float32 else 1e-1
        for n, p in m.named_parameters():
            g = fx["grads"][n]
            e = (p.grad.float().cpu() - g).abs().max().item() / max(g.abs().max().item(), 1e-6)
            assert e < gtol, f"{n}: grad rel err {e}"
    finally:
        Fn.set_compute_dtype(prev)
