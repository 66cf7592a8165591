"""HiFi-GAN generator: the oracle restatement against the HF-generated golden (CPU) and the HIP
implementation against the same golden (GPU; fp32 1e-4, bf16 5e-2 of the waveform scale)."""
import os

import pytest
import torch

from oracle import speecht5_oracle as O
from tests.util import G, close


def _fx():
    return torch.load(os.path.join(G, "tiny_hifigan.pt"), weights_only=False)


def test_oracle_hifigan_matches_hf_golden():
    fx = _fx()
    wav = O.hifigan(fx["state_dict"], fx["config"], fx["spectrogram"])
    close(wav, fx["waveform"], 1e-5, what="hifigan oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_hip_hifigan_matches_hf_golden(cuda, dtype, tol):
    from speecht5_amd import functional as Fn
    from speecht5_amd.hifigan import SpeechT5HifiGan
    fx = _fx()
    Fn.set_compute_dtype(dtype)
    try:
        m = SpeechT5HifiGan(**{k: v for k, v in fx["config"].items()}).to(cuda).eval()
        m.load_state_dict(fx["state_dict"])
        wav = m(fx["spectrogram"].to(cuda))
        assert wav.shape == fx["waveform"].shape
        close(wav, fx["waveform"], tol, what=f"hifigan {dtype}")
        single = m(fx["spectrogram"][0].to(cuda))
        close(single, fx["waveform"][0], tol, what="hifigan unbatched")
    finally:
        Fn.set_compute_dtype(torch.float32)
