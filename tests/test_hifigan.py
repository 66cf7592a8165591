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


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 5e-2)])
def test_hip_hifigan_full_size_matches_the_installed_hf_module(cuda, dtype, tol):
    """The FULL-SIZE generator (HuggingFace `SpeechT5HifiGanConfig()` defaults: 512 initial channels, rates 4-4-4-4, kernels
    3/7/11, dilations 1/3/5 -- what `microsoft/speecht5_hifigan` is): same seeded weights in the installed HF module on the CPU
    (fp32) and in the HIP implementation; a batch of two clips of different content, 48 frames each, plus the unbatched call.
    Exercises every channel width (256 ... 32, the vectorised pad + LeakyReLU pass, convolutions writing straight into the next
    one's padded input) at its real size."""
    transformers = pytest.importorskip("transformers")
    from speecht5_amd import functional as Fn
    from speecht5_amd.hifigan import SpeechT5HifiGan
    torch.manual_seed(11)
    hf = transformers.SpeechT5HifiGan(transformers.SpeechT5HifiGanConfig()).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.normal_(0.0, 0.03)
        hf.conv_post.weight.normal_(0.0, 0.3)      # (a waveform that is not mostly tanh(bias))
        hf.mean.normal_(0.0, 0.5)
        hf.scale.uniform_(0.5, 1.5)
    mel = torch.randn(2, 48, 80) * 0.8 - 0.5
    with torch.no_grad():
        ref = hf(mel)
    Fn.set_compute_dtype(dtype)
    try:
        m = SpeechT5HifiGan().to(cuda).eval()
        missing, unexpected = m.load_state_dict(hf.state_dict(), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        wav = m(mel.to(cuda))
        assert wav.shape == ref.shape == (2, 48 * 256)
        assert float(ref.std()) > 0.05
        close(wav, ref, tol, what=f"full-size hifigan {dtype}")
        close(m(mel[1].to(cuda)), ref[1], tol, what="full-size hifigan unbatched")
    finally:
        Fn.set_compute_dtype(torch.float32)
