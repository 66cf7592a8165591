"""Log-mel front end (reference data/speech_dataset.py:142-181).  CPU: the numpy oracle and the product's filterbank /
basis construction against the HuggingFace-generated golden (oracle/make_golden_logmel.py).  GPU: the HIP path
(speecht5_amd.features.LogMelFilterBank, through the C ABI) against oracle and golden, incl. ragged lengths."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import logmel_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "logmel_hf.pt")


def test_oracle_matches_independent_implementation():
    g = torch.load(GOLD)
    for it in g["items"]:
        mine = O.logmelfilterbank(it["wav"].numpy())
        ref = it["logmel"].numpy()
        assert mine.shape == ref.shape == (1 + it["wav"].numel() // 256, 80)
        assert np.abs(mine - ref).max() < 5e-6                      # float32 fixture vs float64 oracle


def test_product_filterbank_equals_oracle():
    from speecht5_amd.features import slaney_mel_filterbank
    a = slaney_mel_filterbank(16000, 1024, 80, 80, 7600)
    b = O.mel_basis(16000, 1024, 80, 80.0, 7600.0)
    assert a.shape == b.shape == (80, 513)
    assert np.abs(a - b).max() < 1e-12
    assert (a >= 0).all() and (a.sum(1) > 0).all()
    # Slaney normalisation: every triangle has (continuous) area 1 Hz^-1 * Hz => sum * bin width ~ 1
    assert np.allclose(a.sum(1) * (8000.0 / 512), 1.0, atol=0.05)


def test_product_dft_basis_is_the_windowed_rfft():
    """Host logic of features.LogMelFilterBank (no GPU): frames @ basis^T = [Re | Im] of rfft(frames * periodic Hann), the
    padding columns of the basis / filterbank are zero, and basis + filterbank + log10 reproduce the oracle on CPU."""
    from speecht5_amd.features import LogMelFilterBank
    fb = LogMelFilterBank(torch.device("cpu"))
    assert fb.basis.shape == (2 * 520, 1024) and fb.mel.shape == (80, 520)
    rng = np.random.default_rng(0)
    frames = rng.standard_normal((7, 1024))
    reim = frames @ fb.basis.double().numpy().T
    ref = np.fft.rfft(frames * O.hann_periodic(1024), axis=1)
    assert np.abs(reim[:, :513] - ref.real).max() < 1e-4 and np.abs(reim[:, 520:520 + 513] - ref.imag).max() < 1e-4
    assert np.all(reim[:, 513:520] == 0) and np.all(reim[:, 520 + 513:] == 0)
    assert torch.all(fb.mel[:, 513:] == 0)
    # the whole pipeline with numpy matmuls in place of st5_gemm
    wav = (0.2 * rng.standard_normal(3000)).astype(np.float32)
    x = np.pad(wav.astype(np.float64), (512, 512), mode="reflect")
    fr = np.stack([x[i * 256:i * 256 + 1024] for i in range(1 + len(wav) // 256)])
    ri = fr @ fb.basis.double().numpy().T
    mag = np.sqrt(ri[:, :520] ** 2 + ri[:, 520:] ** 2)
    out = np.log10(np.maximum(mag @ fb.mel.double().numpy().T, 1e-10))
    assert np.abs(out - O.logmelfilterbank(wav)).max() < 1e-4


@pytest.mark.gpu
def test_gpu_logmel_matches_oracle_and_golden(cuda):
    from speecht5_amd.features import LogMelFilterBank
    fb = LogMelFilterBank(cuda)
    g = torch.load(GOLD)
    for it in g["items"]:
        out = fb(it["wav"][None].to(cuda))[0].cpu().numpy()
        ref = it["logmel"].numpy()
        orc = O.logmelfilterbank(it["wav"].numpy())
        assert out.shape == ref.shape
        # tolerance: fp32 DFT of 1024 points (abs error ~1e-5 of the frame energy) seen through log10 near the floor
        assert np.abs(out - orc).max() < 2e-3, np.abs(out - orc).max()
        assert np.abs(out - ref).max() < 2e-3


@pytest.mark.gpu
def test_gpu_logmel_batch_and_edges(cuda):
    from speecht5_amd.features import LogMelFilterBank
    fb = LogMelFilterBank(cuda)
    torch.manual_seed(5)
    wav = torch.randn(3, 16000 * 2 + 77) * 0.1
    out = fb(wav.to(cuda)).cpu().numpy()
    assert out.shape == (3, 1 + wav.shape[1] // 256, 80)
    for b in range(3):
        assert np.abs(out[b] - O.logmelfilterbank(wav[b].numpy())).max() < 2e-3
    # silence hits the floor exactly: log10(1e-10) = -10
    z = fb(torch.zeros(1, 4000, device=cuda)).cpu().numpy()
    assert np.allclose(z, -10.0)
    # shortest legal input (reflect padding needs > n_fft / 2 samples), and one sample less must raise
    s = fb(torch.randn(1, 513, device=cuda) * 0.1)
    assert s.shape == (1, 3, 80)
    with pytest.raises(ValueError):
        fb(torch.randn(1, 512, device=cuda))
