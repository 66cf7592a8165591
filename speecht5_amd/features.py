"""Log-mel filterbank features on the GPU: the `logmelfilterbank` of the reference input pipeline
(/root/reference/SpeechT5/speecht5/data/speech_dataset.py:142-181, called per item from `__getitem__` :249-260 through
librosa on the host) as two fp32 MFMA GEMMs and three small kernels of the C ABI:

    frames  = st5_stft_frames(wav)                      [B*L, 1024]   centred, reflect-padded frames (L = 1 + S // 256)
    re|im   = st5_gemm(frames, window * DFT basis)      [B*L, 2*520]  rows of the basis: w[k] cos / -w[k] sin (2 pi n k / 1024)
    mag     = st5_stft_magnitude(re|im)                 [B*L, 520]
    mel     = st5_gemm(mag, mel filterbank)             [B*L, 80]     Slaney scale + Slaney area normalisation, 80-7600 Hz
    logmel  = st5_log10_floor(mel, 1e-10)

The Hann window is folded into the DFT basis, so the framing kernel is a pure gather.  Everything is fp32 (`st5_gemm` in
its exact-fp32 mode): features feed the decoder targets and the MSE/L1 losses at full precision in the reference too.
"""
import math

import numpy as np
import torch

from . import hip


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / math.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (math.log(6.4) / 27.0)), m * 200.0 / 3.0)


def slaney_mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax):
    """[num_mels, 1 + fft_size // 2] triangular filters on the Slaney mel scale, each scaled to unit area
    (what `librosa.filters.mel` returns with its defaults, speech_dataset.py:179)."""
    freqs = np.linspace(0.0, sampling_rate / 2.0, 1 + fft_size // 2)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), num_mels + 2))
    width = np.diff(edges)
    up = (freqs[None, :] - edges[:-2, None]) / width[:-1, None]
    down = (edges[2:, None] - freqs[None, :]) / width[1:, None]
    tri = np.maximum(0.0, np.minimum(up, down))
    return tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]


class LogMelFilterBank:
    """wav fp32 [B, S] on the GPU -> log10-mel fp32 [B, 1 + S // hop, num_mels]."""

    def __init__(self, device, sampling_rate=16000, fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600, eps=1e-10):
        assert fft_size % 32 == 0 and num_mels % 8 == 0
        self.n_fft, self.hop, self.num_mels, self.eps = fft_size, hop_size, num_mels, float(eps)
        self.nbins = 1 + fft_size // 2
        self.ldh = (self.nbins + 7) // 8 * 8
        fmin = 0 if fmin is None else fmin
        fmax = sampling_rate / 2 if fmax is None else fmax
        k = np.arange(fft_size)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / fft_size)                    # periodic Hann (scipy get_window("hann"))
        ang = 2.0 * np.pi * np.outer(np.arange(self.nbins), k) / fft_size
        basis = np.zeros((2 * self.ldh, fft_size))
        basis[:self.nbins] = np.cos(ang) * win
        basis[self.ldh:self.ldh + self.nbins] = -np.sin(ang) * win
        mel = np.zeros((num_mels, self.ldh))
        mel[:, :self.nbins] = slaney_mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax)
        self.basis = torch.from_numpy(basis.astype(np.float32)).to(device)
        self.mel = torch.from_numpy(mel.astype(np.float32)).to(device)

    def __call__(self, wav):
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2, "wav: fp32 [B, S] on the GPU"
        wav = wav.contiguous()
        B, S = wav.shape
        if S <= self.n_fft // 2:
            raise ValueError(f"reflect padding needs more than {self.n_fft // 2} samples, got {S}")
        L = 1 + S // self.hop
        M = B * L
        dev = wav.device
        lib, st = hip.lib(), hip.stream()
        frames = torch.empty(M, self.n_fft, dtype=torch.float32, device=dev)
        hip.check(lib.st5_stft_frames(wav.data_ptr(), frames.data_ptr(), B, S, self.n_fft, self.hop, st), "st5_stft_frames")
        reim = torch.empty(M, 2 * self.ldh, dtype=torch.float32, device=dev)
        hip.gemm(hip.operand(frames, self.n_fft), hip.operand(self.basis, self.n_fft), hip.operand(reim, 2 * self.ldh),
                 M, 2 * self.ldh, self.n_fft, hip.F32)
        mag = torch.empty(M, self.ldh, dtype=torch.float32, device=dev)
        hip.check(lib.st5_stft_magnitude(reim.data_ptr(), mag.data_ptr(), M, self.nbins, self.ldh, st), "st5_stft_magnitude")
        mel = torch.empty(M, self.num_mels, dtype=torch.float32, device=dev)
        hip.gemm(hip.operand(mag, self.ldh), hip.operand(self.mel, self.ldh), hip.operand(mel, self.num_mels),
                 M, self.num_mels, self.ldh, hip.F32)
        out = torch.empty_like(mel)
        hip.check(lib.st5_log10_floor(mel.data_ptr(), out.data_ptr(), mel.numel(), self.eps, st), "st5_log10_floor")
        return out.view(B, L, self.num_mels)
