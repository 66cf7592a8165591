"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md 2): CPU restatement, in numpy float64, of the log-mel feature
extraction of the reference input pipeline -- `logmelfilterbank` in /root/reference/SpeechT5/speecht5/data/speech_dataset.py:142-183:

    x_stft = librosa.stft(audio, n_fft=1024, hop_length=256, win_length=None, window="hann", pad_mode="reflect")   # :172-173
    spc = np.abs(x_stft).T                                                                                          # :174
    mel_basis = librosa.filters.mel(sr, n_fft, n_mels=80, fmin=80, fmax=7600)                                       # :179
    return np.log10(np.maximum(eps, np.dot(spc, mel_basis.T)))                                                      # :181

librosa is third party and absent from this image (un-vendored, unpinned: SpeechT5/README.md:32), so `stft` (centred,
reflect padding, periodic Hann window, frames = 1 + len // hop) and `filters.mel` (Slaney mel scale, Slaney area
normalisation: librosa's defaults htk=False, norm="slaney") are restated from their published definitions.  Pinned
against an independent implementation of the same features that IS installed: HuggingFace's SpeechT5FeatureExtractor
(oracle/make_golden_logmel.py -> tests/golden/logmel_hf.pt, checked by tests/test_logmel.py).
"""
import numpy as np


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=16000, n_fft=1024, n_mels=80, fmin=80.0, fmax=7600.0):
    """librosa.filters.mel(htk=False, norm='slaney'): [n_mels, 1 + n_fft/2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def logmelfilterbank(audio, sampling_rate=16000, fft_size=1024, hop_size=256, num_mels=80, fmin=80, fmax=7600, eps=1e-10):
    """speech_dataset.py:142-181.  audio [S] -> [1 + S // hop, num_mels] float64."""
    audio = np.asarray(audio, dtype=np.float64)
    pad = fft_size // 2
    x = np.pad(audio, (pad, pad), mode="reflect")
    n_frames = 1 + len(audio) // hop_size
    win = hann_periodic(fft_size)
    frames = np.stack([x[i * hop_size:i * hop_size + fft_size] * win for i in range(n_frames)])
    spc = np.abs(np.fft.rfft(frames, n=fft_size, axis=1))
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    mb = mel_basis(sampling_rate, fft_size, num_mels, fmin, fmax)
    return np.log10(np.maximum(eps, spc @ mb.T))
