"""Generates tests/golden/logmel_hf.pt: log-mel features of seeded waveforms from an implementation that is independent of
oracle/logmel_oracle.py -- HuggingFace transformers' SpeechT5FeatureExtractor (the installed port of the same recipe:
1024-point centred STFT, hop 256, periodic Hann, 80 Slaney mel filters 80-7600 Hz, log10 with floor 1e-10).  Run in the
build container:  python oracle/make_golden_logmel.py"""
import os

import numpy as np
import torch
from transformers import SpeechT5FeatureExtractor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    fe = SpeechT5FeatureExtractor(feature_size=1, sampling_rate=16000, num_mel_bins=80, hop_length=16, win_length=64,
                                  win_function="hann_window", fmin=80, fmax=7600, mel_floor=1e-10, do_normalize=False)
    rng = np.random.default_rng(1337)
    items = []
    for n in (16000, 4321, 40000):
        t = np.arange(n) / 16000.0
        wav = (0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        mel = fe._extract_mel_features(wav)          # [frames, 80] float32
        items.append({"wav": torch.from_numpy(wav), "logmel": torch.from_numpy(np.asarray(mel, dtype=np.float32))})
    out = os.path.join(ROOT, "tests", "golden", "logmel_hf.pt")
    torch.save({"items": items, "source": "transformers.SpeechT5FeatureExtractor._extract_mel_features"}, out)
    for it in items:
        print(tuple(it["wav"].shape), "->", tuple(it["logmel"].shape))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
