"""Golden fixture for the HiFi-GAN generator.  The reference tree contains no vocoder; the spec
is HuggingFace `SpeechT5HifiGan` (installed in the build image), run here with seeded random weights on a tiny
configuration.  TEST INFRASTRUCTURE ONLY."""
import os

import torch
from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

if __name__ == "__main__":
    torch.manual_seed(21)
    kw = dict(model_in_dim=80, upsample_initial_channel=64, upsample_rates=[4, 4], upsample_kernel_sizes=[8, 8],
              resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]], normalize_before=True)
    m = SpeechT5HifiGan(SpeechT5HifiGanConfig(**kw)).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.5 / max(1.0, (p[0].numel() if p.dim() > 1 else 1) ** 0.5)))
        m.mean.copy_(torch.randn(80) * 0.3 - 1.0)
        m.scale.copy_(torch.rand(80) + 0.5)
    spec = torch.randn(2, 23, 80) * 0.5 - 1
    with torch.no_grad():
        wav = m(spec)
    torch.save(dict(config=kw, state_dict={k: v.clone() for k, v in m.state_dict().items()}, spectrogram=spec, waveform=wav),
               os.path.join(OUT, "tiny_hifigan.pt"))
    print("hifigan golden:", tuple(wav.shape), float(wav.abs().mean()))
