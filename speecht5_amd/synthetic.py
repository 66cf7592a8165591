"""Seeded synthetic samples with the reference collaters' dict schemas.
There is no network for datasets; bench.py, smoke() and tests feed these."""
import numpy as np
import torch


def speech_pretrain_sample(B=8, seconds=10.0, n_units=500, spk_dim=512, device="cpu", seed=1337, sample_rate=16000):
    g = torch.Generator().manual_seed(seed)
    S = int(seconds * sample_rate)
    src = torch.randn(B, S, generator=g)
    L = 1 + S // 256
    mel = torch.randn(B, L, 80, generator=g) * 0.5 - 1
    r = 2
    Lr = L - L % r
    prev = torch.cat([mel.new_zeros(B, 1, 80), mel[:, r - 1::r][:, :-1]], 1)[:, : Lr // r]
    labels = torch.zeros(B, L)
    labels[:, -1] = 1.0
    s = dict(
        net_input=dict(source=src, padding_mask=torch.zeros(B, S, dtype=torch.bool), prev_output_tokens=prev,
                       tgt_lengths=torch.full((B,), prev.shape[1], dtype=torch.long), spkembs=torch.randn(B, spk_dim, generator=g)),
        target_list=[torch.randint(0, n_units, (B, int(seconds * 50)), generator=g) + 4],
        labels=labels, dec_target=mel, dec_target_lengths=torch.full((B,), L, dtype=torch.long), src_lengths=[S] * B,
        id=torch.arange(B), task_name="speech_pretrain")
    return to_device(s, device)


def text_pretrain_sample(B=16, T=512, vocab=83, mask_idx=None, device="cpu", seed=1338):
    """BART-style text infilling batch: 30 % of tokens replaced by Poisson(3.5)-length <mask> spans."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    mask_idx = vocab - 2 if mask_idx is None else mask_idx
    tgt = torch.randint(4, vocab - 2, (B, T), generator=g)
    tgt[:, -1] = 2
    src = tgt.clone()
    for b in range(B):
        n, t = 0, 0
        while n < 0.3 * T and t < T - 1:
            ln = max(1, int(rng.poisson(3.5)))
            st = int(rng.randint(0, T - 1 - ln)) if T - 1 - ln > 0 else 0
            src[b, st:st + ln] = mask_idx
            n += ln
            t += 1
    prev = torch.cat([torch.full((B, 1), 2, dtype=torch.long), tgt[:, :-1]], 1)
    s = dict(net_input=dict(src_tokens=src, src_lengths=torch.full((B,), T, dtype=torch.long), prev_output_tokens=prev),
             target=tgt, ntokens=int(tgt.ne(1).sum()), id=torch.arange(B), task_name="text_pretrain")
    return to_device(s, device)


def t2s_sample(B=32, T_text=100, L=600, vocab=83, spk_dim=512, r=2, device="cpu", seed=1339):
    """TTS fine-tuning batch (SURVEY.md 8d cfg 3; collater schema of data/text_to_speech_dataset.py:262-281): B texts of
    T_text tokens, B log-mel targets of L frames (80 bins), reduction factor r."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(4, vocab - 2, (B, T_text), generator=g)
    src[:, -1] = 2
    mel = torch.randn(B, L, 80, generator=g) * 0.5 - 1
    prev = torch.cat([mel.new_zeros(B, 1, 80), mel[:, r - 1::r][:, :-1]], 1)[:, : L // r]
    labels = torch.zeros(B, L)
    labels[:, -1] = 1.0
    s = dict(net_input=dict(src_tokens=src, src_lengths=torch.full((B,), T_text, dtype=torch.long), prev_output_tokens=prev,
                            tgt_lengths=torch.full((B,), prev.shape[1], dtype=torch.long), spkembs=torch.randn(B, spk_dim, generator=g),
                            task_name="t2s"),
             labels=labels, dec_target=mel, dec_target_lengths=torch.full((B,), L, dtype=torch.long),
             src_lengths=torch.full((B,), T_text, dtype=torch.long), target=src, ntokens=int(B * L), id=torch.arange(B), task_name="t2s")
    return to_device(s, device)


def to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, list):
        return [to_device(v, device) for v in obj]
    return obj
