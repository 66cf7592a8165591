"""Shared by oracle/make_golden_collate.py (verbatim reference collater), tests/test_collate_cpu.py (oracle restatement) and
tests/test_collate_gpu.py (HIP collater): the configurations and the seeded synthetic items of the speech-pretraining collater
(reference SpeechT5/speecht5/data/speech_dataset.py:302-386)."""
import numpy as np
import torch

# name -> dataset attributes the collater reads (:238-244, :193-199) + item lengths in samples
CASES = {
    # the pre-training recipe's form (README.md:100-119: --max-sample-size 250000, crop to the shortest item, random crop, r = 2) at 1/8 of
    # its lengths (the fixture stays small; the arithmetic -- crop starts, frame alignment, thinning -- is length-independent)
    "recipe_crop":   dict(pad_audio=False, random_crop=True, max_sample_size=31250, reduction_factor=2,
                          sizes=[20480, 12160, 16384, 31360]),
    # cropping to max_sample_size itself (every item longer)
    "max_size_crop": dict(pad_audio=False, random_crop=True, max_sample_size=5000, reduction_factor=2,
                          sizes=[8000, 6000, 6400]),
    # pad to the longest item (pad_audio), no reduction
    "pad_audio":     dict(pad_audio=True, random_crop=False, max_sample_size=31250, reduction_factor=1,
                          sizes=[4000, 2560, 3200]),
    # pad_audio with items on both sides of max_sample_size, deterministic crop (start 0), r = 3
    "pad_and_crop":  dict(pad_audio=True, random_crop=False, max_sample_size=3750, reduction_factor=3,
                          sizes=[4800, 2880, 3750, 3840]),
    # one item
    "single":        dict(pad_audio=False, random_crop=True, max_sample_size=31250, reduction_factor=2, sizes=[6000]),
}
SAMPLE_RATE, LABEL_RATE, HOP, NMEL, PAD_IDX, SPK = 16000, 50.0, 256, 80, 1, 512


def items(case, seed=0):
    """Synthetic items as SpeechPretrainDataset.__getitem__ returns them (:277-285): waveform fp32 [S], log-mel target fp32
    [1 + S // 256, 80] (values only have to be distinguishable), k-means labels int64 at 50 Hz, speaker embedding [512]."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = []
    for i, S in enumerate(CASES[case]["sizes"]):
        wav = torch.randn(S, generator=g)
        L = 1 + S // HOP
        fb = torch.randn(L, NMEL, generator=g)
        nlab = int(S * LABEL_RATE / SAMPLE_RATE) + (1 if i % 2 else 0)      # (label files run a frame long or exact)
        lab = torch.randint(4, 504, (nlab,), generator=g)
        out.append({"id": 10 + i, "source": wav, "target": fb, "label_list": [lab], "spkembs": torch.randn(SPK, generator=g)})
    return out


def seed_numpy(case, seed=0):
    np.random.seed(4000 + 17 * seed + len(case))


KEYS = ("source", "padding_mask", "prev_output_tokens", "spkembs", "tgt_lengths", "labels", "dec_target", "dec_target_lengths",
        "target", "target_lengths")


def flatten(batch):
    """The tensors of a collated batch under fixed names (net_input merged in; single_target form)."""
    ni = batch["net_input"]
    out = {k: ni[k] for k in ("source", "padding_mask", "prev_output_tokens", "spkembs", "tgt_lengths")}
    out.update(labels=batch["labels"], dec_target=batch["dec_target"], dec_target_lengths=batch["dec_target_lengths"],
               target=batch["target_list"][0] if "target_list" in batch else batch["target"],
               target_lengths=batch["target_lengths_list"][0] if "target_lengths_list" in batch else batch["target_lengths"])
    out["ntokens"] = batch["ntokens_list"][0] if "ntokens_list" in batch else batch["ntokens"]
    out["src_lengths"] = list(batch["src_lengths"])
    out["id"] = batch["id"]
    return out
