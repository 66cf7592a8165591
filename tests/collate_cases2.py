"""Shared by oracle/make_golden_collate2.py (verbatim reference collaters + noise), tests/test_text_noise_cpu.py and
tests/test_collate2_gpu.py: seeded synthetic items for TextToSpeechDataset.collater, SpeechToTextDataset.collater and
TextPretrainDataset (__getitem__ noise + collater) of /root/reference/SpeechT5/speecht5/data/."""
import torch

VOCAB, PAD, EOS, BOS, UNK, NMEL, SPK = 81, 1, 2, 0, 3, 80, 512
MASK_IDX = VOCAB          # (the task appends <mask> behind the dictionary's symbols: tasks/speecht5.py:283-291)
VOCAB_WITH_MASK = VOCAB + 2   # <mask>, <ctc_blank>

T2S_CASES = {"r2": dict(r=2, frames=[120, 87, 64, 121], toks=[31, 17, 40, 9]),
             "r1": dict(r=1, frames=[40, 33], toks=[12, 12]),
             "r3_single": dict(r=3, frames=[50], toks=[7])}
S2T_CASES = {"ragged": dict(sizes=[4000, 2560, 3999, 1200], toks=[9, 14, 1, 6]), "single": dict(sizes=[800], toks=[3])}


def t2s_items(case, seed=0):
    g = torch.Generator().manual_seed(2000 + seed)
    c = T2S_CASES[case]
    return [{"id": 5 + i, "audio_name": f"utt{i}", "source": [torch.cat([torch.randint(4, VOCAB, (n - 1,), generator=g), torch.tensor([EOS])])],
             "target": torch.randn(L, NMEL, generator=g), "spkembs": torch.randn(SPK, generator=g)}
            for i, (L, n) in enumerate(zip(c["frames"], c["toks"]))]


def s2t_items(case, seed=0):
    g = torch.Generator().manual_seed(3000 + seed)
    c = S2T_CASES[case]
    return [{"id": 7 + i, "source": torch.randn(S, generator=g), "label_list": [torch.randint(4, VOCAB, (n,), generator=g)]}
            for i, (S, n) in enumerate(zip(c["sizes"], c["toks"]))]


# BART noise configurations (TextPretrainDataset args, tasks/speecht5.py:141-200); "recipe" = the defaults of the pre-training recipe
NOISE = {
    "recipe":         dict(mask=0.3, mask_random=0.1, insert=0.0, rotate=0.0, mask_length="span-poisson", replace_length=1, whole_word=False),
    "recipe_words":   dict(mask=0.3, mask_random=0.1, insert=0.0, rotate=0.0, mask_length="span-poisson", replace_length=1, whole_word=True),
    "span_keep":      dict(mask=0.3, mask_random=0.3, insert=0.0, rotate=0.0, mask_length="span-poisson", replace_length=-1, whole_word=True),
    "span_delete":    dict(mask=0.25, mask_random=0.1, insert=0.05, rotate=0.0, mask_length="span-poisson", replace_length=0, whole_word=False),
    "word":           dict(mask=0.2, mask_random=0.2, insert=0.0, rotate=0.5, mask_length="word", replace_length=1, whole_word=True),
    "word_keep":      dict(mask=0.2, mask_random=0.5, insert=0.1, rotate=0.0, mask_length="word", replace_length=-1, whole_word=True),
    "subword":        dict(mask=0.15, mask_random=0.1, insert=0.0, rotate=0.0, mask_length="subword", replace_length=1, whole_word=False),
    "tiny_budget":    dict(mask=0.01, mask_random=0.1, insert=0.0, rotate=0.0, mask_length="span-poisson", replace_length=1, whole_word=False),
}
POISSON_LAMBDA = 3.5
TEXT_LENGTHS = [512, 300, 64, 17, 512, 130, 5, 256]


def word_start_table():
    """ByteTensor over the vocabulary: which symbols begin a word (get_whole_word_mask's output; here: every third symbol and the specials)."""
    t = torch.zeros(VOCAB_WITH_MASK, dtype=torch.uint8)
    t[4::3] = 1
    t[:4] = 1
    return t


def token_blocks(seed=0):
    g = torch.Generator().manual_seed(5000 + seed)
    return [torch.cat([torch.tensor([BOS]), torch.randint(4, VOCAB, (n - 2,), generator=g), torch.tensor([EOS])]) for n in TEXT_LENGTHS]


def flatten(batch):
    out = {}
    for k, v in batch.items():
        if k == "net_input":
            for k2, v2 in v.items():
                out["net_input." + k2] = v2
        else:
            out[k] = v
    return out
