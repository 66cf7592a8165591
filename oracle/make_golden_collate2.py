"""Generates tests/golden/collate_t2s_s2t_text.pt: batches of the VERBATIM reference collaters
  TextToSpeechDataset.collater      /root/reference/SpeechT5/speecht5/data/text_to_speech_dataset.py:228-298
  SpeechToTextDataset.collater      /root/reference/SpeechT5/speecht5/data/speech_to_text_dataset.py:150-241
  TextPretrainDataset.__getitem__ (BART noise :203-433) + collate() (:18-99) of data/text_dataset.py
on the seeded synthetic items of tests/collate_cases2.py -- the pin for speecht5_amd/collate.py (round 6 collaters, tests/test_collate2_gpu.py)
and speecht5_amd/text_noise.py (tests/test_text_noise_cpu.py).  The dataset objects are created without their disk-reading constructors
where they have one.  Third-party pieces that are not under /root/reference, restated from their published behaviour:
fairseq.data.data_utils.collate_tokens (right-pad; move_eos_to_beginning puts the item's last token -- or eos_idx -- first) and
numpy_seed (seed numpy from hash((seed, *extra)) % 1e6 inside the block, restore afterwards).

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).     python oracle/make_golden_collate2.py"""
import contextlib
import importlib.util
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_stubs  # noqa: E402
import make_golden as mg  # noqa: E402
from tests import collate_cases2 as cc  # noqa: E402


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False, pad_to_length=None, pad_to_multiple=1,
                   pad_to_bsz=None):
    assert not left_pad and pad_to_multiple == 1 and pad_to_length is None and pad_to_bsz is None
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        dst = res[i][: len(v)]
        if move_eos_to_beginning:
            dst[0] = v[-1] if eos_idx is None else eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res


@contextlib.contextmanager
def numpy_seed(seed, *addl_seeds):
    if seed is None:
        yield
        return
    if len(addl_seeds) > 0:
        seed = int(hash((seed, *addl_seeds)) % 1e6)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


class Vocab(list):
    def __init__(self):
        super().__init__(["<s>", "<pad>", "</s>", "<unk>"] + [f"c{i}" for i in range(cc.VOCAB - 4)] + ["<mask>", "<ctc_blank>"])

    def pad(self): return cc.PAD
    def eos(self): return cc.EOS
    def bos(self): return cc.BOS
    def unk(self): return cc.UNK


def load(name):
    ref_stubs.install()
    du = sys.modules["fairseq.data.data_utils"]
    du.collate_tokens, du.numpy_seed = collate_tokens, numpy_seed
    fd = sys.modules["fairseq.data"]
    fd.data_utils = du
    base = type("FairseqDataset", (), {})
    fd.FairseqDataset = base
    fd.Dictionary = Vocab
    for mname, attrs in (("librosa", {}), ("fairseq.data.audio", {}),
                         ("fairseq.data.audio.speech_to_text_dataset", {"get_features_or_waveform": lambda *a, **k: None}),
                         ("fairseq.data.fairseq_dataset", {"FairseqDataset": base})):
        if mname not in sys.modules:
            m = types.ModuleType(mname)
            m.__path__ = []
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[mname] = m
    spec = importlib.util.spec_from_file_location("ref_" + name, f"/root/reference/SpeechT5/speecht5/data/{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def clone(d):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}


def main():
    out = {"t2s": {}, "s2t": {}, "text": {}}
    t2s = load("text_to_speech_dataset")
    for case, c in cc.T2S_CASES.items():
        ds = object.__new__(t2s.TextToSpeechDataset)
        ds.reduction_factor, ds.num_labels, ds.src_dict = c["r"], 1, Vocab()
        out["t2s"][case] = clone(cc.flatten(ds.collater(cc.t2s_items(case))))
    s2t = load("speech_to_text_dataset")
    for case in cc.S2T_CASES:
        ds = object.__new__(s2t.SpeechToTextDataset)
        ds.num_labels, ds.tgt_dict = 1, Vocab()
        out["s2t"][case] = clone(cc.flatten(ds.collater(cc.s2t_items(case))))
    txt = load("text_dataset")
    vocab = Vocab()
    for name, n in cc.NOISE.items():
        args = Namespace(mask=n["mask"], mask_random=n["mask_random"], insert=n["insert"], rotate=n["rotate"], permute_sentences=0.0, bpe="sentencepiece",
                         replace_length=n["replace_length"], mask_length=n["mask_length"], poisson_lambda=cc.POISSON_LAMBDA)
        blocks = cc.token_blocks()
        ds = txt.TextPretrainDataset(blocks, np.array([len(b) for b in blocks]), vocab, cc.MASK_IDX, cc.word_start_table() if n["whole_word"] else None,
                                     shuffle=False, seed=7, args=args)
        torch.manual_seed(100 + len(name))
        np.random.seed(200 + len(name))
        items = []
        for i in range(len(blocks)):
            try:
                items.append(ds[i])
            except (TypeError, ValueError) as e:
                # (reference quirk: with a zero masking budget, or only 0-length spans drawn, add_whole_word_mask returns ONE tensor where
                #  __getitem__ unpacks two -- text_dataset.py:270, :301 against :210; such an item cannot be produced by the reference)
                items.append({"id": i, "error": type(e).__name__})
        probe = (float(torch.rand(1)), float(np.random.rand()))
        good = [it for it in items if "error" not in it]
        out["text"][name] = {"items": [clone(it) for it in items], "rng_after": probe, "batch": clone(cc.flatten(txt.collate(good, vocab.pad(), cc.EOS, vocab)))}
        ch = sum(int((it["source"].numel() != it["target"].numel()) or not torch.equal(it["source"], it["target"])) for it in good)
        print(f"text noise {name:14s}: {len(good)} of {len(items)} items, {ch} changed, lengths {[int(it['source'].numel()) for it in good]}")
    for k in ("t2s", "s2t"):
        for case, f in out[k].items():
            print(k, case, {n: tuple(v.shape) for n, v in f.items() if torch.is_tensor(v)})
    torch.save(out, os.path.join(mg.OUT, "collate_t2s_s2t_text.pt"))


if __name__ == "__main__":
    main()
