"""SURVEY.md section 8 row f4, the remaining collaters (round 6): speecht5_amd.collate.TextToSpeechCollater / SpeechToTextCollater /
TextPretrainCollater -- ragged gathers of the C ABI over items resident in HBM -- against batches of the VERBATIM reference collaters
(tests/golden/collate_t2s_s2t_text.pt from oracle/make_golden_collate2.py; reference data/text_to_speech_dataset.py:228-298,
data/speech_to_text_dataset.py:150-241, data/text_dataset.py:18-99): every tensor bit-identical (values, dtype, shape), every host
field equal; and the text path end to end: token blocks -> BartNoise (host draws) -> GPU collation == noise + collate of the reference."""
import os

import numpy as np
import pytest
import torch

from tests import collate_cases2 as cc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate_t2s_s2t_text.pt")


def _dev(x, dev):
    if torch.is_tensor(x):
        return x.to(dev)
    if isinstance(x, list):
        return [_dev(v, dev) for v in x]
    if isinstance(x, dict):
        return {k: (_dev(v, dev) if k not in ("id",) else v) for k, v in x.items()}
    return x


def _same(got, ref, what):
    got = cc.flatten(got)
    assert set(got) == set(ref), (what, sorted(set(got) ^ set(ref)))
    for k, r in ref.items():
        g = got[k]
        if torch.is_tensor(r):
            g = g.cpu()
            assert g.dtype == r.dtype and g.shape == r.shape, (what, k, g.dtype, r.dtype, tuple(g.shape), tuple(r.shape))
            assert torch.equal(g, r), (what, k, int((g != r).sum()))
        else:
            assert g == r, (what, k, g, r)


@pytest.mark.parametrize("case", list(cc.T2S_CASES))
def test_text_to_speech_collater_equals_the_reference(cuda, case):
    from speecht5_amd.collate import TextToSpeechCollater
    ref = torch.load(GOLD)["t2s"][case]
    batch = TextToSpeechCollater(cuda, pad_idx=cc.PAD, reduction_factor=cc.T2S_CASES[case]["r"]).collater(_dev(cc.t2s_items(case), cuda))
    torch.cuda.synchronize()
    _same(batch, ref, f"t2s {case}")


@pytest.mark.parametrize("case", list(cc.S2T_CASES))
def test_speech_to_text_collater_equals_the_reference(cuda, case):
    from speecht5_amd.collate import SpeechToTextCollater
    ref = torch.load(GOLD)["s2t"][case]
    batch = SpeechToTextCollater(cuda, pad_idx=cc.PAD, eos_idx=cc.EOS).collater(_dev(cc.s2t_items(case), cuda))
    torch.cuda.synchronize()
    _same(batch, ref, f"s2t {case}")


@pytest.mark.parametrize("name", list(cc.NOISE))
def test_text_pretrain_pipeline_equals_the_reference(cuda, name):
    """Token blocks -> BartNoise on the host (the reference's draws, tests/test_text_noise_cpu.py) -> items to the device ->
    TextPretrainCollater: the batch of the verbatim __getitem__ + collate() -- padded source / target, descending-length order incl. its
    tie order, decoder input with the last token moved to the front."""
    from speecht5_amd.collate import TextPretrainCollater
    from speecht5_amd.text_noise import BartNoise, DegenerateItem
    gold = torch.load(GOLD)["text"][name]
    n = cc.NOISE[name]
    noise = BartNoise(cc.VOCAB_WITH_MASK, cc.MASK_IDX, eos=cc.EOS, bos=cc.BOS, mask=n["mask"], mask_random=n["mask_random"], insert=n["insert"],
                      rotate=n["rotate"], poisson_lambda=cc.POISSON_LAMBDA, mask_length=n["mask_length"], replace_length=n["replace_length"],
                      mask_whole_words=cc.word_start_table() if n["whole_word"] else None)
    noise.strict = True
    torch.manual_seed(100 + len(name))
    np.random.seed(200 + len(name))
    items = []
    for i, block in enumerate(cc.token_blocks()):
        try:
            items.append(noise.item(i, block, seed=7))
        except DegenerateItem:
            pass
    batch = TextPretrainCollater(cuda, pad_idx=cc.PAD).collater(_dev(items, cuda))
    torch.cuda.synchronize()
    _same(batch, gold["batch"], f"text {name}")


def test_reference_datasets_wrapped_with_gpu_collaters(cuda):
    """speecht5_amd.data.GpuCollated / wrap_datasets: objects shaped like the reference's dataset classes (the attributes their collaters
    read) get the matching GPU collater, configured from those attributes; items go to the device in __getitem__; everything else is the
    wrapped dataset's.  The batches equal the verbatim collaters' fixtures."""
    from speecht5_amd.data import GpuCollated, wrap_datasets
    gold = torch.load(GOLD)

    class Dict:
        def pad(self): return cc.PAD
        def eos(self): return cc.EOS

    class TextToSpeechDataset:
        def __init__(self, items, r):
            self.items, self.reduction_factor, self.src_dict, self.sizes = items, r, Dict(), [len(i["target"]) for i in items]
        def __getitem__(self, i): return self.items[i]
        def __len__(self): return len(self.items)
        def ordered_indices(self): return list(range(len(self.items)))

    class SpeechToTextDataset(TextToSpeechDataset):
        def __init__(self, items):
            self.items, self.tgt_dict = items, Dict()

    t2s = GpuCollated(TextToSpeechDataset(cc.t2s_items("r2"), 2), cuda)
    assert len(t2s) == 4 and t2s.ordered_indices() == [0, 1, 2, 3] and t2s[0]["target"].is_cuda and t2s[0]["source"][0].is_cuda
    _same(t2s.collater([t2s[i] for i in range(4)]), gold["t2s"]["r2"], "wrapped t2s")

    class MultitaskDataset:
        def __init__(self, ds):
            self.datasets, self.sample_ratios = list(ds), 1
        def collater(self, samples, idx):
            return self.datasets[idx].collater(samples)

    multi = wrap_datasets(MultitaskDataset([TextToSpeechDataset(cc.t2s_items("r1"), 1), SpeechToTextDataset(cc.s2t_items("ragged"))]), cuda)
    assert all(isinstance(m, GpuCollated) for m in multi.datasets)
    _same(multi.collater([multi.datasets[0][i] for i in range(2)], 0), gold["t2s"]["r1"], "wrapped multitask member 0")
    _same(multi.collater([multi.datasets[1][i] for i in range(4)], 1), gold["s2t"]["ragged"], "wrapped multitask member 1")
    torch.cuda.synchronize()
