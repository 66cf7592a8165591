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
