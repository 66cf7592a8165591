"""speecht5_amd.text_noise.BartNoise (host half of the text pre-training input pipeline) against items of the VERBATIM
reference TextPretrainDataset.__getitem__ (text_dataset.py:203-433; tests/golden/collate_t2s_s2t_text.pt from
oracle/make_golden_collate2.py): for every noise configuration the same torch / numpy seeds must give IDENTICAL noised sources and
targets for all items, in order, AND leave both generators where the reference leaves them (same number and kind of draws) -- whole-word
and sub-word masking, Poisson spans replaced by one <mask> / deleted / replaced token by token, random-token substitution, the
insertions that 0-length spans turn into, insertion and rolling noise.  Items the reference itself cannot produce (DegenerateItem)
must fail here too, at the same positions."""
import os

import numpy as np
import pytest
import torch

from tests import collate_cases2 as cc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate_t2s_s2t_text.pt")


@pytest.mark.parametrize("name", list(cc.NOISE))
def test_bart_noise_equals_the_reference_items_and_rng_streams(name):
    from speecht5_amd.text_noise import BartNoise, DegenerateItem
    gold = torch.load(GOLD)["text"][name]
    n = cc.NOISE[name]
    noise = BartNoise(cc.VOCAB_WITH_MASK, cc.MASK_IDX, eos=cc.EOS, bos=cc.BOS, mask=n["mask"], mask_random=n["mask_random"], insert=n["insert"],
                      rotate=n["rotate"], poisson_lambda=cc.POISSON_LAMBDA, mask_length=n["mask_length"], replace_length=n["replace_length"],
                      mask_whole_words=cc.word_start_table() if n["whole_word"] else None)
    noise.strict = True
    torch.manual_seed(100 + len(name))
    np.random.seed(200 + len(name))
    changed = 0
    for i, (block, ref) in enumerate(zip(cc.token_blocks(), gold["items"])):
        try:
            got = noise.item(i, block, seed=7)
        except DegenerateItem:
            assert "error" in ref, (name, i, "the reference produced this item")
            continue
        assert "error" not in ref, (name, i, ref)
        assert torch.equal(got["source"], ref["source"]), (name, i, got["source"].tolist()[:40], ref["source"].tolist()[:40])
        assert torch.equal(got["target"], ref["target"]) and got["id"] == ref["id"]
        changed += int(not torch.equal(got["source"], got["target"]))
    assert changed >= 5
    assert (float(torch.rand(1)), float(np.random.rand())) == tuple(gold["rng_after"]), "a generator stands elsewhere than the reference's"


def test_degenerate_items_are_handled_outside_strict_mode():
    """Default mode: a block too short for any masking budget comes back unchanged instead of raising (the reference raises)."""
    from speecht5_amd.text_noise import BartNoise
    noise = BartNoise(cc.VOCAB_WITH_MASK, cc.MASK_IDX, mask=0.01)
    tok = torch.tensor([cc.BOS, 5, 6, 7, cc.EOS])
    src, tgt = noise(tok.clone())
    assert src.tolist()[0] == cc.BOS and src.tolist()[-1] == cc.EOS and torch.equal(tgt, tok)
