"""The collater restatement (oracle/collate_oracle.py) against batches of the VERBATIM reference collater
(tests/golden/collate_speech_pretrain.pt, oracle/make_golden_collate.py): every tensor of the batch identical, bit for bit, for the
same items and the same numpy stream -- 5 configurations (crop to the shortest item / to max_sample_size, random and fixed crop
starts, pad_audio, reduction factors 1 / 2 / 3) x 2 seeds."""
import os

import pytest
import torch

from oracle import collate_oracle as CO
from tests import collate_cases as cc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate_speech_pretrain.pt")


@pytest.mark.parametrize("case", list(cc.CASES))
def test_collate_oracle_equals_the_reference_collater(case):
    gold = torch.load(GOLD)[case]
    c = cc.CASES[case]
    for seed in (0, 1):
        cc.seed_numpy(case, seed)
        got = CO.collate_speech_pretrain(cc.items(case, seed), pad_audio=c["pad_audio"], random_crop=c["random_crop"],
                                         max_sample_size=c["max_sample_size"], reduction_factor=c["reduction_factor"],
                                         sample_rate=cc.SAMPLE_RATE, label_rate=cc.LABEL_RATE, pad_idx=cc.PAD_IDX)
        ref = gold[seed]
        for k in cc.KEYS:
            assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape, (case, k, got[k].shape, ref[k].shape)
            assert torch.equal(got[k], ref[k]), (case, seed, k)
        assert got["ntokens"] == ref["ntokens"] and got["src_lengths"] == ref["src_lengths"] and torch.equal(got["id"], ref["id"])
