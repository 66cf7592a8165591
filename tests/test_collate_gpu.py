"""SURVEY.md section 8 row f4 (input pipeline on the GPU): speecht5_amd.collate.SpeechPretrainCollater -- ragged gathers of the C ABI
(st5_ragged_rows, st5_tail_mask) over items resident in HBM -- against batches of the VERBATIM reference collater
(tests/golden/collate_speech_pretrain.pt; reference speech_dataset.py:302-446): every tensor bit-identical, same numpy stream position
afterwards; and the whole front of the pipeline, raw waveforms -> GPU log-mel -> collated batch, against the log-mel oracle."""
import os

import numpy as np
import pytest
import torch

from tests import collate_cases as cc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate_speech_pretrain.pt")


def _to(items, dev):
    return [{"id": s["id"], "source": s["source"].to(dev), "target": s["target"].to(dev), "label_list": [t.to(dev) for t in s["label_list"]],
             "spkembs": s["spkembs"].to(dev)} for s in items]


def _collater(case, cuda):
    from speecht5_amd.collate import SpeechPretrainCollater
    c = cc.CASES[case]
    return SpeechPretrainCollater(cuda, sample_rate=cc.SAMPLE_RATE, label_rates=[cc.LABEL_RATE], pad_list=[cc.PAD_IDX],
                                  max_sample_size=c["max_sample_size"], pad_audio=c["pad_audio"], random_crop=c["random_crop"],
                                  reduction_factor=c["reduction_factor"])


@pytest.mark.parametrize("case", list(cc.CASES))
def test_gpu_collater_equals_the_reference_collater(cuda, case):
    gold = torch.load(GOLD)[case]
    for seed in (0, 1):
        cc.seed_numpy(case, seed)
        batch = _collater(case, cuda).collater(_to(cc.items(case, seed), cuda))
        after = float(np.random.rand())
        torch.cuda.synchronize()
        got, ref = cc.flatten(batch), gold[seed]
        for k in cc.KEYS:
            g = got[k].cpu()
            assert g.dtype == ref[k].dtype and g.shape == ref[k].shape, (case, k, g.dtype, ref[k].dtype, g.shape, ref[k].shape)
            assert torch.equal(g, ref[k]), (case, seed, k, int((g != ref[k]).sum()))
        assert got["ntokens"] == ref["ntokens"] and got["src_lengths"] == ref["src_lengths"] and torch.equal(got["id"], ref["id"])
        # the numpy stream stands where the reference's stands (the same number of draws was taken)
        cc.seed_numpy(case, seed)
        from oracle import collate_oracle as CO
        CO.crop_starts([len(s["source"]) for s in cc.items(case, seed)], ref["source"].shape[1], cc.CASES[case]["random_crop"])
        assert after == float(np.random.rand())


def test_raw_waveforms_to_collated_batch_with_gpu_logmel(cuda):
    """Items WITHOUT precomputed targets: the collater computes the log-mel targets from the waveforms on the GPU (the reference's
    __getitem__ does that per item with librosa, :249-260) and collates them; against the log-mel oracle + the collate oracle."""
    from oracle import collate_oracle as CO, logmel_oracle as LO
    case = "recipe_crop"
    c = cc.CASES[case]
    items = cc.items(case, 3)
    for s in items:
        s["source"] = s["source"] * 0.1
    ref_items = [dict(s, target=torch.from_numpy(np.asarray(LO.logmelfilterbank(s["source"].numpy(), cc.SAMPLE_RATE))).float()) for s in items]
    cc.seed_numpy(case, 3)
    ref = CO.collate_speech_pretrain(ref_items, pad_audio=c["pad_audio"], random_crop=c["random_crop"], max_sample_size=c["max_sample_size"],
                                     reduction_factor=c["reduction_factor"], sample_rate=cc.SAMPLE_RATE, label_rate=cc.LABEL_RATE, pad_idx=cc.PAD_IDX)
    dev_items = _to(items, cuda)
    for s in dev_items:
        s["target"] = None
    cc.seed_numpy(case, 3)
    got = cc.flatten(_collater(case, cuda).collater(dev_items))
    torch.cuda.synchronize()
    assert torch.equal(got["source"].cpu(), ref["source"]) and torch.equal(got["target"].cpu(), ref["target"])
    assert got["dec_target"].shape == ref["dec_target"].shape and torch.equal(got["dec_target_lengths"].cpu(), ref["dec_target_lengths"])
    for k in ("dec_target", "prev_output_tokens"):
        err = float((got[k].cpu() - ref[k]).abs().max())
        assert err <= 2e-3, (k, err)        # (the GPU log-mel's own bar against the oracle: tests/test_logmel.py)
