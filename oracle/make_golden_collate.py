"""Generates tests/golden/collate_speech_pretrain.pt: batches of the VERBATIM reference collater
(/root/reference/SpeechT5/speecht5/data/speech_dataset.py:302-446, SpeechPretrainDataset.collater / collater_audio / crop_to_max_size
/ collater_frm_label) on seeded synthetic items, numpy RNG seeded per case -- the pin for oracle/collate_oracle.py
(tests/test_collate_cpu.py) and for the HIP collater speecht5_amd/collate.py (tests/test_collate_gpu.py).

The dataset object is created without its constructor (which reads manifests from disk); the attributes the collater reads are set
by hand.  Third-party pieces it calls that are not under /root/reference: fairseq.data.data_utils.collate_tokens (restated below from
its published behaviour: right-pad 1-D tensors with pad_idx to the longest), librosa (imported by the module, unused by the collater).

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).     python oracle/make_golden_collate.py"""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_stubs  # noqa: E402
import make_golden as mg  # noqa: E402
from tests import collate_cases as cc  # noqa: E402


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False, pad_to_length=None, pad_to_multiple=1,
                   pad_to_bsz=None):
    """fairseq.data.data_utils.collate_tokens for the arguments the reference passes (pad_idx, left_pad=False)."""
    assert not left_pad and not move_eos_to_beginning and pad_to_multiple == 1 and pad_to_length is None and pad_to_bsz is None
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        res[i, : len(v)].copy_(v)
    return res


def load_dataset_module():
    ref_stubs.install()
    du = sys.modules["fairseq.data.data_utils"]
    du.collate_tokens = collate_tokens
    for name, attrs in (("librosa", {}), ("fairseq.data.audio", {}),
                        ("fairseq.data.audio.speech_to_text_dataset", {"get_features_or_waveform": lambda *a, **k: None}),
                        ("fairseq.data.fairseq_dataset", {"FairseqDataset": type("FairseqDataset", (), {})})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    sys.modules["fairseq.data"].data_utils = du
    path = "/root/reference/SpeechT5/speecht5/data/speech_dataset.py"
    spec = importlib.util.spec_from_file_location("ref_speech_dataset", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dataset(mod, case):
    c = cc.CASES[case]
    ds = object.__new__(mod.SpeechPretrainDataset)
    ds.pad_audio, ds.random_crop = c["pad_audio"], c["random_crop"]
    ds.max_sample_size, ds.reduction_factor = c["max_sample_size"], c["reduction_factor"]
    ds.sample_rate, ds.num_labels, ds.label_rates, ds.pad_list, ds.single_target = cc.SAMPLE_RATE, 1, [cc.LABEL_RATE], [cc.PAD_IDX], False
    return ds


def main():
    mod = load_dataset_module()
    out = {}
    for case in cc.CASES:
        res = []
        for seed in (0, 1):
            cc.seed_numpy(case, seed)
            batch = dataset(mod, case).collater(cc.items(case, seed))
            f = cc.flatten(batch)
            res.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in f.items()})
        out[case] = res
        f = res[0]
        print(f"{case:14s} source {tuple(f['source'].shape)} dec_target {tuple(f['dec_target'].shape)} prev {tuple(f['prev_output_tokens'].shape)} "
              f"target {tuple(f['target'].shape)} padded {int(f['padding_mask'].sum())} stop ones {int(f['labels'].sum())} ntokens {f['ntokens']}")
    torch.save(out, os.path.join(mg.OUT, "collate_speech_pretrain.pt"))


if __name__ == "__main__":
    main()
