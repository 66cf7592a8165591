"""Generates tests/golden/fake_beam.pt: hypotheses of the VERBATIM reference generator (SpeechT5/speecht5/sequence_generator.py
through oracle/ref_stubs.load_reference_generator) on the plain-torch stand-in model of tests/fake_seq_model.py -- the pin for
the CPU test of the product generator's host logic (tests/test_generator_cpu.py).

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).

    python oracle/make_golden_beam_fake.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_stubs  # noqa: E402
import make_golden as mg  # noqa: E402
from make_golden_beam import FairseqLikeDictionary  # noqa: E402
from tests.fake_seq_model import CASES, FakeSeqModel, fake_sample  # noqa: E402


def main():
    ref_stubs.load_reference_models()
    gen_mod = ref_stubs.load_reference_generator()
    d = FairseqLikeDictionary(30, ["<mask>", "<ctc_blank>"])
    model = FakeSeqModel(len(d)).eval()
    sample = fake_sample()
    out = dict(cases={})
    for name, c in CASES.items():
        g = gen_mod.SequenceGenerator([model], d, **c["kw"])
        prefix = torch.tensor(c["prefix"]) if "prefix" in c else None
        with torch.no_grad():
            hyps = g.generate([model], dict(sample), prefix_tokens=prefix)
        out["cases"][name] = [[dict(tokens=h["tokens"].clone(), score=float(h["score"]), positional_scores=h["positional_scores"].clone())
                               for h in sent] for sent in hyps]
        for si, sent in enumerate(out["cases"][name]):
            print(f"{name:22s} sent {si}: {len(sent)} hyps, best {sent[0]['tokens'].tolist()} score {sent[0]['score']:.5f}")
    torch.save(out, os.path.join(mg.OUT, "fake_beam.pt"))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
