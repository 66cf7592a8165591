"""Host logic of the beam-search generator (speecht5_amd/sequence_generator.py: beam bookkeeping, length / unk penalties, minimum
length, prefix forcing, n-gram blocking, temperature, finalisation order) on the CPU: the product generator and the VERBATIM
reference generator (SpeechT5/speecht5/sequence_generator.py, golden tests/golden/fake_beam.pt from
oracle/make_golden_beam_fake.py) decode the same plain-torch stand-in model (tests/fake_seq_model.py) -- token ids must be
identical, scores equal to fp32 round-off.  (Joint CTC scoring needs the GPU scorer: tests/test_generator_gpu.py.)"""
import os

import pytest
import torch

from tests.fake_seq_model import CASES, FakeSeqModel, fake_sample
from tests.util import Task

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_generator_host_logic_matches_reference(name):
    from speecht5_amd.sequence_generator import SequenceGenerator
    fx = torch.load(os.path.join(G, "fake_beam.pt"), weights_only=False)["cases"][name]
    d = Task().dicts["text"]
    model = FakeSeqModel(len(d)).eval()
    c = CASES[name]
    g = SequenceGenerator([model], d, **c["kw"])
    prefix = torch.tensor(c["prefix"]) if "prefix" in c else None
    with torch.no_grad():
        hyps = g.generate([model], dict(fake_sample()), prefix_tokens=prefix)
    assert len(hyps) == len(fx)
    for si, (got, ref) in enumerate(zip(hyps, fx)):
        assert len(got) == len(ref), (name, si)
        for hi, (h, r) in enumerate(zip(got, ref)):
            assert h["tokens"].tolist() == r["tokens"].tolist(), (name, si, hi)
            assert abs(float(h["score"]) - r["score"]) <= 1e-5 * max(1.0, abs(r["score"])), (name, si, hi)
            assert torch.allclose(h["positional_scores"], r["positional_scores"], atol=1e-5), (name, si, hi)
