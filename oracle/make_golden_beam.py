"""Generates tests/golden/tiny_s2t_beam.pt: hypotheses of the VERBATIM reference generator
(SpeechT5/speecht5/sequence_generator.py, imported through oracle/ref_stubs.load_reference_generator) on the shared tiny model.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).

    python oracle/make_golden_beam.py

Cases: plain beam search (beam 1 and 3, two sentences of different length), joint CTC / attention scoring (--ctc-weight 0.3 and
0.5, one sentence as the README recipe decodes: --batch-size 1), n-gram blocking, a forced prefix.  The reference's CTC path
moves numpy scores with `.to(device="cuda")` (:385-388); this container has no GPU, so that one keyword is dropped while the
reference runs (the arithmetic is unchanged: everything stays fp32 on the host)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402
import make_golden as mg  # noqa: E402


class FairseqLikeDictionary(mg.Dictionary):
    def index(self, sym):          # fairseq Dictionary.index: <unk> for an unknown symbol (the generator probes "<mask>0")
        try:
            return list.index(self, sym)
        except ValueError:
            return self.unk()


class _CpuTo:
    """Drop `device="cuda"` from Tensor.to while the reference generator runs on the host."""

    def __enter__(self):
        self.orig = torch.Tensor.to
        orig = self.orig

        def to(t, *a, **k):
            if k.get("device") == "cuda":
                k = {x: y for x, y in k.items() if x != "device"}
                if not a and not k:
                    return t
            return orig(t, *a, **k)
        torch.Tensor.to = to

    def __exit__(self, *exc):
        torch.Tensor.to = self.orig


def main():
    ref = ref_stubs.load_reference_models()
    gen_mod = ref_stubs.load_reference_generator()
    args, task, model = mg.shared_model(ref)
    saved = torch.load(os.path.join(mg.OUT, "tiny_model.pt"))["state_dict"]
    for k, v in model.state_dict().items():
        assert torch.equal(v, saved[k]), f"shared tiny model drifted from tests/golden/tiny_model.pt at {k}"
    d = FairseqLikeDictionary(30, ["<mask>", "<ctc_blank>"])
    assert list(d) == list(task.dicts["text"])
    model.eval()
    sb = mg.speech_batch(args, B=2, S=7000, n_units=20, pad_last=900, seed=7)
    two = dict(net_input=dict(source=sb["net_input"]["source"], padding_mask=sb["net_input"]["padding_mask"]), id=torch.arange(2))
    one = dict(net_input=dict(source=sb["net_input"]["source"][:1], padding_mask=sb["net_input"]["padding_mask"][:1]), id=torch.arange(1))
    cases = dict(
        beam1=dict(sample="two", kw=dict(beam_size=1, max_len_b=12)),
        beam3=dict(sample="two", kw=dict(beam_size=3, max_len_b=12)),
        beam3_unnorm_minlen=dict(sample="two", kw=dict(beam_size=3, max_len_b=10, min_len=4, normalize_scores=False, unk_penalty=0.5)),
        beam3_ngram2=dict(sample="two", kw=dict(beam_size=3, max_len_b=12, no_repeat_ngram_size=2)),
        beam2_prefix=dict(sample="two", kw=dict(beam_size=2, max_len_b=10), prefix=torch.tensor([[7, 9], [11, 1]])),
        beam3_ctc03=dict(sample="one", kw=dict(beam_size=3, max_len_b=12, ctc_weight=0.3)),
        beam4_ctc05=dict(sample="one", kw=dict(beam_size=4, max_len_b=14, ctc_weight=0.5, len_penalty=0.8)),
        beam1_ctc05=dict(sample="one", kw=dict(beam_size=1, max_len_b=12, ctc_weight=0.5)),
    )
    out = dict(samples=dict(two=two, one=one), cases={})
    for name, c in cases.items():
        g = gen_mod.SequenceGenerator([model], d, **c["kw"])
        with torch.no_grad(), _CpuTo():
            hyps = g.generate([model], dict(out["samples"][c["sample"]]), prefix_tokens=c.get("prefix"))
        rec = [[dict(tokens=h["tokens"].clone(), score=float(h["score"]), positional_scores=h["positional_scores"].clone())
                for h in sent] for sent in hyps]
        out["cases"][name] = dict(sample=c["sample"], kw=c["kw"], prefix=c.get("prefix"), hyps=rec)
        for si, sent in enumerate(rec):
            gap = sent[0]["score"] - sent[1]["score"] if len(sent) > 1 else float("nan")
            print(f"{name:22s} sent {si}: best {sent[0]['tokens'].tolist()} score {sent[0]['score']:.5f}  (margin to 2nd {gap:.2e}, {len(sent)} hyps)")
    torch.save(out, os.path.join(mg.OUT, "tiny_s2t_beam.pt"))


if __name__ == "__main__":
    np.random.seed(0)
    torch.manual_seed(0)
    main()
