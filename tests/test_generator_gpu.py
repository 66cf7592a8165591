"""Beam search + joint CTC / attention scoring (speecht5_amd/sequence_generator.py, csrc/ctc_prefix.hip) against the hypotheses of
the VERBATIM reference generator on the tiny model (tests/golden/tiny_s2t_beam.pt, written by oracle/make_golden_beam.py):
token ids of EVERY returned hypothesis bit-exact and in the same order, scores within fp32 round-off of the model's logits."""
import numpy as np
import pytest
import torch

from tests.util import Task, build_tiny, load_golden, to_dev

pytestmark = pytest.mark.gpu

SCORE_TOL = 2e-4     # absolute, on length-normalised log-probabilities (logits differ from the CPU reference by ~1e-5)


@pytest.fixture(scope="module")
def setup(cuda):
    from speecht5_amd import functional as Fn
    _, fx = load_golden("tiny_s2t_beam.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.eval()
    yield model, fx, Task().dicts["text"]
    Fn.set_compute_dtype(torch.float32)


CASES = ["beam1", "beam3", "beam3_unnorm_minlen", "beam3_ngram2", "beam2_prefix", "beam3_ctc03", "beam4_ctc05", "beam1_ctc05"]


@pytest.mark.parametrize("name", CASES)
def test_hypotheses_equal_reference_generator(cuda, setup, name):
    from speecht5_amd.sequence_generator import SequenceGenerator
    model, fx, d = setup
    case = fx["cases"][name]
    gen = SequenceGenerator([model], d, **case["kw"])
    sample = to_dev(fx["samples"][case["sample"]], cuda)
    prefix = case["prefix"].to(cuda) if case["prefix"] is not None else None
    hyps = gen.generate([model], sample, prefix_tokens=prefix)
    assert len(hyps) == len(case["hyps"])
    for si, (got, ref) in enumerate(zip(hyps, case["hyps"])):
        assert len(got) == len(ref), f"sentence {si}: {len(got)} hypotheses, reference {len(ref)}"
        for hi, (g, r) in enumerate(zip(got, ref)):
            assert g["tokens"].cpu().tolist() == r["tokens"].tolist(), f"sentence {si} hypothesis {hi}"
            assert abs(float(g["score"]) - r["score"]) <= SCORE_TOL * max(1.0, abs(r["score"])), (si, hi, float(g["score"]), r["score"])
            np.testing.assert_allclose(g["positional_scores"].cpu().numpy(), r["positional_scores"].numpy(), atol=2e-3, rtol=0)


def test_ctc_prefix_kernel_equals_numpy_restatement(cuda):
    """csrc/ctc_prefix.hip against a numpy restatement of the prefix-score recursion (ctc_prefix_score.py:40-112) on random
    posteriors: first step (empty prefix), a repeated label among the candidates, eos and blank candidates."""
    from speecht5_amd.sequence_generator import CTCPrefixScorer
    rng = np.random.RandomState(0)
    T, V, blank, eos = 37, 19, 18, 2
    x = torch.log_softmax(torch.tensor(rng.randn(T, V), dtype=torch.float32), -1)
    xn = x.numpy()
    LOGZERO = np.float32(-1e10)

    def np_initial():
        r = np.full((T, 2), LOGZERO, dtype=np.float32)
        r[0, 1] = xn[0, blank]
        for t in range(1, T):
            r[t, 1] = r[t - 1, 1] + xn[t, blank]
        return r

    def np_score(y, cs, r_prev):
        out_len = len(y) - 1
        r = np.full((T, 2, len(cs)), LOGZERO, dtype=np.float32)
        xs = xn[:, cs]
        if out_len == 0:
            r[0, 0] = xs[0]
        r_sum = np.logaddexp(r_prev[:, 0], r_prev[:, 1])
        log_phi = np.stack([r_sum if (out_len == 0 or c != y[-1]) else r_prev[:, 1] for c in cs], 1)
        start = max(out_len, 1)
        log_psi = r[start - 1, 0].copy()
        for t in range(start, T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + xn[t, blank]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        for i, c in enumerate(cs):
            if c == eos:
                log_psi[i] = r_sum[-1]
            if c == blank:
                log_psi[i] = LOGZERO
        return log_psi, np.moveaxis(r, 2, 0)

    sc = CTCPrefixScorer(x.to(cuda), blank, eos)
    r0 = sc.initial_state()
    np.testing.assert_allclose(r0.cpu().numpy(), np_initial(), rtol=1e-6, atol=1e-5)
    y = [eos]
    r_prev_np = np_initial()
    r_prev = r0
    for step in range(6):
        cs = rng.permutation(V)[:7]
        if step == 2:
            cs[0] = y[-1]        # the label that repeats the prefix's last one
        if step == 3:
            cs[1], cs[2] = eos, blank
        psi_np, r_np = np_score(y, cs, r_prev_np)
        psi, r = sc.score(torch.tensor([y[-1]], device=cuda), len(y) - 1, torch.tensor(cs[None], device=cuda), r_prev[None])
        np.testing.assert_allclose(psi.cpu().numpy()[0], psi_np, rtol=2e-5, atol=2e-4)
        start = max(len(y) - 1, 1)
        np.testing.assert_allclose(r.cpu().numpy()[0][:, start - 1:], r_np[:, start - 1:], rtol=2e-5, atol=2e-4)
        pick = 0 if step == 2 else 3
        y.append(int(cs[pick]))
        r_prev_np, r_prev = r_np[pick], r[0, pick]
