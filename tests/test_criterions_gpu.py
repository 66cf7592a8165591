"""The PRODUCT criteria (speecht5_amd/criterions.py, row a18 of SURVEY.md 8a) on the GPU model against the losses the
verbatim reference criteria produced for the same weights, inputs and random draws (tests/golden/*.pt):
model forward through the HIP kernels + criterion arithmetic + backward, all product code."""
from types import SimpleNamespace

import pytest
import torch

from tests.util import Task, build_tiny, check_grads, close, injected_randomness, load_golden, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_speech_pretrain_criterion(cuda):
    from speecht5_amd.criterions import SpeechPretrainCriterion
    _, fx = load_golden("tiny_speech_pretrain.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.train()
    sample = to_dev(fx["sample"], cuda)
    crit = SpeechPretrainCriterion(Task(), False, 1.0, 0.0, loss_weights=[10, 0.1])
    with injected_randomness(model, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        loss, sample_size, log = crit(model, sample)
    close(loss, fx["loss"], 2e-4, what="speech_pretrain loss")
    assert sample_size == fx["sample_size"]
    for k, v in fx["log"].items():
        if k in log and isinstance(v, float) and isinstance(log[k], (int, float)):
            assert abs(log[k] - v) <= 3e-4 * max(abs(v), 1.0), (k, log[k], v)
    (loss / sample_size).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)


def test_text_pretrain_criterion(cuda):
    from speecht5_amd.criterions import TextPretrainCriterion
    _, fx = load_golden("tiny_text_pretrain.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.train()
    sample = to_dev(fx["sample"], cuda)
    crit = TextPretrainCriterion(Task(), False, 1.0, loss_weights=[0.1])
    with injected_randomness(model, None, fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        loss, sample_size, log = crit(model, sample)
    close(loss, fx["loss"], 2e-4, what="text_pretrain loss")
    assert sample_size == fx["sample_size"]
    (loss / sample_size).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)


def test_s2t_criterion(cuda):
    from speecht5_amd.criterions import SpeechtoTextLoss
    _, fx = load_golden("tiny_s2t.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.speech_encoder_prenet.mask_prob = 0.5
    model.train()
    sample = to_dev(fx["sample"], cuda)
    cfg = SimpleNamespace(zero_infinity=True)
    crit = SpeechtoTextLoss(cfg, Task(), sentence_avg=False, label_smoothing=0.1, ce_weight=0.5, ctc_weight=0.5)
    with injected_randomness(model, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        loss, sample_size, log = crit(model, sample)
    close(loss, fx["loss"], 2e-4, what="s2t loss")
    assert sample_size == fx["sample_size"]
    for k in ("ce_loss", "ctc_loss", "nll_loss"):
        if k in fx["log"] and isinstance(fx["log"][k], float):
            assert abs(log[k] - fx["log"][k]) <= 3e-4 * max(abs(fx["log"][k]), 1.0), (k, log[k], fx["log"][k])
    (loss / sample_size).backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)


def test_t2s_criterion(cuda):
    from speecht5_amd.criterions import TexttoSpeechLoss
    _, fx = load_golden("tiny_t2s.pt")
    model, args = build_tiny(cuda, torch.float32)
    model.use_codebook = False
    model.train()
    sample = to_dev(fx["sample"], cuda)
    crit = TexttoSpeechLoss(Task(), False, use_guided_attn_loss=True, guided_attn_loss_sigma=0.4, guided_attn_loss_lambda=10.0,
                            num_layers_applied_guided_attn=2, num_heads_applied_guided_attn=2)
    net_output = model(**sample["net_input"])
    loss, l1, l2, bce, ga = crit.compute_loss(model, net_output, sample)
    close(loss, fx["loss"], 2e-4, what="t2s loss")
    close(l1, fx["l1"], 2e-4, what="t2s l1")
    close(bce, fx["bce"], 2e-4, what="t2s bce")
    close(ga, fx["guided"], 2e-4, what="t2s guided attention")
    loss.backward()
    torch.cuda.synchronize()
    check_grads(model, fx, 2e-3)
