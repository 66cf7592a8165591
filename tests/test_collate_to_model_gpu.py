"""The input pipeline END TO END (SURVEY.md 8 row f4 -> rows a16 / a18): items resident in HBM -> the round-6 collaters -> the batch
dictionary goes straight into the model + criterion mirrors, as fairseq's trainer hands a collated batch to `criterion(model, sample)`
(tasks/speecht5.py:519-556).  What this pins is the SCHEMA: every key, dtype, device and shape the criteria and T5TransformerModel.forward
read is what the collaters produce; one training step per task on the tiny golden model, loss finite, gradients finite
and non-zero.  (Bit-equality of the batches themselves: tests/test_collate2_gpu.py; of the model: tests/test_model_gpu.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import Task, build_tiny  # noqa: E402

VOCAB = 30          # tests.util.Task: 4 specials + 30 symbols + <mask> + <ctc_blank>


def _step(model, crit_out):
    loss, ss, log = crit_out
    assert torch.isfinite(loss.detach()).all(), log
    (loss / ss).backward()
    torch.cuda.synchronize()
    g = [p.grad for p in model.parameters() if p.grad is not None]
    assert g and all(torch.isfinite(x).all() for x in g)
    assert sum(float(x.double().abs().sum()) for x in g) > 0
    return float(loss.detach())


def test_text_to_speech_batch_trains(cuda):
    from speecht5_amd.collate import TextToSpeechCollater
    from speecht5_amd.criterions import TexttoSpeechLoss
    model, args = build_tiny(cuda, torch.float32)
    model.use_codebook = False
    model.train()
    g = torch.Generator().manual_seed(1)
    items = [{"id": i, "audio_name": f"u{i}", "source": [torch.cat([torch.randint(4, 4 + VOCAB, (n - 1,), generator=g), torch.tensor([2])]).to(cuda)],
              "target": (torch.randn(L, 80, generator=g) * 0.5 - 1).to(cuda), "spkembs": torch.randn(args.spk_embed_dim, generator=g).to(cuda)}
             for i, (L, n) in enumerate([(44, 9), (31, 12), (50, 5)])]
    batch = TextToSpeechCollater(cuda, pad_idx=1, reduction_factor=args.reduction_factor).collater(items)
    assert batch["net_input"]["prev_output_tokens"].shape == (3, 25, 80) and batch["dec_target"].shape == (3, 50, 80)
    crit = TexttoSpeechLoss(Task(), False, use_guided_attn_loss=True, guided_attn_loss_sigma=0.4, guided_attn_loss_lambda=10.0)
    _step(model, crit(model, batch))


def test_speech_to_text_batch_trains(cuda):
    from types import SimpleNamespace
    from speecht5_amd.collate import SpeechToTextCollater
    from speecht5_amd.criterions import SpeechtoTextLoss
    model, args = build_tiny(cuda, torch.float32)
    model.train()
    g = torch.Generator().manual_seed(2)
    items = [{"id": i, "source": (torch.randn(S, generator=g) * 0.1).to(cuda), "label_list": [torch.randint(4, 4 + VOCAB, (n,), generator=g).to(cuda)]}
             for i, (S, n) in enumerate([(7000, 6), (5200, 9)])]
    batch = SpeechToTextCollater(cuda, pad_idx=1, eos_idx=2).collater(items)
    assert batch["net_input"]["padding_mask"].dtype == torch.bool and int(batch["net_input"]["padding_mask"].sum()) == 1800
    assert batch["target"].shape == (2, 10) and batch["net_input"]["prev_output_tokens"][:, 0].tolist() == [2, 2]
    np.random.seed(3)
    crit = SpeechtoTextLoss(SimpleNamespace(zero_infinity=True), Task(), sentence_avg=False, label_smoothing=0.1, ce_weight=0.5, ctc_weight=0.5)
    _step(model, crit(model, batch))


def test_text_pretrain_batch_trains(cuda):
    from speecht5_amd.collate import TextPretrainCollater
    from speecht5_amd.criterions import TextPretrainCriterion
    from speecht5_amd.text_noise import BartNoise
    model, args = build_tiny(cuda, torch.float32)
    model.train()
    task = Task()
    mask_idx = task.dicts["text"].index("<mask>")
    noise = BartNoise(len(task.dicts["text"]), mask_idx, mask=0.3, mask_random=0.1, poisson_lambda=3.5)
    torch.manual_seed(4)
    np.random.seed(4)
    g = torch.Generator().manual_seed(4)
    blocks = [torch.cat([torch.tensor([0]), torch.randint(4, 4 + VOCAB, (n - 2,), generator=g), torch.tensor([2])]) for n in (40, 25, 33)]
    items = []
    for i, b in enumerate(blocks):
        it = noise.item(i, b, seed=1)
        items.append({"id": i, "source": it["source"].to(cuda), "target": it["target"].to(cuda)})
    batch = TextPretrainCollater(cuda, pad_idx=1).collater(items)
    assert batch["target"].shape[0] == 3 and batch["net_input"]["src_lengths"].tolist() == sorted(batch["net_input"]["src_lengths"].tolist(), reverse=True)
    assert int((batch["net_input"]["src_tokens"] == mask_idx).sum()) > 0
    crit = TextPretrainCriterion(task, False, 1.0, loss_weights=[0.1])
    _step(model, crit(model, batch))
