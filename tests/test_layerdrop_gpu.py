"""LayerDrop (t5_transformer_base ships 0.05 on both stacks, models/speecht5.py:1397-1398; applied at modules/encoder.py:251-257
and through LayerDropModuleList at modules/decoder.py:64-67): the device-side select form that a replayed step uses
(functional.layerdrop_select: every layer runs, a dropped layer's output and gradients are discarded on the device) against
the reference form (the layer is skipped on the host), same numpy / torch CPU draws on both sides."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from tests.util import Task, load_golden, to_dev

pytestmark = pytest.mark.gpu


def _loss_and_grads(cuda, static, layerdrop, seed):
    from speecht5_amd import functional as Fn
    from speecht5_amd.criterions import SpeechT5Criterion
    from speecht5_amd.speecht5 import T5TransformerModel
    from speecht5_amd.task import SpeechT5Task
    m, fx_s = load_golden("tiny_speech_pretrain.pt")
    _, fx_t = load_golden("tiny_text_pretrain.pt")
    args = Namespace(**m["args"])
    args.encoder_layerdrop = args.decoder_layerdrop = layerdrop     # (dropout stays 0 as in the golden configuration)
    Fn.set_compute_dtype(torch.float32)
    task = SpeechT5Task(args, Task().dicts)
    model = T5TransformerModel.build_model(args, task)
    torch.nn.Module.load_state_dict(model, m["state_dict"], strict=True)
    model = model.to(cuda).train()
    crit = SpeechT5Criterion(task, loss_weights=[10, 0.1], sync_logging=False)
    Fn.manual_seed(3)
    np.random.seed(seed)
    torch.manual_seed(seed)
    Fn._S.force_static = static
    try:
        out = []
        for fx in (fx_s, fx_t):
            model.zero_grad(set_to_none=True)
            loss, ss, _ = crit(model, to_dev(fx["sample"], cuda))
            (loss / ss).backward()
            torch.cuda.synchronize()
            out.append((float(loss.detach()) / float(ss), {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}))
        return out
    finally:
        Fn._S.force_static = False
        Fn.weight_cache.clear()


@pytest.mark.parametrize("seed", [1, 2, 5])
def test_select_form_equals_skipping(cuda, seed):
    skip = _loss_and_grads(cuda, False, 0.4, seed)
    sel = _loss_and_grads(cuda, True, 0.4, seed)
    full = _loss_and_grads(cuda, True, 0.0, seed)
    dropped_any = False
    for (ls, gs), (lt, gt), (lf, gf) in zip(skip, sel, full):
        assert abs(ls - lt) <= 2e-5 * abs(ls), (ls, lt)
        dropped_any |= abs(lf - ls) > 1e-4 * abs(ls)
        total = sum(float(g.pow(2).sum()) for g in gs.values()) ** 0.5
        for n, g in gt.items():
            r = gs.get(n)
            if r is None:       # skipped on the host: no gradient at all; selected away on the device: exactly zero
                assert float(g.abs().max()) == 0.0, (n, float(g.abs().max()))
                continue
            assert float((g - r).norm()) <= 2e-5 * float(r.norm()) + 2e-7 * total, (n, float((g - r).norm()), float(r.norm()))
    assert dropped_any, "no layer was dropped with these seeds: the test compared nothing"
