"""Learning-rate schedules of the reference recipes (SpeechT5/README.md:114-119,188-191,305-307).  The schedulers are
fairseq's (un-vendored): these tests pin the defining properties of the published definitions, with the README flags."""
import math
import os
import sys
from argparse import Namespace

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht5_amd.lr_scheduler import (InverseSqrtSchedule, PolynomialDecaySchedule, TriStageSchedule,  # noqa: E402
                                       build_lr_scheduler)


def test_polynomial_decay_pretrain_recipe():
    s = PolynomialDecaySchedule(2e-4, warmup_updates=64000, total_num_update=800000)
    assert s.lr == pytest.approx(2e-4 / 64000)
    assert s.step_update(32000) == pytest.approx(1e-4)
    assert s.step_update(64000) == pytest.approx(2e-4)                     # peak at the end of warm-up
    assert s.step_update(64001) < 2e-4
    mid = 64000 + (800000 - 64000) // 2
    assert s.step_update(mid) == pytest.approx(1e-4)                        # linear (power 1) half way down
    assert s.step_update(800000) == 0.0 and s.step_update(900000) == 0.0
    lrs = [s.step_update(n) for n in range(64000, 800001, 9200)]
    assert all(a > b for a, b in zip(lrs, lrs[1:]))                          # strictly decreasing after warm-up
    q = PolynomialDecaySchedule(1.0, 0, 100, end_learning_rate=0.1, power=2.0)
    assert q.step_update(50) == pytest.approx(0.9 * 0.25 + 0.1)


def test_tri_stage_finetune_recipe():
    s = TriStageSchedule(6e-5, max_update=80000, phase_ratio=[0.1, 0.4, 0.5], final_lr_scale=0.05)
    assert (s.warmup_steps, s.hold_steps, s.decay_steps) == (8000, 32000, 40000)
    assert s.step_update(0) == pytest.approx(0.01 * 6e-5)
    assert s.step_update(4000) == pytest.approx(0.01 * 6e-5 + (6e-5 - 0.01 * 6e-5) / 2)
    assert s.step_update(8000) == pytest.approx(6e-5) and s.step_update(39999) == pytest.approx(6e-5)
    assert s.step_update(40000) == pytest.approx(6e-5)                      # decay starts continuously at the peak
    assert s.step_update(60000) == pytest.approx(6e-5 * math.sqrt(0.05))     # half of the exponential decay
    assert s.step_update(80000) == pytest.approx(6e-5 * 0.05)
    assert s.step_update(80001) == pytest.approx(6e-5 * 0.05) and s.step_update(10 ** 6) == pytest.approx(6e-5 * 0.05)


def test_inverse_sqrt_recipe():
    s = InverseSqrtSchedule(1e-4, warmup_updates=10000)
    assert s.step_update(0) == 0.0
    assert s.step_update(5000) == pytest.approx(5e-5)
    assert s.step_update(10000) == pytest.approx(1e-4)
    assert s.step_update(40000) == pytest.approx(5e-5)                      # lr * sqrt(warmup / n)


def test_build_from_flags():
    a = Namespace(lr=[2e-4], lr_scheduler="polynomial_decay", warmup_updates=64000, total_num_update=800000)
    assert isinstance(build_lr_scheduler(a), PolynomialDecaySchedule)
    b = Namespace(lr=[6e-5], lr_scheduler="tri_stage", max_update=80000, phase_ratio=[0.1, 0.4, 0.5], final_lr_scale=0.05)
    assert build_lr_scheduler(b).step_update(80000) == pytest.approx(3e-6)
    with pytest.raises(ValueError):
        build_lr_scheduler(Namespace(lr=1.0, lr_scheduler="cosine"))


def _hf_curve(make, lr, steps):
    """Learning rates an independent, installed implementation of the same published schedule yields (HuggingFace transformers
    `optimization.py`: a LambdaLR over a dummy optimizer), at the given update counts."""
    import torch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=lr)
    sched = make(opt)
    out, n = {}, 0
    want = set(steps)
    while n <= max(steps):
        if n in want:
            out[n] = opt.param_groups[0]["lr"]
        opt.step()
        sched.step()
        n += 1
    return out


def test_polynomial_decay_and_inverse_sqrt_agree_with_an_independent_implementation():
    """fairseq is un-vendored (the schedules here are restatements of its definitions), but the two published schedules
    `polynomial_decay` (with warm-up) and `inverse_sqrt` also exist in HuggingFace transformers, written independently: same curve
    to fp64 round-off at the README's hyper-parameters (scaled down 100x in length so that the loop stays short)."""
    from transformers import get_inverse_sqrt_schedule, get_polynomial_decay_schedule_with_warmup
    lr, warm, total = 2e-4, 640, 8000           # README.md:114-119 has 64000 / 800000
    steps = [0, 1, 2, 100, 639, 640, 641, 1000, 4000, 7999, 8000]
    ours = PolynomialDecaySchedule(lr, warmup_updates=warm, total_num_update=total, end_learning_rate=0.0, power=1.0)
    hf = _hf_curve(lambda o: get_polynomial_decay_schedule_with_warmup(o, warm, total, lr_end=0.0, power=1.0), lr, steps)
    for n in steps:
        assert abs(ours.step_update(n) - hf[n]) <= 1e-12 * lr + 1e-18, ("polynomial_decay", n, ours.lr, hf[n])
    ours = PolynomialDecaySchedule(lr, warmup_updates=warm, total_num_update=total, end_learning_rate=1e-6, power=2.0)
    hf = _hf_curve(lambda o: get_polynomial_decay_schedule_with_warmup(o, warm, total, lr_end=1e-6, power=2.0), lr, steps)
    for n in steps:
        assert abs(ours.step_update(n) - hf[n]) <= 1e-12 * lr + 1e-18, ("polynomial_decay power 2", n, ours.lr, hf[n])
    lr, warm = 1e-3, 400                           # README.md:305-307 has 4000
    steps = [0, 1, 100, 399, 400, 401, 1600, 6400]
    ours = InverseSqrtSchedule(lr, warmup_updates=warm)
    hf = _hf_curve(lambda o: get_inverse_sqrt_schedule(o, warm), lr, steps)
    for n in steps:
        assert abs(ours.step_update(n) - hf[n]) <= 1e-12 * lr + 1e-18, ("inverse_sqrt", n, ours.lr, hf[n])
