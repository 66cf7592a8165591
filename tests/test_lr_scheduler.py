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
