"""Learning-rate schedules of the reference recipes (SpeechT5/README.md:114-119 pre-training `polynomial_decay`,
:188-191 ASR/TTS fine-tuning `tri_stage`, :305-307 `inverse_sqrt`), host-side scalars fed to `ddp.FusedAdam.lr`.

The schedulers themselves live in fairseq (third party, un-vendored, version unpinned: SURVEY.md 8c), so these are
restatements of fairseq's published definitions, not of code under /root/reference.  tests/test_lr_scheduler.py checks the
defining properties (end points, continuity, the README flag values) and, for `polynomial_decay` and `inverse_sqrt`, agreement
with an independent installed implementation of the same schedules (HuggingFace transformers) to fp64 round-off; `tri_stage` has
no second implementation here and stays "parity unpinned".

Usage:  sched = PolynomialDecaySchedule(2e-4, warmup_updates=64000, total_num_update=800000)
        opt.lr = sched.step_update(num_updates)      # once per optimizer step, before opt.step()
"""
import math


class PolynomialDecaySchedule:
    """fairseq `polynomial_decay`: linear warm-up from lr/warmup_updates to lr over `warmup_updates`, then
    (lr - end_lr) * (1 - (n - warmup) / (total - warmup)) ** power + end_lr, and end_lr from `total_num_update` on."""

    def __init__(self, lr, warmup_updates=0, total_num_update=1000000, end_learning_rate=0.0, power=1.0):
        assert total_num_update > 0
        self.peak, self.warmup, self.total = float(lr), int(warmup_updates), int(total_num_update)
        self.end, self.power = float(end_learning_rate), float(power)
        self.lr = self.peak / self.warmup if self.warmup > 0 else self.peak   # value before the first update

    def step_update(self, num_updates):
        n = num_updates
        if self.warmup > 0 and n <= self.warmup:
            self.lr = self.peak * n / float(self.warmup)
        elif n >= self.total:
            self.lr = self.end
        else:
            pct_remaining = 1.0 - (n - self.warmup) / float(self.total - self.warmup)
            self.lr = (self.peak - self.end) * pct_remaining ** self.power + self.end
        return self.lr


class TriStageSchedule:
    """fairseq `tri_stage` (SpecAugment paper schedule): linear warm-up from init_lr_scale*lr to lr, hold, exponential
    decay to final_lr_scale*lr, then constant.  `phase_ratio` = fractions of `max_update` for the three stages
    (README.md:190 "[0.1, 0.4, 0.5]"), or give the step counts explicitly."""

    def __init__(self, lr, max_update=None, phase_ratio=None, warmup_steps=0, hold_steps=0, decay_steps=0,
                 init_lr_scale=0.01, final_lr_scale=0.01):
        self.peak = float(lr)
        self.init_lr = init_lr_scale * self.peak
        self.final_lr = final_lr_scale * self.peak
        if phase_ratio is not None:
            assert max_update is not None and max_update > 0 and abs(sum(phase_ratio) - 1.0) < 1e-6
            self.warmup_steps = int(max_update * phase_ratio[0])
            self.hold_steps = int(max_update * phase_ratio[1])
            self.decay_steps = int(max_update * phase_ratio[2])
        else:
            self.warmup_steps, self.hold_steps, self.decay_steps = int(warmup_steps), int(hold_steps), int(decay_steps)
        assert self.warmup_steps + self.hold_steps + self.decay_steps > 0
        self.warmup_rate = (self.peak - self.init_lr) / self.warmup_steps if self.warmup_steps != 0 else 0.0
        self.decay_factor = -math.log(final_lr_scale) / self.decay_steps if self.decay_steps != 0 else 0.0
        self.lr = self.init_lr

    def _stage(self, n):
        if n < self.warmup_steps:
            return 0, n
        off = self.warmup_steps
        if n < off + self.hold_steps:
            return 1, n - off
        off += self.hold_steps
        if n <= off + self.decay_steps:
            return 2, n - off
        return 3, n - off - self.decay_steps

    def step_update(self, num_updates):
        stage, k = self._stage(num_updates)
        if stage == 0:
            self.lr = self.init_lr + self.warmup_rate * k
        elif stage == 1:
            self.lr = self.peak
        elif stage == 2:
            self.lr = self.peak * math.exp(-self.decay_factor * k)
        else:
            self.lr = self.final_lr
        return self.lr


class InverseSqrtSchedule:
    """fairseq `inverse_sqrt`: linear warm-up from warmup_init_lr (0 when warming up) to lr, then lr * sqrt(warmup / n)."""

    def __init__(self, lr, warmup_updates=4000, warmup_init_lr=-1.0):
        self.peak, self.warmup = float(lr), int(warmup_updates)
        if warmup_init_lr < 0:
            warmup_init_lr = 0.0 if self.warmup > 0 else self.peak
        self.init = float(warmup_init_lr)
        self.lr_step = (self.peak - self.init) / self.warmup if self.warmup > 0 else 0.0
        self.decay_factor = self.peak * max(self.warmup, 1) ** 0.5
        self.lr = self.init

    def step_update(self, num_updates):
        if num_updates < self.warmup:
            self.lr = self.init + num_updates * self.lr_step
        else:
            self.lr = self.decay_factor * max(num_updates, 1) ** -0.5
        return self.lr


def build_lr_scheduler(args):
    """From the recipe flags (argparse Namespace with fairseq's names: lr, lr_scheduler, warmup_updates, total_num_update,
    end_learning_rate, power, max_update, phase_ratio, init_lr_scale, final_lr_scale, warmup_init_lr)."""
    lr = args.lr[0] if isinstance(args.lr, (list, tuple)) else args.lr
    name = getattr(args, "lr_scheduler", "polynomial_decay")
    if name == "polynomial_decay":
        return PolynomialDecaySchedule(lr, getattr(args, "warmup_updates", 0), getattr(args, "total_num_update", 1000000),
                                       getattr(args, "end_learning_rate", 0.0), getattr(args, "power", 1.0))
    if name == "tri_stage":
        return TriStageSchedule(lr, getattr(args, "max_update", None), getattr(args, "phase_ratio", None),
                                getattr(args, "warmup_steps", 0), getattr(args, "hold_steps", 0), getattr(args, "decay_steps", 0),
                                getattr(args, "init_lr_scale", 0.01), getattr(args, "final_lr_scale", 0.01))
    if name == "inverse_sqrt":
        return InverseSqrtSchedule(lr, getattr(args, "warmup_updates", 4000), getattr(args, "warmup_init_lr", -1.0))
    raise ValueError(f"unknown --lr-scheduler {name!r} (the SpeechT5 recipes use polynomial_decay, tri_stage, inverse_sqrt)")
