"""TEST / MEASUREMENT INFRASTRUCTURE (build container only: needs /root/reference).  Times the VERBATIM reference stack
(microsoft/SpeechT5 modules + criterion under oracle/ref_stubs.py) on the host cores next to the oracle port that bench.py's
`cpu_baseline` times on the GPU box (kind "port"), on the same workload: SpeechT5-Base, one 4 s clip, speech_pretrain
forward + criterion + backward, fp32.  Writes the ratio to profiles/r6_cpu_reference_vs_port.json (4 s and 10 s clips) (BASELINE.md section 4 promised
the verbatim modules as the CPU baseline; /root/reference does not exist on the GPU box, so the ratio is how the two connect)."""
import json
import os
import sys
import time
from argparse import Namespace
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_stubs  # noqa: E402
from make_golden import Task  # noqa: E402


def main(seconds=4.0, runs=5):
    ref = ref_stubs.load_reference_models()
    crit = ref_stubs.load_reference_criterions()
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True, share_input_output_embed=True)
    ref_stubs.ARCH_REGISTRY["t5_transformer_base"](args)
    task = Task(vocab=77, n_units=500)
    torch.manual_seed(1337)
    model = ref.T5TransformerModel.build_model(args, task)
    model.train()
    c = crit.speech_pretrain.SpeechPretrainCriterion(task, False, 1.0, 0.0, loss_weights=[10, 0.1])
    from speecht5_amd.synthetic import speech_pretrain_sample

    def run_ref(secs):
        s = speech_pretrain_sample(B=1, seconds=secs, device="cpu", seed=7)
        model.zero_grad(set_to_none=True)
        np.random.seed(1); torch.manual_seed(1)
        t0 = time.perf_counter()
        loss, ss, _ = c(model, s)
        (loss / ss).backward()
        return time.perf_counter() - t0

    # the port, exactly as bench.cpu_baseline runs it
    from oracle import speecht5_oracle as O
    sd = {k: v.detach().float().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(**vars(args))

    def run_port(secs):
        s = speech_pretrain_sample(B=1, seconds=secs, device="cpu", seed=7)
        T = int(secs * 50) - 1
        mask = torch.zeros(1, T, dtype=torch.bool)
        mask[:, : int(0.6 * T)] = True
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.forward_speech_pretrain(sd, cfg, s, mask_indices=mask, mix_idx=torch.arange(0, T, 2)[: T // 2], gumbel_noise=None)
        loss, ss, _ = O.speech_pretrain_loss(out, s, cfg, loss_weights=(10, 0.1))
        (loss / ss).backward()
        return time.perf_counter() - t0

    def measure(secs):
        res = {}
        for name, fn in (("reference", run_ref), ("port", run_port)):
            fn(1.0); fn(1.0)
            ts = sorted(fn(secs) for _ in range(runs))
            res[name] = dict(median_s=ts[len(ts) // 2], min_s=ts[0], max_s=ts[-1], audio_s_per_s=secs / ts[len(ts) // 2])
        return dict(workload=f"SpeechT5-Base speech_pretrain fwd + criterion + bwd, 1 x {secs:g} s clip, fp32, dropout / LayerDrop as shipped",
                    **res, reference_over_port=res["reference"]["median_s"] / res["port"]["median_s"])
    # round 6: refreshed at the current oracle, on the 4 s clip bench.py's cpu_baseline times AND on the 10 s clip of
    # the benched configuration; the 4 s entry keeps the top-level keys bench.py reads
    out = dict(cores=torch.get_num_threads(), where="build container (the GPU box has no /root/reference)", **measure(seconds))
    out["clip_10s"] = measure(10.0)
    path = os.path.join(ROOT, "profiles", "r6_cpu_reference_vs_port.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
