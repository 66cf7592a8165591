"""Generates tests/golden/span_masks.pt: span masks of the reference's own copy of fairseq's compute_mask_indices
(/root/reference/SpeechLM/modules.py:219, the function SpeechT5/speecht5/models/modules/speech_encoder_prenet.py:237-263 calls)
for seeded numpy streams -- the pin for speecht5_amd/data_utils.compute_mask_indices (tests/test_masks_cpu.py).

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).

    python oracle/make_golden_masks.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_stubs  # noqa: E402
import make_golden as mg  # noqa: E402

from tests.mask_cases import CASES, padding  # noqa: E402


def main():
    ref_stubs.load_reference_models()
    from fairseq.data.data_utils import compute_mask_indices   # (the stub package re-exports the vendored copy)
    out = {}
    for name, c in CASES.items():
        res = []
        for seed in (0, 1, 2):
            np.random.seed(seed)
            m1 = compute_mask_indices(c["shape"], padding(c), **c["kw"])
            m2 = compute_mask_indices(c["shape"], padding(c), **c["kw"])    # second draw from the same stream: the stream position is pinned too
            res.append((torch.from_numpy(m1.copy()), torch.from_numpy(m2.copy()), float(np.random.rand())))
        out[name] = res
        print(f"{name:16s} masked fraction {float(res[0][0].float().mean()):.3f}, spans per row (seed 0): {res[0][0].sum(1).tolist()}")
    torch.save(out, os.path.join(mg.OUT, "span_masks.pt"))


if __name__ == "__main__":
    main()
