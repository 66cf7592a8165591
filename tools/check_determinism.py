"""Run-to-run determinism of the forward pass: the tiny speech-pretrain criterion evaluated on 5 freshly built models in one
process must give bit-identical loss terms (run it in several fresh processes: the first launch of a kernel has different
wave timing, which is what exposed fp32 LDS atomics in the conv0 statistics kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import build_tiny, load_golden, injected_randomness, to_dev, Task
from speecht5_amd import functional as Fn
from speecht5_amd.criterions import SpeechPretrainCriterion
dev = torch.device("cuda:0")
_, fx = load_golden("tiny_speech_pretrain.pt")
def ref_run():
    ref, _ = build_tiny(dev, torch.bfloat16); ref.train()
    crit = SpeechPretrainCriterion(Task(), False, 1.0, 0.0, loss_weights=[10, 0.1], sync_logging=False)
    sample = to_dev(fx["sample"], dev)
    with injected_randomness(ref, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        loss, ss, log = crit(ref, sample)
    torch.cuda.synchronize()
    return {k: (float(v) if torch.is_tensor(v) else v) for k, v in log.items()}
runs = [ref_run() for _ in range(5)]
for i, r in enumerate(runs[1:], 1):
    diff = {k: (runs[0][k], r[k]) for k in r if r[k] != runs[0][k]}
    if diff:
        print("RUN", i, "differs:", diff)
print("done", {k: v for k, v in runs[0].items() if k in ("loss", "loss_m_0", "loss_u_0", "dec_loss")})
