"""Where do the 1.7e-3 .. 3.2e-3 post-net gradient errors of tests/test_cfg3_shape_gpu.py (fp32 parity mode) come from?
The product's post-net (kernels) against torch fp64 on the GPU, fed the product's own `before` and d(after)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tests.test_cfg3_shape_gpu import build, t2s_batch
from tests.util import to_dev
from speecht5_amd.criterions import TexttoSpeechLoss
dev = torch.device("cuda:0")
args, task, model = build(dev, torch.float32)
sample = to_dev(t2s_batch(len(task.dicts["text"])), dev)
model.train()
crit = TexttoSpeechLoss(task, False, use_guided_attn_loss=True, guided_attn_loss_sigma=0.4, guided_attn_loss_lambda=10.0, bce_pos_weight=5.0, sync_logging=False)
before, after, logits, attn = model(**sample["net_input"])
before.retain_grad(); after.retain_grad()
loss, l1, l2, bce, ga = crit.compute_loss(model, (before, after, logits, attn), sample)
loss.backward()
torch.cuda.synchronize()
pn = model.speech_decoder_postnet.postnet
d_after = after.grad.double()
print("d_after: nonzero", int((d_after != 0).sum()), "abs values", d_after.abs().unique()[:5].tolist(), "sum", float(d_after.sum()))
# torch fp64 reference of the post-net alone, on the product's `before` and d(after)
b64 = before.detach().double().clone().requires_grad_(True)
ws = []
x = b64.transpose(1, 2)
n = len(pn.postnet)
for i, blk in enumerate(pn.postnet):
    w = blk[0].weight.detach().double().requires_grad_(True); g = blk[1].weight.detach().double().requires_grad_(True); b = blk[1].bias.detach().double().requires_grad_(True)
    ws.append((w, g, b))
    x = F.conv1d(x, w, padding=2)
    x = F.batch_norm(x, None, None, g, b, training=True, eps=blk[1].eps)
    if i < n - 1:
        x = torch.tanh(x)
ref_after = b64 + x.transpose(1, 2)
print("after: max err", float((after.detach().double() - ref_after).abs().max()), "scale", float(ref_after.abs().max()))
ref_after.backward(d_after)
for i, blk in enumerate(pn.postnet):
    for nm, p, r in (("conv.w", blk[0].weight, ws[i][0]), ("bn.w", blk[1].weight, ws[i][1]), ("bn.b", blk[1].bias, ws[i][2])):
        gp = p.grad.double()
        print(f"layer {i} {nm}: rel {float((gp - r.grad).norm() / r.grad.norm()):.3e}  |ref| {float(r.grad.norm()):.3e}")
# d(before) through the post-net only: the product's before.grad also holds the direct l1 term, so compare the difference of both sides to it
print("d_before (total) product norm", float(before.grad.norm()))
