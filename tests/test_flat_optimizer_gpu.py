"""FlatGradDataParallel + FusedAdam on the GPU: the fused (stacked-projection) weight-gradient GEMMs that the contiguous flat
gradient layout enables, and the bf16 parameter image the Adam kernel maintains (no cast launches), against the plain
per-parameter path on the same tiny model."""
import pytest
import torch

from tests.util import build_tiny, injected_randomness, load_golden, to_dev

pytestmark = pytest.mark.gpu


def _loss(model, fx, dev):
    from speecht5_amd.criterions import SpeechPretrainCriterion
    from tests.util import Task
    crit = SpeechPretrainCriterion(Task(), False, 1.0, 0.0, loss_weights=[10, 0.1])
    sample = to_dev(fx["sample"], dev)
    with injected_randomness(model, fx["mask_indices"], fx["mix_idx"], fx["gumbel_noise"], fx["tau"]):
        loss, ss, _ = crit(model, sample)
    return loss / ss


def test_flat_layout_fused_wgrad_and_bf16_image():
    from speecht5_amd import functional as Fn
    from speecht5_amd.ddp import FlatGradDataParallel, FusedAdam
    dev = torch.device("cuda:0")
    _, fx = load_golden("tiny_speech_pretrain.pt")
    try:
        # reference run: separate .grad tensors, per-weight gradient GEMMs, cast-kernel weight copies
        ref, _ = build_tiny(dev, torch.bfloat16)
        ref.train()
        _loss(ref, fx, dev).backward()
        gref = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        # flat run: one gradient buffer, stacked q/k/v (k/v) gradients by ONE GEMM each, Adam-maintained bf16 image
        Fn.weight_cache.clear()
        model, _ = build_tiny(dev, torch.bfloat16)
        model.train()
        ddp = FlatGradDataParallel(model)
        opt = FusedAdam(ddp, lr=1e-3, clip_norm=0.0, weight_decay=0.0)
        assert Fn.bf16_mirror.flat is not None
        att = model.encoder.layers[0].self_attn
        assert Fn._adjacent_grads([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight])
        assert Fn.bf16_mirror.stacked([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight]) is not None
        ddp.zero_grad()
        l0 = _loss(model, fx, dev)
        l0.backward()
        ddp.finish()
        torch.cuda.synchronize()
        tot = sum(float(g.double().pow(2).sum()) for g in gref.values()) ** 0.5
        for n, p in model.named_parameters():
            if n in gref:
                err = float((p.grad - gref[n]).double().norm())
                assert err <= 3e-2 * max(float(gref[n].double().norm()), 1e-3 * tot), (n, err)
        # one optimizer step, then the forward through the bf16 image must equal a forward that re-casts the fp32 masters
        opt.step(1.0)
        with torch.no_grad():
            l_img = float(_loss(model, fx, dev))
            for p in model.parameters():
                p._st5_mver = -1          # invalidate the image: cast-kernel path
            Fn.weight_cache.clear()
            l_cast = float(_loss(model, fx, dev))
        assert abs(l_img - l_cast) <= 2e-3 * abs(l_cast), (l_img, l_cast)
        assert abs(l_img - float(l0.detach())) > 0   # the step changed something
    finally:
        if "ddp" in locals():
            ddp.close()
        Fn.bf16_mirror.__init__()
        Fn.weight_cache.clear()
        Fn.set_layer_boundary_hook(None)
        Fn.set_compute_dtype(torch.float32)
