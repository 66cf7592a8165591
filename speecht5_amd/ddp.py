"""Data parallelism for the SpeechT5 hot path: one process per GPU, RCCL (torch.distributed `nccl`) over xGMI.

The reference trains with fairseq's legacy_ddp (README.md:86-88): after backward, ONE all-reduce of a flat
buffer holding every parameter's gradient (zeros for parameters the micro-batch did not touch, which is why
the recipe needs --find-unused-parameters).  Same semantics here, re-designed for overlap:

* all gradients live in one flat fp32 buffer; `param.grad` are views into it, and the wgrad GEMM epilogues
  accumulate into those views directly (speecht5_amd/functional.py), so there is no copy-in/copy-out;
* the buffer is cut into buckets along module boundaries in backward order (post-nets, decoder layers 5..0,
  encoder layers 11..0, pre-nets).  `functional.layer_boundary()` plants an identity autograd node at each
  layer input; its backward fires when that layer's gradients are complete and launches the bucket's
  all-reduce (async, on the process group's RCCL stream) while backward continues on the compute stream;
* `finish()` launches whatever was not triggered (unused or boundary-less parameters are still reduced: every
  parameter takes part in every step), waits, and leaves the MEAN over ranks in the buffer.

xGMI note: the 8-GPU node is a full mesh of point-to-point links, so a ring all-reduce is bound
by one link (~153 GB/s).  Buckets default to 64 MB so that several collectives are in flight during backward.
"""
import os

import torch
import torch.distributed as dist

from . import functional as Fn


class _Trigger(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, bucket_id):
        ctx.owner, ctx.bucket_id = owner, bucket_id
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.owner._bucket_ready(ctx.bucket_id)
        return g, None, None


class FlatGradDataParallel:
    def __init__(self, model, process_group=None, bucket_groups=None):
        """bucket_groups: list of lists of modules, in the order their backward completes; parameters not covered by
        any group form a final bucket.  Default: derived from a T5TransformerModel (see `default_buckets`)."""
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # ST5_DDP_FORCE_COLLECTIVES=1 runs the bucketed async all-reduce path even in a 1-rank group (used to exercise the
        # RCCL stream / event plumbing on a single-GPU box; a 1-rank all-reduce leaves the data unchanged)
        import os
        self.collectives = self.world > 1 or (os.environ.get("ST5_DDP_FORCE_COLLECTIVES") == "1" and dist.is_initialized())
        params, seen = [], set()
        groups = bucket_groups if bucket_groups is not None else default_buckets(model)
        self.module_bucket = {}
        order = []
        for bi, mods in enumerate(groups):
            for m in mods:
                self.module_bucket[id(m)] = bi
                for p in _fusion_ordered_parameters(m):
                    if id(p) not in seen and p.requires_grad:
                        seen.add(id(p))
                        order.append((bi, p))
        nb = len(groups)
        for p in model.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                order.append((nb, p))
        order.sort(key=lambda t: t[0])
        # matrices start on 64-element boundaries: 128-byte aligned rows in the bf16 parameter image (the GEMM loaders
        # fetch 128-byte row segments; 16-byte-only alignment made them straddle cache lines and cost ~10 % in the NT
        # GEMMs); the few padding elements stay zero in the gradient / optimizer buffers
        offs, off = [], 0
        for _, p in order:
            if p.dim() >= 2:
                off = (off + 63) // 64 * 64
            offs.append(off)
            off += p.numel()
        total = (off + 7) // 8 * 8
        dev = order[0][1].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []  # (start, end)
        cur, start = order[0][0], 0
        for (bi, p), o in zip(order, offs):
            if bi != cur:
                self.buckets.append((start, o))
                start, cur = o, bi
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.buckets.append((start, total))
        self.params = [p for _, p in order]
        self.offsets = offs
        self._launched = [None] * len(self.buckets)
        Fn.set_layer_boundary_hook(self._boundary)
        if dev.type == "cuda":
            # batched split-K reductions: this wrapper owns the points where gradients must be complete (bucket
            # all-reduce, finish()), so the per-GEMM slab reductions can be deferred and folded in one launch
            from . import hip
            hip.check(hip.lib().st5_gemm_defer_splitk(1, hip.stream()), "st5_gemm_defer_splitk")
            if os.environ.get("ST5_LN_DEFER", "1") == "1":   # same idea for the LayerNorm dgamma/dbeta reductions
                hip.check(hip.lib().st5_layernorm_defer(1, hip.stream()), "st5_layernorm_defer")
            # weight-gradient GEMMs of the transformer layers on their own stream (functional.set_wgrad_stream): their
            # gradient buffers have no other writer (no tied weights inside a layer)
            if os.environ.get("ST5_WGRAD_STREAM", "1") == "1":
                from .modules.transformer_layer import TransformerSentenceEncoderLayer, TransformerDecoderLayer
                from .modules.speech_encoder_prenet import ConvFeatureExtractionModel
                for m in model.modules():
                    if isinstance(m, (TransformerSentenceEncoderLayer, TransformerDecoderLayer)):
                        for p in m.parameters():
                            p._st5_side_ok = True
                    elif isinstance(m, ConvFeatureExtractionModel) and os.environ.get("ST5_WGRAD_CONV", "1") == "1":
                        for p in m.parameters():
                            if p.dim() == 3:   # convolution weights (GroupNorm / LayerNorm parameters stay on the main stream)
                                p._st5_side_ok = True
                self._side = torch.cuda.Stream(device=dev)
                Fn.set_wgrad_stream(self._side)
            if os.environ.get("ST5_ATTN_STREAM", "1") == "1":
                self._attn_side = torch.cuda.Stream(device=dev)   # dq / dkv kernels of the attention backward side by side
                Fn.set_attention_stream(self._attn_side)

    # -- hooks -------------------------------------------------------------------------------------
    def _boundary(self, x, module):
        bi = self.module_bucket.get(id(module))
        if bi is None or not self.collectives or not x.requires_grad:
            return x
        return _Trigger.apply(x, self, bi)

    def _flush_splitk(self):
        """Gradients complete on the current stream: batched split-K reductions folded, weight-gradient stream joined."""
        from . import hip
        if self.flat.is_cuda:
            hip.check(hip.lib().st5_layernorm_flush(hip.stream()), "st5_layernorm_flush")
            if Fn.wgrad_stream() is not None:
                Fn.join_wgrad_stream()   # the deferred reductions belong to the side stream: folded there, then joined
            else:
                hip.check(hip.lib().st5_gemm_flush_splitk(hip.stream()), "st5_gemm_flush_splitk")

    def _bucket_ready(self, bi):
        if self._launched[bi] is None and self.collectives:
            self._flush_splitk()   # the bucket's weight gradients may still be slabs awaiting their batched reduction
            s, e = self.buckets[bi]
            self._launched[bi] = dist.all_reduce(self.flat[s:e], group=self.pg, async_op=True)

    # -- step API ----------------------------------------------------------------------------------
    def close(self):
        """Undo the process-wide switches this wrapper turned on (deferred split-K reductions, layer-boundary hook)."""
        from . import hip
        if self.flat.is_cuda:
            Fn.set_wgrad_stream(None)
            Fn.set_attention_stream(None)
            hip.check(hip.lib().st5_layernorm_defer(0, hip.stream()), "st5_layernorm_defer")
            hip.check(hip.lib().st5_gemm_defer_splitk(0, hip.stream()), "st5_gemm_defer_splitk")
        Fn.set_layer_boundary_hook(None)

    def zero_grad(self):
        self.flat.zero_()
        self._launched = [None] * len(self.buckets)

    def finish(self):
        """Call after backward: reduce the remaining buckets, wait for all, average over ranks."""
        self._flush_splitk()
        if self.collectives:
            for bi in range(len(self.buckets)):
                self._bucket_ready(bi)
            for w in self._launched:
                w.wait()
            if self.world > 1:
                self.flat.mul_(1.0 / self.world)
        self._launched = [None] * len(self.buckets)


def _fusion_ordered_parameters(module):
    """module.parameters() with the projections an attention block fuses into one GEMM made ADJACENT, in the order they are
    stacked ([Wq; Wk; Wv] / [Wk; Wv] and their biases), so that the optimizer's bf16 parameter image holds the stacked
    weight as one contiguous matrix (functional.bf16_mirror) and their gradients are one contiguous block too."""
    from .modules.multihead_attention import MultiheadAttention
    params = list(module.parameters())
    pos = {id(p): i for i, p in enumerate(params)}
    groups = []
    for m in module.modules():
        if isinstance(m, MultiheadAttention):
            projs = [m.q_proj, m.k_proj, m.v_proj] if m.self_attention else [m.k_proj, m.v_proj]
            groups.append([q.weight for q in projs])
            if all(q.bias is not None for q in projs):
                groups.append([q.bias for q in projs])
    out, done = [], set()
    head = {id(g[0]): g for g in groups}
    member = {id(p) for g in groups for p in g[1:]}
    for p in params:
        if id(p) in done or id(p) in member:
            continue
        if id(p) in head:
            for q in head[id(p)]:
                out.append(q); done.add(id(q))
        else:
            out.append(p); done.add(id(p))
    for p in params:   # members whose head was not in this module's list (cannot happen for whole attention blocks)
        if id(p) not in done:
            out.append(p); done.add(id(p))
    return out


def default_buckets(model):
    """Backward-completion order of a T5TransformerModel: post-nets + NCE head + quantizer, decoder layers (last
    first), decoder pre-nets, encoder layers (last first), encoder pre-nets.  The remaining parameters (layer-less
    encoder/decoder members, tied embeddings, ...) fall into a final bucket reduced by finish()."""
    groups = []
    head = [m for m in (getattr(model, n, None) for n in ("speech_decoder_postnet", "text_decoder_postnet", "hubert_layer",
                                                            "quantizer")) if m is not None]
    if head:
        groups.append(head)
    dec = getattr(model, "decoder", None)
    if dec is not None and hasattr(dec, "layers"):
        groups += [[l] for l in reversed(list(dec.layers))]
    pre = [m for m in (getattr(model, n, None) for n in ("speech_decoder_prenet", "text_decoder_prenet")) if m is not None]
    if pre:
        groups.append(pre)
    enc = getattr(model, "encoder", None)
    if enc is not None and hasattr(enc, "layers"):
        groups += [[l] for l in reversed(list(enc.layers))]
    return groups


class FusedAdam:
    """fairseq `adam` (decoupled weight decay) + global-norm clipping on the flat buffers, one HIP kernel per step.
    Parameters are re-pointed at views of one flat fp32 buffer so that the update is a single launch."""

    def __init__(self, ddp: FlatGradDataParallel, lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=5.0):
        self.ddp = ddp
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        total = ddp.flat.numel()
        dev = ddp.flat.device
        self.pflat = torch.zeros(total, dtype=torch.float32, device=dev)
        offsets = list(ddp.offsets)
        for p, off in zip(ddp.params, offsets):
            n = p.numel()
            self.pflat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.pflat[off:off + n].view_as(p)
        # bf16 image of the parameters, written by the Adam kernel itself (no per-weight cast launches), plus a pool for
        # the transposed copies of the >= 2-D ones (refreshed by one batched transpose per step)
        self.mirror = None
        if dev.type == "cuda" and Fn._S.dtype == torch.bfloat16:
            self.wflat = self.pflat.to(torch.bfloat16)
            Fn.bf16_mirror.attach(self.wflat, ddp.params, offsets)
            cap = sum(p.numel() + 64 for p in ddp.params if p.dim() >= 2)
            Fn.bf16_mirror.tflat = torch.empty(cap, dtype=torch.bfloat16, device=dev)
            Fn.bf16_mirror.tcap = cap
            self.mirror = Fn.bf16_mirror
        self.m = torch.zeros_like(self.pflat)
        self.v = torch.zeros_like(self.pflat)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = 0

    def backward(self, loss):
        loss.backward()

    def step(self, grad_scale=1.0):
        from . import hip
        self.t += 1
        L = hip.lib()
        g = self.ddp.flat
        if self.clip > 0:
            hip.check(L.st5_sumsq(g.data_ptr(), self.gnorm_sq.data_ptr(), g.numel(), 1.0, 0, hip.F32, hip.stream()), "st5_sumsq")
        hip.check(L.st5_adam_step(self.pflat.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), g.numel(), self.lr,
                                  self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                  self.gnorm_sq.data_ptr() if self.clip > 0 else 0, self.clip, grad_scale,
                                  self.wflat.data_ptr() if self.mirror is not None else 0, hip.stream()),
                  "st5_adam_step")
        # parameters changed in place through the flat view: invalidate the compute-dtype weight cache (entries that do
        # not come from the bf16 image: conv / fp32 / non-adjacent stacks) and refresh the transposed copies
        Fn.weight_cache.clear()
        if self.mirror is not None:
            for p in self.ddp.params:
                p._st5_mver = p._version
            self.mirror.refresh_transposes()
