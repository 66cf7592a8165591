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
import contextlib
import os

import torch
import torch.distributed as dist

from . import functional as Fn



_SBS_STREAM_POOL = {}   # (device index, priority, position) -> the second micro-batch's stream, shared by every wrapper of the process

class _Trigger(torch.autograd.Function):
    """Identity whose backward tells the owner that every gradient produced by the graph BEHIND this point (the layers
    that run after it in the forward) is complete for this backward pass."""

    @staticmethod
    def forward(ctx, x, owner, bucket_id):
        ctx.owner, ctx.bucket_id = owner, bucket_id
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.owner._bucket_ready(ctx.bucket_id)
        return g, None, None


class FlatGradDataParallel:
    def __init__(self, model, process_group=None, bucket_groups=None, wgrad_stream=None):
        """bucket_groups: list of lists of modules, in the order their backward completes; parameters not covered by
        any group form a final bucket.  Default: derived from a T5TransformerModel (see `default_buckets`)."""
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # ST5_DDP_FORCE_COLLECTIVES=1 runs the bucketed async all-reduce path even in a 1-rank group (used to exercise the
        # RCCL stream / event plumbing on a single-GPU box; a 1-rank all-reduce leaves the data unchanged)
        self.collectives = self.world > 1 or (os.environ.get("ST5_DDP_FORCE_COLLECTIVES") == "1" and dist.is_initialized())
        # May collectives run underneath the backward?  Decided ONCE, here, and agreed over the group: every rank must
        # issue the same sequence of collectives -- per-bucket / per-range all-reduces when the answer is yes, one message behind
        # the backward when it is no -- so a rank whose environment differs (NCCL_ALGO missing on one of them) pulls all to "no".
        self.overlap_exchange = agreed_exchange_overlap(process_group, next(model.parameters()).device) if self.collectives else True
        groups = [g if isinstance(g, BucketGroup) else BucketGroup(list(g)) for g in
                  (bucket_groups if bucket_groups is not None else default_buckets(model))]
        nb = len(groups)
        # a parameter belongs to bucket i only if every module that holds it lies inside group i: tied weights (the text
        # embedding is shared by two pre-nets and a post-net) and everything outside the groups go to the final bucket,
        # which only finish() reduces -- otherwise a late contribution could land in a bucket already handed to RCCL
        sub_bucket = {}
        for bi, g in enumerate(groups):
            for m in g.modules:
                for sub in m.modules():
                    sub_bucket[id(sub)] = bi
        owners = {}
        for mod in model.modules():
            for q in mod.parameters(recurse=False):
                owners.setdefault(id(q), set()).add(sub_bucket.get(id(mod), nb))
        self.module_bucket = {}
        seen, order = set(), []
        for bi, g in enumerate(groups):
            for key in g.trigger_keys():
                self.module_bucket[key] = bi
            for m in g.modules:
                for q in _fusion_ordered_parameters(m):
                    if id(q) not in seen and q.requires_grad and owners.get(id(q), {nb}) == {bi}:
                        seen.add(id(q))
                        order.append((bi, q))
        for q in model.parameters():
            if id(q) not in seen and q.requires_grad:
                seen.add(id(q))
                order.append((nb, q))
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
        # Gradient accumulation (--update-freq U): triggers only count during the LAST micro-batch of an update
        # (`no_sync()` around the others), so a bucket is reduced once, after its last local contribution.
        # Collective ORDER must be identical on every rank (RCCL matches collectives by issue order and the layer
        # buckets all have the same size): buckets are launched strictly in index order -- bucket i goes out as soon as
        # it AND every bucket before it has been reported ready -- and finish() launches the rest, again in index order.
        # A rank that skipped a layer (LayerDrop, another modality) simply defers from that bucket on; it never reorders.
        self._accumulating = False
        self._local_phase = False   # local_phase(): no bucket triggers, gradients reduced afterwards by all_reduce_gradients()
        self._cut_set = None        # cut_points(): bucket indices at whose boundary the autograd graph of the forward is cut
        self._cuts = None
        self._fwd_streams = []   # streams of micro-batches 1.. (accumulate_overlapped)
        self.flat2 = None        # second gradient buffer (accumulate_overlapped), allocated on first use
        self._pair_pending = False   # flat2 holds gradients not yet summed into flat
        self._grads_zeroed = False   # the optimizer left both buffers zeroed: the next zero_grad() has nothing to do
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._sum_next = 0       # sum_bucket_range(): first bucket whose two gradient buffers are not summed yet
        self._works = []
        Fn.set_layer_boundary_hook(self._boundary)
        # A TIED parameter (the text embedding = the decoder's output projection) has several writers -- the embedding backward's
        # scatter-add at once, the projection's weight-gradient GEMM through a deferred reduction -- and the order of two fp32
        # accumulations is visible in the last bit: such a gradient is reduced where it is computed, in program order, so that the
        # result does not depend on where the next flush point happens to be (phased or not, queue full or not).
        owners = {}
        for m in model.modules():
            for p in m._parameters.values():
                if p is not None:
                    owners.setdefault(id(p), []).append(p)
        for lst in owners.values():
            if len(lst) > 1:
                lst[0]._st5_multi_writer = True
        if dev.type == "cuda":
            # batched split-K reductions: this wrapper owns the points where gradients must be complete (bucket
            # all-reduce, finish()), so the per-GEMM slab reductions can be deferred and folded in one launch
            from . import hip
            hip.check(hip.lib().st5_gemm_defer_splitk(1, hip.stream()), "st5_gemm_defer_splitk")
            Fn.set_wgrad_grouping(True)     # (same ownership: a layer's weight gradients queued and launched as one GEMM group)
            if os.environ.get("ST5_LN_DEFER", "1") == "1":   # same idea for the LayerNorm dgamma/dbeta reductions
                hip.check(hip.lib().st5_layernorm_defer(1, hip.stream()), "st5_layernorm_defer")
            # weight-gradient GEMMs of the transformer layers on their own stream (functional.set_wgrad_stream): their
            # gradient buffers have no other writer (no tied weights inside a layer)
            # wgrad_stream: None = the ST5_WGRAD_STREAM switch (default on).  Worth 3.5 ms per update when the step is enqueued
            # eagerly; inside a replayed graph it brings nothing (48.05 vs 47.9 ms) and, worse, the graph executor then puts
            # the second micro-batch's forward stream (accumulate_overlapped) behind the whole weight-gradient branch
            # (measured: 48.3 ms with both, 45.0 ms with the forward overlap alone) -- bench.py turns it off for replay.
            # Default OFF: with the micro-batches side by side it costs time (42.0 against 31.5 ms per update,
            # profiles/r5_replay_hunt.txt: ONE side stream turns every weight-gradient GEMM of both backward passes into one serial
            # chain); results are bit-identical with and without it.  A side stream PER micro-batch stream (round 5, tried) cannot be
            # captured: a helper stream forked from the second parent inside one capture crashes hipStreamEndCapture on ROCm 7.2.
            if (os.environ.get("ST5_WGRAD_STREAM", "0") == "1") if wgrad_stream is None else wgrad_stream:
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
            if os.environ.get("ST5_ATTN_STREAM", "0") == "1":
                self._attn_side = torch.cuda.Stream(device=dev)   # dq / dkv kernels of the attention backward side by side
                Fn.set_attention_stream(self._attn_side)

    # -- hooks -------------------------------------------------------------------------------------
    def _boundary(self, x, module, tag=None):
        # a forward through the model precedes every backward that writes the gradient buffers: whatever FusedAdam.step left
        # zeroed is not known to be zero any more (zero_grad() must fill again unless another step() intervenes)
        self._grads_zeroed = False
        if tag == "shared":
            # A tensor every layer of a stack reads (the relative-position keys; the caller passes the PRODUCER's output each
            # time).  Uncut, autograd folds the consumers' gradients one after the other, in arrival order, into the producer's
            # input buffer.  A backward cut in the middle of the stack would instead push each phase's partial sum through the
            # producer separately: same gradient, another bf16 summation order (measured: 1 ulp on single elements).  So under
            # cut_points() the consumers of each REGION (the stretch of forward between two cuts) read a leaf of their own, and
            # backward_phases() seeds a region's leaf with the sum its successors left behind before that region's gradients
            # arrive -- the fold continues where it stopped -- and runs the producer's backward once, at the end of its region.
            if not self._cut_set or not x.requires_grad:
                return x
            ent = next((c for c in self._cuts if c[0] == "shared" and c[1] is x), None)
            if ent is None:
                ent = ("shared", x, [])
                self._cuts.append(ent)
            region = sum(1 for c in self._cuts if c[0] not in ("shared", "bypass"))
            if not ent[2] or ent[2][-1][0] != region:
                ent[2].append((region, x.detach().requires_grad_(True)))
            return ent[2][-1][1]
        if tag == "bypass":
            # A loss term that reads a tensor produced EARLY in the forward directly (the feature penalty on the convolution stack's
            # output, speech_encoder_prenet.py:172-176): its gradient path crosses no layer boundary, so the first phase of a cut
            # backward (loss.backward()) would run the producer's backward at once, with the penalty's gradient alone, and the phase
            # that brings the main gradient would run it a second time.  Under cut_points() the term reads a leaf; backward_phases()
            # hands the leaf's gradient to the producer as an extra root of the phase that runs the producer's region: one backward
            # through the producer, the two contributions summed by autograd as in the uncut graph (two addends: order-free).
            if not self._cut_set or not x.requires_grad:
                return x
            leaf = x.detach().requires_grad_(True)
            self._cuts.append(("bypass", x, leaf, sum(1 for c in self._cuts if c[0] not in ("shared", "bypass"))))
            return leaf
        bi = self.module_bucket.get((id(module), tag))
        if bi is not None and self._cut_set is not None and bi in self._cut_set and x.requires_grad:
            # phased backward (cut_points): everything BEHIND this point hangs off a fresh leaf, so a backward pass stops here
            # with bucket bi (and every bucket before it) complete; the next phase continues from (x, leaf.grad)
            leaf = x.detach().requires_grad_(True)
            self._cuts.append((bi, x, leaf))
            return leaf
        if bi is None or not self.collectives or self._local_phase or not x.requires_grad:
            return x
        return _Trigger.apply(x, self, bi)

    def _flush_splitk(self):
        """Gradients complete on the current stream: batched split-K reductions folded, weight-gradient stream joined."""
        from . import hip
        if self.flat.is_cuda:
            Fn.flush_wgrads()
            hip.check(hip.lib().st5_layernorm_flush(hip.stream()), "st5_layernorm_flush")
            if Fn.wgrad_stream() is not None:
                Fn.join_wgrad_stream()   # the deferred reductions belong to the side stream: folded there, then joined
            else:
                hip.check(hip.lib().st5_gemm_flush_splitk(hip.stream()), "st5_gemm_flush_splitk")

    def _bucket_ready(self, bi):
        """A layer trigger fired: bucket bi has received its last local contribution of this update."""
        if not self.collectives or self._accumulating:
            return
        self._ready[bi] = True
        if self.overlap_exchange:           # else: finish() launches every bucket, in index order, behind the backward
            self._launch_in_order(False)

    def _launch_in_order(self, everything):
        nb = len(self.buckets)
        if self._next >= nb or not (everything or self._ready[self._next]):
            return
        # the buckets' weight gradients may still be slabs awaiting their batched reduction, or in flight on the
        # weight-gradient stream: fold and join first, so that the collective (ordered after the current stream by the
        # process group) only ever sees complete gradients
        self._flush_splitk()
        while self._next < nb and (everything or self._ready[self._next]):
            s, e = self.buckets[self._next]
            self._works.append(dist.all_reduce(self.flat[s:e], group=self.pg, async_op=True))
            self._next += 1

    @contextlib.contextmanager
    def no_sync(self):
        """Wrap the forward+backward of every micro-batch of an update EXCEPT the last one (torch DDP's no_sync /
        fairseq's accumulation, trainer semantics of --update-freq): gradients accumulate locally, nothing is reduced."""
        old, self._accumulating = self._accumulating, True
        try:
            yield
        finally:
            self._accumulating = old

    def accumulate(self, micro_batches, fn):
        """fn(sample) = forward + backward of one micro-batch; runs all of them with the right sync mode."""
        n = len(micro_batches)
        out = []
        for i, mb in enumerate(micro_batches):
            with (self.no_sync() if i + 1 < n else contextlib.nullcontext()):
                out.append(fn(mb))
        return out

    def accumulate_overlapped(self, micro_batches, forward_loss, backward="side_by_side"):
        """The (two) micro-batches of one update on two streams: forward_loss(sample) -> normalised loss tensor (forward only).
        Micro-batch 0 runs on the current stream, micro-batch 1 on its own stream, forked here; autograd runs a node's backward
        on the stream of its forward.  backward = "side_by_side": micro-batch 1 accumulates into a SECOND flat gradient buffer
        (its param.grad views are switched while its backward is enqueued), so the two backward passes are independent and run
        concurrently; the buffers are summed afterwards.  backward = "in_turn": same two buffers, same arithmetic, the second
        backward ordered behind the first (bit-identical results: the race check of tests/test_graph_gpu.py).  Several ranks: only
        inside local_phase() (the bucket all-reduces of the overlapped path need summed gradients: that path uses accumulate()).
        Measured on the pre-training update (speech 8 x 10 s + text 16 x 512, graph replay): in turn 47.8 ms, forward passes side
        by side 44.8 ms, forward and backward side by side 37.4 ms -- on their own the micro-batches are strings of kernels that
        leave much of the chip idle (the reference's trainer runs them in turn, trainer semantics unchanged)."""
        from . import hip
        n = len(micro_batches)
        assert n <= 2, "accumulate_overlapped: at most two micro-batches per update"
        assert not self.collectives or self._local_phase, \
            "accumulate_overlapped with several ranks: inside local_phase(), followed by all_reduce_gradients()"
        cur = self.pair_streams(n)
        # Which micro-batch owns the update's own stream (`cur`)?  Default: the first.  ST5_SBS_OWNER=1 (A/B) gives it to the SECOND
        # (host enqueue order, gradient buffers and random draws unchanged -- only the stream assignment swaps), and with a
        # weight-gradient stream (ST5_WGRAD_STREAM=1) that micro-batch's weight-gradient GEMMs fork to it: a third concurrent chain
        # forked from the capture's FIRST parent, the only kind ROCm 7.2 captures.
        own = int(os.environ.get("ST5_SBS_OWNER", "0")) if (n == 2 and backward == "side_by_side") else 0
        stream_of = (lambda i: cur if i == own else self._fwd_streams[0]) if n == 2 else (lambda i: cur)
        if Fn.wgrad_stream() is not None and n == 2:
            Fn.set_wgrad_owner(cur)
        # ST5_WGRAD_MOVE=1 (A/B switch, OFF by default): the weight-gradient groups of the micro-batch on the SECOND stream are launched
        # on the update's own stream, behind the first micro-batch's backward (functional.set_wgrad_target).  The idea: the second
        # micro-batch -- text -- is the longer chain (21.1 against 17.2 ms alone, tools/r6/chains.py), its weight gradients are off its
        # critical path and the first stream runs dry early.  Measured (round 6, same box, alternating): 35.0 ms per update against
        # 29.8 -- every cross-stream edge inside a replayed graph costs more than the idle time it fills, like the third chain of
        # round 5.  Same bits either way.
        move = (n == 2 and backward == "side_by_side" and own == 0 and Fn.wgrad_stream() is None and stream_of(1) is not cur
                and os.environ.get("ST5_WGRAD_MOVE", "0") == "1")
        losses = []
        for i, mb in enumerate(micro_batches):
            with torch.cuda.stream(stream_of(i)):
                losses.append(forward_loss(mb))
        try:
            for i, loss in enumerate(losses):
                st = stream_of(i)
                if i > 0 and backward == "in_turn":
                    st.wait_stream(cur)      # (behind micro-batch 0's backward)
                if i == 1 and move:
                    Fn.set_wgrad_target(st, cur)
                with self._grad_slot(i), torch.cuda.stream(st):
                    loss.backward()          # (root gradient on `st`: no dependence on the other micro-batch's stream)
                    # the last layers' queued weight gradients and this stream's deferred reductions are folded on this stream
                    self.flush_stream_deferred()
        except BaseException:
            Fn.drop_wgrads()             # (a backward that raised leaves nothing behind for the next update)
            raise
        finally:
            Fn.set_wgrad_target(None, None)
        for st in self._fwd_streams[: n - 1]:
            cur.wait_stream(st)
        if move:
            # (groups that fell back to single split-K launches left their slabs pending on `cur`: folded where this wrapper folds
            #  `cur`'s own -- finish() / sum_gradient_buffers() / FusedAdam.step's callers -- before anybody reads the gradients)
            Fn.release_wgrad_holds()     # (joined: the operands of the moved groups may go back to the allocator)
        # the update's gradient is flat + flat2: FusedAdam.step reads both (and leaves both zeroed); anyone else gets the sum
        # through sum_gradient_buffers()
        self._pair_pending = n > 1
        return [l.detach() for l in losses]

    def pair_streams(self, n):
        """Streams and buffers of n (<= 2) micro-batches side by side: the current stream for micro-batch 0, a stream of its own for
        micro-batch 1 (created once, ordered behind the current stream here), the second gradient buffer.  Returns the current stream."""
        from . import hip
        cur = torch.cuda.current_stream()
        if not self._fwd_streams and n > 1 and os.environ.get("ST5_DEEP_RING") is None:   # (the env switch: bench.py A/B)
            # side by side, a one-block-per-CU GEMM grid gets its latency cover from the other stream's blocks; the deep operand
            # ring of csrc/gemm.hip (128 KB of LDS per block) would keep those off the CU: 37.7 vs 38.3 ms per update
            hip.check(hip.lib().st5_gemm_set_deep_ring(0, 2), "st5_gemm_set_deep_ring")
        while len(self._fwd_streams) < n - 1:
            # ST5_SERIAL_MICRO=1 (debug): the "second stream" IS the current stream -- same program, same two gradient buffers,
            # no concurrency at all (the reference point when hunting a race between the micro-batches' kernels)
            # ST5_SBS_PRIORITY (A/B): HIP priority of the second micro-batch's stream (-1 = high).  The text micro-batch is the longer
            # chain (8.1 against 4.75 TFLOP): whatever it gains while both are resident comes off the update's critical path.
            prio = int(os.environ.get("ST5_SBS_PRIORITY", "0"))
            # The stream comes from a process-wide pool (one per device, priority and position), not from a fresh torch.cuda.Stream()
            # per wrapper: the library keeps per-stream state in small tables keyed by the stream handle (8 split-K slab workspaces,
            # 32 deferred-reduction and LayerNorm states, csrc/gemm.hip / norm.hip) whose recycling paths are for streams that come and
            # go -- a process that builds many wrappers (a test session) walked torch's pool of 32 streams through them
            # (profiles/r6b_side_by_side_flake.txt).  ST5_SBS_STREAM_POOL=0 restores a stream per wrapper (A/B).
            if os.environ.get("ST5_SERIAL_MICRO") == "1":
                self._fwd_streams.append(cur)
            elif os.environ.get("ST5_SBS_STREAM_POOL", "1") == "0":
                self._fwd_streams.append(torch.cuda.Stream(device=self.flat.device, priority=prio))
            else:
                key = (self.flat.device.index, prio, len(self._fwd_streams))
                if key not in _SBS_STREAM_POOL:
                    _SBS_STREAM_POOL[key] = torch.cuda.Stream(device=self.flat.device, priority=prio)
                self._fwd_streams.append(_SBS_STREAM_POOL[key])
        if n > 1 and self.flat2 is None:
            # the second gradient buffer is created (and zero-filled) HERE, on the current stream, before the streams fork: created
            # lazily inside the second backward it was zero-filled on this stream while the other stream already accumulated into it
            self._make_flat2()
        for st in self._fwd_streams[: n - 1]:
            st.wait_stream(cur)          # every forward / phase starts from here (what precedes: zero_grad, the previous phase)
        return cur

    def flush_stream_deferred(self):
        """Fold what the CURRENT stream has deferred (queued weight-gradient groups, LayerNorm partials, split-K slabs): the end of a
        micro-batch's backward -- or of one phase of it -- on that micro-batch's stream."""
        from . import hip
        Fn.flush_wgrads()
        Fn.join_wgrad_stream()
        hip.check(hip.lib().st5_layernorm_flush(hip.stream()), "st5_layernorm_flush")
        hip.check(hip.lib().st5_gemm_flush_splitk(hip.stream()), "st5_gemm_flush_splitk")

    def sum_bucket_range(self, upto):
        """flat[range] += flat2[range] (and flat2[range] = 0) for the buckets [first not yet summed .. upto] (None: the rest): the
        side-by-side phased exchange hands a bucket range to the process group as soon as BOTH micro-batches' backward passes have
        completed it, and the group reduces ONE buffer.  x + y in fp32, like the pair kernels of FusedAdam: same bits as the one-rank
        update.  On the current stream (the caller has joined the second micro-batch's stream)."""
        from . import hip
        nb = len(self.buckets)
        hi = nb - 1 if upto is None else min(upto, nb - 1)
        lo = self._sum_next
        if lo > hi or self.flat2 is None:
            self._sum_next = max(lo, hi + 1)
            return
        s, e = self.buckets[lo][0], self.buckets[hi][1]
        if self.flat.is_cuda:
            es = self.flat.element_size()
            hip.check(hip.lib().st5_axpby(self.flat2.data_ptr() + s * es, self.flat.data_ptr() + s * es, e - s, 1.0, 1.0, hip.F32, hip.stream()),
                      "st5_axpby")
        else:
            self.flat[s:e].add_(self.flat2[s:e])
        self.flat2[s:e].zero_()
        self._sum_next = hi + 1

    @contextlib.contextmanager
    def local_phase(self):
        """Forward + backward of ALL micro-batches of an update with the bucket triggers off: gradients stay local (both
        buffers of accumulate_overlapped allowed), nothing is enqueued on the process group -- so the whole phase can be captured
        into a HIP graph on several ranks too.  Follow with all_reduce_gradients() (eager)."""
        old, self._local_phase = self._local_phase, True
        try:
            yield
        finally:
            self._local_phase = old

    @contextlib.contextmanager
    def cut_points(self, bucket_ids):
        """Around the forward of the LAST micro-batch of an update (inside local_phase()): the autograd graph is cut at the
        boundaries that report the given buckets ready, so that the backward can be run -- and captured -- in PHASES, with a
        bucket range handed to the process group between two phases while the next phase computes (reduce_bucket_range).
        Yields the list of cuts, filled by the forward in forward order: [(bucket, tensor, leaf)]; run_phases() consumes it."""
        assert self._cut_set is None
        self._cut_set, self._cuts = set(bucket_ids), []
        try:
            yield self._cuts
        finally:
            self._cut_set = None

    @staticmethod
    def backward_phases(loss, cuts):
        """The backward of a forward built under cut_points(), as a list of callables in execution order: phase 0 runs from
        the loss to the LAST cut of the forward, phase k continues behind the k-th cut from the end.  Returns [(fn, bucket)]:
        after fn() every bucket up to and including `bucket` is complete (None for the final phase: everything is)."""
        real = [c for c in cuts if c[0] not in ("shared", "bypass")]
        shared = [c for c in cuts if c[0] == "shared"]      # (tag "shared" of _boundary: (_, producer output, [(region, leaf)]))
        bypass = [c for c in cuts if c[0] == "bypass"]      # (tag "bypass": (_, producer output, the loss term's leaf, region))
        m = len(real)
        carry = [None] * len(shared)     # per shared tensor: the folded gradient of the regions already run

        def phase(k):
            r = m - k                    # the region of the forward this phase runs backward through (r cuts precede it)

            def fn():
                roots, grads = [], []
                if k > 0:
                    _, x, leaf = real[r]
                    if leaf.grad is not None:          # (None: nothing behind this cut needed a gradient)
                        roots.append(x)
                        grads.append(leaf.grad)
                        leaf.grad = None
                mine = [next((lf for rg, lf in ent[2] if rg == r), None) for ent in shared]
                for i, lf in enumerate(mine):
                    if lf is not None and carry[i] is not None:
                        roots.append(lf)               # a root's gradient reaches the leaf's buffer before anything this phase computes
                        grads.append(carry[i])
                        carry[i] = None
                for _, bx, bleaf, rg in bypass:
                    if rg == r and k > 0 and bleaf.grad is not None:
                        roots.append(bx)               # the held-back gradient of a loss term's tap, with the region's main gradient
                        grads.append(bleaf.grad)
                        bleaf.grad = None
                if k == 0:
                    assert not roots
                    here = [c for c in bypass if c[3] == r]
                    if here:
                        # a tap in the loss's OWN region (no cut between them): first the taps' gradients alone, then one pass from the
                        # loss and the tapped tensors together (the short loss -> tap paths are walked twice; nothing else is)
                        leaves = [c[2] for c in here]
                        torch.autograd.backward([loss], inputs=leaves, retain_graph=True)
                        gs = [lf.grad for lf in leaves]
                        for lf in leaves:
                            lf.grad = None
                        keep = [(c[1], g) for c, g in zip(here, gs) if g is not None]
                        torch.autograd.backward([loss] + [x for x, _ in keep], [None] + [g for _, g in keep])
                        for lf in leaves:
                            lf.grad = None
                    else:
                        loss.backward()
                elif roots:
                    torch.autograd.backward(roots, grads)
                for i, (lf, ent) in enumerate(zip(mine, shared)):
                    if lf is None or lf.grad is None:
                        continue
                    g, lf.grad = lf.grad, None
                    if lf is ent[2][0][1]:             # the producer's own region: its backward, once, with the complete sum
                        torch.autograd.backward([ent[1]], [g])
                    else:
                        carry[i] = g
            return fn
        return [(phase(k), real[m - k - 1][0] if k < m else None) for k in range(m + 1)]

    def flush_deferred(self):
        """Fold the deferred reductions (split-K slabs, LayerNorm partials) queued on the current stream: gradients written so
        far are complete in the flat buffer afterwards (call at the end of a phase, inside the capture)."""
        self._flush_splitk()

    def reduce_bucket_range(self, upto):
        """Hand buckets [next .. upto] (None: all the rest) to the process group as ONE asynchronous all-reduce of their
        contiguous range of the flat buffer; the collective is ordered behind the current stream and runs on the group's own
        stream while the caller enqueues the next phase.  SUM over ranks (fold 1/world into the optimizer's grad_scale)."""
        nb = len(self.buckets)
        hi = nb - 1 if upto is None else min(upto, nb - 1)
        if self._next > hi:
            return
        s, e = self.buckets[self._next][0], self.buckets[hi][1]
        if self.collectives:
            self._works.append(dist.all_reduce(self.flat[s:e], group=self.pg, async_op=True))
        self._next = hi + 1

    def exchange_plan(self, phased=False, cuts=None):
        """Bytes of every all-reduce message of one update, in launch order: the bucket ranges of the phased exchange (cuts =
        PretrainUpdate.cut_buckets()), the buckets of the eager overlapped path, or the whole buffer (one message)."""
        es = self.flat.element_size()
        if phased and cuts is not None:
            out, lo = [], 0
            for hi in list(cuts) + [len(self.buckets) - 1]:
                hi = min(hi, len(self.buckets) - 1)
                if hi >= lo:
                    out.append((self.buckets[hi][1] - self.buckets[lo][0]) * es)
                    lo = hi + 1
            return out
        return [self.flat.numel() * es]

    def wait_reductions(self):
        for w in self._works:
            w.wait()
        self._reset_round()

    def all_reduce_gradients(self, average=True, payload=None):
        """The exchange step after a local_phase(): ONE all-reduce over the whole flat buffer (same call on every rank, so
        the order question of the bucketed path does not arise; 617 MB for Base: a single large message is what the xGMI
        links move fastest).  No overlap with the backward -- the price of replaying the backward as a graph, which saves more
        (bench.py).  average=False leaves the SUM (the caller folds 1/world into the optimizer's grad_scale)."""
        assert not self._accumulating and not self._works
        self.check_grad_views()
        self._flush_splitk()
        self.sum_gradient_buffers()
        if self.collectives:
            # async_op + wait(): the collective runs on the process group's OWN stream.  A synchronous call is enqueued on the
            # CURRENT stream and leaves its completion event there; when that stream later begins a graph capture, the group's
            # watchdog thread polls the event ("operation not permitted on an event last recorded in a capturing stream") and
            # takes the process down (measured: bench.py --exchange one_message, 1-rank RCCL group)
            if payload == torch.bfloat16 and self.flat.is_cuda:
                # bf16 gradient payload: half the bytes on the links -- 309 instead of 617 MB for Base.  The local
                # sum of the micro-batches stays fp32; it is rounded ONCE to bf16, summed over the ranks in bf16 by the collective and
                # widened again, and Adam accumulates its moments in fp32 as before.  An option (PretrainUpdate(exchange_payload=)):
                # the default exchange is exact fp32.
                from . import hip
                n = self.flat.numel()
                if getattr(self, "_flat_bf16", None) is None:
                    self._flat_bf16 = torch.empty(n, dtype=torch.bfloat16, device=self.flat.device)
                hip.check(hip.lib().st5_cast_from_f32(self.flat.data_ptr(), self._flat_bf16.data_ptr(), 1, n, 0, hip.BF16, hip.stream()), "st5_cast_from_f32")
                dist.all_reduce(self._flat_bf16, group=self.pg, async_op=True).wait()
                hip.check(hip.lib().st5_cast_to_f32(self._flat_bf16.data_ptr(), self.flat.data_ptr(), n, hip.BF16, hip.stream()), "st5_cast_to_f32")
            else:
                dist.all_reduce(self.flat, group=self.pg, async_op=True).wait()
            if average and self.world > 1:
                self.flat.mul_(1.0 / self.world)
        self._reset_round()

    def sum_gradient_buffers(self):
        """flat += flat2 (second micro-batch's gradients of accumulate_overlapped) when that sum is still outstanding."""
        from . import hip
        if self._pair_pending:
            if self.flat.is_cuda:
                # both buffers complete first: weight gradients may still sit in split-K slabs / on the weight-gradient stream
                self._flush_splitk()
                hip.check(hip.lib().st5_axpby(self.flat2.data_ptr(), self.flat.data_ptr(), self.flat.numel(), 1.0, 1.0, hip.F32, hip.stream()),
                          "st5_axpby")
            else:   # (host buffers: the gloo tests of this wrapper's bookkeeping on a stand-in model)
                self.flat.add_(self.flat2)
            self.flat2.zero_()
            self._pair_pending = False

    @contextlib.contextmanager
    def _grad_slot(self, slot):
        """param.grad -> views of gradient buffer `slot` (0 = the flat buffer, 1 = its twin) while a backward is enqueued."""
        self._grads_zeroed = False
        if slot == 0:
            yield
            return
        if self.flat2 is None:
            self._make_flat2()
        for p, v in zip(self.params, self._views2):
            p.grad = v
        try:
            yield
        finally:
            for p, v in zip(self.params, self._views1):
                p.grad = v

    def _make_flat2(self):
        self.flat2 = torch.zeros_like(self.flat)
        self._views2 = [self.flat2[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self._views1 = [p.grad for p in self.params]

    # -- step API ----------------------------------------------------------------------------------
    def close(self):
        """Undo the process-wide switches this wrapper turned on (deferred split-K reductions, layer-boundary hook)."""
        from . import hip
        if self.flat.is_cuda:
            Fn.set_wgrad_stream(None)
            Fn.set_attention_stream(None)
            if self._fwd_streams:
                hip.check(hip.lib().st5_gemm_set_deep_ring(256, 4), "st5_gemm_set_deep_ring")   # (the library default)
            Fn.set_wgrad_grouping(False)
            hip.check(hip.lib().st5_layernorm_defer(0, hip.stream()), "st5_layernorm_defer")
            hip.check(hip.lib().st5_gemm_defer_splitk(0, hip.stream()), "st5_gemm_defer_splitk")
        Fn.set_layer_boundary_hook(None)

    def _reset_round(self):
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._sum_next = 0
        self._works = []

    def zero_grad(self):
        assert not self._works, "zero_grad() between backward and finish(): all-reduces are in flight"
        Fn.drop_wgrads()     # (nothing may be queued across updates: an update that raised half way must not leak into this one)
        if self._grads_zeroed:
            self._grads_zeroed = False      # (FusedAdam's kernel zeroed them while reading)
        else:
            self.flat.zero_()
            if self.flat2 is not None:
                self.flat2.zero_()
        self._pair_pending = False
        self._reset_round()

    def check_grad_views(self):
        """`param.grad` must still be the view of the flat buffer this wrapper installed: model.zero_grad() /
        optimizer.zero_grad(set_to_none=True) drop it, after which the kernels would accumulate into fresh private tensors
        and the all-reduce / optimizer would keep reading zeros.  A dropped view is re-installed; a foreign tensor raises."""
        es = self.flat.element_size()
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
            elif g.data_ptr() != base + o * es:
                raise RuntimeError("a parameter's .grad no longer aliases FlatGradDataParallel.flat (was it replaced by "
                                   "loss.backward() after zero_grad(set_to_none=True)?): use ddp.zero_grad()")

    def finish(self):
        """Call after the LAST micro-batch's backward: reduce the remaining buckets (index order), wait for all, average
        over ranks."""
        assert not self._accumulating, "finish() inside no_sync()"
        self.check_grad_views()
        self._flush_splitk()
        if self.collectives:
            self.sum_gradient_buffers()
        if self.collectives:
            self._launch_in_order(True)
            for w in self._works:
                w.wait()
            if self.world > 1:
                self.flat.mul_(1.0 / self.world)
        self._reset_round()


def exchange_overlap_allowed(pg=None, warn=True):
    """May a collective run on the process group's stream UNDERNEATH this library's MFMA kernels?  On RCCL only with
    NCCL_ALGO=Ring: the ring Sum<float> device functions of the librccl torch ships contain no packed-fp32 VALU ops, the tree /
    PreMulSum ones do (profiles/r3_rccl_packed_fp32.txt), and packed-fp32 math beside another stream's MFMA waves returned stale
    lanes on this hardware.  Anything else (gloo: host-side reduction) is safe.  When the answer is no
    the callers serialise: the exchange goes out behind the backward (update.PretrainUpdate falls back to one message,
    FlatGradDataParallel._bucket_ready leaves every bucket to finish())."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return True
    if dist.get_backend(pg) != "nccl":
        return True
    ok = os.environ.get("NCCL_ALGO", "").strip().lower() == "ring"
    if not ok and warn:
        import warnings
        warnings.warn("speecht5_amd: NCCL_ALGO is not 'Ring' -- the gradient exchange will NOT be overlapped with the backward "
                      "(RCCL's tree / PreMulSum kernels contain packed-fp32 ops that are unsafe beside MFMA kernels on gfx950; "
                      "export NCCL_ALGO=Ring before the process group is created to get the overlapped exchange)")
    return ok


def agreed_exchange_overlap(pg=None, device=None):
    """exchange_overlap_allowed() of THIS rank, MIN-reduced over the group: the one answer every rank acts on."""
    ok = exchange_overlap_allowed(pg)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(pg) == 1:
        return ok
    on_dev = dist.get_backend(pg) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
    agreed = bool(int(flag.item()))
    if ok and not agreed:
        import warnings
        warnings.warn("speecht5_amd: another rank does not allow the overlapped gradient exchange (NCCL_ALGO differs between ranks): "
                      "every rank falls back to one message behind the backward")
    return agreed


def _fusion_ordered_parameters(module):
    """module.parameters() with the projections an attention block fuses into one GEMM made ADJACENT, in the order they are
    stacked ([Wq; Wk; Wv] / [Wk; Wv] and their biases), so that the optimizer's bf16 parameter image holds the stacked
    weight as one contiguous matrix (functional.bf16_mirror) and their gradients are one contiguous block too."""
    from .modules.multihead_attention import MultiheadAttention
    params = list(module.parameters())
    pos = {id(p): i for i, p in enumerate(params)}
    groups = []
    for m in module.modules():
        if isinstance(m, _CrossKV):
            groups += m.fused_groups()
        elif isinstance(m, MultiheadAttention):
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


class _CrossKV(torch.nn.Module):
    """Ordering / bucket holder (not part of the model tree): the k_proj / v_proj of every decoder layer's cross-attention.
    The decoder projects them with ONE GEMM (modules/decoder.py), so their weights (and biases) sit next to each other in the
    flat buffers in stacking order, and their gradients are complete only when the gradient of the encoder output is."""

    def __init__(self, layers):
        super().__init__()
        self.lins = torch.nn.ModuleList([lin for l in layers for lin in (l.encoder_attn.k_proj, l.encoder_attn.v_proj)])

    def fused_groups(self):
        g = [[lin.weight for lin in self.lins]]
        if all(lin.bias is not None for lin in self.lins):
            g.append([lin.bias for lin in self.lins])
        return g


class BucketGroup:
    """Modules whose parameters form one all-reduce bucket + the layer boundaries whose backward means "this bucket has
    received its last contribution": `triggers` = [(module, tag)] as passed to functional.layer_boundary(x, module, tag);
    default: the input boundary (tag None) of every module of the group."""

    def __init__(self, modules, triggers=None):
        self.modules = list(modules)
        self.triggers = triggers

    def trigger_keys(self):
        if self.triggers is None:
            return [(id(m), None) for m in self.modules]
        return [(id(m), tag) for m, tag in self.triggers]


def default_buckets(model):
    """Backward-completion order of a T5TransformerModel:
      0      decoder-side heads (mel / text post-nets)      ready when the gradient of the decoder output exists
      1..Ld  decoder layers, last first                      ready at each layer's input boundary
      Ld+1   encoder-side heads (HuBERT NCE head, quantizer) ready when the gradient of the encoder output is complete
             (every decoder layer's cross-attention and both heads hang off that tensor)
      ...    encoder layers, last first
    Everything else (pre-nets, the tied text embedding, layer-less encoder / decoder members) falls into the final bucket,
    which finish() reduces."""
    groups = []
    dec = getattr(model, "decoder", None)
    enc = getattr(model, "encoder", None)
    head = [m for m in (getattr(model, n, None) for n in ("speech_decoder_postnet", "text_decoder_postnet")) if m is not None]
    if head and dec is not None:
        groups.append(BucketGroup(head, triggers=[(dec, "out")]))
    if dec is not None and hasattr(dec, "layers"):
        groups += [BucketGroup([l]) for l in reversed(list(dec.layers))]
    ehead = [m for m in (getattr(model, n, None) for n in ("hubert_layer", "quantizer")) if m is not None]
    if dec is not None and hasattr(dec, "layers") and len(dec.layers) > 1 and all(getattr(l, "encoder_attn", None) is not None for l in dec.layers):
        ehead.append(_CrossKV(dec.layers))   # (listed after the decoder layers: this group owns these projections)
    # --unb-enc-layer >= 0: the decoder (cross-attention K/V) and the quantizer read an INTERMEDIATE encoder state taken before
    # the (enc, "out") boundary (speecht5.py: decoder_input), so that boundary's backward does not imply their gradients are
    # complete: those parameters then stay in the final bucket, which only finish() reduces
    if ehead and enc is not None and getattr(enc, "unb_enc_layer", -1) < 0:
        groups.append(BucketGroup(ehead, triggers=[(enc, "out")]))
    if enc is not None and hasattr(enc, "layers"):
        groups += [BucketGroup([l]) for l in reversed(list(enc.layers))]
    return groups


class FusedAdam:
    """fairseq `adam` (decoupled weight decay) + global-norm clipping on the flat buffers, one HIP kernel per step.
    Parameters are re-pointed at views of one flat fp32 buffer so that the update is a single launch."""

    def __init__(self, ddp: FlatGradDataParallel, lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=5.0):
        self.ddp = ddp
        self.lr, self.betas, self.eps, self.wd, self.clip = lr, betas, eps, weight_decay, clip_norm
        self.lr_step = None    # graph.StepGraph: this step's learning rate when self.lr already belongs to the next one
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
        # {lr, step} in device memory: what the kernel reads when the step runs from a captured graph (graph.StepGraph), whose
        # kernel arguments are frozen; refreshed from the host image before every replay
        self.hyper_dev = None
        self.hyper_host = None

    def enable_device_hyper(self):
        dev = self.pflat.device
        self.hyper_dev = torch.zeros(2, dtype=torch.float32, device=dev)
        self.hyper_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        self.hyper_host2 = None

    def ensure_second_hyper_image(self):
        if self.hyper_host2 is None:
            self.hyper_host2 = torch.zeros(2, dtype=torch.float32).pin_memory()

    def push_hyper(self, slot=0, lr=None):
        """Host -> device copy of (lr, next step count); call before the step that will read it.  `slot` picks one of two
        pinned images (the copy is asynchronous: a caller that runs ahead of the GPU alternates them, graph.StepGraph).  `lr`:
        the value to push when it is not self.lr at this moment (a schedule advanced on another thread)."""
        if slot:
            self.ensure_second_hyper_image()
        h = self.hyper_host2 if slot else self.hyper_host
        h[0] = float(self.lr if lr is None else lr)
        h[1] = float(self.t + 1)
        self.hyper_dev.copy_(h, non_blocking=True)

    def backward(self, loss):
        loss.backward()

    def step(self, grad_scale=1.0):
        """One update; CONSUMES the gradients (both buffers are left zeroed by the kernel).  The device copy of (lr, step) is
        read only by a step that is being captured into a HIP graph (its kernel arguments are frozen; graph.StepGraph refreshes
        the two floats before every replay): an eagerly enqueued step -- warm-up, a batch that cannot be replayed, the eager tail
        of a several-rank replay -- passes them by value, so it can never run with a stale or never-pushed device image."""
        from . import hip
        self.t += 1
        L = hip.lib()
        lr = self.lr if self.lr_step is None else self.lr_step
        g = self.ddp.flat
        hyper = hip.ptr(self.hyper_dev) if (self.hyper_dev is not None and g.is_cuda and torch.cuda.is_current_stream_capturing()) else 0
        if self.ddp._pair_pending and self.ddp.world == 1:
            # two gradient buffers (ddp.accumulate_overlapped): norm and update over g + g2 in the kernels themselves, both
            # buffers left zeroed -- no "g += g2" pass, no fills before the next update
            g2 = self.ddp.flat2
            if self.clip > 0:
                hip.check(L.st5_sumsq_pair(g.data_ptr(), g2.data_ptr(), self.gnorm_sq.data_ptr(), g.numel(), 1.0, 0, hip.stream()), "st5_sumsq_pair")
            hip.check(L.st5_adam_step_pair(self.pflat.data_ptr(), g.data_ptr(), g2.data_ptr(), 1, self.m.data_ptr(), self.v.data_ptr(), g.numel(),
                                           lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                           self.gnorm_sq.data_ptr() if self.clip > 0 else 0, self.clip, grad_scale,
                                           self.wflat.data_ptr() if self.mirror is not None else 0, hyper, hip.stream()),
                      "st5_adam_step_pair")
            self.ddp._pair_pending = False
            self.ddp._grads_zeroed = True
        else:
            self.ddp.sum_gradient_buffers()
            if self.clip > 0:
                hip.check(L.st5_sumsq(g.data_ptr(), self.gnorm_sq.data_ptr(), g.numel(), 1.0, 0, hip.F32, hip.stream()), "st5_sumsq")
            # (one buffer; the kernel leaves it zeroed in passing -- the second buffer, if there is one, is zero whenever no sum
            # is pending -- so the next zero_grad() has no fill to launch)
            hip.check(L.st5_adam_step_pair(self.pflat.data_ptr(), g.data_ptr(), 0, 1, self.m.data_ptr(), self.v.data_ptr(), g.numel(),
                                           lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                           self.gnorm_sq.data_ptr() if self.clip > 0 else 0, self.clip, grad_scale,
                                           self.wflat.data_ptr() if self.mirror is not None else 0, hyper, hip.stream()),
                      "st5_adam_step_pair")
            self.ddp._grads_zeroed = True
        # parameters changed in place through the flat view: invalidate the compute-dtype weight cache (entries that do
        # not come from the bf16 image: conv / fp32 / non-adjacent stacks) and refresh the transposed copies
        Fn.weight_cache.clear()
        if self.mirror is not None:
            for p in self.ddp.params:
                p._st5_mver = p._version
            if os.environ.get("ST5_LAZY_TRANSPOSES") == "1":
                # (A/B, round 6: leave the refresh to the first data-gradient GEMM of the next update -- off the serial tail, onto the
                #  shorter micro-batch chain.  Measured: 30.5 against 30.15 ms per update, same box, alternating: the one extra
                #  cross-stream event inside the replayed graph costs more than the 0.24 ms it moves.  Off.)
                self.mirror.mark_stale()
                Fn.fp8_mirror.off = True
            else:
                self.mirror.refresh_transposes()
                Fn.fp8_mirror.refresh()     # (fp8 mode: every weight's e4m3 image + block scales in one launch; no-op otherwise)
