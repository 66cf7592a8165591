"""One optimizer update of the pre-training recipe (`--update-freq 2`: a speech micro-batch and a text micro-batch, SURVEY.md
3.1-3.2) the way bench.py times it: forward + backward of both micro-batches, the gradient exchange over the ranks, global-norm clip
and the fused Adam step -- enqueued eagerly or replayed as a HIP graph.  bench.py and tests/test_bench_update_gpu.py build their step
from THIS class, so the thing that is timed is the thing whose results are checked.

Modes (what the reference's trainer does in turn -- tasks/speecht5.py:519-556 called once per micro-batch by fairseq's
Trainer.train_step, gradients summed, one optimizer step -- is `micro="in_turn"`; the other modes reorder execution, never
arithmetic: every mode produces the SAME bits, tests/test_bench_update_gpu.py and tests/test_replay_long_gpu.py):

  micro      "side_by_side"  (default) both micro-batches on two streams, forward and backward, two gradient buffers summed inside the
                             Adam kernel (a + b == b + a bit for bit).  31.5 ms per update against 39.5 in turn on one MI355X: the
                             two branches fill each other's tile-quantisation tails and launch gaps.  Rounds 3-4 kept this mode off
                             because ~1 % of its replays differed; round 5 found the cause -- one missing `s_waitcnt lgkmcnt(0)` in
                             front of a barrier of the attention backward -- and 300 replayed updates now
                             reproduce the in-turn trajectory bit for bit in every process
             "in_turn"       one stream, one gradient buffer, ddp.accumulate: the reference trainer's order
             "in_turn_2buf"  two streams / two gradient buffers, the second backward ordered behind the first
  graph      True: the update (one rank) or its local phase (several ranks) is captured once and replayed
  several ranks, graph: the captured part is the LOCAL phase (zero_grad, both micro-batches); exchange="phased" (default) cuts it
             into three graphs and all-reduces each completed bucket range under the next one -- with micro="side_by_side" (round 6)
             both micro-batches run inside every phase on two streams and a range is summed over the two gradient buffers before it
             goes out; "one_message": one graph and one all-reduce of the whole buffer behind it; Adam follows eagerly
"""
import os

import torch
import torch.distributed as dist

from . import functional as Fn


class PretrainUpdate:
    def __init__(self, task, model, criterion, micro_batches, *, lr=2e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.01, clip_norm=5.0,
                 graph=True, micro="side_by_side", wgrad_stream=None, prefetch_host=True, device=None, lr_fn=None, exchange="phased", exchange_payload="fp32"):
        from .ddp import FlatGradDataParallel, FusedAdam
        assert micro in ("side_by_side", "in_turn_2buf", "in_turn")
        self.task, self.model, self.crit, self.micro = task, model, criterion, list(micro_batches)
        self.mode = micro
        self.use_graph = graph
        self.device = device if device is not None else next(model.parameters()).device
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # no weight-gradient stream, no attention helper stream: beside two micro-batch streams they cost time (42.0 against 31.5 ms,
        # profiles/r5_replay_hunt.txt); results are the same bits with and without them
        if wgrad_stream is None:
            wgrad_stream = False
        self.ddp = FlatGradDataParallel(model, wgrad_stream=wgrad_stream)
        self.opt = FusedAdam(self.ddp, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_norm=clip_norm)
        self.lr_fn = lr_fn
        self.prefetch_host = prefetch_host
        assert exchange_payload in ("fp32", "bf16")
        # several ranks, one-message exchange: the gradient buffer travels as bf16 (half the link bytes; local sums and Adam stay fp32)
        self.exchange_payload = torch.bfloat16 if exchange_payload == "bf16" else None
        self._payload_asked = exchange_payload
        # several ranks + graph: graph = local phase, eager tail = all-reduce + Adam.  Several ranks WITHOUT a graph and two gradient
        # buffers (micro != "in_turn"): the same split, enqueued eagerly -- the bucket triggers of the eager overlapped path need ONE
        # buffer of summed gradients, so accumulate_overlapped() is only legal inside local_phase() (`--gpus N --no-graph`
        # with the default micro-batch mode used to trip that assert)
        self.split = self.ddp.collectives and (graph or micro != "in_turn")
        # several ranks + graph + one stream: the exchange is OVERLAPPED with the last micro-batch's backward -- the local phase
        # is captured as three graphs cut at bucket boundaries (ddp.cut_points), and after each of them the bucket range it
        # completed goes to the process group as one asynchronous all-reduce (RCCL's stream) while the next graph runs
        # (exchange="one_message": one graph + one all-reduce of the whole buffer afterwards, the round-2 form)
        # (ST5_EAGER_PHASED=1: the same three phases enqueued eagerly -- the tests' reference point)
        # micro="side_by_side" (round 6, the default form): the same phases with BOTH micro-batches inside each of them, on two
        # streams forked from and joined into the phase's graph; a bucket range goes out once both backward passes have completed it
        # and its two gradient buffers are summed (phase_fns_side_by_side)
        self.phased = ((self.split or (self.ddp.collectives and os.environ.get("ST5_EAGER_PHASED") == "1"))
                       and micro in ("in_turn", "side_by_side") and exchange == "phased" and len(self.micro) >= 1
                       and (micro == "in_turn" or len(self.micro) == 2))
        if self.phased and not self.ddp.overlap_exchange:    # (decided once per group in FlatGradDataParallel, the same on every rank)
            self.phased = False      # one message behind the local phase: no collective beside the backward's kernels
        if self.exchange_payload is not None and (self.phased or not self.split):
            # (the bf16 payload exists for the one-message exchange only; silently exchanging fp32 is not what was asked)
            import warnings
            warnings.warn("speecht5_amd: exchange_payload='bf16' applies to the one-message exchange only (exchange='one_message' with a "
                          "replayed or split update); this update exchanges fp32 gradients "
                          f"({'phased' if self.phased else 'bucketed, eager'} form)")
            self.exchange_payload = None
        self._ph = None
        self.n = 0            # update counter (fairseq's num_updates)
        self.sg = None
        # Every update is enqueued on a stream of its own, never on the legacy NULL stream: on this runtime (ROCm 7.2) work issued
        # to the NULL stream from two host threads (the forward from the caller's thread, the backward from autograd's worker
        # thread) is NOT kept in issue order once another stream is busy -- measured: with the speech micro-batch on the NULL
        # stream and the text micro-batch beside it, the reduction kernel of conv layer 0's backward read partials its
        # predecessor in the same stream had not written yet (round 3, docs/HISTORY.md 4a: NaN-poisoned partials came through in 4 of 4
        # updates; 0 of 4 on an explicit stream).
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    # -- pieces ---------------------------------------------------------------------------------------------------------
    def _fwd(self, s):
        return self.task.forward_loss(s, self.model, self.crit, self.n)

    def local_part(self):
        """(several ranks, graph) what the graph holds: gradients of this rank's micro-batches, summed into ddp.flat."""
        self.ddp.zero_grad()
        with self.ddp.local_phase():
            if self.mode == "in_turn":
                self.ddp.accumulate(self.micro, lambda s: self.task.train_step(s, self.model, self.crit, None, self.n, sync=False))
            else:
                self.ddp.accumulate_overlapped(self.micro, self._fwd, backward="in_turn" if self.mode == "in_turn_2buf" else "side_by_side")
        self.ddp.sum_gradient_buffers()
        # the deferred reductions (split-K slabs, LayerNorm partials) are folded INSIDE the captured part: their descriptors are
        # host state of the library, consumed by the first flush -- left to the eager tail, the first replay's tail would fold
        # them and every later replay's slabs would never reach the gradient buffer
        self.ddp.flush_deferred()

    # -- phased local part (several ranks, overlapped exchange) ------------------------------------------------------------
    def cut_buckets(self):
        """Where the backward is cut: behind the decoder (+ both heads, the shared cross-attention K/V projection: ~60 M parameters
        complete) and inside the encoder stack -- at the input of layer L/3 (Base: layer 4), so that the LAST message, the one nothing
        is left to hide, is the smallest: 234 / 227 / 156 MB for Base (round 5 cut at L/2: 234 / 170 / 213 MB).  ST5_ENC_CUT=<layer>
        moves the second cut (A/B)."""
        mb = self.ddp.module_bucket
        enc = getattr(self.model, "encoder", None)
        cuts = []
        if enc is not None:
            if (id(enc), "out") in mb:
                cuts.append(mb[(id(enc), "out")])
            layers = list(getattr(enc, "layers", []))
            at = int(os.environ.get("ST5_ENC_CUT", len(layers) // 3))
            if len(layers) >= 4 and 0 < at < len(layers) and (id(layers[at]), None) in mb:
                cuts.append(mb[(id(layers[at]), None)])
        return sorted(set(cuts))

    def phase_fns(self):
        """The local phase as len(cuts) + 1 callables and, behind each, the hand-over of a bucket range to the process group.
        The bucket ranges are STATIC (cut_buckets() is a function of the model alone, the same list on every rank): message k
        covers the buckets up to cuts[k], the last one the rest.  What this rank's forward materialised only decides how many of
        its own backward phases run inside nominal phase k (a cut whose input needed no gradient on this rank does not
        materialise: the phase in front of it then runs through, and the nominal phase behind it is empty) -- the number and the
        extent of the collectives never depend on per-rank state."""
        ddp, cuts_b = self.ddp, self.cut_buckets()
        nph = len(cuts_b) + 1
        nb = len(ddp.buckets)

        def run_until(b):     # this rank's backward phases until bucket b (None: every bucket) is complete
            while self._ph_done < len(self._ph) and (b is None or self._ph_complete < b):
                fn, upto = self._ph[self._ph_done]
                with ddp.local_phase():
                    fn()
                self._ph_complete = nb - 1 if upto is None else upto
                self._ph_done += 1

        def first():
            ddp.zero_grad()
            with ddp.local_phase():
                for mb in self.micro[:-1]:
                    self.task.train_step(mb, self.model, self.crit, None, self.n, sync=False)
                with ddp.cut_points(cuts_b) as cuts:
                    loss = self._fwd(self.micro[-1])
                self._ph = ddp.backward_phases(loss, cuts)
            self._ph_done, self._ph_complete = 0, -1
            fn, upto = self._ph[0]           # (from the loss to the last materialised cut: always runs here)
            with ddp.local_phase():
                fn()
            self._ph_complete = nb - 1 if upto is None else upto
            self._ph_done = 1
            run_until(cuts_b[0] if cuts_b else None)
            ddp.flush_deferred()

        def later(k):
            def fn():
                run_until(cuts_b[k] if k < len(cuts_b) else None)
                ddp.flush_deferred()
            return fn

        def reduce_after(k):
            def fn():
                ddp.reduce_bucket_range(cuts_b[k] if k < len(cuts_b) else None)
            return fn
        return [first] + [later(k) for k in range(1, nph)], [reduce_after(k) for k in range(nph)]

    def phase_fns_side_by_side(self):
        """phase_fns() for the two micro-batches SIDE BY SIDE: every phase forks the second micro-batch's stream from the phase's own
        and joins it again, so each phase is one graph with two branches (the form the one-graph local phase already has); both
        forward passes run under cut_points() and keep their own list of backward phases; nominal phase k runs each micro-batch's
        backward, on its stream and into its gradient buffer, until bucket cuts[k] is complete in BOTH, folds each stream's deferred
        reductions, joins, sums the two buffers over the completed bucket range (ddp.sum_bucket_range: flat += flat2, fp32, the sum the
        pair kernels of the one-rank optimizer step form) -- and the caller hands that range of `flat` to the process group while the
        next phase's graph runs.  Bucket ranges are static (see phase_fns)."""
        import torch
        ddp, cuts_b = self.ddp, self.cut_buckets()
        nph = len(cuts_b) + 1
        nb = len(ddp.buckets)
        st = {}

        def streams():
            cur = ddp.pair_streams(2)           # (forks the second stream from the current one)
            return cur, [cur, ddp._fwd_streams[0]]

        def run_until(i, b):                    # micro-batch i's backward phases until bucket b (None: every bucket) is complete
            ph = st["ph"][i]
            while st["done"][i] < len(ph) and (b is None or st["complete"][i] < b):
                fn, upto = ph[st["done"][i]]
                fn()
                st["complete"][i] = nb - 1 if upto is None else upto
                st["done"][i] += 1

        def backward_to(k, first=False, forked=None):
            cur, ss = forked if forked is not None else streams()
            for i in (0, 1):
                with ddp._grad_slot(i), torch.cuda.stream(ss[i]), ddp.local_phase():
                    if first:
                        fn, upto = st["ph"][i][0]      # (from the loss to the last materialised cut: always runs here)
                        fn()
                        st["complete"][i] = nb - 1 if upto is None else upto
                        st["done"][i] = 1
                    run_until(i, cuts_b[k] if k < len(cuts_b) else None)
                    ddp.flush_stream_deferred()
            cur.wait_stream(ss[1])
            ddp.sum_bucket_range(cuts_b[k] if k < len(cuts_b) else None)

        def first():
            ddp.zero_grad()
            cur, ss = streams()
            st["ph"], st["done"], st["complete"] = [None, None], [0, 0], [-1, -1]
            with ddp.local_phase():
                for i, mb in enumerate(self.micro):
                    with torch.cuda.stream(ss[i]), ddp.cut_points(cuts_b) as cuts:
                        loss = self._fwd(mb)
                    st["ph"][i] = ddp.backward_phases(loss, cuts)
            backward_to(0, first=True, forked=(cur, ss))     # (no join between forward and backward: each chain runs on)

        def later(k):
            def fn():
                backward_to(k)
                if k == nph - 1:
                    st["ph"] = None             # (the autograd graphs of this update may go)
            return fn

        defer = os.environ.get("ST5_PHASED_DEFER_EXCHANGE") == "1"    # (diagnostic: the same three graphs, ONE message behind the last)

        def reduce_after(k):
            def fn():
                if defer and k < nph - 1:
                    return
                ddp.reduce_bucket_range(cuts_b[k] if (k < len(cuts_b) and not defer) else None)
            return fn
        self._ph_sbs = st
        return [first] + [later(k) for k in range(1, nph)], [reduce_after(k) for k in range(nph)]

    def finish_exchange_and_update(self):
        self.ddp.check_grad_views()
        self.ddp.wait_reductions()
        self.opt.step(grad_scale=1.0 / (len(self.micro) * self.world))

    def exchange_and_update(self):
        """(several ranks, graph) eager tail: sum over ranks, then mean over ranks and micro-batches inside Adam."""
        self.ddp.all_reduce_gradients(average=False, payload=self.exchange_payload)
        self.opt.step(grad_scale=1.0 / (len(self.micro) * self.world))

    def step(self):
        """One update enqueued on the current stream (no host synchronisation)."""
        if self.phased:
            for fn, bt in zip(*(self.phase_fns() if self.mode == "in_turn" else self.phase_fns_side_by_side())):
                fn()
                bt()
            self.finish_exchange_and_update()
            return
        if self.split:
            self.local_part()
            self.exchange_and_update()
            return
        ddp = self.ddp
        ddp.zero_grad()
        if self.mode == "in_turn":
            # --update-freq 2: gradients of the first micro-batch only accumulate (no_sync); the bucket all-reduces are launched
            # from the backward of the LAST micro-batch, each bucket once, after its last local contribution
            ddp.accumulate(self.micro, lambda s: self.task.train_step(s, self.model, self.crit, None, self.n, sync=False))
        else:
            ddp.accumulate_overlapped(self.micro, self._fwd, backward="in_turn" if self.mode == "in_turn_2buf" else "side_by_side")
        ddp.finish()
        self.opt.step(grad_scale=1.0 / len(self.micro))

    def advance(self):
        """Host-side state a replayed step does not touch: the update counter (quantizer temperature, freeze counters), lr."""
        self.n += 1
        self.model.set_num_updates(self.n)
        if self.lr_fn is not None:
            self.opt.lr = self.lr_fn(self.n)

    # -- driving --------------------------------------------------------------------------------------------------------
    def eager_update(self):
        self.advance()
        if self.stream is None:
            self.step()
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.step()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def prepare_graph(self):
        """Two recording updates + capture (three updates' worth of host random draws; two of them executed)."""
        from .graph import StepGraph
        if self.phased:
            phases, between = self.phase_fns() if self.mode == "in_turn" else self.phase_fns_side_by_side()
            self.sg = StepGraph(None, opt=self.opt, model=self.model, device=self.device, on_step=self.advance,
                                prefetch_host=self.prefetch_host, after_fn=self.finish_exchange_and_update, stream=self.stream,
                                phases=phases, between=between)
        else:
            self.sg = StepGraph(self.local_part if self.split else self.step, opt=self.opt, model=self.model, device=self.device,
                                on_step=self.advance, prefetch_host=self.prefetch_host,
                                after_fn=self.exchange_and_update if self.split else None, stream=self.stream)
        self.sg.record()
        self.sg.record()
        self.sg.capture()
        return self.sg

    def update(self):
        """The next update: a replay when a graph exists, else eager."""
        if self.sg is not None:
            with torch.cuda.stream(self.sg.stream):
                self.sg.replay()
        else:
            self.eager_update()

    def finish(self):
        if self.sg is not None:
            self.sg.drain()
            torch.cuda.current_stream(self.device).wait_stream(self.sg.stream)

    def close(self):
        self.finish()
        self.ddp.close()
        if self.sg is not None:      # break the cycle update -> graph -> bound step function -> update: a HIP graph object must
            self.sg.step_fn = self.sg.after_fn = self.sg.on_step = self.sg.phases = self.sg.between = None   # not wait for the cyclic
            self.sg.graph = self.sg.graphs = None    # collector (a collection during a LATER capture destroys it mid-capture)
            self._ph = None
            self._ph_sbs = None
            self.sg = None

    def state(self):
        torch.cuda.synchronize(self.device)
        return self.opt.pflat.clone(), self.opt.m.clone(), self.opt.v.clone(), self.opt.t
