"""One optimizer update as a HIP graph (hipStreamBeginCapture through torch.cuda.CUDAGraph): the step's ~2000 kernel launches
are enqueued once and replayed with a single call, which takes the host (36-48 ms of Python / ctypes enqueue per step, more
than the GPU needs once the kernels are fast) off the critical path and removes the launch gaps between dependent kernels.

What makes a training step replayable:
  * dropout seeds live in device memory (functional.SeedSlots: a `seed` argument with bit 63 set is a pointer, csrc/common.h
    resolve_seed), refilled before every replay with the values the eager path's host counter would have produced;
  * host-produced step inputs (HuBERT span mask and code-book time-mix from the CPU RNGs, target index maps) go through
    functional.stage_host: persistent device buffers + pinned host images, refreshed before every replay in recording order,
    so the numpy / torch CPU random streams advance exactly as in eager mode;
  * data-dependent shapes are replaced by fixed-shape forms while recording (functional.static_shapes(): the NCE head scores
    every frame and passes selection masks; the sample size becomes a device scalar);
  * the optimizer reads the learning rate and step count from device memory (FusedAdam.enable_device_hyper).
  * LayerDrop: the per-layer host draws are staged to the device as keep flags and the layer outputs are selected on the
    device (functional.layerdrop_select; modules/encoder.py, modules/decoder.py), so the recipe's --encoder-layerdrop /
    --decoder-layerdrop 0.05 replays (a dropped layer still runs; its output and gradients are discarded).
Not supported in a captured step: host reads of device values, collectives.  Several ranks: the captured part is the LOCAL phase of the update (zero_grad, forward / backward of every
micro-batch: ddp.local_phase()), and `after_fn` -- gradient all-reduce + optimizer step -- is enqueued eagerly behind every
replay (a handful of launches)."""
import torch

from . import functional as Fn


class StepGraph:
    def __init__(self, step_fn, opt=None, model=None, device=None, seed_slots=1024, on_step=None, prefetch_host=False, after_fn=None,
                 stream=None, phases=None, between=None):
        """step_fn(): one full update on the current stream -- zero_grad, forward/backward of every micro-batch, finish,
        optimizer step -- without host synchronisation.  `opt`: the FusedAdam whose (lr, step) must follow the host.
        after_fn(): optional eager tail of the update, NOT captured (then step_fn must not contain the optimizer step: after_fn
        does, after the collectives)."""
        # phases / between (several ranks, overlapped exchange): the update's local part as a LIST of callables, each captured
        # into its own graph (one memory pool, replayed in order), and after each of them an eagerly enqueued callable -- the
        # asynchronous all-reduce of the bucket range that phase completed -- so the collective runs on the process group's
        # stream while the next phase's graph executes.  step_fn is then None.
        self.phases = list(phases) if phases is not None else None
        self.between = list(between) if between is not None else None
        assert self.phases is None or (step_fn is None and len(self.phases) == len(self.between))
        self.graphs = None
        self.step_fn = step_fn
        self.after_fn = after_fn
        self.on_step = on_step   # host-side bookkeeping a replay skips (e.g. model.set_num_updates(n)): called before every step
        self.opt = opt
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.graph = None
        self._assumes_zeroed = False
        # prefetch_host: hipGraphLaunch keeps the calling thread until the previous launch of the same graph has drained (measured:
        # graph.replay() returns after 40-60 ms), so the host part of the NEXT step's preparation (on_step, seed values, the CPU
        # producers of the staged inputs: ~1.2 ms) would sit between two replays with the GPU idle.  With prefetch_host it runs on
        # a helper thread while the launch call blocks (the call releases the GIL), into the second pinned image of every staged
        # buffer; replay() then only enqueues the uploads.  The CPU random streams are drawn in the same order, one step EARLIER
        # than without (after the last replay one prepared, unused step has advanced them).
        self.prefetch_host = prefetch_host
        self._ahead = None      # (thread, slot) preparing the next step
        self._slot = 0
        self._lr = [None, None]  # the learning rate on_step() left behind, per pinned slot (read on the thread that ran on_step)
        self.trace = None        # tests: a list -> one record per replay of what was uploaded (digests of the pinned images)
        self._uploaded = [None, None]   # event after the last upload from each pinned slot (the slot may be rewritten only after it)
        self.slots = Fn.SeedSlots(seed_slots, self.device)
        self.staging = Fn.HostStaging()   # per graph: two graphs (two tests, two shapes) never share a staging sequence
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)   # (never the NULL stream)
        if opt is not None and opt.hyper_dev is None:
            opt.enable_device_hyper()

    # -- one step in a given staging mode ------------------------------------------------------------------------------
    def _enter(self, mode):
        Fn._S.slots = self.slots
        self._prev_staging, Fn.staging = Fn.staging, self.staging     # this graph's own sequence of staged inputs / host draws
        self.staging.begin_step(mode)

    def _exit(self):
        Fn._S.slots = None
        self.staging.mode = None
        Fn.staging = self._prev_staging

    def _run(self, mode):
        self._enter(mode)
        try:
            if self.phases is None:
                self.step_fn()
            else:
                for fn, bt in zip(self.phases, self.between):
                    fn()
                    if mode == "record":
                        bt()
        finally:
            self._exit()

    def record(self):
        """Eager step that allocates the static buffers and counts the seed slots (also the last warm-up step)."""
        # on the capture stream: per-stream state of the library (split-K slab workspaces) and of the allocator is created
        # here, eagerly -- nothing may allocate device memory once capture has begun
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self.on_step is not None:
                self.on_step()
            if self.opt is not None:
                self.opt.push_hyper()
            self.slots.begin_step()
            self._run("record")
            if self.after_fn is not None:
                self.after_fn()
        cur.wait_stream(self.stream)
        if not self.slots.used:
            self.slots.used = self.slots.k
            Fn._S.counter -= self.slots.n - self.slots.used   # begin_step() reserved all slots: give the unused ones back
        else:
            assert self.slots.k == self.slots.used, "the step's dropout call sequence changed"

    def capture(self):
        torch.cuda.synchronize(self.device)
        self.staging.pack()      # (all staged inputs in one block: one upload per replay instead of one per tensor)
        with torch.cuda.stream(self.stream):
            self._pre_replay()
        torch.cuda.synchronize(self.device)
        import gc
        gc.collect()    # (a graph object freed by the cyclic collector DURING a capture aborts the process)
        self.graph = torch.cuda.CUDAGraph()
        t0 = self.opt.t if self.opt is not None else 0
        # does the captured zero_grad() contain the fills, or does it rely on the previous optimizer step having left the
        # gradient buffers zeroed (ddp._grads_zeroed at this moment)?  replay() re-establishes that precondition when the host
        # flag says the buffers were written in between (an eager backward without a step, a failed step ...)
        ddp = getattr(self.opt, "ddp", None)
        self._assumes_zeroed = bool(ddp is not None and ddp._grads_zeroed)
        try:
            if self.phases is None:
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self._run("capture")
            else:
                pool = torch.cuda.graph_pool_handle()
                self.graphs = []
                self._enter("capture")
                try:
                    for fn in self.phases:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool, stream=self.stream, capture_error_mode="thread_local"):
                            fn()
                        self.graphs.append(g)
                finally:
                    self._exit()
        except BaseException:
            # a failed capture usually ends the process in the graph's destructor ("operation not permitted when stream is
            # capturing") before Python prints the cause: show it here, while the graph object is still referenced
            import sys
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            raise
        # the capture pass itself executed nothing: the host-side step counter it advanced is rolled back
        if self.opt is not None and self.after_fn is None:
            self.opt.t = t0
        assert self.slots.k == self.slots.used, "the captured step used a different number of dropout seeds than the recorded one"
        if self.prefetch_host:
            # every pinned image the helper thread will write exists before it first runs: the helper never allocates pinned
            # memory (hipHostMalloc from a second host thread beside a graph launch) and never touches the HIP runtime at all
            self.slots.ensure_second_image()
            self.staging.ensure_second_images()
            if self.opt is not None:
                self.opt.ensure_second_hyper_image()
        self._pending = True   # buffers are already staged for the first replay

    def _produce(self, slot):
        """Host half of the preparation of one step (no device work)."""
        if self.on_step is not None:
            self.on_step()
        if self.opt is not None:
            self._lr[slot] = float(self.opt.lr)   # (snapshot on the producing thread: the upload never reads a half-advanced schedule)
        self.slots.produce(slot)
        self.staging.produce(slot)

    def _upload(self, slot):
        self.slots.upload(slot)
        self.staging.upload(slot)
        if self.opt is not None:
            self.opt.push_hyper(slot, lr=self._lr[slot])
        if self.trace is not None:
            self.trace.append({"slot": slot, "seeds": self.slots.digest(slot), "staged": self.staging.digests(slot),
                               "lr": self._lr[slot], "t": (self.opt.t + 1) if self.opt is not None else None})
        # the uploads are asynchronous reads of pinned host images: remember when this slot's have executed (a host that runs
        # ahead of the GPU -- small steps -- must not rewrite the images before that)
        ev = torch.cuda.Event()
        ev.record()
        self._uploaded[slot] = ev

    def _pre_replay(self):
        if self._uploaded[0] is not None:
            self._uploaded[0].synchronize()
        self._produce(0)
        self._upload(0)

    def replay(self):
        """One replayed update.  It is ALWAYS enqueued on this graph's own stream (uploads of the staged inputs, the graph
        launch, the eager tail), whatever stream the caller is on -- in particular never on the legacy NULL stream, which this
        runtime does not keep in issue order when more than one host thread is alive beside it; the
        caller's stream is ordered before and behind the replay, so for the caller it behaves like work on its own stream."""
        cur = torch.cuda.current_stream(self.device)
        if cur == self.stream:
            return self._replay()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._replay()
        cur.wait_stream(self.stream)

    def _replay(self):
        used = 0            # the pinned slot this step's inputs came from
        if self._pending:
            self._pending = False
        elif self._ahead is not None:
            th, used = self._ahead
            th.join()
            self._ahead = None
            if self._ahead_error is not None:
                raise self._ahead_error
            self._upload(used)
        else:
            self._pre_replay()
        step_lr = self._lr[used]
        if self.prefetch_host:
            import threading
            self._slot ^= 1
            self._ahead_error = None
            if self._uploaded[self._slot] is not None:
                self._uploaded[self._slot].synchronize()

            def work(slot=self._slot):
                try:
                    self._produce(slot)
                except BaseException as e:   # surfaced by the next replay()
                    self._ahead_error = e
            th = threading.Thread(target=work, daemon=True)
            th.start()
            self._ahead = (th, self._slot)
        ddp = getattr(self.opt, "ddp", None)
        if self._assumes_zeroed and ddp is not None and not ddp._grads_zeroed:
            ddp.flat.zero_()
            if ddp.flat2 is not None:
                ddp.flat2.zero_()
            ddp._pair_pending = False
        if self.graphs is not None:
            for g, bt in zip(self.graphs, self.between):
                g.replay()
                bt()
        else:
            self.graph.replay()
        if self.after_fn is not None:
            # the eager tail's optimizer step passes lr BY VALUE; by now the helper thread may have advanced the schedule to the
            # next step (on_step runs one step early there): the tail reads this step's snapshot
            if self.opt is not None:
                self.opt.lr_step = step_lr
            try:
                self.after_fn()          # (its optimizer step advances opt.t itself)
            finally:
                if self.opt is not None:
                    self.opt.lr_step = None
        elif self.opt is not None:
            self.opt.t += 1
            if ddp is not None:
                ddp._grads_zeroed = True     # the replayed Adam kernel left both buffers zeroed

    def drain(self):
        """Wait for the helper thread that prepares the next step (prefetch_host): after the last replay it has drawn one unused
        step's worth of host random numbers; whoever re-seeds the CPU generators afterwards must not race with it."""
        if self._ahead is not None:
            self._ahead[0].join()

    def __call__(self):
        if self.graph is None:
            self.record()
            self.record()   # second recording step: confirms that the slot / staging sequence repeats
            self.capture()
        self.replay()
