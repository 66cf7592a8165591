"""Autograd functions of the SpeechT5 hot path, implemented on libspeecht5_hip.so (no torch math).

Every forward/backward below is a sequence of C-ABI kernel launches on the current HIP stream;
PyTorch only owns the buffers and threads the autograd graph.  Activations live in the global
compute dtype (bf16 for training/benchmarks, fp32 for the parity mode); parameters, statistics and
weight gradients are fp32.  Weight gradients are accumulated straight into `param.grad` by the
wgrad GEMM epilogue (beta = 1), so no temporary full-size gradient tensors exist; a registered
callback (`set_grad_ready_hook`) tells the data-parallel layer when a parameter's gradient is final.

Row convention: activations are 2-D `[rows, channels]` with rows ordered batch-major (b, t).
"""
import math
from types import SimpleNamespace

import os
import threading

import torch

from . import hip
from .hip import ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH  # noqa: F401

_S = SimpleNamespace(dtype=torch.float32, seed=0x5EED, counter=0, grad_hook=None, boundary_hook=None, side=None, side_raw=0,
                     side_keep=[], side_gens=[], side_n=0, attn_side=None, attn_side_raw=None, slots=None, force_static=False)


def set_compute_dtype(dtype):
    assert dtype in (torch.float32, torch.bfloat16)
    _S.dtype = dtype


def compute_dtype():
    return _S.dtype


def manual_seed(seed):
    _S.seed = int(seed) & 0x7FFFFFFF   # 31 bits: bit 63 of a kernel's seed argument tags a seed-slot pointer (next_seed)
    _S.counter = 0


def next_seed():
    """A fresh 64-bit dropout seed (host-side counter; the kernels hash (seed, element index)).  While a step is recorded /
    captured for graph replay (speecht5_amd/graph.py) the return value is instead a tagged device pointer to a seed SLOT (bit
    63 set, csrc/common.h resolve_seed); the slots are refilled before every replay with exactly the values the host
    counter would have produced, so eager and replayed steps draw the same masks."""
    if _S.slots is not None:
        return _S.slots.take()
    _S.counter += 1
    return ((_S.seed << 32) | (_S.counter & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


class SeedSlots:
    """Device array of 64-bit seeds + its pinned host image; slot k of a step holds the k-th next_seed() of that step."""

    def __init__(self, n, device):
        self.n = n
        self.dev = torch.zeros(n, dtype=torch.int64, device=device)
        self.host = torch.zeros(n, dtype=torch.int64).pin_memory()
        self.host2 = None   # second pinned image (graph.StepGraph prefetch_host)
        self.k = 0          # slots handed out in the step being recorded / captured
        self.used = 0       # slots one step uses (fixed after the recording step)

    def take(self):
        assert self.k < self.n, "seed slots exhausted: raise StepGraph(seed_slots=...)"
        v = (1 << 63) | (self.dev.data_ptr() + 8 * self.k)
        self.k += 1
        return v

    def begin_step(self):
        """Fill the slots with the seeds the host counter yields for this step (same values as the eager path)."""
        self.produce(0)
        self.upload(0)

    def produce(self, slot):
        n = self.used if self.used else self.n
        base = _S.counter
        import numpy as np
        vals = ((np.uint64(_S.seed) << np.uint64(32)) | ((np.arange(1, n + 1, dtype=np.uint64) + np.uint64(base)) & np.uint64(0xFFFFFFFF)))
        if slot and self.host2 is None:
            self.ensure_second_image()
        (self.host2 if slot else self.host)[:n] = torch.from_numpy(vals.astype(np.int64))
        _S.counter += n

    def upload(self, slot):
        self.dev.copy_(self.host2 if slot else self.host, non_blocking=True)
        self.k = 0

    def ensure_second_image(self):
        if self.host2 is None:
            self.host2 = torch.zeros(self.n, dtype=torch.int64).pin_memory()

    def digest(self, slot):
        import zlib
        n = self.used if self.used else self.n
        return zlib.crc32((self.host2 if slot else self.host)[:n].numpy().tobytes())


# Host-produced step inputs (random span masks, time-mix indices, ...): in eager mode a plain host->device copy; while a
# step is RECORDED the tensor gets a persistent device buffer + pinned host image and the producer is remembered; while the
# step is CAPTURED the buffer is returned as is (a captured graph must not copy from pageable host memory, and must read
# the same addresses on every replay); before every REPLAY graph.StepGraph re-runs the producers in recording order and
# refreshes the buffers.
class HostStaging:
    def __init__(self):
        self.mode = None          # None | "record" | "capture"
        self.entries = []         # [dev, pinned, producer]
        self.i = 0
        self.block = None         # (device block, [pinned image 0, pinned image 1]) once pack() has run
        self.packed = set()       # ids of the entries that live inside the block

    def pack(self):
        """ONE device block and one pinned image (per slot) for all staged inputs, the entries re-pointed at 256-byte-aligned views of
        them -- before the step is captured, so that the captured kernels read the views' addresses.  upload() is then a single copy:
        the benched update stages ~55 small tensors (span masks, LayerDrop flags, time-mix draws, temperatures ...), and 55 separate
        4-us blit kernels in front of every replay were 0.3-0.4 ms of an otherwise idle GPU per update (round 6)."""
        ents = [e for e in self.entries if e[0] is not None and id(e) not in self.packed]
        if not ents or self.block is not None or os.environ.get("ST5_STAGING_PACK", "1") == "0":
            return
        dev = ents[0][0].device
        offs, tot = [], 0
        for e in ents:
            offs.append(tot)
            tot += (e[0].numel() * e[0].element_size() + 255) // 256 * 256
        D = torch.zeros(max(tot, 256), dtype=torch.uint8, device=dev)
        H = [torch.zeros(max(tot, 256), dtype=torch.uint8).pin_memory() for _ in range(2)]
        for e, off in zip(ents, offs):
            nb, dt, shp = e[0].numel() * e[0].element_size(), e[0].dtype, e[0].shape
            view = lambda buf: buf[off:off + nb].view(dt).view(shp)
            h0, h1 = view(H[0]), view(H[1])
            h0.copy_(e[1])
            if len(e) == 4:
                h1.copy_(e[3])
            d = view(D)
            d.copy_(e[0])
            e[0], e[1] = d, h0
            if len(e) == 3:
                e.append(h1)
            else:
                e[3] = h1
            self.packed.add(id(e))
        self.block = (D, H)

    def begin_step(self, mode):
        if mode == "record" and self.entries and torch.cuda.is_available():
            # a second recording step rewrites the pinned images of the first one: their asynchronous uploads (enqueued on the
            # micro-batches' streams, possibly still queued behind the previous step's kernels) must have executed
            torch.cuda.synchronize()
        self.mode = mode
        self.i = 0

    def get(self, producer, device):
        if self.mode is None:
            return producer().to(device, non_blocking=True)
        if self.mode == "record":
            h = producer()
            if self.i == len(self.entries):
                self.entries.append([torch.empty(h.shape, dtype=h.dtype, device=device), torch.empty(h.shape, dtype=h.dtype).pin_memory(), producer])
            e = self.entries[self.i]
            assert e[0] is not None and e[1].shape == h.shape and e[1].dtype == h.dtype, "a staged step input changed shape between steps"
            e[2] = producer
            e[1].copy_(h)
            e[0].copy_(e[1], non_blocking=True)
            self.i += 1
            return e[0]
        e = self.entries[self.i]   # capture: contents were refreshed by refresh()
        self.i += 1
        return e[0]

    def draw(self, fn):
        """A host-side random draw whose VALUE steers nothing in a replayed step (LayerDrop probabilities at LayerDrop 0, which
        graph.StepGraph asserts) but which CONSUMES the CPU random stream: recorded like a staged input without a device buffer
        and repeated before every replay, so the stream stays in step with the eager path (the span masks drawn after it come
        out the same)."""
        if self.mode is None:
            return fn()
        if self.mode == "record":
            v = fn()
            if self.i == len(self.entries):
                self.entries.append([None, v, fn])
            e = self.entries[self.i]
            assert e[0] is None, "the step's sequence of staged inputs / host draws changed between steps"
            e[1], e[2] = v, fn
            self.i += 1
            return v
        e = self.entries[self.i]   # capture: nothing is drawn; the recorded value stands in
        assert e[0] is None
        self.i += 1
        return e[1]

    def refresh(self):
        self.produce(0)
        self.upload(0)

    # The two halves of refresh(), for a replay loop that produces the NEXT step's host inputs on a helper thread while the
    # current step runs (graph.StepGraph(prefetch_host=True)): `slot` selects one of two pinned images per entry, so the host
    # may write step k+1's data while step k's upload (queued behind step k-1 on the stream) has not executed yet.
    def produce(self, slot):
        for e in self.entries:
            h = e[2]()
            if e[0] is None:   # (draw(): consumed, not used)
                continue
            assert e[1].shape == h.shape
            if slot and len(e) == 3:
                e.append(torch.empty(h.shape, dtype=h.dtype).pin_memory())
            (e[3] if slot else e[1]).copy_(h)

    def upload(self, slot):
        if self.block is not None:
            self.block[0].copy_(self.block[1][slot], non_blocking=True)
        for e in self.entries:
            if e[0] is not None and id(e) not in self.packed:
                e[0].copy_(e[3] if slot else e[1], non_blocking=True)

    def ensure_second_images(self):
        """Second pinned image of every staged input, allocated on the calling thread (graph.StepGraph.capture)."""
        for e in self.entries:
            if e[0] is not None and len(e) == 3:
                e.append(torch.empty(e[1].shape, dtype=e[1].dtype).pin_memory())

    def digests(self, slot):
        import zlib
        return [zlib.crc32((e[3] if slot else e[1]).numpy().tobytes()) for e in self.entries if e[0] is not None]


staging = HostStaging()


def stage_host(producer, device):
    """Device copy of the CPU tensor `producer()` returns (see HostStaging)."""
    return staging.get(producer, device)


def host_draw(fn):
    """fn() now, and again before every replay of a captured step (see HostStaging.draw)."""
    return staging.draw(fn)


def static_shapes():
    """True while a step is recorded / captured for graph replay: data-dependent shapes (boolean-index gathers) are replaced by
    their fixed-shape forms (all rows + per-row weights)."""
    return staging.mode is not None or _S.force_static


def set_grad_ready_hook(fn):
    """fn(param) is called when a parameter's .grad has received its last contribution of this backward
    from this library (used to launch bucketed all-reduces while backward is still running)."""
    _S.grad_hook = fn


def set_layer_boundary_hook(fn):
    """fn(x, module, tag) -> x, called by the layer mirrors at their input (tag None) and by the encoder / decoder stacks
    at their output (tag "out"); see speecht5_amd/ddp.py."""
    _S.boundary_hook = fn


def layer_boundary(x, module, tag=None):
    """Idempotent per (tensor, module, tag): a caller that needs the post-boundary tensor itself (the LayerDrop select takes the
    layer's INPUT as its skip operand -- it must be the tensor behind the boundary, or a backward cut at that boundary would leak
    through the skip path) applies the boundary first; the layer's own call then returns the same tensor."""
    if _S.boundary_hook is None:
        return x
    if tag in ("shared", "bypass"):     # (shared: a tensor read by every layer of a stack, asked for once per layer; bypass: a loss
        return _S.boundary_hook(x, module, tag)   # term's tap on an early tensor -- ddp._boundary)
    key = (id(module), tag)
    if getattr(x, "_st5_boundary", None) == key:
        return x
    y = _S.boundary_hook(x, module, tag)
    y._st5_boundary = key
    return y


def _ceil8(n):
    return (n + 7) // 8 * 8


def _dt(t):
    return hip.dt(t)


# -------------------------------------------------------------------------------------------------
# parameter handling: compute-dtype weight cache + direct gradient accumulation
# -------------------------------------------------------------------------------------------------
class _WeightCache:
    """Compute-dtype (optionally fused / re-laid-out) copies of fp32 parameters, rebuilt when any
    source parameter changes (torch's in-place version counter)."""

    def __init__(self):
        self.store = {}

    def get(self, key, params, build):
        ver = tuple((p.data_ptr(), p._version) for p in params)
        hit = self.store.get(key)
        cur = hip.stream() if params and params[0].is_cuda else None
        if hit is not None and hit[0] == ver:
            if hit[2] is not None and cur != hit[3] and cur not in hit[4]:
                # built on another stream (two micro-batches forward side by side): this stream waits for the build once
                torch.cuda.current_stream().wait_event(hit[2])
                hit[4].add(cur)
            return hit[1]
        val = build()
        ev = None
        if cur is not None:
            ev = torch.cuda.Event()
            ev.record()
        self.store[key] = (ver, val, ev, cur, set())
        return val

    def clear(self):
        self.store.clear()


weight_cache = _WeightCache()


def _gather3(src, dst, dims, strides, off=0, accumulate=False):
    """dst[a, b, c] (+)= src.flat[off + a*sa + b*sb + c*sc]: one-pass re-layout (+ cast to dst's dtype) of an fp32 tensor."""
    A, B, C = dims
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst.numel() == A * B * C
    hip.check(hip.lib().st5_gather3(src.data_ptr(), dst.data_ptr(), A, B, C, strides[0], strides[1], strides[2], off,
                                    1 if accumulate else 0, _dt(dst), hip.stream()), "st5_gather3")
    return dst


def _cast_into(src, dst, transpose=False):
    """dst (compute dtype) <- src (fp32 [rows, cols]); transpose writes dst[c, r]."""
    rows, cols = src.shape
    hip.check(hip.lib().st5_cast_from_f32(src.data_ptr(), dst.data_ptr(), rows, cols, 1 if transpose else 0,
                                          _dt(dst), hip.stream()), "st5_cast_from_f32")


class _Bf16Mirror:
    """bf16 copies of the parameters kept current by the fused optimizer step (ddp.FusedAdam): the Adam kernel writes the
    updated fp32 master AND its bf16 image in one pass, and every transposed weight copy the data-gradient GEMMs need
    is refreshed by ONE batched transpose launch per step -- instead of ~300 cast / transpose launches per step.
    A parameter's image is used only while `p._version` still equals the version recorded at the last refresh."""

    def __init__(self):
        if globals().get("fp8_mirror") is not None:
            fp8_mirror.reset()  # (its jobs point into this mirror's pools)
        self.flat = None        # bf16 [total], same element layout as the optimizer's flat fp32 buffer
        self.tflat = None       # bf16 pool of transposed copies
        self.tcap = 0
        self.tused = 0
        self.jobs = {}          # (src_off, rows, cols) -> (dst_off, view [cols, rows])
        self.first = {}         # key -> (event, stream) of a copy produced by a first-use transpose since the last batched refresh
        self.jobs_dev = None
        self.ntiles = 0
        self.dirty = False      # job table changed since it was last uploaded
        # Round 6, an A/B mode (ddp.FusedAdam, ST5_LAZY_TRANSPOSES=1; measured slower, off): the batched refresh can be LAZY.  Only the
        # data-gradient GEMMs read the transposed copies, i.e. nothing before the first backward of the NEXT update -- the optimizer step
        # then only marks them stale (mark_stale) and the first reader refreshes them, on its stream, at the head of that update's
        # backward.  A reader on another stream waits for that refresh's event once.
        self.stale = False
        self.refreshed = None   # (event, raw stream) of the last lazy refresh
        self.waited = set()     # raw streams already ordered behind it

    def attach(self, flat_bf16, params, offsets):
        self.__init__()
        self.flat = flat_bf16
        for p, off in zip(params, offsets):
            p._st5_moff = off
            p._st5_mver = p._version

    def valid(self, p):
        return self.flat is not None and getattr(p, "_st5_moff", None) is not None and p._st5_mver == p._version and \
            p.device == self.flat.device

    def stacked(self, weights):
        """View [sum N_i, K] of the mirror if the weights are valid, 2-D-able and adjacent in this order, else None."""
        if not all(self.valid(w) for w in weights):
            return None
        K = weights[0].numel() // weights[0].shape[0]
        off = weights[0]._st5_moff
        nxt = off
        for w in weights:
            if w._st5_moff != nxt or w.numel() // w.shape[0] != K:
                return None
            nxt += w.numel()
        if off % 8:   # 16-byte rows for the GEMM loaders
            return None
        return off, self.flat[off:nxt].view(-1, K)

    def mark_stale(self):
        """The parameters (and their bf16 image) changed: every transposed copy is out of date until the next reader refreshes them."""
        self.stale = True

    def _sync_transposes(self):
        if self.stale:
            self.stale = False
            self.refresh_transposes()
            ev = torch.cuda.Event()
            ev.record()
            self.refreshed, self.waited = (ev, hip.stream()), {hip.stream()}
        elif self.refreshed is not None and hip.stream() not in self.waited:
            torch.cuda.current_stream().wait_event(self.refreshed[0])
            self.waited.add(hip.stream())

    def transposed(self, off, rows, cols):
        self._sync_transposes()
        key = (off, rows, cols)
        hit = self.jobs.get(key)
        if hit is not None:
            ev = self.first.get(key)
            if ev is not None and ev[1] != hip.stream():
                # produced by a single-job transpose on ANOTHER stream earlier in this very step (the two micro-batches of an
                # update backward side by side: the second stream's data-gradient GEMM must not read the copy before the first
                # stream's transpose kernel has written it); from the next optimizer step on the batched refresh produces
                # every copy on the main stream before the streams fork
                torch.cuda.current_stream().wait_event(ev[0])
            return hit[1]
        n = rows * cols
        need = (self.tused + n + 63) // 64 * 64
        if need > self.tcap:
            return None   # pool exhausted (sized for every >= 2-D parameter once): caller falls back to the cast path
        view = self.tflat[self.tused:self.tused + n].view(cols, rows)
        self.jobs[key] = (self.tused, view)
        self.tused = need
        self.dirty = True
        # first use: produce this copy now (the batched refresh only runs after optimizer steps)
        _cast_free_transpose(self.flat[off:off + n].view(rows, cols), view)
        ev = torch.cuda.Event()
        ev.record()
        self.first[key] = (ev, hip.stream())
        return view

    def refresh_transposes(self):
        self.first = {}         # (the caller -- the optimizer step -- runs after the micro-batch streams have joined)
        if not self.jobs:
            return
        if self.dirty or self.jobs_dev is None:
            import struct
            recs, tile0 = [], 0
            for (off, rows, cols), (doff, _v) in self.jobs.items():
                recs.append(struct.pack("<qqiiii", off, doff, rows, cols, tile0, 0))
                tile0 += ((rows + 63) // 64) * ((cols + 63) // 64)
            self.ntiles = tile0
            raw = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)
            self.jobs_dev = raw.to(self.flat.device)
            self.dirty = False
        hip.check(hip.lib().st5_multi_transpose_bf16(self.flat.data_ptr(), self.tflat.data_ptr(), self.jobs_dev.data_ptr(),
                                                     len(self.jobs), self.ntiles, hip.stream()), "st5_multi_transpose_bf16")


bf16_mirror = _Bf16Mirror()


def _cast_free_transpose(src_bf16, dst_bf16):
    """dst [cols, rows] <- src [rows, cols]^T for one bf16 matrix (single-job batched transpose)."""
    import struct
    rows, cols = src_bf16.shape
    rec = struct.pack("<qqiiii", 0, 0, rows, cols, 0, 0)
    job = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(src_bf16.device)
    hip.check(hip.lib().st5_multi_transpose_bf16(src_bf16.data_ptr(), dst_bf16.data_ptr(), job.data_ptr(), 1,
                                                 ((rows + 63) // 64) * ((cols + 63) // 64), hip.stream()), "st5_multi_transpose_bf16")


def fused_weight(weights, dtype):
    """[sum N_i, K] compute-dtype matrix stacking the given nn.Linear weights (cached)."""
    if len(weights) == 1 and dtype == torch.float32 and weights[0].is_contiguous():
        return weights[0].detach()
    if dtype == torch.bfloat16:
        hit = bf16_mirror.stacked(weights)
        if hit is not None:
            return hit[1]

    def build():
        K = weights[0].shape[1]
        out = torch.empty(sum(w.shape[0] for w in weights), K, dtype=dtype, device=weights[0].device)
        o = 0
        for w in weights:
            src = w.detach().reshape(w.shape[0], -1).contiguous()
            _cast_into(src, out[o:o + w.shape[0]])
            o += w.shape[0]
        return out

    return weight_cache.get(("w", dtype) + tuple(id(w) for w in weights), weights, build)


def fused_weight_t(weights, dtype):
    """[K, sum N_i] compute-dtype TRANSPOSE of the stacked weights (cached): the B operand of the data-gradient GEMM
    dX = G . W in K-major form, so that dgrad runs on the same LDS-DMA NT kernel as the forward."""
    if dtype == torch.bfloat16:
        hit = bf16_mirror.stacked(weights)
        if hit is not None:
            off, st = hit
            tv = bf16_mirror.transposed(off, st.shape[0], st.shape[1])
            if tv is not None:
                return tv

    def build():
        src = weights[0].detach() if len(weights) == 1 else torch.cat([w.detach() for w in weights], 0)
        src = src.reshape(src.shape[0], -1).contiguous().float()
        out = torch.empty(src.shape[1], src.shape[0], dtype=dtype, device=src.device)
        _cast_into(src, out, transpose=True)
        return out

    return weight_cache.get(("wt", dtype) + tuple(id(w) for w in weights), weights, build)


# -------------------------------------------------------------------------------------------------
# MX-fp8 compute mode for the large Linears (BASELINE.json configs[4]: SpeechT5-Large, arch models/speecht5.py:1402-1425):
# forward and data-gradient GEMMs of LinearFunction / FFNFunction run on st5_gemm_mxfp8 (block-scaled fp8 MFMA, twice the
# bf16 rate), weight gradients stay bf16 (dW = dY^T X on the TN kernel, fp32 accumulation into the flat gradient buffer).
# Activations / gradients are quantised per call (st5_quant_mxfp8: 32-element blocks along the reduction index), weights once
# per optimizer step (cached with the other compute-dtype copies, dropped by weight_cache.clear()).
# -------------------------------------------------------------------------------------------------
_FP8 = SimpleNamespace(enabled=False, min_rows=512, min_n=512, launches=0, fuse=int(os.environ.get("ST5_FP8_FUSE_QUANT", "7")))   # fuse: bit mask of the producers that write fp8 images -- 1 LayerNorm forward, 2 GELU epilogue (fc1), 4 GELU-derivative epilogue


def set_fp8(enabled):
    """MX-fp8 forward / data-gradient GEMMs for eligible Linears (bf16 compute mode only)."""
    _FP8.enabled = bool(enabled)


def fp8_enabled():
    return _FP8.enabled


def _fp8_ok(M, N, K, dtype):
    """K = the reduction length: MX blocks of 32 along it, 128-byte k-tiles."""
    return _FP8.enabled and dtype == torch.bfloat16 and K % 128 == 0 and N % 8 == 0 and M >= _FP8.min_rows and N >= _FP8.min_n


def _quant_cached(kind, weights, mat):
    """(q, scales) of a cached compute-dtype weight matrix, themselves cached until the parameters change."""
    hit = fp8_mirror.quantised(mat)
    if hit is not None:
        return hit
    return weight_cache.get((kind,) + tuple(id(w) for w in weights), weights, lambda: hip.quant_mxfp8(mat))


class _Fp8Mirror:
    """fp8 images (e4m3 bytes + e8m0 block scales) of the weight matrices the fp8 GEMMs read, kept current by ONE batched
    quantisation launch per optimizer step (st5_multi_quant_mxfp8, behind the batched transpose of _Bf16Mirror whose copies are the
    data-gradient sources) -- round 6.  Before, every weight and every transposed weight was quantised by a launch of its own, once per
    update, on whichever micro-batch stream reached the Linear first (~280 launches of 2-5 us and as many cross-stream event waits
    inside the replayed graph).  Only matrices that live in the bf16 mirror's two pools (stable addresses, refreshed by the optimizer
    step) are registered; everything else keeps the per-step cache."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.jobs = {}          # (address, rows, cols) -> (source view, q, scales)
        self.first = {}         # key -> (event, raw stream) of an image produced by a first-use launch since the last batched refresh
        self.jobs_dev = None
        self.nblocks = 0
        self.dirty = False
        self.off = False        # (set when the optimizer step does not refresh: lazy-transpose A/B mode)

    def quantised(self, W):
        m = bf16_mirror
        if self.off or m.flat is None or m.stale or not W.is_cuda or W.dtype != torch.bfloat16:
            return None
        st = W.untyped_storage().data_ptr()
        if st != m.flat.untyped_storage().data_ptr() and (m.tflat is None or st != m.tflat.untyped_storage().data_ptr()):
            return None
        if W.dim() != 2 or not W.is_contiguous() or W.shape[1] % 32:
            return None
        key = (W.data_ptr(), W.shape[0], W.shape[1])
        hit = self.jobs.get(key)
        if hit is not None:
            ev = self.first.get(key)
            if ev is not None and ev[1] != hip.stream():
                # produced on ANOTHER stream earlier in this very update (first use, before any batched refresh has covered it)
                torch.cuda.current_stream().wait_event(ev[0])
            return hit[1], hit[2]
        q, sc = hip.quant_mxfp8(W)
        ev = torch.cuda.Event()
        ev.record()
        self.first[key] = (ev, hip.stream())
        self.jobs[key] = (W, q, sc)
        self.dirty = True
        return q, sc

    def refresh(self):
        """Re-quantise every registered matrix from its (just refreshed) bf16 source: one launch, on the optimizer's stream."""
        self.first = {}
        if self.off or not self.jobs or not _FP8.enabled:
            if self.jobs and not _FP8.enabled:
                self.reset()    # (mode switched off: the images would go stale)
            return
        if self.dirty or self.jobs_dev is None:
            import struct
            recs, blk0 = [], 0
            for (W, q, sc) in self.jobs.values():
                n = W.numel()
                recs.append(struct.pack("<QQQqii", W.data_ptr(), q.data_ptr(), sc.data_ptr(), n, W.shape[1], blk0))
                blk0 += (n + 2047) // 2048
            self.nblocks = blk0
            dev = next(iter(self.jobs.values()))[0].device
            self.jobs_dev = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
            self.dirty = False
        hip.check(hip.lib().st5_multi_quant_mxfp8(self.jobs_dev.data_ptr(), len(self.jobs), self.nblocks, hip.stream()),
                  "st5_multi_quant_mxfp8")


fp8_mirror = _Fp8Mirror()


# fp8 images made by a PRODUCER (round 6): the LayerNorm forward of a pre-LN layer, the fc1 GEMM's GELU epilogue and the data gradient
# through the GELU write the MX-fp8 image of their bf16 output beside it (st5_layernorm_fwd_q8, st5_gemm_mxfp8_q), so the fp8 GEMM that
# consumes the tensor launches no quantisation pass.  Keyed by the output's address; an entry HOLDS the output tensor, so the address
# cannot be handed to another tensor while the entry exists, and is dropped when its consumer takes it (or by the small FIFO bound).
_q8_tags = {}      # y.data_ptr() -> (y, q, scales, rows, cols)
_Q8_MAX = 8
_q8_lock = threading.Lock()      # (producers and consumers run on the main thread AND on autograd's backward threads)


def _q8_tag(y, q, sc, rows, cols):
    with _q8_lock:
        while len(_q8_tags) >= _Q8_MAX:
            _q8_tags.pop(next(iter(_q8_tags)), None)
        _q8_tags[y.data_ptr()] = (y, q, sc, rows, cols)


def _q8_take(a2, rows, cols):
    with _q8_lock:
        hit = _q8_tags.pop(a2.data_ptr(), None)
    if hit is not None and hit[3:] == (rows, cols) and a2.is_contiguous():
        return hit[1], hit[2]
    return None


def _q8_buffers(rows, cols, device):
    return (torch.empty(rows, cols, dtype=torch.uint8, device=device), torch.empty(rows, cols // 32, dtype=torch.uint8, device=device))


def _nt_gemm(a2, weights, transposed, C, M, N, K, dtype, quant_out=None, **epi):
    """C = epilogue(a2 . W^T) with W = stacked weights [N, K] (transposed: W = stack^T, the data-gradient form) -- on the MX-fp8
    kernel when the mode is on and the shapes allow, else st5_gemm in the compute dtype.  quant_out = the output TENSOR when its
    consumer is another fp8 GEMM over all N columns (the FFN's hidden activations and their gradients): the epilogue then writes the
    output's fp8 image too."""
    W = fused_weight_t(weights, dtype) if transposed else fused_weight(weights, dtype)
    if _fp8_ok(M, N, K, dtype) and W.is_contiguous() and a2.is_contiguous():
        Wq, Ws = _quant_cached("qt" if transposed else "q", weights, W)
        pre = _q8_take(a2, M, K)
        aq, as_ = pre if pre is not None else hip.quant_mxfp8(a2)
        out_q = None
        if quant_out is not None and (_FP8.fuse & (4 if (epi.get("flags", 0) & hip.DACT) else 2)) and N % 128 == 0 and epi.get("dropout_p", 0.0) == 0 and epi.get("R") is None and \
                epi.get("beta", 0.0) == 0 and quant_out.is_contiguous() and quant_out.shape == (M, N) and \
                ((epi.get("flags", 0) & hip.DACT) or (epi.get("act", ACT_NONE) == ACT_GELU and epi.get("Cpre") is not None)):
            out_q = _q8_buffers(M, N, a2.device)
        hip.gemm_mxfp8(aq, as_, Wq, Ws, C, M, N, K, out_q=out_q, **epi)
        if out_q is not None:
            _q8_tag(quant_out, out_q[0], out_q[1], M, N)
        _FP8.launches += 1
        return
    hip.gemm(hip.operand(a2, a2.stride(0)), hip.operand(W, W.shape[1]), C, M, N, K, _dt(dtype), **epi)


def fused_bias(biases):
    if all(b is None for b in biases):
        return None
    if len(biases) == 1:
        return biases[0].detach()
    if all(b is not None and b.is_contiguous() for b in biases):
        # adjacent in the optimizer's flat buffer (ddp._fusion_ordered_parameters): the stacked bias is a view, no cat
        nxt, st = biases[0].data_ptr(), biases[0].untyped_storage().data_ptr()
        for b in biases:
            if b.data_ptr() != nxt or b.untyped_storage().data_ptr() != st or b.dtype != torch.float32:
                nxt = None
                break
            nxt += b.numel() * 4
        if nxt is not None:
            return torch.as_strided(biases[0].detach(), (sum(b.numel() for b in biases),), (1,))
    return weight_cache.get(("b",) + tuple(id(b) for b in biases), biases,
                            lambda: torch.cat([b.detach() for b in biases]))


def cast_param(p, dtype):
    """compute-dtype copy of a (small) fp32 parameter tensor, cached."""
    if dtype == torch.float32:
        return p.detach()

    def build():
        out = torch.empty(p.shape, dtype=dtype, device=p.device)
        src = p.detach().reshape(-1, p.shape[-1]).contiguous()
        _cast_into(src, out.view(src.shape))
        return out

    return weight_cache.get(("c", dtype, id(p)), [p], build)


def grad_buffer(p):
    """fp32 .grad of a parameter, allocated (zeroed) on first use; kernels accumulate into it."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def _grad_done(p):
    if _S.grad_hook is not None:
        _S.grad_hook(p)


def to_compute(x):
    """Cast an input tensor (fp32) to the compute dtype through the library's cast kernel."""
    if x.dtype == _S.dtype:
        return x.contiguous()
    assert x.dtype == torch.float32, x.dtype
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=_S.dtype, device=x.device)
    hip.check(hip.lib().st5_cast_from_f32(x.data_ptr(), out.data_ptr(), 1, x.numel(), 0, _dt(out), hip.stream()),
              "st5_cast_from_f32")
    return out


def to_float(x):
    if x.dtype == torch.float32:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().st5_cast_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _dt(x), hip.stream()),
              "st5_cast_to_f32")
    return out


class _ToCompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_compute(x)

    @staticmethod
    def backward(ctx, g):
        return to_float(g)


class _ToFloat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_float(x)

    @staticmethod
    def backward(ctx, g):
        return to_compute(g)


def as_compute(x):
    """Differentiable cast fp32 -> compute dtype."""
    return x if x.dtype == _S.dtype else _ToCompute.apply(x)


def as_float(x):
    """Differentiable cast compute dtype -> fp32 (model outputs consumed by the criterion)."""
    return x if x.dtype == torch.float32 else _ToFloat.apply(x)


# -------------------------------------------------------------------------------------------------
# raw kernel wrappers (no autograd)
# -------------------------------------------------------------------------------------------------
def _colsum_into(x2d, ld, cols, out, col_off=0, scale=1.0):
    """out[c] += scale * sum_r x2d[r, col_off + c]"""
    rows = x2d.shape[0]
    L = hip.lib()
    ws = hip.workspace(L.st5_colsum_ws_bytes(rows, cols), x2d.device)
    hip.check(L.st5_colsum_ws(x2d.data_ptr() + col_off * x2d.element_size(), out.data_ptr(), ws.data_ptr(), rows, cols,
                              ld, scale, 1, _dt(x2d), hip.stream()), "st5_colsum_ws")


# Dropout-epilogue producers (Linear / FFN with dropout_p > 0) tag their output buffer; the LayerNorm that reads that
# buffer then emits, from its backward kernel, BOTH dX and dX * mask(seed) -- the masked copy is what the producer's
# backward needs, so no separate dropout kernel runs over the gradient (post-LN layers: ~110 launches per step).
# Matching is by buffer address and verified by (p, seed, shape) at the consumer, so a stale or recycled address can only
# cost a wasted second output, never a wrong gradient.
_drop_tags = {}      # forward:  y.data_ptr() -> (p, seed, rows, cols)
_drop_grads = {}     # backward: dX.data_ptr() -> (p, seed, rows, cols, dX_dropped)


def _tag_dropout_output(y, p, seed, rows, cols):
    if len(_drop_tags) > 64:
        _drop_tags.clear()
    _drop_tags[y.data_ptr()] = (p, seed, rows, cols)


def _dropped_grad(g, p, seed):
    """g * dropout_mask(p, seed): taken from the LayerNorm backward that produced g when it made one, else computed."""
    hit = _drop_grads.pop(g.data_ptr(), None)
    if hit is not None and hit[:4] == (p, seed, g.shape[0], g.shape[1]):
        return hit[4]
    return _dropout(g, p, seed)


# Weight-gradient stream.  dW = G^T X of a Linear is off the backward critical path (only the optimizer / the bucket
# all-reduce needs it) and is a short-M x N, long-K problem that cannot fill the chip on its own; launched on a second
# stream it runs beside the data-gradient chain instead of in front of it.  The owner of the gradient buffers
# (ddp.FlatGradDataParallel) switches it on, marks the parameters whose gradients ONLY these GEMMs write (`_st5_side_ok`:
# no tied weights, no other writer) and joins the streams wherever gradients must be complete.
def set_wgrad_stream(stream):
    join_wgrad_stream()
    _S.side = stream
    _S.side_raw = stream.cuda_stream if stream is not None else 0


def wgrad_stream():
    return _S.side


def join_wgrad_stream():
    """The current stream waits for every weight-gradient GEMM issued so far (and their batched split-K reduction)."""
    if not _side_here():
        return
    L = hip.lib()
    hip.check(L.st5_gemm_flush_splitk(_S.side_raw), "st5_gemm_flush_splitk")
    hip.check(L.st5_stream_fork(_S.side_raw, hip.stream()), "st5_stream_fork")
    _S.side_keep = []
    _S.side_gens = []


def _side_hold(keep):
    """Keep the tensors a side-stream GEMM reads alive until that GEMM has run.  Generations of 16 launches are closed by an
    event on the side stream and dropped once the event has completed, so at most a few dozen gradient activations are
    pinned at a time even when no bucket boundary joins the streams (one rank: the only join is in finish())."""
    _S.side_keep.extend(keep)
    _S.side_n += 1
    if staging.mode == "capture":
        return   # (event queries are illegal during stream capture; the graph's private pool keeps the memory anyway)
    if _S.side_n % 16 == 0:
        ev = torch.cuda.Event()
        ev.record(_S.side)
        _S.side_gens.append((ev, _S.side_keep))
        _S.side_keep = []
        while _S.side_gens and _S.side_gens[0][0].query():
            _S.side_gens.pop(0)


def _side_here():
    """Does the weight-gradient stream serve the CURRENT stream?  With an owner set (set_wgrad_owner) only work issued on the owner
    stream forks to it -- the micro-batches side by side: the one on the update's own stream; a helper stream forked from the second
    micro-batch's stream cannot be captured on ROCm 7.2."""
    if _S.side is None:
        return False
    owner = _S.__dict__.get("side_owner")
    return owner is None or hip.stream() == owner


def set_wgrad_owner(stream):
    """Restrict the weight-gradient stream to work issued on `stream` (None: whoever issues weight-gradient GEMMs)."""
    _S.__dict__["side_owner"] = stream.cuda_stream if stream is not None else None


# Weight-gradient groups.  The weight gradients of a transformer layer are 36-144 output tiles each with a reduction over every
# token: launched one by one they need split-K (fp32 slabs + a reduction kernel) to fill the chip.  While the owner of the gradient
# buffers has grouping on (set_wgrad_grouping: ddp.FlatGradDataParallel, which also owns the points where gradients must be
# complete and calls flush_wgrads() there), _wgrad_gemm QUEUES the problems of the current stream -- keeping their operands alive --
# and hands them to st5_gemm_tn_group as soon as they amount to one round of the chip (~400-512 tiles of 128^2: the four / six weight
# gradients of one layer): one launch, whole reductions, no slabs.  233 -> 162 us per encoder layer at 8192 tokens.
WGRAD_GROUP = os.environ.get("ST5_WGRAD_GROUP", "1") != "0"     # A/B switch
_WG_ROUND, _WG_FLUSH_AT, _WG_MAX = [int(v) for v in os.environ.get("ST5_WGRAD_ROUND", "512,400").split(",")] + [8]   # tiles: never above / launch at
# (round 6) problems whose M, N are multiples of 256 -- every Linear of the transformer -- run on the phased 256 x 256 grouped kernel, one
# block per CU: a round is 256 tiles of 256^2 (two Base layers are 216, a Large layer and the next one's first problems 208-240)
_WG_ROUND_P, _WG_FLUSH_AT_P = [int(v) for v in os.environ.get("ST5_WGRAD_ROUND_P", "256,200").split(",")]
# (problems per phased launch: the library takes up to sixteen -- compact 72-byte records in the kernel arguments; a decoder layer has six.
#  Sixteen measured no better than eight on the benched update -- 29.38 / 29.64 / 29.46 against 29.51 / 29.34 / 29.11 ms, same box -- so eight it stays)
_WG_MAX_P = int(os.environ.get("ST5_WG_MAX_P", "8"))


def set_wgrad_grouping(on):
    flush_wgrads()
    _S.__dict__["wgroup"] = bool(on) and WGRAD_GROUP


# Where a stream's queued weight-gradient groups are LAUNCHED (round 6; an A/B mode, ddp.accumulate_overlapped / ST5_WGRAD_MOVE=1).
# With the two micro-batches of an update side by side the text micro-batch is the longer chain (8.1 against 4.75 TFLOP).
# A layer's weight gradients hang off the backward's critical path -- only the optimizer reads them -- so the groups the TEXT stream
# queues can be launched on the SPEECH stream instead: ordered behind their producers by an event (st5_stream_fork), behind everything
# the speech stream was given before (its own backward is enqueued first), writing the text micro-batch's own gradient buffer.  No
# third stream, same kernels, same bits.  Measured: SLOWER inside a replayed graph (35.0 against 29.8 ms per update): off by default.
# The operands stay alive until the streams join (release_wgrad_holds): the caching allocator hands a freed block to the next
# allocation of the stream that OWNS it, which knows nothing about a reader on another stream.
def set_wgrad_target(src, dst):
    """Groups queued on torch stream `src` are launched on `dst` (None, None: every stream launches its own)."""
    _S.__dict__["wq_target"] = None if src is None else (src.cuda_stream, dst)


def release_wgrad_holds():
    _S.__dict__["wq_hold"] = []


def hold_wgrads(stream):
    """Weight-gradient groups queued on torch stream `stream` are not launched but HELD (with their operands) until
    launch_held_wgrads() -- on whatever stream is current then, e.g. inside another graph (tools/r6/multi_graph.py: a micro-batch's weight
    gradients as graphs of their own on a third stream, ordered behind the backward phase that produced their operands by a plain stream
    event).  None: off."""
    _S.__dict__["wq_holding"] = None if stream is None else (stream.cuda_stream, [])


def launch_held_wgrads(keep_alive):
    """Launch every held group on the current stream; their operand tensors are appended to `keep_alive` (the caller decides how long the
    allocator must not reuse them)."""
    h = _S.__dict__.get("wq_holding")
    if h is None:
        return 0
    n = 0
    for problems, dt, keeps in h[1]:
        hip.gemm_tn_group(problems, dt)
        keep_alive.append(keeps)
        n += 1
    del h[1][:]
    return n


def _launch_wgrad_group(st, ent):
    h = _S.__dict__.get("wq_holding")
    if h is not None and h[0] == st:
        h[1].append((ent[0], ent[3], list(ent[1])))
        ent[1].clear()
        return
    tgt = _S.__dict__.get("wq_target")
    if tgt is None or tgt[0] != st:
        hip.gemm_tn_group(ent[0], ent[3])
        ent[1].clear()
        return
    hip.check(hip.lib().st5_stream_fork(st, tgt[1].cuda_stream), "st5_stream_fork")
    hip.gemm_tn_group(ent[0], ent[3], on=tgt[1])
    _S.__dict__.setdefault("wq_hold", []).append(ent[1])


def flush_wgrads():
    """Launch what the current stream has queued.  (Every stream that issues weight gradients reaches a flush point of its own
    before anybody reads the gradients: ddp.accumulate_overlapped per micro-batch stream, ddp._flush_splitk on the update's stream.)"""
    q = _S.__dict__.get("wq")
    if not q:
        return
    st = hip.stream()
    ent = q.pop(st, None)
    if ent and ent[0]:
        _launch_wgrad_group(st, ent)
    assert not any(e[0] for e in q.values()), "weight gradients still queued on another stream at a point where gradients must be complete"


def drop_wgrads():
    """Discard every queued weight-gradient problem and held operand (an update that failed half way: ADVICE r5 -- stale entries
    would otherwise be launched into the NEXT update's freshly zeroed gradient buffers)."""
    q = _S.__dict__.get("wq")
    if q:
        q.clear()
    _S.__dict__["wq_hold"] = []


def _is_queued_wgrad_operand(t):
    """Does a queued (not yet launched) or held weight-gradient problem read tensor t's memory?"""
    p = t.data_ptr()
    q = _S.__dict__.get("wq") or {}
    for ent in q.values():
        for keep in ent[1]:
            if any(torch.is_tensor(k) and k.data_ptr() == p for k in keep):
                return True
    for keeps in _S.__dict__.get("wq_hold") or []:
        for keep in keeps:
            if any(torch.is_tensor(k) and k.data_ptr() == p for k in keep):
                return True
    return False


def _wgrad_queue(A, B, C, M, N, K, dt, flags, asum, keep):
    q = _S.__dict__.setdefault("wq", {})
    st = hip.stream()
    ent = q.get(st)
    phased = bool(hip.lib().st5_gemm_tn_group_is_phased(M, N, K))       # (the library's own per-problem rule)
    tiles = (M // 256) * (N // 256) if phased else ((M + 127) // 128) * ((N + 127) // 128)
    round_, flush_at = (_WG_ROUND_P, _WG_FLUSH_AT_P) if phased else (_WG_ROUND, _WG_FLUSH_AT)

    def launch():
        _launch_wgrad_group(st, q.pop(st))
    if ent is not None and ent[0] and (ent[2] + tiles > round_ or len(ent[0]) == (_WG_MAX_P if ent[4] else _WG_MAX) or ent[3] != dt or ent[4] != phased
                                       or any(p[2].ptr == C.ptr or (asum is not None and p[8] is not None and p[8].data_ptr() == asum.data_ptr())
                                              for p in ent[0])):
        launch()            # (the round is full, another block tile, or the same gradient twice -- tied weights -- would race inside one launch)
        ent = None
    if ent is None:
        ent = q[st] = [[], [], 0, dt, phased]
    ent[0].append((A, B, C, M, N, K, flags, 1.0, asum))
    ent[1].append(keep)
    ent[2] += tiles
    if ent[2] >= flush_at:
        launch()


def _wgrad_gemm(params, A, B, C, M, N, K, dt, asum, keep):
    """C[M,N] += A^T B (fp32, both operands k-strided), bias-gradient column into `asum`.  `keep`: the tensors the GEMM
    reads -- held until the next join so that the caching allocator cannot hand their memory to a main-stream kernel
    while the side stream still reads them."""
    flags = hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32
    if any(getattr(q, "_st5_multi_writer", False) for q in params):
        # a gradient with other writers (tied weights, ddp marks them): accumulated here and now, in program order
        hip.gemm(A, B, C, M, N, K, dt, flags=flags, beta=1.0, asum=asum)
        return
    if not _side_here():
        if _S.__dict__.get("wgroup") and dt == hip.BF16 and M % 8 == 0 and N % 8 == 0:
            _wgrad_queue(A, B, C, M, N, K, dt, flags | hip.DEFERRABLE, asum, keep)
            return
        hip.gemm(A, B, C, M, N, K, dt, flags=flags | hip.DEFERRABLE, beta=1.0, asum=asum)
        return
    for q in params:
        if not getattr(q, "_st5_side_ok", False):
            # another kernel may write this gradient on the main stream (tied weights): stay there, reduce at once
            hip.gemm(A, B, C, M, N, K, dt, flags=flags, beta=1.0, asum=asum)
            return
    hip.check(hip.lib().st5_stream_fork(hip.stream(), _S.side_raw), "st5_stream_fork")
    hip.gemm(A, B, C, M, N, K, dt, flags=flags | hip.DEFERRABLE, beta=1.0, asum=asum, on=_S.side)
    _side_hold(keep)


def _conv_wgrad(w, opA, opB, Cout, k, Cin, Kred, dt, keep):
    """grad(w)[Cout, Cin, k] += (A^T B)[Cout, k*Cin] re-laid out (implicit-GEMM convolution weight gradient); on the
    weight-gradient stream when nothing else writes w's gradient."""
    def run():
        tmpw = torch.empty(Cout, k * Cin, dtype=torch.float32, device=w.device)
        hip.gemm(opA, opB, hip.operand(tmpw, k * Cin), Cout, k * Cin, Kred, dt, flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
        _gather3(tmpw, grad_buffer(w), (Cout, Cin, k), (k * Cin, 1, Cin), accumulate=True)   # grad[co, ci, j] += tmpw[co, j, ci]
    if _side_here() and getattr(w, "_st5_side_ok", False):
        hip.check(hip.lib().st5_stream_fork(hip.stream(), _S.side_raw), "st5_stream_fork")
        with torch.cuda.stream(_S.side):
            run()
        _side_hold(keep)
    else:
        run()


def _dropout(x, p, seed):
    y = torch.empty_like(x)
    hip.check(hip.lib().st5_dropout(x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _dt(x), hip.stream()), "st5_dropout")
    return y


def _padded(x2d, ld):
    """Copy [M, N] into a zero-padded [M, ld] buffer (only for odd widths such as the vocabulary)."""
    M, N = x2d.shape
    if x2d.is_contiguous() and N == ld:
        return x2d
    out = torch.zeros(M, ld, dtype=x2d.dtype, device=x2d.device)
    out[:, :N] = x2d  # strided device copy (glue; tiny tensors only)
    return out


def _rows(x):
    """Dense row view [rows, C] of a contiguous tensor."""
    assert x.is_contiguous()
    return x.view(-1, x.shape[-1])


# -------------------------------------------------------------------------------------------------
# Linear (+ fused bias / activation / dropout / residual), N-fused over several weight matrices
# -------------------------------------------------------------------------------------------------
def _adjacent_grads(params):
    """True when every parameter wants a gradient and their gradient buffers follow each other in memory (same storage)."""
    if any(p is None or not p.requires_grad for p in params):
        return False
    bufs = [grad_buffer(p) for p in params]
    st = bufs[0].untyped_storage().data_ptr()
    nxt = bufs[0].data_ptr()
    for b in bufs:
        if b.data_ptr() != nxt or b.untyped_storage().data_ptr() != st or not b.is_contiguous():
            return False
        nxt += b.numel() * b.element_size()
    return True


class LinearFunction(torch.autograd.Function):
    """y = dropout(act(x W^T + b)) + residual with W = stack(weights).  Reference call sites: nn.Linear
    in multihead_attention.py:213-231,397, speech_encoder_prenet.py:177, speech_decoder_prenet.py,
    speech_decoder_postnet.py:61-63, text_decoder_postnet.py:59-65, speech_encoder_postnet.py:88."""

    @staticmethod
    def forward(ctx, x, residual, act, dropout_p, nw, *wb):
        relay_in = None
        if isinstance(nw, tuple):
            nw, relay_in = nw
        relay_out = None
        if isinstance(residual, _RelayedResidual):
            residual, relay_out = residual.tensor, residual.relay
        weights, biases = wb[:nw], wb[nw:]
        dtype = x.dtype
        x2 = _rows(x)
        M, K = x2.shape
        N = sum(w.shape[0] for w in weights)
        ldn = _ceil8(N)
        Wc = fused_weight(weights, dtype)
        bc = fused_bias(biases)
        y = torch.empty(M, ldn, dtype=dtype, device=x.device)
        need_pre = act != ACT_NONE and any(ctx.needs_input_grad)
        pre = torch.empty(M, ldn, dtype=dtype, device=x.device) if need_pre else None
        seed = next_seed() if dropout_p > 0 else 0
        if dropout_p > 0:
            assert ldn == N, "dropout epilogue requires an 8-aligned output width"
        res2 = None
        if residual is not None:
            res2 = _rows(residual)
            assert res2.shape == (M, N) and ldn == N
        if M > 0:
            _nt_gemm(x2, weights, False, hip.operand(y, ldn), M, N, K, dtype,
                     R=hip.operand(res2, ldn) if res2 is not None else None,
                     Cpre=hip.operand(pre, ldn) if pre is not None else None,
                     bias=bc, act=act, dropout_p=dropout_p, seed=seed)
        if dropout_p > 0 and act == ACT_NONE and M > 0:
            _tag_dropout_output(y, dropout_p, seed, M, N)
        ctx.save_for_backward(x2, pre, Wc)
        ctx.meta = (weights, biases, act, dropout_p, seed, M, N, K, ldn, x.shape, residual is not None and relay_out is None)
        ctx.relays = (relay_out, relay_in)
        out = y if ldn == N else y[:, :N]
        return out.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, pre, Wc = ctx.saved_tensors
        weights, biases, act, p, seed, M, N, K, ldn, xshape, has_res = ctx.meta
        dtype = x2.dtype
        dy2 = dy.reshape(M, N)
        d_res = dy if has_res else None
        relay_out, relay_in = ctx.relays
        if relay_out is not None:
            relay_out.grad = dy.contiguous().view(M, N)   # picked up by the block's first Linear (same backward pass)
        extra = None
        if relay_in is not None and relay_in.grad is not None:
            extra, relay_in.grad = relay_in.grad, None
        g = _padded(dy2, ldn) if (not dy2.is_contiguous() or ldn != N) else dy2
        if M == 0:
            return (torch.zeros(xshape, dtype=dtype, device=x2.device), d_res, None, None, None) + (None,) * (2 * len(weights))
        if p > 0:
            g = _dropped_grad(g, p, seed)
        if act != ACT_NONE:
            g2 = torch.empty_like(g)
            hip.check(hip.lib().st5_act_bwd(g.data_ptr(), pre.data_ptr(), g2.data_ptr(), g.numel(), act, _dt(g),
                                            hip.stream()), "st5_act_bwd")
            g = g2
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=dtype, device=x2.device)
            if N % 8 == 0:
                # dX = G . W as an NT GEMM against the cached transposed weights W^T [K, N]
                _nt_gemm(g, weights, True, hip.operand(dx, K), M, K, N, dtype,
                         R=hip.operand(extra, K) if extra is not None else None)
                extra = None
            else:  # odd widths (vocabulary): B operand k-strided, W is [N, K], reduction over N
                hip.gemm(hip.operand(g, ldn), hip.operand(Wc, Wc.shape[1]), hip.operand(dx, K), M, K, N, _dt(dtype),
                         flags=hip.B_KSTRIDED, R=hip.operand(extra, K) if extra is not None else None)
                extra = None
            dx = dx.view(xshape)
        assert extra is None, "a relayed residual gradient reached a Linear whose input needs no gradient"
        if len(weights) > 1 and _adjacent_grads(weights) and (all(b is None for b in biases) or _adjacent_grads(biases)):
            # the stacked projections' gradients are ONE contiguous [sum N_i, K] block of the flat gradient buffer
            # (ddp._fusion_ordered_parameters): a single weight-gradient GEMM (+ bias column) for all of them
            gw = torch.as_strided(grad_buffer(weights[0]), (N, K), (K, 1))
            gb = None
            if biases[0] is not None:
                gb = torch.as_strided(grad_buffer(biases[0]), (N,), (1,))
            _wgrad_gemm(weights, hip.operand(g, ldn), hip.operand(x2, K), hip.operand(gw, K), N, K, M, _dt(dtype), gb, (g, x2))
            for w in weights:
                _grad_done(w)
            for b in biases:
                if b is not None:
                    _grad_done(b)
            return (dx, d_res, None, None, None) + (None,) * (2 * len(weights))
        off = 0
        for i, w in enumerate(weights):
            n_i = w.shape[0]
            b = biases[i]
            want_db = b is not None and b.requires_grad
            if w.requires_grad:
                gw = grad_buffer(w)
                # dW_i += G[:, off:off+n_i]^T . X   (both operands k-strided, fp32 accumulate in place); the bias
                # gradient db_i += colsum(G_i) rides along as one extra MFMA column of the same kernel
                _wgrad_gemm((w, b) if want_db else (w,), hip.operand(g, ldn, off=off), hip.operand(x2, K), hip.operand(gw, K),
                            n_i, K, M, _dt(dtype), grad_buffer(b) if want_db else None, (g, x2))
                _grad_done(w)
            elif want_db:
                _colsum_into(g, ldn, n_i, grad_buffer(b), col_off=off)
            if want_db:
                _grad_done(b)
            off += n_i
        return (dx, d_res, None, None, None) + (None,) * (2 * len(weights))


class GradRelay:
    """Carries the gradient of a residual connection from the Linear that ADDS the residual (end of an attention block) to
    the Linear that CONSUMES the same tensor (start of the block): the consumer's dX GEMM adds it in its epilogue, so
    autograd never has to sum the two gradients of the forked tensor with a separate kernel.  Valid when the residual IS
    the block input (post-LN layers); backward visits the adding Linear first by construction of the graph."""

    def __init__(self):
        self.grad = None


def linear(x, weights, biases=None, act=ACT_NONE, residual=None, dropout_p=0.0, relay_out=None, relay_in=None):
    """x [..., K] (compute dtype); weights: a Parameter [N,K] or a list of them (outputs concatenated).
    relay_out: this call adds `residual`; hand its gradient to the relay instead of returning it to autograd.
    relay_in: add the relayed gradient to this call's input gradient."""
    if isinstance(weights, torch.Tensor):
        weights, biases = [weights], [biases]
    if biases is None:
        biases = [None] * len(weights)
    x = x.contiguous()
    if relay_out is not None and residual is not None:
        residual = _RelayedResidual(residual.detach(), relay_out)
    y = LinearFunction.apply(x, residual, act, float(dropout_p), len(weights) if relay_in is None else (len(weights), relay_in),
                             *weights, *biases)
    return y


class _RelayedResidual:
    def __init__(self, tensor, relay):
        self.tensor, self.relay = tensor, relay


# -------------------------------------------------------------------------------------------------
# Feed-forward block: fc1 + GELU (+dropout) + fc2 (+dropout) + residual, activation derivative fused
# into the fc2 dgrad epilogue (transformer_layer.py:127-131, 385-389)
# -------------------------------------------------------------------------------------------------
class FFNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, act, p_act, p_out, relay_out=None):
        dtype = x.dtype
        x2 = _rows(x)
        M, d = x2.shape
        Fd = w1.shape[0]
        W1, W2 = fused_weight([w1], dtype), fused_weight([w2], dtype)
        h = torch.empty(M, Fd, dtype=dtype, device=x.device)
        hpre = torch.empty(M, Fd, dtype=dtype, device=x.device)
        y = torch.empty(M, w2.shape[0], dtype=dtype, device=x.device)
        s1 = next_seed() if p_act > 0 else 0
        s2 = next_seed() if p_out > 0 else 0
        # residual == "x": the block input itself is the residual (post-LN layers).  Its gradient dY is then added in the
        # epilogue of the dX GEMM instead of by a separate autograd accumulation kernel.
        res_is_x = isinstance(residual, str)
        res2 = x2 if res_is_x else (_rows(residual) if residual is not None else None)
        _nt_gemm(x2, [w1], False, hip.operand(h, Fd), M, Fd, d, dtype, quant_out=h, Cpre=hip.operand(hpre, Fd),
                 bias=b1.detach(), act=act, dropout_p=p_act, seed=s1)
        _nt_gemm(h, [w2], False, hip.operand(y, w2.shape[0]), M, w2.shape[0], Fd, dtype,
                 R=hip.operand(res2, w2.shape[0]) if res2 is not None else None, bias=b2.detach(), dropout_p=p_out, seed=s2)
        if p_out > 0:
            _tag_dropout_output(y, p_out, s2, M, w2.shape[0])
        ctx.save_for_backward(x2, h, hpre, W1, W2)
        ctx.meta = (w1, b1, w2, b2, act, p_act, p_out, s1, s2, x.shape, residual is not None and not res_is_x, res_is_x)
        ctx.relay_out = relay_out if (residual is not None and not res_is_x) else None
        return y.view(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, h, hpre, W1, W2 = ctx.saved_tensors
        w1, b1, w2, b2, act, p_act, p_out, s1, s2, xshape, has_res, res_is_x = ctx.meta
        dtype = x2.dtype
        M, d = x2.shape
        Fd, dout = w1.shape[0], w2.shape[0]
        g = dy.contiguous().view(M, dout)
        g_in = g
        d_res = dy if has_res else None
        if ctx.relay_out is not None:      # (pre-LN block: the residual's gradient goes to the LayerNorm that opened the block)
            ctx.relay_out.grad = g
            d_res = None
        if p_out > 0:
            g = _dropped_grad(g, p_out, s2)
        # dHpre = (G . W2) * act'(Hpre) [* activation-dropout mask]   (fused epilogue)
        dh = torch.empty(M, Fd, dtype=dtype, device=x2.device)
        _nt_gemm(g, [w2], True, hip.operand(dh, Fd), M, Fd, dout, dtype,     # (B operand: W2^T [Fd, dout])
                 quant_out=dh if ctx.needs_input_grad[0] else None,
                 P=hip.operand(hpre, Fd), act=act, flags=hip.DACT, dropout_p=p_act, seed=s1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, d, dtype=dtype, device=x2.device)
            _nt_gemm(dh, [w1], True, hip.operand(dx, d), M, d, Fd, dtype,     # (B operand: W1^T [d, Fd])
                     R=hip.operand(g_in, d) if res_is_x else None)
            dx = dx.view(xshape)
        # weight gradients last: both data-gradient GEMMs of the block are enqueued before the side stream forks
        if w2.requires_grad:
            _wgrad_gemm((w2, b2), hip.operand(g, dout), hip.operand(h, Fd), hip.operand(grad_buffer(w2), Fd), dout, Fd, M,
                        _dt(dtype), grad_buffer(b2) if b2.requires_grad else None, (g, h))
            _grad_done(w2)
        elif b2.requires_grad:
            _colsum_into(g, dout, dout, grad_buffer(b2))
        if b2.requires_grad:
            _grad_done(b2)
        if w1.requires_grad:
            _wgrad_gemm((w1, b1), hip.operand(dh, Fd), hip.operand(x2, d), hip.operand(grad_buffer(w1), d), Fd, d, M,
                        _dt(dtype), grad_buffer(b1) if b1.requires_grad else None, (dh, x2))
            _grad_done(w1)
        elif b1.requires_grad:
            _colsum_into(dh, Fd, Fd, grad_buffer(b1))
        if b1.requires_grad:
            _grad_done(b1)
        return dx, d_res, None, None, None, None, None, None, None, None


def ffn(x, residual, fc1, fc2, act=ACT_GELU, p_act=0.0, p_out=0.0, relay_out=None):
    """relay_out (GradRelay): hand the residual's gradient to the relay (the LayerNorm that opened this pre-LN block adds it to its dX)
    instead of returning it to autograd."""
    xc = x.contiguous()
    if residual is x and fc2.weight.shape[0] == x.shape[-1]:
        residual = "x"
    if relay_out is not None and isinstance(residual, torch.Tensor):
        return FFNFunction.apply(xc, residual.detach(), fc1.weight, fc1.bias, fc2.weight, fc2.bias, act, float(p_act), float(p_out), relay_out)
    return FFNFunction.apply(xc, residual, fc1.weight, fc1.bias, fc2.weight, fc2.bias, act, float(p_act), float(p_out))


# -------------------------------------------------------------------------------------------------
# (Label-smoothed) cross entropy, summed over rows: log-softmax, gather, smoothing term and the logit gradient in one
# kernel (speech_pretrain_criterion.py:98-141, text_pretrain_criterion.py:56-60, speech_to_text_loss.py:93-110)
# -------------------------------------------------------------------------------------------------
class CrossEntropySumFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, eps, ignore_index):
        rows, V = logits.shape
        dev = logits.device
        ld = logits.stride(0)
        assert logits.stride(1) == 1 and ld >= V
        t32 = target.to(torch.int32)
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        row_nll = torch.empty(rows, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        dlogits = torch.empty(rows, ld, dtype=logits.dtype, device=dev) if need else None
        hip.check(hip.lib().st5_cross_entropy_rows(logits.data_ptr(), t32.data_ptr(), row_loss.data_ptr(), row_nll.data_ptr(),
                                                   hip.ptr(dlogits), rows, V, ld, float(eps),
                                                   -1 if ignore_index is None else int(ignore_index), 1.0, _dt(logits),
                                                   hip.stream()), "st5_cross_entropy_rows")
        ctx.save_for_backward(dlogits)
        ctx.V = V
        nll = row_nll.sum()
        ctx.mark_non_differentiable(nll)
        return row_loss.sum(), nll

    @staticmethod
    def backward(ctx, g_loss, _g_nll):
        (dlogits,) = ctx.saved_tensors
        g = dlogits[:, :ctx.V] if dlogits.shape[1] != ctx.V else dlogits
        return g * g_loss.to(g.dtype), None, None, None


def cross_entropy_sum(logits, target, label_smoothing=0.0, ignore_index=None):
    """logits [rows, V] (fp32 or the compute dtype; -inf entries allowed), target int64 [rows] -> (sum of row losses,
    sum of row NLLs), both fp32 scalars on the device; rows whose target is ignore_index (or negative) contribute 0.
    loss_r = (1 - eps - eps_i) nll_r + eps_i smooth_r with eps_i = eps / (V - 1) (fairseq label_smoothed_nll_loss)."""
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    if logits.shape[0] == 0:
        z = logits.sum()
        return z, z.detach()
    return CrossEntropySumFunction.apply(logits if logits.stride(-1) == 1 else logits.contiguous(), target, float(label_smoothing),
                                         ignore_index)


class TacotronLossFunction(torch.autograd.Function):
    """(l1, mse, bce) of Tacotron2Loss with masking in one reduction pass (csrc/losses.hip); the gradient pass produces
    d_after / d_before / d_logits from the three upstream scalar gradients without leaving the device."""

    @staticmethod
    def forward(ctx, after, before, logits, ys, labels, olens, r, pos_weight):
        B, L, C = after.shape
        after, before, logits = after.contiguous(), before.contiguous(), logits.contiguous()
        assert after.dtype == before.dtype == logits.dtype == ys.dtype == labels.dtype == torch.float32
        assert ys.stride(2) == 1 and ys.stride(1) == C and labels.stride(1) == 1 and ys.shape[1] >= L and labels.shape[1] >= L
        olens = olens.to(device=after.device, dtype=torch.int64).contiguous()
        L_ = hip.lib()
        out = torch.empty(4, dtype=torch.float32, device=after.device)
        ws = hip.workspace(L_.st5_tacotron_loss_ws_bytes(), after.device)
        hip.check(L_.st5_tacotron_loss_fwd(after.data_ptr(), before.data_ptr(), logits.data_ptr(), ys.data_ptr(), ys.stride(0),
                                           labels.data_ptr(), labels.stride(0), olens.data_ptr(), B, L, C, int(r), float(pos_weight),
                                           out.data_ptr(), ws.data_ptr(), hip.stream()), "st5_tacotron_loss_fwd")
        ctx.save_for_backward(after, before, logits, ys, labels, olens, out)
        ctx.meta = (int(r), float(pos_weight))
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g1, g2, g3):
        after, before, logits, ys, labels, olens, out = ctx.saved_tensors
        r, pw = ctx.meta
        B, L, C = after.shape
        need = ctx.needs_input_grad
        da = torch.empty_like(after) if need[0] else None
        db = torch.empty_like(before) if need[1] else None
        dl = torch.empty_like(logits) if need[2] else None
        gs = [g.contiguous() if g is not None else None for g in (g1, g2, g3)]
        hip.check(hip.lib().st5_tacotron_loss_bwd(after.data_ptr(), before.data_ptr(), logits.data_ptr(), ys.data_ptr(), ys.stride(0),
                                                  labels.data_ptr(), labels.stride(0), olens.data_ptr(), B, L, C, r, pw, out.data_ptr(),
                                                  hip.ptr(gs[0]), hip.ptr(gs[1]), hip.ptr(gs[2]), hip.ptr(da), hip.ptr(db), hip.ptr(dl),
                                                  hip.stream()), "st5_tacotron_loss_bwd")
        return da, db, dl, None, None, None, None, None


def tacotron_loss(after, before, logits, ys, labels, olens, r, pos_weight):
    """Masked L1 / MSE / stop-token BCE of the speech decoder (text_to_speech_loss.py:186-216, :296-345): `olens` are the
    untrimmed target lengths, trimmed to a multiple of `r` inside (with the last valid frame's stop label forced to 1)."""
    return TacotronLossFunction.apply(after, before, logits, ys, labels, olens, r, pos_weight)


class CTCLossFunction(torch.autograd.Function):
    """Sum of the CTC negative log-likelihoods (csrc/ctc_loss.hip): one block per sentence walks the alpha recursion, the
    backward the beta recursion and the per-class posteriors.  Gradient in the form torch's ctc_loss returns."""

    @staticmethod
    def forward(ctx, lprobs, targets, target_offsets, input_lengths, target_lengths, max_target_len, blank, zero_infinity):
        T, B, V = lprobs.shape
        lprobs = lprobs.contiguous()
        assert lprobs.dtype == torch.float32, "CTC runs on fp32 log-probabilities (speech_to_text_loss.py:324: log_softmax(...).float())"
        dev = lprobs.device
        idx = [t.to(device=dev, dtype=torch.int64).contiguous() for t in (targets, target_offsets, input_lengths, target_lengths)]
        L_ = hip.lib()
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        alpha = torch.empty(max(int(L_.st5_ctc_loss_ws_bytes(T, B, int(max_target_len))), 4) // 4, dtype=torch.float32, device=dev)
        hip.check(L_.st5_ctc_loss_fwd(lprobs.data_ptr(), hip.ptr(idx[0]) if idx[0].numel() else 0, idx[1].data_ptr(), idx[2].data_ptr(),
                                      idx[3].data_ptr(), T, B, V, int(max_target_len), int(blank), int(bool(zero_infinity)),
                                      nll.data_ptr(), loss.data_ptr(), alpha.data_ptr(), hip.stream()), "st5_ctc_loss_fwd")
        ctx.save_for_backward(lprobs, *idx, nll, alpha)
        ctx.meta = (int(max_target_len), int(blank), int(bool(zero_infinity)))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lprobs, targets, offsets, in_len, tg_len, nll, alpha = ctx.saved_tensors
        maxL, blank, zi = ctx.meta
        T, B, V = lprobs.shape
        grad = torch.empty_like(lprobs)
        g = g.contiguous().float()
        hip.check(hip.lib().st5_ctc_loss_bwd(lprobs.data_ptr(), hip.ptr(targets) if targets.numel() else 0, offsets.data_ptr(),
                                             in_len.data_ptr(), tg_len.data_ptr(), T, B, V, maxL, blank, zi, nll.data_ptr(), g.data_ptr(),
                                             alpha.data_ptr(), grad.data_ptr(), hip.stream()), "st5_ctc_loss_bwd")
        return grad, None, None, None, None, None, None, None


def ctc_loss_sum(lprobs, targets_flat, input_lengths, target_lengths, blank, zero_infinity=True, max_target_len=None):
    """F.ctc_loss(lprobs [T, B, V], flat targets, input_lengths, target_lengths, blank, reduction="sum", zero_infinity)
    (speech_to_text_loss.py:333-337).  max_target_len: an upper bound known on the host (default: read from the device)."""
    target_lengths = target_lengths.to(lprobs.device)
    offsets = torch.cumsum(target_lengths, 0) - target_lengths
    if max_target_len is None:
        max_target_len = int(target_lengths.max()) if target_lengths.numel() else 0
    return CTCLossFunction.apply(lprobs, targets_flat, offsets, input_lengths, target_lengths, max_target_len, blank, zero_infinity)


class GuidedAttentionFunction(torch.autograd.Function):
    """alpha * mean of the guided-attention penalty over the valid (to, ti) region (csrc/ctc_loss.hip)."""

    @staticmethod
    def forward(ctx, att, ilens, olens, sigma, alpha):
        B, H, To, Ti = att.shape
        att = att.contiguous()
        assert att.dtype == torch.float32
        dev = att.device
        ilens = ilens.to(device=dev, dtype=torch.int64).contiguous()
        olens = olens.to(device=dev, dtype=torch.int64).contiguous()
        L_ = hip.lib()
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ws = hip.workspace(L_.st5_guided_attn_ws_bytes(), dev)
        hip.check(L_.st5_guided_attn_fwd(att.data_ptr(), ilens.data_ptr(), olens.data_ptr(), B, H, To, Ti, float(sigma), float(alpha),
                                         out.data_ptr(), ws.data_ptr(), hip.stream()), "st5_guided_attn_fwd")
        ctx.save_for_backward(ilens, olens, out)
        ctx.meta = (B, H, To, Ti, float(sigma))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        ilens, olens, out = ctx.saved_tensors
        B, H, To, Ti, sigma = ctx.meta
        datt = torch.empty(B, H, To, Ti, dtype=torch.float32, device=out.device)
        g = g.contiguous().float()
        hip.check(hip.lib().st5_guided_attn_bwd(ilens.data_ptr(), olens.data_ptr(), B, H, To, Ti, sigma, out.data_ptr(), g.data_ptr(),
                                                datt.data_ptr(), hip.stream()), "st5_guided_attn_bwd")
        return datt, None, None, None, None


def guided_attention_loss(att, ilens, olens, sigma, alpha):
    """GuidedMultiHeadAttentionLoss (text_to_speech_loss.py:370-427) on att [B, H, To, Ti] fp32."""
    return GuidedAttentionFunction.apply(att, ilens, olens, sigma, alpha)


# -------------------------------------------------------------------------------------------------
# LayerNorm
# -------------------------------------------------------------------------------------------------
class LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, gate=None, q8=False, relay=None):
        x2 = _rows(x)
        rows, cols = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        if q8 and gate is None and _FP8.enabled and (_FP8.fuse & 1) and x2.dtype == torch.bfloat16 and cols % 128 == 0 and cols <= 2048 and \
                rows >= _FP8.min_rows:
            # fp8 mode: the consumer of a pre-LN layer's LayerNorm is an fp8 GEMM (QKV / fc1) -- its operand image comes out of this pass
            q, sc = _q8_buffers(rows, cols, x.device)
            hip.check(hip.lib().st5_layernorm_fwd_q8(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                     rstd.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, cols, eps, hip.stream()),
                      "st5_layernorm_fwd_q8")
            _q8_tag(y, q, sc, rows, cols)
        elif gate is None:
            hip.check(hip.lib().st5_layernorm_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                  rstd.data_ptr(), rows, cols, eps, _dt(x), hip.stream()), "st5_layernorm_fwd")
        else:       # LayerDrop gate: y = keep ? LN(x) : the layer's input
            skip = gate.skip
            assert skip.shape == x2.shape and skip.dtype == x2.dtype and skip.is_contiguous()
            hip.check(hip.lib().st5_layernorm_gated_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                        rstd.data_ptr(), rows, cols, eps, gate.keep.data_ptr(), skip.data_ptr(), _dt(x),
                                                        hip.stream()), "st5_layernorm_gated_fwd")
        tag = _drop_tags.pop(x2.data_ptr(), None)
        if tag is not None and (tag[2:] != (rows, cols) or cols % 4 or cols > 2048):
            tag = None
        ctx.save_for_backward(x2, mean, rstd)
        ctx.meta = (weight, bias, x.shape, tag)
        ctx.gate = gate
        ctx.relay = relay
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        weight, bias, xshape, tag = ctx.meta
        rows, cols = x2.shape
        g = dy.contiguous().view(rows, cols)
        L = hip.lib()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dxd = None
        if tag is not None and dx is not None:
            dxd = torch.empty_like(x2)
            if len(_drop_grads) > 64:
                _drop_grads.clear()
            _drop_grads[dx.data_ptr()] = (tag[0], tag[1], rows, cols, dxd)
        gw = grad_buffer(weight) if weight.requires_grad else None
        gb = grad_buffer(bias) if bias.requires_grad else None
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), x2.device)
        gate = ctx.gate
        relay, addend = ctx.relay, None
        if relay is not None and relay.grad is not None:
            # a pre-LN block's residual gradient (handed over by the Linear / FFN that added the residual, earlier in this backward pass)
            addend, relay.grad = relay.grad, None
        if addend is not None and gate is None and dx is not None and dxd is None:
            assert addend.shape == (rows, cols) and addend.dtype == x2.dtype and addend.is_contiguous()
            hip.check(L.st5_layernorm_bwd_add(g.data_ptr(), x2.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                              hip.ptr(gw), hip.ptr(gb), ws.data_ptr(), rows, cols, addend.data_ptr(), _dt(x2), hip.stream()),
                      "st5_layernorm_bwd_add")
            addend = None
        elif gate is None:
            hip.check(L.st5_layernorm_bwd(g.data_ptr(), x2.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          hip.ptr(dx), hip.ptr(gw), hip.ptr(gb), ws.data_ptr(), rows, cols, hip.ptr(dxd),
                                          tag[0] if dxd is not None else 0.0, tag[1] if dxd is not None else 0, _dt(x2),
                                          hip.stream()), "st5_layernorm_bwd")
        else:       # a dropped layer's output gradient counts as zero here and reaches the layer's input through the gate
            assert dx is not None
            gate.g_out = g
            hip.check(L.st5_layernorm_gated_bwd(g.data_ptr(), x2.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                hip.ptr(dx), hip.ptr(gw), hip.ptr(gb), ws.data_ptr(), rows, cols, hip.ptr(dxd),
                                                tag[0] if dxd is not None else 0.0, tag[1] if dxd is not None else 0,
                                                gate.keep.data_ptr(), _dt(x2), hip.stream()), "st5_layernorm_gated_bwd")
        if gw is not None:
            _grad_done(weight)
        if gb is not None:
            _grad_done(bias)
        if addend is not None:      # (a form the fused kernel does not cover: add it here, as autograd would have)
            assert dx is not None, "a relayed residual gradient reached a LayerNorm whose input needs no gradient"
            dx = dx + addend
        return (dx.view(xshape) if dx is not None else None), None, None, None, None, None, None


class LayerNormGeluFunction(torch.autograd.Function):
    """GELU(LayerNorm(x)) as one forward and one backward pass (st5_layernorm_gelu_fwd / _bwd): the layer-norm feature extractor of
    t5_transformer_large (speech_encoder_prenet.py:318-331) ran LayerNorm and GELU as two passes each way over [B * T, 512] rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x2 = _rows(x)
        rows, cols = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().st5_layernorm_gelu_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), rows, cols, eps, _dt(x), hip.stream()), "st5_layernorm_gelu_fwd")
        ctx.save_for_backward(x2, mean, rstd)
        ctx.meta = (weight, bias, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        weight, bias, xshape = ctx.meta
        rows, cols = x2.shape
        g = dy.contiguous().view(rows, cols)
        L = hip.lib()
        dx = torch.empty_like(x2)
        gw = grad_buffer(weight) if weight.requires_grad else None
        gb = grad_buffer(bias) if bias.requires_grad else None
        ws = hip.workspace(L.st5_layernorm_bwd_ws_bytes(rows, cols), x2.device)
        hip.check(L.st5_layernorm_gelu_bwd(g.data_ptr(), x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           dx.data_ptr(), hip.ptr(gw), hip.ptr(gb), ws.data_ptr(), rows, cols, _dt(x2), hip.stream()),
                  "st5_layernorm_gelu_bwd")
        if gw is not None:
            _grad_done(weight)
        if gb is not None:
            _grad_done(bias)
        return dx.view(xshape), None, None, None


def layer_norm_gelu(x, weight, bias, eps=1e-5):
    """GELU(LayerNorm(x)); one fused pass each way when the row width allows (<= 512, a multiple of 4), else the composition."""
    if x.shape[-1] % 4 == 0 and x.shape[-1] <= 512 and os.environ.get("ST5_LN_GELU_FUSED", "1") != "0":
        return LayerNormGeluFunction.apply(x.contiguous(), weight, bias, float(eps))
    return activation(layer_norm(x, weight, bias, eps), ACT_GELU)


def layer_norm(x, weight, bias, eps=1e-5, gate=None, q8=False, relay=None):
    """q8: the output's consumer is a Linear that may run on the fp8 GEMM (a pre-LN layer's QKV projection / fc1): in fp8 compute mode
    the kernel then writes the output's MX-fp8 image beside it (st5_layernorm_fwd_q8) and that Linear launches no quantiser.
    relay (GradRelay): this LayerNorm opens a pre-LN block y = x + f(LN(x)); the Linear / FFN that adds the residual hands dY to the relay
    and this function's backward adds it to dX inside its kernel (st5_layernorm_bwd_add) -- no autograd accumulation kernel."""
    if gate is not None:
        gate.used = True
        return LayerNormFunction.apply(x.contiguous(), weight, bias, float(eps), gate)
    if relay is not None or (q8 and _FP8.enabled):
        return LayerNormFunction.apply(x.contiguous(), weight, bias, float(eps), None, bool(q8 and _FP8.enabled), relay)
    return LayerNormFunction.apply(x.contiguous(), weight, bias, float(eps))


# -------------------------------------------------------------------------------------------------
# Attention core (scores, rel-pos bias, masks, softmax, dropout, P.V) on projected q / k / v buffers
# -------------------------------------------------------------------------------------------------
def _attn_fwd(q, k, v, B, H, T, S, hd, pe, maxrel, kpm, causal, p_drop, seed):
    """q/k/v: (tensor, ld, col_off) of projected activations, rows batch-major.  Returns ctx [B*T, H*hd],
    probs [B*H, T, lds], probs_drop (or None)."""
    qt, qld, qoff = q
    kt, kld, koff = k
    vt, vld, voff = v
    dtype = qt.dtype
    dev = qt.device
    d = H * hd
    lds = _ceil8(S)
    alpha = hd ** -0.5
    BH = B * H
    scores = torch.empty(BH, T, lds, dtype=dtype, device=dev)
    if lds != S:
        scores[:, :, S:].zero_()
    hip.gemm(hip.operand(qt, qld, off=qoff, zs0=T * qld, zs1=hd), hip.operand(kt, kld, off=koff, zs0=S * kld, zs1=hd),
             hip.operand(scores, lds, zs0=H * T * lds, zs1=T * lds), T, S, hd, _dt(dtype), batch=BH, zdiv=H, alpha=alpha)
    qp, nb = None, 0
    if pe is not None:
        nb = pe.shape[0]
        qp = torch.empty(BH, T, nb, dtype=dtype, device=dev)
        hip.gemm(hip.operand(qt, qld, off=qoff, zs0=T * qld, zs1=hd), hip.operand(pe, hd),
                 hip.operand(qp, nb, zs0=H * T * nb, zs1=T * nb), T, nb, hd, _dt(dtype), batch=BH, zdiv=H, alpha=alpha)
    probs = torch.empty(BH, T, lds, dtype=dtype, device=dev)
    pdrop = torch.empty(BH, T, lds, dtype=dtype, device=dev) if p_drop > 0 else None
    hip.check(hip.lib().st5_softmax_fwd(scores.data_ptr(), hip.ptr(qp), hip.ptr(kpm), probs.data_ptr(), hip.ptr(pdrop), BH, H,
                                        T, S, lds, nb, maxrel, 1 if causal else 0, p_drop, seed, _dt(dtype), hip.stream()),
              "st5_softmax_fwd")
    ctx = torch.empty(B * T, d, dtype=dtype, device=dev)
    pa = pdrop if pdrop is not None else probs
    hip.gemm(hip.operand(pa, lds, zs0=H * T * lds, zs1=T * lds), hip.operand(vt, vld, off=voff, zs0=S * vld, zs1=hd),
             hip.operand(ctx, d, zs0=T * d, zs1=hd), T, hd, S, _dt(dtype), batch=BH, zdiv=H, flags=hip.B_KSTRIDED)
    return ctx, probs, pdrop


def _attn_bwd(dctx, q, k, v, dq, dk, dv, probs, pdrop, dP_extra, B, H, T, S, hd, pe, want_dpe, maxrel, p_drop, seed):
    """Writes dq/dk/dv = (tensor, ld, col_off) slices; returns dPE (fp32 [nb, hd]) or None."""
    qt, qld, qoff = q
    kt, kld, koff = k
    vt, vld, voff = v
    dqt, dqld, dqoff = dq
    dkt, dkld, dkoff = dk
    dvt, dvld, dvoff = dv
    dtype = qt.dtype
    dev = qt.device
    d = H * hd
    lds = probs.shape[2]
    alpha = hd ** -0.5
    BH = B * H
    zP = dict(zs0=H * T * lds, zs1=T * lds)
    # dP = dCtx . V^T
    dP = torch.empty(BH, T, lds, dtype=dtype, device=dev)
    hip.gemm(hip.operand(dctx, d, zs0=T * d, zs1=hd), hip.operand(vt, vld, off=voff, zs0=S * vld, zs1=hd),
             hip.operand(dP, lds, **zP), T, S, hd, _dt(dtype), batch=BH, zdiv=H)
    # dV = Pdrop^T . dCtx
    pa = pdrop if pdrop is not None else probs
    hip.gemm(hip.operand(pa, lds, **zP), hip.operand(dctx, d, zs0=T * d, zs1=hd),
             hip.operand(dvt, dvld, off=dvoff, zs0=S * dvld, zs1=hd), S, hd, T, _dt(dtype), batch=BH, zdiv=H,
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED)
    nb = pe.shape[0] if pe is not None else 0
    dqp = torch.empty(BH, T, nb, dtype=dtype, device=dev) if pe is not None else None
    hip.check(hip.lib().st5_softmax_bwd(dP.data_ptr(), probs.data_ptr(), hip.ptr(dP_extra), hip.ptr(dqp), BH, T, S, lds, nb,
                                        maxrel, p_drop, seed, _dt(dtype), hip.stream()), "st5_softmax_bwd")
    dS = dP
    # dQ = alpha * dS . K  (+ alpha * dQP . PE)
    hip.gemm(hip.operand(dS, lds, **zP), hip.operand(kt, kld, off=koff, zs0=S * kld, zs1=hd),
             hip.operand(dqt, dqld, off=dqoff, zs0=T * dqld, zs1=hd), T, hd, S, _dt(dtype), batch=BH, zdiv=H,
             flags=hip.B_KSTRIDED, alpha=alpha)
    if pe is not None:
        hip.gemm(hip.operand(dqp, nb, zs0=H * T * nb, zs1=T * nb), hip.operand(pe, hd),
                 hip.operand(dqt, dqld, off=dqoff, zs0=T * dqld, zs1=hd), T, hd, nb, _dt(dtype), batch=BH, zdiv=H,
                 flags=hip.B_KSTRIDED, alpha=alpha, beta=1.0)
    # dK = alpha * dS^T . Q
    hip.gemm(hip.operand(dS, lds, **zP), hip.operand(qt, qld, off=qoff, zs0=T * qld, zs1=hd),
             hip.operand(dkt, dkld, off=dkoff, zs0=S * dkld, zs1=hd), S, hd, T, _dt(dtype), batch=BH, zdiv=H,
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED, alpha=alpha)
    if pe is not None and want_dpe:
        # dPE = alpha * sum_{b,h} dQP[b,h]^T . Q[b,h]: one [nb x hd] product per (b,h) (K = T), then a column
        # reduction over the B*H partials (fp32)
        part = torch.empty(BH, nb, hd, dtype=torch.float32, device=dev)
        hip.gemm(hip.operand(dqp, nb, zs0=H * T * nb, zs1=T * nb),
                 hip.operand(qt, qld, off=qoff, zs0=T * qld, zs1=hd),
                 hip.operand(part, hd, zs0=H * nb * hd, zs1=nb * hd), nb, hd, T, _dt(dtype), batch=BH, zdiv=H,
                 flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, alpha=alpha)
        g = torch.zeros(nb, hd, dtype=torch.float32, device=dev)
        L = hip.lib()
        ws = hip.workspace(L.st5_colsum_ws_bytes(BH, nb * hd), dev)
        hip.check(L.st5_colsum_ws(part.data_ptr(), g.data_ptr(), ws.data_ptr(), BH, nb * hd, nb * hd, 1.0, 1, hip.F32,
                                  hip.stream()), "st5_colsum_ws")
        return g
    return None



# -------------------------------------------------------------------------------------------------
# fused (flash) attention path: bf16, head_dim 64, no probability output
# -------------------------------------------------------------------------------------------------
_FLASH = SimpleNamespace(enabled=True)


def set_flash_attention(enabled):
    """Switch between the fused attention kernels and the unfused GEMM+softmax path (bf16 only; fp32 is always unfused)."""
    _FLASH.enabled = bool(enabled)


def _can_flash(dtype, hd, want_probs):
    return _FLASH.enabled and dtype == torch.bfloat16 and hd == 64 and not want_probs


def _eptr(t3):
    t, _ld, off = t3
    return t.data_ptr() + off * t.element_size()


_LOG2E = 1.4426950408889634


def _flash_fwd(q, k, v, B, H, T, S, hd, pe, maxrel, kpm, causal, p_drop, seed, save_qp=False):
    """Returns ctx, lse, qp.  save_qp: keep the bucket table scale*log2e*q.pe^T [B*H, T, nb] the kernel builds (30 MB per
    encoder layer at cfg 2) for the backward instead of recomputing it there with a batched GEMM."""
    dev = q[0].device
    d = H * hd
    ctx = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H, T, dtype=torch.float32, device=dev)
    nb = pe.shape[0] if pe is not None else 0
    # relative-position table workspace ([.., 8 | nb | 8] rows for the second-generation kernels); kept for the backward
    qp = torch.empty(B * H, T, hip.lib().st5_flash_attn_qp_row(nb), dtype=torch.bfloat16, device=dev) if pe is not None else None
    hip.check(hip.lib().st5_flash_attn_fwd_qp(_eptr(q), q[1], _eptr(k), k[1], _eptr(v), v[1], ctx.data_ptr(), d, lse.data_ptr(),
                                              hip.ptr(pe), hip.ptr(kpm), B, H, T, S, hd, nb, maxrel, 1 if causal else 0, _ceil8(S),
                                              hd ** -0.5, p_drop, seed, hip.ptr(qp), hip.BF16, hip.stream()), "st5_flash_attn_fwd_qp")
    return ctx, lse, (qp if save_qp else None)


def set_attention_stream(stream):
    """Second stream for the attention backward (dq and dkv kernels side by side, st5_flash_attn_bwd_2s); None = off."""
    _S.attn_side = stream
    _S.attn_side_raw = stream.cuda_stream if stream is not None else None
    _S.attn_owner = None


def _attn_side_raw():
    """Second stream of the attention backward (None = single-stream backward).  Only the stream that first asks gets it: with
    the micro-batches' backward passes on different streams (ddp.accumulate_overlapped) a helper stream forked from a second
    parent inside one graph capture crashed hipStreamEndCapture (also with one helper per parent), so backward passes running
    on another stream do dq and dkv in turn (the side-by-side form is worth ~0.3 ms per micro-batch)."""
    if _S.attn_side is None:
        return None
    cur = hip.stream()
    owner = _S.__dict__.setdefault("attn_owner", None)
    if owner is None:
        _S.attn_owner = owner = cur
    return _S.attn_side_raw if cur == owner else None


def _flash_bwd(dctx, ctx, lse, q, k, v, dq, dk, dv, B, H, T, S, hd, pe, want_dpe, maxrel, kpm, causal, p_drop, seed, qp=None,
               pe_t=None):
    """Writes dq/dk/dv slices; returns dPE (fp32) or None."""
    dev = dctx.device
    dtype = torch.bfloat16
    d = H * hd
    BH = B * H
    alpha = hd ** -0.5
    dvec = torch.empty(BH * T, dtype=torch.float32, device=dev)
    dqp = None
    nb = 0
    qt, qld, qoff = q
    if pe is not None:
        nb = pe.shape[0]
        if qp is None:   # the kernels work in the log2 domain: qp = scale * log2(e) * q.pe^T (+ replicated end chunks)
            qp = torch.empty(BH, T, hip.lib().st5_flash_attn_qp_row(nb), dtype=dtype, device=dev)
            hip.check(hip.lib().st5_flash_attn_qp_table(_eptr(q), q[1], pe.data_ptr(), qp.data_ptr(), B, H, T, nb, alpha, hip.BF16,
                                                        hip.stream()), "st5_flash_attn_qp_table")
        dqp = torch.empty(BH, T, nb, dtype=dtype, device=dev)
    hip.check(hip.lib().st5_flash_attn_bwd_2s(_eptr(q), q[1], _eptr(k), k[1], _eptr(v), v[1], ctx.data_ptr(), d, dctx.data_ptr(), d,
                                              _eptr(dq), dq[1], _eptr(dk), dk[1], _eptr(dv), dv[1], lse.data_ptr(), dvec.data_ptr(),
                                              hip.ptr(pe), hip.ptr(qp), hip.ptr(dqp), hip.ptr(kpm), B, H, T, S, hd, nb, maxrel,
                                              1 if causal else 0, _ceil8(S), alpha, p_drop, seed, hip.BF16, hip.stream(),
                                              _attn_side_raw()), "st5_flash_attn_bwd_2s")
    if pe is None:
        return None
    dqt, dqld, dqoff = dq
    # dQ += alpha * dQP . PE  (NT form on the LDS-DMA kernel when the caller made the K-major copy PE^T [hd, nb])
    if pe_t is not None:
        hip.gemm(hip.operand(dqp, nb, zs0=H * T * nb, zs1=T * nb), hip.operand(pe_t, nb),
                 hip.operand(dqt, dqld, off=dqoff, zs0=T * dqld, zs1=hd), T, hd, nb, hip.BF16, batch=BH, zdiv=H, alpha=alpha, beta=1.0)
    else:
        hip.gemm(hip.operand(dqp, nb, zs0=H * T * nb, zs1=T * nb), hip.operand(pe, hd),
                 hip.operand(dqt, dqld, off=dqoff, zs0=T * dqld, zs1=hd), T, hd, nb, hip.BF16, batch=BH, zdiv=H,
                 flags=hip.B_KSTRIDED, alpha=alpha, beta=1.0)
    if not want_dpe:
        return None
    part = torch.empty(BH, nb, hd, dtype=torch.float32, device=dev)
    hip.gemm(hip.operand(dqp, nb, zs0=H * T * nb, zs1=T * nb), hip.operand(qt, qld, off=qoff, zs0=T * qld, zs1=hd),
             hip.operand(part, hd, zs0=H * nb * hd, zs1=nb * hd), nb, hd, T, hip.BF16, batch=BH, zdiv=H,
             flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, alpha=alpha)
    g = torch.empty(nb, hd, dtype=torch.float32, device=dev)
    L = hip.lib()
    ws = hip.workspace(L.st5_colsum_ws_bytes(BH, nb * hd), dev)
    hip.check(L.st5_colsum_ws(part.data_ptr(), g.data_ptr(), ws.data_ptr(), BH, nb * hd, nb * hd, 1.0, 0, hip.F32, hip.stream()),
              "st5_colsum_ws")   # (overwrite form: two-stage, no zero fill, fixed summation order)
    return g


class SelfAttentionFunction(torch.autograd.Function):
    """Fused-QKV self-attention core.  qkv [B*T, 3d] -> ctx [B*T, d].  `pe` is the relative-position key
    table [2*maxrel, hd] in the compute dtype (a differentiable cast / norm_k of the parameter) or None."""

    @staticmethod
    def forward(ctx_, qkv, pe, kpm, cfg):
        B, H, T, hd, maxrel, causal, p_drop = cfg[:7]
        d = H * hd
        seed = next_seed() if p_drop > 0 else 0
        if _can_flash(qkv.dtype, hd, False):
            ctx, lse, qp = _flash_fwd((qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d), B, H, T, T, hd, pe, maxrel, kpm, causal,
                                      p_drop, seed, save_qp=any(ctx_.needs_input_grad))
            ctx_.save_for_backward(qkv, ctx, lse, pe, kpm, qp)
            ctx_.meta = (cfg, seed, True)
            return ctx
        ctx, probs, pdrop = _attn_fwd((qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d), B, H, T, T, hd, pe, maxrel,
                                      kpm, causal, p_drop, seed)
        ctx_.save_for_backward(qkv, probs, pdrop, pe)
        ctx_.meta = (cfg, seed, False)
        return ctx

    @staticmethod
    def backward(ctx_, dctx):
        cfg, seed, flash = ctx_.meta
        B, H, T, hd, maxrel, causal, p_drop = cfg[:7]
        pe_t = cfg[7] if len(cfg) > 7 else None
        d = H * hd
        if flash:
            qkv, ctx, lse, pe, kpm, qp = ctx_.saved_tensors
            dqkv = torch.empty_like(qkv)
            dpe = _flash_bwd(dctx.contiguous(), ctx, lse, (qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d), (dqkv, 3 * d, 0),
                             (dqkv, 3 * d, d), (dqkv, 3 * d, 2 * d), B, H, T, T, hd, pe, pe is not None and ctx_.needs_input_grad[1],
                             maxrel, kpm, causal, p_drop, seed, qp=qp, pe_t=pe_t)
            if dpe is not None and pe.dtype != torch.float32:
                dpe = to_compute(dpe)
            return dqkv, dpe, None, None
        qkv, probs, pdrop, pe = ctx_.saved_tensors
        dqkv = torch.empty_like(qkv)
        dpe = _attn_bwd(dctx.contiguous(), (qkv, 3 * d, 0), (qkv, 3 * d, d), (qkv, 3 * d, 2 * d), (dqkv, 3 * d, 0),
                        (dqkv, 3 * d, d), (dqkv, 3 * d, 2 * d), probs, pdrop, None, B, H, T, T, hd, pe,
                        pe is not None and ctx_.needs_input_grad[1], maxrel, p_drop, seed)
        if dpe is not None and pe.dtype != torch.float32:
            dpe = to_compute(dpe)
        return dqkv, dpe, None, None


class KVShare:
    """Keys / values of EVERY decoder layer's cross-attention projected by one GEMM (decoder.py: the encoder output is the same
    for all layers): kv_all [B*S, L*2d], layer l reads columns [l*2d, (l+1)*2d).  In the backward every layer writes its
    dK / dV into its column slice of ONE gradient buffer; only the layer that ran first in the forward (the last one to run
    backward) hands that buffer to autograd, the others return None -- so autograd adds nothing and the projection's data
    gradient is one GEMM over K = L*2d."""

    def __init__(self, n_layers, width, may_skip_layers):
        self.n_layers, self.width, self.may_skip = n_layers, width, may_skip_layers
        self.first = None      # id of the forward call that ran first
        self.grad = None

    def grad_buffer(self, like):
        if self.grad is None:
            self.grad = (torch.zeros_like if self.may_skip else torch.empty_like)(like)
        return self.grad


class CrossAttentionFunction(torch.autograd.Function):
    """q [B*T, d], kv [B*S, 2d] -> ctx [B*T, d] (+ probabilities [B*H, T, lds], differentiable: the
    guided-attention loss of TTS fine-tuning back-propagates through them).  cfg may carry (kv_ld, kv_off, share): kv is then
    the all-layer projection [B*S, kv_ld] and this layer's keys / values start at column kv_off (KVShare)."""

    @staticmethod
    def forward(ctx_, q, kv, kpm, cfg):
        B, H, T, S, hd, p_drop, want_probs = cfg[:7]
        d = H * hd
        kld, koff, share = cfg[7:10] if len(cfg) > 7 else (2 * d, 0, None)
        if share is not None and share.first is None:
            share.first = id(ctx_)
        seed = next_seed() if p_drop > 0 else 0
        if _can_flash(q.dtype, hd, want_probs):
            ctx, lse, _ = _flash_fwd((q, d, 0), (kv, kld, koff), (kv, kld, koff + d), B, H, T, S, hd, None, 0, kpm, False, p_drop, seed)
            ctx_.save_for_backward(q, kv, ctx, lse, kpm)
            ctx_.meta = (cfg, seed, True)
            return ctx, None
        ctx, probs, pdrop = _attn_fwd((q, d, 0), (kv, kld, koff), (kv, kld, koff + d), B, H, T, S, hd, None, 0, kpm, False, p_drop, seed)
        ctx_.save_for_backward(q, kv, probs, pdrop)
        ctx_.meta = (cfg, seed, False)
        if want_probs:
            pf = to_float(probs)[:, :, :S] if probs.shape[2] != S or probs.dtype != torch.float32 else probs
            return ctx, pf.reshape(B, H, T, S)
        return ctx, None

    @staticmethod
    def backward(ctx_, dctx, dprobs):
        cfg, seed, flash = ctx_.meta
        B, H, T, S, hd, p_drop, _ = cfg[:7]
        d = H * hd
        kld, koff, share = cfg[7:10] if len(cfg) > 7 else (2 * d, 0, None)
        kv = ctx_.saved_tensors[1]
        dq = torch.empty_like(ctx_.saved_tensors[0])
        dkv = share.grad_buffer(kv) if share is not None else torch.empty_like(kv)
        if flash:
            q, kv, ctx, lse, kpm = ctx_.saved_tensors
            _flash_bwd(dctx.contiguous(), ctx, lse, (q, d, 0), (kv, kld, koff), (kv, kld, koff + d), (dq, d, 0), (dkv, kld, koff),
                       (dkv, kld, koff + d), B, H, T, S, hd, None, False, 0, kpm, False, p_drop, seed)
        else:
            q, kv, probs, pdrop = ctx_.saved_tensors
            extra = None
            if dprobs is not None:
                extra = dprobs.reshape(B * H, T, S).float().contiguous()
            if dctx is None:
                dctx = torch.zeros(B * T, d, dtype=q.dtype, device=q.device)
            _attn_bwd(dctx.contiguous(), (q, d, 0), (kv, kld, koff), (kv, kld, koff + d), (dq, d, 0), (dkv, kld, koff), (dkv, kld, koff + d),
                      probs, pdrop, extra, B, H, T, S, hd, None, False, 0, p_drop, seed)
        if share is not None and share.first != id(ctx_):
            dkv = None    # (this layer's slice is in the shared buffer; the first-run layer returns it)
        return dq, dkv, None, None


# -------------------------------------------------------------------------------------------------
# element-wise helpers with autograd
# -------------------------------------------------------------------------------------------------
class ActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        y = torch.empty_like(x)
        hip.check(hip.lib().st5_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), act, _dt(x), hip.stream()), "st5_act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        hip.check(hip.lib().st5_act_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), ctx.act, _dt(x), hip.stream()),
                  "st5_act_bwd")
        return dx, None


def activation(x, act):
    return ActFunction.apply(x.contiguous(), act)


class DropoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        ctx.p, ctx.seed = p, next_seed()
        return _dropout(x, p, ctx.seed)

    @staticmethod
    def backward(ctx, dy):
        return _dropout(dy.contiguous(), ctx.p, ctx.seed), None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFunction.apply(x.contiguous(), float(p))


class LayerDropSelectFunction(torch.autograd.Function):
    """LayerDrop inside a step that is recorded for HIP-graph replay: the host's per-layer draw cannot steer control flow in a
    captured graph, so the layer always runs and its output is SELECTED on the device -- y = keep ? layer(x) : x with `keep` a
    one-float device tensor staged from the host draw (stage_host) -- and the backward routes the gradient accordingly
    (a dropped layer receives zeros: its parameters get exactly zero gradients, as when it is skipped).  Same loss and
    gradients as skipping (modules/encoder.py:251-257, decoder.py:64-67 of the reference); the dropped layer's work is not saved."""

    @staticmethod
    def forward(ctx, x_in, x_out, keep):
        assert x_in.shape == x_out.shape and x_in.dtype == x_out.dtype and x_in.is_contiguous() and x_out.is_contiguous()
        y = torch.empty_like(x_in)
        nbytes = y.numel() * y.element_size()
        assert nbytes % 16 == 0, "layerdrop_select: row size must be a multiple of 16 bytes"
        hip.check(hip.lib().st5_select(keep.data_ptr(), x_in.data_ptr(), x_out.data_ptr(), y.data_ptr(), nbytes, hip.stream()), "st5_select")
        ctx.keep = keep
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        hip.check(hip.lib().st5_select_bwd(ctx.keep.data_ptr(), g.data_ptr(), ga.data_ptr(), gb.data_ptr(), g.numel() * g.element_size(),
                                           hip.stream()), "st5_select_bwd")
        return ga, gb, None


def layerdrop_select(x_in, x_out, keep):
    return LayerDropSelectFunction.apply(x_in.contiguous(), x_out.contiguous(), keep)


class LayerDropGate:
    """The select of a post-LN layer folded into the kernels on either side of it (one launch instead of three per layer and
    direction): the layer's LAST LayerNorm writes  keep ? LN(z) : x_in  (st5_layernorm_gated_fwd) and, in the backward, counts its
    incoming gradient g_out as zero when the layer is dropped -- every gradient inside the layer is then exactly zero, as with
    layerdrop_select -- while LayerDropEnterFunction at the layer's input replaces the (zero) input gradient by g_out
    (st5_skip_grad: a kept layer, 19 of 20, costs one early-exit launch instead of a select and an autograd accumulation).
    Usage (modules/encoder.py, modules/decoder.py):  gate, x = layerdrop_gate(x, keep);  y = layer.forward_rows(x, ..., gate=gate)."""

    def __init__(self, keep, skip):
        self.keep, self.skip = keep, skip      # one device float; the layer's input rows (detached, contiguous)
        self.g_out = None                      # the gradient of the gated LayerNorm's output, set by its backward
        self.used = False


class LayerDropEnterFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        ctx.gate = gate
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, dx):
        gate = ctx.gate
        g, gate.g_out = gate.g_out, None
        assert g is not None, "LayerDropGate: the layer's input gradient arrived before its gated LayerNorm's backward ran"
        dx = dx.contiguous()
        assert g.numel() == dx.numel() and g.dtype == dx.dtype
        if _is_queued_wgrad_operand(dx):
            # (the kernel below overwrites dx IN PLACE for a dropped layer.  Today dx is a fresh data-gradient GEMM output;
            #  should it ever alias an operand a queued weight-gradient GEMM has yet to read, that GEMM must keep seeing the zeros)
            dx = dx.clone()
        hip.check(hip.lib().st5_skip_grad(gate.keep.data_ptr(), g.data_ptr(), dx.data_ptr(), dx.numel() * dx.element_size(), hip.stream()),
                  "st5_skip_grad")
        return dx, None


LAYERDROP_GATE = os.environ.get("ST5_LAYERDROP_GATE", "1") != "0"    # A/B switch: 0 = the stand-alone select kernels everywhere


def layerdrop_gate(x, keep):
    """(gate, x') for a layer whose last operation is a LayerNorm that accepts `gate=`; x' aliases x."""
    x = x.contiguous()
    gate = LayerDropGate(keep, _rows(x.detach()))
    y = LayerDropEnterFunction.apply(x, gate)
    b = getattr(x, "_st5_boundary", None)
    if b is not None:
        y._st5_boundary = b          # (layer_boundary stays idempotent: the layer's own call must not cut the graph again)
    return gate, y


def layerdrop_on_device(x):
    """True when LayerDrop decisions must be device-side selects: while a step is recorded / captured for replay (and in the
    eager fixed-shape form the replay is compared against)."""
    return x.is_cuda and static_shapes()


class AddScaledFunction(torch.autograd.Function):
    """y = a + s * b (same shapes, compute dtype); s is a python float."""

    @staticmethod
    def forward(ctx, a, b, s):
        y = a.clone()
        hip.check(hip.lib().st5_axpby(b.data_ptr(), y.data_ptr(), y.numel(), s, 1.0, _dt(y), hip.stream()), "st5_axpby")
        ctx.s = s
        return y

    @staticmethod
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            db = torch.zeros_like(dy, memory_format=torch.contiguous_format)
            dyc = dy.contiguous()
            hip.check(hip.lib().st5_axpby(dyc.data_ptr(), db.data_ptr(), db.numel(), ctx.s, 0.0, _dt(db), hip.stream()), "st5_axpby")
        return dy, db, None


def add(a, b, scale=1.0):
    return AddScaledFunction.apply(a.contiguous(), b.contiguous(), float(scale))


class MaskedFillRowsFunction(torch.autograd.Function):
    """x[mask] = v   (apply_hubert_mask, speech_encoder_prenet.py:249); v fp32 parameter [C]."""

    @staticmethod
    def forward(ctx, x, mask_u8, v):
        y = x.clone()
        rows = y.numel() // y.shape[-1]
        hip.check(hip.lib().st5_masked_fill_rows(y.data_ptr(), mask_u8.data_ptr(), v.data_ptr(), rows, y.shape[-1], _dt(y),
                                                 hip.stream()), "st5_masked_fill_rows")
        ctx.save_for_backward(mask_u8)
        ctx.v = v
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask_u8,) = ctx.saved_tensors
        dx = dy.clone(memory_format=torch.contiguous_format)
        rows = dx.numel() // dx.shape[-1]
        v = ctx.v
        gv = grad_buffer(v) if v.requires_grad else None
        hip.check(hip.lib().st5_masked_fill_rows_bwd(dx.data_ptr(), mask_u8.data_ptr(), hip.ptr(gv), rows, dx.shape[-1], _dt(dx),
                                                     hip.stream()), "st5_masked_fill_rows_bwd")
        if gv is not None:
            _grad_done(v)
        return dx, None, None


def masked_fill_rows(x, mask_bool, v):
    return MaskedFillRowsFunction.apply(x.contiguous(), mask_bool.to(torch.uint8).contiguous(), v)


class ChannelMaskFunction(torch.autograd.Function):
    """x[b, :, c] = 0 where the channel mask is set (apply_hubert_mask, speech_encoder_prenet.py:253-270): a per-clip
    per-channel 0/1 scale applied with st5_channel_affine; the gradient is scaled the same way."""

    @staticmethod
    def _apply(x, keep):
        B, T, C = x.shape
        y = torch.empty_like(x)
        zero = torch.zeros(C, dtype=torch.float32, device=x.device)
        L = hip.lib()
        for b in range(B):
            hip.check(L.st5_channel_affine(x[b].data_ptr(), keep[b].data_ptr(), zero.data_ptr(), y[b].data_ptr(), T, C, ACT_NONE,
                                           _dt(x), hip.stream()), "st5_channel_affine")
        return y

    @staticmethod
    def forward(ctx, x, keep):
        ctx.save_for_backward(keep)
        return ChannelMaskFunction._apply(x, keep)

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        return ChannelMaskFunction._apply(dy.contiguous(), keep), None


def mask_channels(x, channel_mask_bool):
    """x [B,T,C]; channel_mask_bool [B,C] (True = zero that channel for the whole clip)."""
    keep = (~channel_mask_bool).to(torch.float32).contiguous()
    return ChannelMaskFunction.apply(x.contiguous(), keep)


class AddTableRowsFunction(torch.autograd.Function):
    """y[r] = x[r] + scale * table[idx[r]] with a constant fp32 table (sinusoidal positions)."""

    @staticmethod
    def forward(ctx, x, table, idx, scale):
        y = torch.empty_like(x)
        rows = x.numel() // x.shape[-1]
        hip.check(hip.lib().st5_add_table_rows(x.data_ptr(), table.data_ptr(), idx.data_ptr(), y.data_ptr(), rows, x.shape[-1],
                                               scale, _dt(x), hip.stream()), "st5_add_table_rows")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None


def add_table_rows_dev(x, table, idx, scale_dev):
    """y[r] = x[r] + scale_dev[0] * table[idx[r]] with the scale read from device memory (no autograd)."""
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    hip.check(hip.lib().st5_add_table_rows_dev(x.data_ptr(), table.data_ptr(), idx.data_ptr(), y.data_ptr(), rows, x.shape[-1],
                                               scale_dev.data_ptr(), _dt(x), hip.stream()), "st5_add_table_rows_dev")
    return y


def add_table_rows(x, table, idx, scale=1.0):
    return AddTableRowsFunction.apply(x.contiguous(), table, idx.to(torch.int32).contiguous(), float(scale))


class EmbedRowsFunction(torch.autograd.Function):
    """y[r] = emb_scale * table[tok[r]] + pos_scale * pos[pidx[r]] (table: fp32 parameter, pos: constant)."""

    @staticmethod
    def forward(ctx, table, tok, pos, pidx, emb_scale, pos_scale, dtype):
        rows, cols = tok.numel(), table.shape[1]
        y = torch.empty(tuple(tok.shape) + (cols,), dtype=dtype, device=table.device)
        hip.check(hip.lib().st5_embed_rows(table.data_ptr(), tok.data_ptr(), hip.ptr(pos), hip.ptr(pidx), y.data_ptr(), rows, cols,
                                           emb_scale, pos_scale, _dt(dtype), hip.stream()), "st5_embed_rows")
        ctx.save_for_backward(tok)
        ctx.meta = (table, emb_scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (tok,) = ctx.saved_tensors
        table, emb_scale = ctx.meta
        if table.requires_grad:
            dy = dy.contiguous()
            g = grad_buffer(table)
            # (the deterministic form: the atomic one made two runs of the same step differ in the last bits of this gradient)
            hip.check(hip.lib().st5_embed_rows_bwd_det(dy.data_ptr(), tok.data_ptr(), g.data_ptr(), tok.numel(), table.shape[1],
                                                       table.shape[0], emb_scale, _dt(dy), hip.stream()), "st5_embed_rows_bwd_det")
            _grad_done(table)
        return None, None, None, None, None, None, None


def embed_rows(table, tok, pos=None, pidx=None, emb_scale=1.0, pos_scale=1.0):
    return EmbedRowsFunction.apply(table, tok.to(torch.int32).contiguous(), pos,
                                   pidx.to(torch.int32).contiguous() if pidx is not None else None,
                                   float(emb_scale), float(pos_scale), _S.dtype)


class GumbelVQFunction(torch.autograd.Function):
    """Gumbel-softmax code selection + code-book lookup + time-wise mix with the encoder states + both perplexities
    (csrc/vq.hip).  logits [N, G*V] fp32; vars_p the [1, G*V, Dg] fp32 code-book parameter; vars_c its compute-dtype copy (for
    the backward GEMM); enc [N, G*Dg] compute dtype or None; gumbel [N, G*V] fp32 or None (eval); mix_w [T] fp32 or None;
    tau float or 1-element device tensor.  Returns (out [N, G*Dg], code_perplexity, prob_perplexity, code rows [N, G] int32)."""

    @staticmethod
    def forward(ctx, logits, vars_p, vars_c, enc, gumbel, mix_w, tau, training, G, V, T, out_dtype):
        L = hip.lib()
        N = logits.shape[0]
        Dg = vars_p.shape[-1]
        dev = logits.device
        logits = logits.contiguous()
        assert logits.dtype == torch.float32 and logits.shape[1] == G * V and V <= L.st5_vq_vpad()
        if enc is not None:
            enc = enc.contiguous()
            out_dtype = enc.dtype
        out = torch.empty(N, G * Dg, dtype=out_dtype, device=dev)
        idx = torch.empty(N, G, dtype=torch.int32, device=dev)
        avg = torch.empty(G, L.st5_vq_vpad(), dtype=torch.float32, device=dev)
        perp = torch.empty(2, dtype=torch.float32, device=dev)
        ws = hip.workspace(L.st5_vq_ws_bytes(), dev)
        tau_dev = tau if torch.is_tensor(tau) else None
        hip.check(L.st5_vq_fwd(logits.data_ptr(), hip.ptr(gumbel), vars_p.data_ptr(), hip.ptr(enc), hip.ptr(mix_w),
                               0.0 if tau_dev is not None else float(tau), hip.ptr(tau_dev), 1 if training else 0, out.data_ptr(),
                               idx.data_ptr(), avg.data_ptr(), perp.data_ptr(), ws.data_ptr(), N, G, V, Dg, T, _dt(out), hip.stream()),
                  "st5_vq_fwd")
        ctx.save_for_backward(logits, gumbel, mix_w, avg, idx, vars_c, tau_dev)
        ctx.meta = (vars_p, None if tau_dev is not None else float(tau), bool(training), G, V, T, Dg, enc is not None)
        ctx.mark_non_differentiable(idx)
        return out, perp[0], perp[1], idx

    @staticmethod
    def backward(ctx, dout, g_code, g_prob, _g_idx):
        logits, gumbel, mix_w, avg, idx, vars_c, tau_dev = ctx.saved_tensors
        vars_p, tau, training, G, V, T, Dg, has_enc = ctx.meta
        L = hip.lib()
        N = logits.shape[0]
        dev = logits.device
        VP = L.st5_vq_vpad()
        dout = dout.contiguous()
        dt = _dt(dout)
        dsel = None
        if training and ctx.needs_input_grad[0]:
            # dsel[n, g, v] = <dOut[n, g*Dg:(g+1)*Dg], vars[g, v, :]>   (one batched NT GEMM over the groups, fp32 out)
            dsel = torch.empty(N, G * VP, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(dout, G * Dg, zs0=Dg), hip.operand(vars_c, Dg, zs0=V * Dg), hip.operand(dsel, G * VP, zs0=VP),
                     N, V, Dg, dt, batch=G, flags=hip.OUT_F32)
        dlogits = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        denc = torch.empty_like(dout) if (has_enc and ctx.needs_input_grad[3]) else None
        if dlogits is not None or denc is not None:
            if dlogits is None:
                dlogits = torch.empty_like(logits)
            hip.check(L.st5_vq_bwd(logits.data_ptr(), hip.ptr(gumbel), hip.ptr(dsel), G * VP, avg.data_ptr(),
                                   hip.ptr(g_prob.contiguous() if g_prob is not None else None), dout.data_ptr(), hip.ptr(mix_w),
                                   0.0 if tau_dev is not None else tau, hip.ptr(tau_dev), 1 if training else 0, dlogits.data_ptr(),
                                   hip.ptr(denc), N, G, V, Dg, T, dt, hip.stream()), "st5_vq_bwd")
        if vars_p.requires_grad:
            g = grad_buffer(vars_p)
            hip.check(L.st5_embed_rows_bwd_det_w(dout.data_ptr(), idx.data_ptr(), g.data_ptr(), N * G, Dg, G * V, 1.0, hip.ptr(mix_w), G, T,
                                                 dt, hip.stream()), "st5_embed_rows_bwd_det_w")
            _grad_done(vars_p)
        return dlogits, None, None, denc, None, None, None, None, None, None, None, None


class NCELogitsFunction(torch.autograd.Function):
    """logits [S, 1 + V] of the HuBERT NCE head from the projected frames x [S, D] (any float dtype), the code-book rows
    emb_param[row0 : row0 + V] (fp32 parameter) and the per-frame class ids (speech_encoder_postnet.py:56-76); fp32 like the
    reference.  Gradients: dx is returned, the code-book gradient is accumulated into the parameter's gradient buffer."""

    @staticmethod
    def forward(ctx, x, emb_param, row0, V, target, temp):
        L = hip.lib()
        S, D = x.shape
        dev = x.device
        x = x.contiguous()
        emb = emb_param.detach()[row0:row0 + V]
        assert emb.is_contiguous() and emb.dtype == torch.float32
        tgt = target.to(torch.int32).contiguous()
        st = hip.stream()
        en = torch.empty(V, D, dtype=torch.float32, device=dev)
        inv_e = torch.empty(V, dtype=torch.float32, device=dev)
        canon = torch.empty(V, dtype=torch.int32, device=dev)
        hip.check(L.st5_norm_rows(emb.data_ptr(), en.data_ptr(), inv_e.data_ptr(), V, D, hip.F32, st), "st5_norm_rows")
        hip.check(L.st5_canon_rows(emb.data_ptr(), canon.data_ptr(), V, D, st), "st5_canon_rows")
        xn = torch.empty(S, D, dtype=torch.float32, device=dev)
        inv_x = torch.empty(S, dtype=torch.float32, device=dev)
        logits = torch.empty(S, V + 1, dtype=torch.float32, device=dev)
        if S > 0:
            hip.check(L.st5_norm_rows(x.data_ptr(), xn.data_ptr(), inv_x.data_ptr(), S, D, _dt(x), st), "st5_norm_rows")
            sim = torch.empty(S, V, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(xn, D), hip.operand(en, D), hip.operand(sim, V), S, V, D, hip.F32)
            hip.check(L.st5_nce_logits(sim.data_ptr(), tgt.data_ptr(), canon.data_ptr(), logits.data_ptr(), S, V, float(temp), st), "st5_nce_logits")
        ctx.save_for_backward(xn, inv_x, en, inv_e, canon, tgt)
        ctx.meta = (emb_param, row0, V, float(temp), x.dtype)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        xn, inv_x, en, inv_e, canon, tgt = ctx.saved_tensors
        emb_param, row0, V, temp, xdtype = ctx.meta
        L = hip.lib()
        S, D = xn.shape
        dev = xn.device
        st = hip.stream()
        dx = torch.empty(S, D, dtype=xdtype, device=dev) if ctx.needs_input_grad[0] else None
        if S == 0:
            return dx, None, None, None, None, None
        dlogits = dlogits.contiguous().float()
        dsim = torch.empty(S, V, dtype=torch.float32, device=dev)
        hip.check(L.st5_nce_logits_bwd(dlogits.data_ptr(), tgt.data_ptr(), canon.data_ptr(), dsim.data_ptr(), S, V, temp, st), "st5_nce_logits_bwd")
        if dx is not None:
            dxn = torch.empty(S, D, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(dsim, V), hip.operand(en, D), hip.operand(dxn, D), S, D, V, hip.F32, flags=hip.B_KSTRIDED)
            hip.check(L.st5_norm_rows_bwd(xn.data_ptr(), inv_x.data_ptr(), dxn.data_ptr(), dx.data_ptr(), S, D, 0, hip.dt(xdtype), st),
                      "st5_norm_rows_bwd")
        if emb_param.requires_grad:
            den = torch.empty(V, D, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(dsim, V), hip.operand(xn, D), hip.operand(den, D), V, D, S, hip.F32,
                     flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)   # (OUT_F32: the split-K path, few output tiles / long K)
            g = grad_buffer(emb_param)[row0:row0 + V]
            hip.check(L.st5_norm_rows_bwd(en.data_ptr(), inv_e.data_ptr(), den.data_ptr(), g.data_ptr(), V, D, 1, hip.F32, st), "st5_norm_rows_bwd")
            _grad_done(emb_param)
        return dx, None, None, None, None, None


class SumSqFunction(torch.autograd.Function):
    """mean(x^2) as an fp32 scalar (features_pen, speech_encoder_prenet.py:172)."""

    @staticmethod
    def forward(ctx, x):
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().st5_sumsq(x.data_ptr(), out.data_ptr(), x.numel(), 1.0 / x.numel(), 0, _dt(x), hip.stream()), "st5_sumsq")
        ctx.save_for_backward(x)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        # d/dx mean(x^2) = 2x/n * g : g is a device scalar -> scale on device without a host sync
        s = (g.float() * (2.0 / x.numel())).to(x.dtype)
        return x * s  # (glue: one fused elementwise torch op on a tensor we already hold)


def mean_square(x):
    return SumSqFunction.apply(x.contiguous())


# -------------------------------------------------------------------------------------------------
# Speech feature extractor: conv0+GroupNorm+GELU kernel, then 6 strided Conv1d+GELU layers as implicit
# GEMMs on channels-last activations (speech_encoder_prenet.py:290-354, mode "default")
# -------------------------------------------------------------------------------------------------
def _conv_w_fwd(w, dtype):
    """[Cout, Cin, k] -> [Cout, k*Cin] compute dtype (cached)."""
    def build():
        Cout, Cin, k = w.shape   # out[co, j, ci] = w[co, ci, j]
        return _gather3(w.detach(), torch.empty(Cout, k * Cin, dtype=dtype, device=w.device), (Cout, k, Cin), (Cin * k, 1, k))
    return weight_cache.get(("convw", dtype, id(w)), [w], build)


def _conv_w_dgrad_k2(w, dtype):
    """k = 2, stride 2 transposed convolution as ONE NT GEMM: B[j = jj*Cin + ci][k = co] = W[co, ci, jj], K-major ([2*Cin, Cout])."""
    def build():
        Cout, Cin, k = w.shape   # out[j, ci, co] = w[co, ci, j]
        return _gather3(w.detach(), torch.empty(k * Cin, Cout, dtype=dtype, device=w.device), (k, Cin, Cout), (1, k, Cin * k))
    return weight_cache.get(("convw_d2", dtype, id(w)), [w], build)


def _conv_w_even_odd_t(w, dtype):
    """K-major (transposed) forms of _conv_w_even_odd for the NT kernels: even rows [Cin, 2*Cout] = [W2 | W0]^T, odd rows
    [Cin, Cout] = W1^T."""
    def build():
        Cout, Cin, k = w.shape
        wd = w.detach()
        # e[ci, h, co] = w[co, ci, 2 - 2h] (h = 0: tap 2, h = 1: tap 0);  o[ci, co] = w[co, ci, 1]
        e = _gather3(wd, torch.empty(Cin, 2 * Cout, dtype=dtype, device=w.device), (Cin, 2, Cout), (k, -2, Cin * k), off=2)
        o = _gather3(wd, torch.empty(Cin, Cout, dtype=dtype, device=w.device), (Cin, 1, Cout), (k, 0, Cin * k), off=1)
        return e, o
    return weight_cache.get(("convw_eo_t", dtype, id(w)), [w], build)


def _conv_w_even_odd(w, dtype):
    """k=3, stride 2 transposed-conv weights: even rows use [W2; W0] ([2*Cout, Cin]), odd rows W1 ([Cout, Cin])."""
    def build():
        Cout, Cin, k = w.shape
        wd = w.detach()
        # e[h, co, ci] = w[co, ci, 2 - 2h];  o[co, ci] = w[co, ci, 1]
        e = _gather3(wd, torch.empty(2 * Cout, Cin, dtype=dtype, device=w.device), (2, Cout, Cin), (-2, Cin * k, k), off=2)
        o = _gather3(wd, torch.empty(Cout, Cin, dtype=dtype, device=w.device), (1, Cout, Cin), (0, Cin * k, k), off=1)
        return e, o
    return weight_cache.get(("convw_eo", dtype, id(w)), [w], build)


class ConvFeatureExtractorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, layers, gscale, w0, gn_w, gn_b, *ws):
        """wav fp32 [B,S]; layers = [(dim,k,stride)]*; returns channels-last features [B,T,C] (compute dtype)."""
        dtype = _S.dtype
        dev = wav.device
        B, S = wav.shape
        L = hip.lib()
        C0, k0, s0 = layers[0]
        wav = wav.contiguous()
        L0 = (S - k0) // s0 + 1
        y = torch.empty(B, L0, C0, dtype=dtype, device=dev)
        stats = torch.empty(B, C0, 2, dtype=torch.float32, device=dev)
        # the waveform moments travel to the backward with the statistics (B x 65 doubles): it then needs no pass over the waveform
        mom = torch.empty(B, L.st5_conv0_mom_count(k0), dtype=torch.float64, device=dev)
        wsb = hip.workspace(L.st5_conv0_ws_bytes(B, S, C0, k0, s0), dev)
        w0f = w0.detach().reshape(C0, k0).contiguous()
        # the frontend AS A UNIT: algorithmic bytes = sum over layers of (input + output) once, GroupNorm / GELU
        # counted as fused (fwd 130.2 MB per 10 s clip in bf16); flops = sum 2 Cout Cin k Lout
        es_ = y.element_size()
        lens_, Lc_ = [], S
        for (C_, k_, s_) in layers:
            Lc_ = (Lc_ - k_) // s_ + 1
            lens_.append(Lc_)
        xb_ = [B * S * 4] + [B * lens_[i] * layers[i][0] * es_ for i in range(len(layers) - 1)]
        yb_ = [B * lens_[i] * layers[i][0] * es_ for i in range(len(layers))]
        fl_ = [2.0 * B * lens_[i] * layers[i][0] * (layers[i - 1][0] if i else 1) * layers[i][1] for i in range(len(layers))]
        ctx.front = (sum(yb_) + 2 * sum(xb_) - xb_[0], 2 * sum(fl_) - fl_[0])   # backward: Y + 2X (layer 0: Y + X), 2x flops (layer 0: 1x)
        front_region = hip.profiler.region("conv_frontend_fwd", sum(xb_) + sum(yb_), flops=sum(fl_))
        front_region.__enter__()
        # algorithmic HBM bytes: waveform in (fp32) + channels-last output once
        with hip.profiler.region("conv0_gn_gelu_fwd", B * S * 4 + B * L0 * C0 * y.element_size()):
            hip.check(L.st5_conv0_gn_gelu_fwd_m(wav.data_ptr(), w0f.data_ptr(), gn_w.data_ptr(), gn_b.data_ptr(), y.data_ptr(),
                                                stats.data_ptr(), mom.data_ptr(), wsb.data_ptr(), B, S, C0, k0, s0, 1e-5, _dt(dtype),
                                                hip.stream()), "st5_conv0_gn_gelu_fwd_m")
        acts, pres, lens = [y], [None], [L0]
        x, Lin, Cin = y, L0, C0
        for i, (C, k, s) in enumerate(layers[1:]):
            Lo = (Lin - k) // s + 1
            Wk = _conv_w_fwd(ws[i], dtype)
            out = torch.empty(B, Lo, C, dtype=dtype, device=dev)
            pre = torch.empty(B, Lo, C, dtype=dtype, device=dev)
            hip.gemm(hip.operand(x, s * Cin, rpb=Lo, bstride=Lin * Cin), hip.operand(Wk, k * Cin), hip.operand(out, C),
                     B * Lo, C, k * Cin, _dt(dtype), Cpre=hip.operand(pre, C), act=ACT_GELU)
            acts.append(out); pres.append(pre); lens.append(Lo)
            x, Lin, Cin = out, Lo, C
        front_region.__exit__(None, None, None)
        ctx.save_for_backward(wav, stats, mom, *acts[:-1], *pres[1:])
        ctx.meta = (layers, gscale, w0, gn_w, gn_b, ws, lens, B, S)
        return x

    @staticmethod
    def backward(ctx, dy):
        layers, gscale, w0, gn_w, gn_b, ws, lens, B, S = ctx.meta
        saved = ctx.saved_tensors
        wav, stats, mom = saved[0], saved[1], saved[2]
        n = len(layers)
        acts = saved[3:3 + n - 1]          # outputs of layers 0..n-2 (inputs of layers 1..n-1)
        pres = (None,) + tuple(saved[3 + n - 1:])  # pre-activations of layers 1..n-1
        dtype = dy.dtype
        dev = dy.device
        L = hip.lib()
        C = layers[-1][0]
        front_region = hip.profiler.region("conv_frontend_bwd", ctx.front[0], flops=ctx.front[1])
        front_region.__enter__()
        # top gradient: scale by feature_grad_mult (GradMultiply, :158-160) and apply GELU' of the last layer,
        # written into the interior of a time-padded buffer (one zero row each side, used by the k=3 layers)
        Ln = lens[-1]
        g = dy.contiguous()
        if gscale != 1.0:
            g2 = torch.zeros_like(g)
            hip.check(L.st5_axpby(g.data_ptr(), g2.data_ptr(), g.numel(), gscale, 0.0, _dt(g), hip.stream()), "st5_axpby")
            g = g2
        dpre = torch.empty(B, Ln + 2, C, dtype=dtype, device=dev)
        dpre[:, 0].zero_(); dpre[:, -1].zero_()
        tmp = torch.empty(B, Ln, C, dtype=dtype, device=dev)
        hip.check(L.st5_act_bwd(g.data_ptr(), pres[-1].data_ptr(), tmp.data_ptr(), g.numel(), ACT_GELU, _dt(g), hip.stream()),
                  "st5_act_bwd")
        dpre[:, 1:-1] = tmp  # (glue copy of the smallest activation in the stack)
        for li in range(n - 1, 0, -1):
            Cout, k, s = layers[li]
            Cin = layers[li - 1][0]
            Lo, Lin = lens[li], lens[li - 1]
            w = ws[li - 1]
            xin = acts[li - 1]  # [B, Lin, Cin]
            dint = dpre[:, 1:-1]  # interior view, row stride Cout, batch stride (Lo+2)*Cout
            ioff = Cout  # element offset of the interior
            if w.requires_grad:
                _conv_wgrad(w, hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout),
                            hip.operand(xin, s * Cin, rpb=Lo, bstride=Lin * Cin), Cout, k, Cin, B * Lo, _dt(dtype), (dpre, xin))
                _grad_done(w)
            # data gradient into the (padded) pre-activation gradient of the previous layer
            last = li == 1
            if last:
                nxt = torch.empty(B, Lin, Cin, dtype=dtype, device=dev)
                noff, nbs = 0, Lin * Cin
                P = None
            else:
                nxt = torch.empty(B, Lin + 2, Cin, dtype=dtype, device=dev)   # (halo rows zeroed below, with the uncovered tail)
                noff, nbs = Cin, (Lin + 2) * Cin
                P = pres[li - 1]
            flags_d = 0 if last else hip.DACT
            act_d = ACT_NONE if last else ACT_GELU
            Wk = _conv_w_fwd(w, dtype)
            # the transposed-convolution weights are cached K-major, so these data gradients run on the LDS-DMA NT kernels
            # (256^2 tile for the long layers) instead of the register-staged general kernel (125-160 vs 600+ TFLOP/s)
            if k == 2 and s == 2:
                covered = 2 * Lo
                Wd = _conv_w_dgrad_k2(w, dtype)   # [2*Cin, Cout]
                hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wd, Cout),
                         hip.operand(nxt, 2 * Cin, off=noff, rpb=Lo, bstride=nbs), B * Lo, 2 * Cin, Cout, _dt(dtype),
                         P=hip.operand(P, 2 * Cin, rpb=Lo, bstride=Lin * Cin) if P is not None else None,
                         act=act_d, flags=flags_d)
            elif k == 3 and s == 2:
                covered = 2 * Lo + 1
                Wet, Wot = _conv_w_even_odd_t(w, dtype)
                # even rows 2t' (t' = 0..Lo): [dpre[t'-1], dpre[t']] . [W2; W0]
                hip.gemm(hip.operand(dpre, Cout, rpb=Lo + 1, bstride=(Lo + 2) * Cout), hip.operand(Wet, 2 * Cout),
                         hip.operand(nxt, 2 * Cin, off=noff, rpb=Lo + 1, bstride=nbs), B * (Lo + 1), Cin, 2 * Cout, _dt(dtype),
                         P=hip.operand(P, 2 * Cin, rpb=Lo + 1, bstride=Lin * Cin) if P is not None else None,
                         act=act_d, flags=flags_d)
                # odd rows 2t'+1: dpre[t'] . W1
                hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wot, Cout),
                         hip.operand(nxt, 2 * Cin, off=noff + Cin, rpb=Lo, bstride=nbs), B * Lo, Cin, Cout, _dt(dtype),
                         P=hip.operand(P, 2 * Cin, off=Cin, rpb=Lo, bstride=Lin * Cin) if P is not None else None,
                         act=act_d, flags=flags_d)
            else:
                raise NotImplementedError(f"conv feature layer (k={k}, stride={s}) backward")
            # halo rows + the trailing input rows no output window touches, one launch
            if last:
                if covered < Lin:
                    hip.check(L.st5_zero_time_edges(nxt.data_ptr(), B, Lin, Cin, 0, covered, _dt(dtype), hip.stream()), "st5_zero_time_edges")
            else:
                hip.check(L.st5_zero_time_edges(nxt.data_ptr(), B, Lin + 2, Cin, 1, 1 + covered, _dt(dtype), hip.stream()), "st5_zero_time_edges")
            dpre = nxt
        # layer 0: conv(1->C) + GroupNorm + GELU backward (weights only; the waveform needs no gradient)
        C0, k0, s0 = layers[0]
        w0f = w0.detach().reshape(C0, k0).contiguous()
        need = w0.requires_grad or gn_w.requires_grad or gn_b.requires_grad
        if need:
            wsb = hip.workspace(L.st5_conv0_ws_bytes(B, S, C0, k0, s0), dev)
            gw0 = grad_buffer(w0) if w0.requires_grad else None
            # algorithmic HBM bytes: waveform + ONE read of dY (the kernels read dY twice: GroupNorm sums, then dW)
            with hip.profiler.region("conv0_gn_gelu_bwd", B * S * 4 + dpre.numel() * dpre.element_size()):
                hip.check(L.st5_conv0_gn_gelu_bwd_m(wav.data_ptr(), w0f.data_ptr(), gn_w.data_ptr(), gn_b.data_ptr(), stats.data_ptr(),
                                                    mom.data_ptr(), dpre.data_ptr(), hip.ptr(gw0),
                                                    hip.ptr(grad_buffer(gn_w)) if gn_w.requires_grad else 0,
                                                    hip.ptr(grad_buffer(gn_b)) if gn_b.requires_grad else 0, wsb.data_ptr(), B, S, C0, k0, s0,
                                                    1.0, _dt(dtype), hip.stream()), "st5_conv0_gn_gelu_bwd_m")
            for p in (w0, gn_w, gn_b):
                if p.requires_grad:
                    _grad_done(p)
        front_region.__exit__(None, None, None)   # (weight-gradient GEMMs on the side stream, if any, are not inside the events)
        return (None, None, None, None, None, None) + (None,) * len(ws)


def conv_feature_extractor(wav, layers, gscale, w0, gn_w, gn_b, ws):
    return ConvFeatureExtractorFunction.apply(wav, tuple(layers), float(gscale), w0, gn_w, gn_b, *ws)


# -------------------------------------------------------------------------------------------------
# extractor_mode=layer_norm (SpeechT5-Large recipe, speech_encoder_prenet.py:290-354): every block is
# Conv1d(bias) -> LayerNorm over channels -> GELU.  Channels-last makes the LayerNorm a plain row LayerNorm; the
# convolutions are differentiable autograd functions on the GEMM kernels, composed with layer_norm()/activation().
# -------------------------------------------------------------------------------------------------
class Conv0UnfoldFunction(torch.autograd.Function):
    """Conv1d(1 -> C, k, stride, bias) on the waveform: windows unfolded to [B*L, 16] rows (st5_unfold_rows) x W^T."""

    @staticmethod
    def forward(ctx, wav, w, bias, k, stride):
        dtype = _S.dtype
        B, S = wav.shape
        C = w.shape[0]
        L0 = (S - k) // stride + 1
        kpad = (k + 15) // 16 * 16
        wav = wav.contiguous().float()
        rows = torch.empty(B * L0, kpad, dtype=dtype, device=wav.device)
        hip.check(hip.lib().st5_unfold_rows(wav.data_ptr(), rows.data_ptr(), B, S, k, stride, kpad, _dt(dtype), hip.stream()),
                  "st5_unfold_rows")
        wk = torch.zeros(C, kpad, dtype=torch.float32, device=wav.device)
        wk[:, :k] = w.detach().reshape(C, k)
        wk = wk.to(dtype)
        y = torch.empty(B * L0, C, dtype=dtype, device=wav.device)
        hip.gemm(hip.operand(rows, kpad), hip.operand(wk, kpad), hip.operand(y, C), B * L0, C, kpad, _dt(dtype),
                 bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(rows)
        ctx.meta = (w, bias, k, kpad, B, L0, C)
        return y.view(B, L0, C)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        w, bias, k, kpad, B, L0, C = ctx.meta
        g = dy.contiguous().view(B * L0, C)
        if w.requires_grad:
            tmp = torch.zeros(C, kpad, dtype=torch.float32, device=g.device)
            want_db = bias is not None and bias.requires_grad
            hip.gemm(hip.operand(g, C), hip.operand(rows, kpad), hip.operand(tmp, kpad), C, kpad, B * L0, _dt(g.dtype),
                     flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32, asum=grad_buffer(bias) if want_db else None)
            grad_buffer(w).add_(tmp[:, :k].reshape(w.shape))
            _grad_done(w)
            if want_db:
                _grad_done(bias)
        elif bias is not None and bias.requires_grad:
            _colsum_into(g, C, C, grad_buffer(bias))
            _grad_done(bias)
        return None, None, None, None, None


class Conv1dStridedFunction(torch.autograd.Function):
    """Channels-last Conv1d(Cin -> Cout, k in {2,3}, stride 2, bias) as an implicit GEMM (overlapping rows); the data
    gradient is the transposed convolution split into its even / odd output phases (as in ConvFeatureExtractorFunction)."""

    @staticmethod
    def forward(ctx, x, w, bias, k, s):
        dtype = x.dtype
        B, Lin, Cin = x.shape
        Cout = w.shape[0]
        Lo = (Lin - k) // s + 1
        x = x.contiguous()
        Wk = _conv_w_fwd(w, dtype)
        y = torch.empty(B, Lo, Cout, dtype=dtype, device=x.device)
        hip.gemm(hip.operand(x, s * Cin, rpb=Lo, bstride=Lin * Cin), hip.operand(Wk, k * Cin), hip.operand(y, Cout),
                 B * Lo, Cout, k * Cin, _dt(dtype), bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x)
        ctx.meta = (w, bias, k, s, B, Lin, Cin, Cout, Lo)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, bias, k, s, B, Lin, Cin, Cout, Lo = ctx.meta
        dtype, dev = dy.dtype, dy.device
        # one zero row each side (k = 3 even phase) in ONE pass (was a zero fill of the whole buffer + a strided torch copy)
        dyc = dy.contiguous()
        onepass = os.environ.get("ST5_LNX_ONEPASS", "1") != "0"      # (A/B)
        if onepass:
            dpre = torch.empty(B, Lo + 2, Cout, dtype=dtype, device=dev)
            hip.check(hip.lib().st5_pad_time(dyc.data_ptr(), dpre.data_ptr(), B, Lo, Cout, 1, 1, _dt(dtype), hip.stream()), "st5_pad_time")
        else:
            dpre = torch.zeros(B, Lo + 2, Cout, dtype=dtype, device=dev)
            dpre[:, 1:-1] = dy
        ioff = Cout
        want_db = bias is not None and bias.requires_grad
        if w.requires_grad:
            _conv_wgrad(w, hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout),
                        hip.operand(x, s * Cin, rpb=Lo, bstride=Lin * Cin), Cout, k, Cin, B * Lo, _dt(dtype), (dpre, x))
            _grad_done(w)
        if want_db:
            _colsum_into(dyc.view(B * Lo, Cout), Cout, Cout, grad_buffer(bias))
            _grad_done(bias)
        dx = None
        if ctx.needs_input_grad[0]:
            # the GEMMs below write rows 0 .. covered - 1 of every utterance; only the uncovered tail rows are zero-filled
            dx = torch.empty(B, Lin, Cin, dtype=dtype, device=dev) if onepass else torch.zeros(B, Lin, Cin, dtype=dtype, device=dev)
            covered = 2 * Lo if k == 2 else 2 * Lo + 1
            if onepass and covered < Lin:
                hip.check(hip.lib().st5_zero_time_edges(dx.data_ptr(), B, Lin, Cin, 0, covered, _dt(dtype), hip.stream()), "st5_zero_time_edges")
            # (round 6: the K-major cached weight forms of ConvFeatureExtractorFunction -- these data gradients, the layer-norm extractor's
            #  of t5_transformer_large, ran on the register-staged general kernel with a k-strided B operand: 16 launches of ~400 us per
            #  Large B = 32 update)
            if os.environ.get("ST5_LNX_DGRAD_KMAJOR", "1") == "0":       # (A/B: the k-strided forms on the general kernel)
                if k == 2 and s == 2:
                    hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(_conv_w_fwd(w, dtype), k * Cin),
                             hip.operand(dx, 2 * Cin, rpb=Lo, bstride=Lin * Cin), B * Lo, 2 * Cin, Cout, _dt(dtype), flags=hip.B_KSTRIDED)
                else:
                    We, Wo = _conv_w_even_odd(w, dtype)
                    hip.gemm(hip.operand(dpre, Cout, rpb=Lo + 1, bstride=(Lo + 2) * Cout), hip.operand(We, Cin),
                             hip.operand(dx, 2 * Cin, rpb=Lo + 1, bstride=Lin * Cin), B * (Lo + 1), Cin, 2 * Cout, _dt(dtype), flags=hip.B_KSTRIDED)
                    hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wo, Cin),
                             hip.operand(dx, 2 * Cin, off=Cin, rpb=Lo, bstride=Lin * Cin), B * Lo, Cin, Cout, _dt(dtype), flags=hip.B_KSTRIDED)
            elif k == 2 and s == 2:
                Wd = _conv_w_dgrad_k2(w, dtype)   # [2*Cin, Cout]
                hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wd, Cout),
                         hip.operand(dx, 2 * Cin, rpb=Lo, bstride=Lin * Cin), B * Lo, 2 * Cin, Cout, _dt(dtype))
            elif k == 3 and s == 2:
                Wet, Wot = _conv_w_even_odd_t(w, dtype)
                hip.gemm(hip.operand(dpre, Cout, rpb=Lo + 1, bstride=(Lo + 2) * Cout), hip.operand(Wet, 2 * Cout),
                         hip.operand(dx, 2 * Cin, rpb=Lo + 1, bstride=Lin * Cin), B * (Lo + 1), Cin, 2 * Cout, _dt(dtype))
                hip.gemm(hip.operand(dpre, Cout, off=ioff, rpb=Lo, bstride=(Lo + 2) * Cout), hip.operand(Wot, Cout),
                         hip.operand(dx, 2 * Cin, off=Cin, rpb=Lo, bstride=Lin * Cin), B * Lo, Cin, Cout, _dt(dtype))
            else:
                raise NotImplementedError(f"conv feature layer (k={k}, stride={s}) backward")
        return dx, None, None, None, None


def conv_feature_extractor_layer_norm(wav, layers, gscale, params):
    """params[i] = (conv_weight, conv_bias or None, ln_weight, ln_bias); returns channels-last features [B,T,C]."""
    x = None
    for i, (dim, k, s) in enumerate(layers):
        w, b, lw, lb = params[i]
        if i == 0:
            x = Conv0UnfoldFunction.apply(wav, w, b, k, s)
        else:
            x = Conv1dStridedFunction.apply(x, w, b, k, s)
        x = layer_norm_gelu(x, lw, lb)
    if gscale != 1.0:
        x = GradScaleFunction.apply(x, float(gscale))
    return x


class GradScaleFunction(torch.autograd.Function):
    """fairseq GradMultiply (speech_encoder_prenet.py:158-160): identity forward, gradient scaled."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        g = dy.contiguous()
        out = torch.empty_like(g)
        hip.check(hip.lib().st5_axpby(g.data_ptr(), out.data_ptr(), g.numel(), ctx.scale, 0.0, _dt(g), hip.stream()), "st5_axpby")
        return out, None


# -------------------------------------------------------------------------------------------------
# Positional convolution: x + GELU(grouped Conv1d(k, groups, pad k//2) [SamePad]) (speech_encoder_prenet.py:187-192)
# -------------------------------------------------------------------------------------------------
class PosConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, groups):
        """x [B,T,d] compute dtype; w fp32 [d, d/groups, k] (already weight-normalised, differentiable)."""
        dtype = x.dtype
        dev = x.device
        B, T, d = x.shape
        k = w.shape[2]
        cg = d // groups
        assert k % 2 == 0 and cg % 8 == 0
        L = hip.lib()
        pl, pr = k // 2, k // 2
        xp = torch.empty(B, T + pl + pr, d, dtype=dtype, device=dev)
        hip.check(L.st5_pad_time(x.data_ptr(), xp.data_ptr(), B, T, d, pl, pr, _dt(dtype), hip.stream()), "st5_pad_time")
        # [G][cg_out][k][cg_in]
        wf = w.detach().view(groups, cg, cg, k).permute(0, 1, 3, 2).contiguous().view(groups * cg, k * cg)
        Wg = torch.empty(wf.shape, dtype=dtype, device=dev)
        _cast_into(wf, Wg)
        y = torch.empty(B, T, d, dtype=dtype, device=dev)
        pre = torch.empty(B, T, d, dtype=dtype, device=dev)
        hip.gemm(hip.operand(xp, d, rpb=T, bstride=(T + pl + pr) * d, seg=cg, seg_stride=d, zs0=cg),
                 hip.operand(Wg, k * cg, zs0=cg * k * cg), hip.operand(y, d, zs0=cg), B * T, cg, k * cg, _dt(dtype), batch=groups,
                 R=hip.operand(x, d, zs0=cg), Cpre=hip.operand(pre, d, zs0=cg), bias=bias.detach(), bias_zs=cg, act=ACT_GELU)
        ctx.save_for_backward(xp, pre, w)
        ctx.meta = (bias, groups, B, T, d, k, cg)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, pre, w = ctx.saved_tensors
        bias, groups, B, T, d, k, cg = ctx.meta
        dtype = dy.dtype
        dev = dy.device
        L = hip.lib()
        dy = dy.contiguous()
        dpre = torch.empty_like(dy)
        hip.check(L.st5_act_bwd(dy.data_ptr(), pre.data_ptr(), dpre.data_ptr(), dy.numel(), ACT_GELU, _dt(dtype), hip.stream()),
                  "st5_act_bwd")
        if bias.requires_grad:
            _colsum_into(dpre.view(B * T, d), d, d, grad_buffer(bias))
            _grad_done(bias)
        dw = None
        if ctx.needs_input_grad[1]:
            dwg = torch.empty(groups, cg, k * cg, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(dpre, d, zs0=cg),
                     hip.operand(xp, d, rpb=T, bstride=xp.shape[1] * d, seg=cg, seg_stride=d, zs0=cg),
                     hip.operand(dwg, k * cg, zs0=cg * k * cg), cg, k * cg, B * T, _dt(dtype), batch=groups,
                     flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
            dw = dwg.view(groups, cg, k, cg).permute(0, 1, 3, 2).reshape(d, cg, k)
        dx = None
        if ctx.needs_input_grad[0]:
            # dx[t'] = dy[t'] + sum_{jj,o} dpre_pad[t'+jj, o] * W[o, c, k-1-jj]   (window of k rows starting at t')
            pl = k // 2 - 1
            pr = k // 2
            dpp = torch.empty(B, T + pl + pr, d, dtype=dtype, device=dev)
            hip.check(L.st5_pad_time(dpre.data_ptr(), dpp.data_ptr(), B, T, d, pl, pr, _dt(dtype), hip.stream()), "st5_pad_time")
            wf = w.detach().view(groups, cg, cg, k).flip(-1).permute(0, 2, 3, 1).contiguous().view(groups * cg, k * cg)
            Wf = torch.empty(wf.shape, dtype=dtype, device=dev)
            _cast_into(wf, Wf)
            dx = torch.empty(B, T, d, dtype=dtype, device=dev)
            hip.gemm(hip.operand(dpp, d, rpb=T, bstride=(T + pl + pr) * d, seg=cg, seg_stride=d, zs0=cg),
                     hip.operand(Wf, k * cg, zs0=cg * k * cg), hip.operand(dx, d, zs0=cg), B * T, cg, k * cg, _dt(dtype),
                     batch=groups, R=hip.operand(dy, d, zs0=cg))
        return dx, dw, None, None


def pos_conv(x, w, bias, groups):
    return PosConvFunction.apply(x.contiguous(), w, bias, int(groups))


# -------------------------------------------------------------------------------------------------
# Conv1d(k odd, stride 1, "same" zero padding, no bias) on channels-last activations as implicit GEMM
# (espnet Postnet convs, speech_decoder_postnet.py:39-51)
# -------------------------------------------------------------------------------------------------
class Conv1dSameFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        """x [B,L,Cin] compute dtype; w fp32 parameter [Cout,Cin,k] -> [B,L,Cout]."""
        dtype = x.dtype
        B, Lx, Cin = x.shape
        Cout, _, k = w.shape
        p = (k - 1) // 2
        xp = torch.empty(B, Lx + 2 * p, Cin, dtype=dtype, device=x.device)
        hip.check(hip.lib().st5_pad_time(x.data_ptr(), xp.data_ptr(), B, Lx, Cin, p, p, _dt(dtype), hip.stream()), "st5_pad_time")
        Wk = _conv_w_fwd(w, dtype)
        y = torch.empty(B, Lx, Cout, dtype=dtype, device=x.device)
        hip.gemm(hip.operand(xp, Cin, rpb=Lx, bstride=(Lx + 2 * p) * Cin), hip.operand(Wk, k * Cin), hip.operand(y, Cout),
                 B * Lx, Cout, k * Cin, _dt(dtype))
        ctx.save_for_backward(xp)
        ctx.meta = (w, B, Lx, Cin, Cout, k, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xp,) = ctx.saved_tensors
        w, B, Lx, Cin, Cout, k, p = ctx.meta
        dtype = dy.dtype
        dev = dy.device
        dy = dy.contiguous()
        if w.requires_grad:
            tmpw = torch.empty(Cout, k * Cin, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(dy, Cout), hip.operand(xp, Cin, rpb=Lx, bstride=(Lx + 2 * p) * Cin), hip.operand(tmpw, k * Cin),
                     Cout, k * Cin, B * Lx, _dt(dtype), flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
            _gather3(tmpw, grad_buffer(w), (Cout, Cin, k), (k * Cin, 1, Cin), accumulate=True)   # grad[co, ci, j] += tmpw[co, j, ci]
            _grad_done(w)
        dx = None
        if ctx.needs_input_grad[0]:
            dyp = torch.empty(B, Lx + 2 * p, Cout, dtype=dtype, device=dev)
            hip.check(hip.lib().st5_pad_time(dy.data_ptr(), dyp.data_ptr(), B, Lx, Cout, p, p, _dt(dtype), hip.stream()), "st5_pad_time")

            def build():   # out[ci, jj, co] = w[co, ci, k - 1 - jj]
                return _gather3(w.detach(), torch.empty(Cin, k * Cout, dtype=dtype, device=dev), (Cin, k, Cout), (k, -1, Cin * k), off=k - 1)
            Wd = weight_cache.get(("convw_d", dtype, id(w)), [w], build)
            dx = torch.empty(B, Lx, Cin, dtype=dtype, device=dev)
            hip.gemm(hip.operand(dyp, Cout, rpb=Lx, bstride=(Lx + 2 * p) * Cout), hip.operand(Wd, k * Cout), hip.operand(dx, Cin),
                     B * Lx, Cin, k * Cout, _dt(dtype))
        return dx, None


def conv1d_same(x, w):
    return Conv1dSameFunction.apply(x.contiguous(), w)


# -------------------------------------------------------------------------------------------------
# Mel post-net (espnet Tacotron Postnet as used at speech_decoder_postnet.py:39-51,65-70):
#   after = before + dropout(BN(conv_5( ... dropout(tanh(BN(conv_1(before)))) ... )))
# Convolutions = implicit GEMMs handing over their fp32 accumulators; BatchNorm statistics / affine / tanh / dropout /
# residual = st5_batchnorm_act_* (csrc/batchnorm.hip).  Everything between two convolutions -- activations AND gradients --
# stays fp32; only the MFMA operands are rounded to the compute dtype.
# -------------------------------------------------------------------------------------------------
class PostnetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, before, cfg, bufs, *params):
        """before [B, L, odim] (compute dtype); cfg = (training, dropout_p, [(eps, momentum)] per layer); bufs = per layer
        (running_mean, running_var, num_batches_tracked); params = per layer (conv_weight [Cout,Cin,k], bn_weight, bn_bias).
        Returns after fp32 [B, L, odim]."""
        training, p_drop, bn_cfg = cfg
        dtype = before.dtype
        dev = before.device
        B, Lx, odim = before.shape
        n = len(params) // 3
        L = hip.lib()
        before = before.contiguous()
        k0 = params[0].shape[2]
        pad = (k0 - 1) // 2
        xp = torch.empty(B, Lx + 2 * pad, odim, dtype=dtype, device=dev)
        hip.check(L.st5_pad_time(before.data_ptr(), xp.data_ptr(), B, Lx, odim, pad, pad, _dt(dtype), hip.stream()), "st5_pad_time")
        saved, seeds = [], []
        after = None
        for i in range(n):
            w, gamma, beta = params[3 * i: 3 * i + 3]
            rm, rv, nbt = bufs[i]
            eps, mom = bn_cfg[i]
            Cout, Cin, k = w.shape
            assert (k - 1) // 2 == pad and Cout % 4 == 0
            Wk = _conv_w_fwd(w, dtype)
            x32 = torch.empty(B * Lx, Cout, dtype=torch.float32, device=dev)
            hip.gemm(hip.operand(xp, Cin, rpb=Lx, bstride=(Lx + 2 * pad) * Cin), hip.operand(Wk, k * Cin), hip.operand(x32, Cout),
                     B * Lx, Cout, k * Cin, _dt(dtype), flags=hip.OUT_F32)
            stats = torch.empty(2 * Cout, dtype=torch.float32, device=dev)
            ws = hip.workspace(L.st5_batchnorm_ws_bytes(Cout), dev)
            last = i == n - 1
            p = p_drop if training else 0.0
            seed = next_seed() if p > 0 else 0
            mom_eff = mom if mom is not None else 0.1
            if last:
                after = torch.empty(B, Lx, Cout, dtype=torch.float32, device=dev)
                hip.check(L.st5_batchnorm_act_fwd(x32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), hip.ptr(rm), hip.ptr(rv), hip.ptr(nbt),
                                                  mom_eff, eps, 1 if training else 0, ACT_NONE, p, seed, before.data_ptr(), after.data_ptr(),
                                                  1, stats.data_ptr(), ws.data_ptr(), B * Lx, Cout, 0, 0, 0, 0, _dt(dtype), hip.stream()),
                          "st5_batchnorm_act_fwd")
                nxt = None
            else:
                nxt = torch.empty(B, Lx + 2 * pad, Cout, dtype=dtype, device=dev)
                hip.check(L.st5_batchnorm_act_fwd(x32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), hip.ptr(rm), hip.ptr(rv), hip.ptr(nbt),
                                                  mom_eff, eps, 1 if training else 0, ACT_TANH, p, seed, 0, nxt.data_ptr(), 0,
                                                  stats.data_ptr(), ws.data_ptr(), B * Lx, Cout, Lx, (Lx + 2 * pad) * Cout, pad * Cout, pad,
                                                  _dt(dtype), hip.stream()), "st5_batchnorm_act_fwd")
            saved += [xp, x32, stats]
            seeds.append((p, seed))
            xp = nxt
        ctx.save_for_backward(*saved)
        ctx.meta = (params, training, seeds, B, Lx, pad, dtype)
        return after

    @staticmethod
    def backward(ctx, d_after):
        params, training, seeds, B, Lx, pad, dtype = ctx.meta
        saved = ctx.saved_tensors
        n = len(params) // 3
        dev = d_after.device
        L = hip.lib()
        d_after = d_after.contiguous().float()
        dy = d_after.view(B * Lx, -1)
        d_before = None
        for i in range(n - 1, -1, -1):
            w, gamma, beta = params[3 * i: 3 * i + 3]
            xp, x32, stats = saved[3 * i: 3 * i + 3]
            Cout, Cin, k = w.shape
            p, seed = seeds[i]
            dxp = torch.empty(B, Lx + 2 * pad, Cout, dtype=dtype, device=dev)
            ws = hip.workspace(L.st5_batchnorm_ws_bytes(Cout), dev)
            gg = grad_buffer(gamma) if gamma.requires_grad else None
            gb = grad_buffer(beta) if beta.requires_grad else None
            hip.check(L.st5_batchnorm_act_bwd(x32.data_ptr(), dy.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), hip.ptr(gg),
                                              hip.ptr(gb), 1 if training else 0, ACT_NONE if i == n - 1 else ACT_TANH, p, seed, dxp.data_ptr(),
                                              ws.data_ptr(), B * Lx, Cout, Lx, (Lx + 2 * pad) * Cout, pad * Cout, pad, _dt(dtype), hip.stream()),
                      "st5_batchnorm_act_bwd")
            for q in (gamma, beta):
                if q.requires_grad:
                    _grad_done(q)
            if w.requires_grad:
                tmpw = torch.empty(Cout, k * Cin, dtype=torch.float32, device=dev)
                hip.gemm(hip.operand(dxp, Cout, off=pad * Cout, rpb=Lx, bstride=(Lx + 2 * pad) * Cout),
                         hip.operand(xp, Cin, rpb=Lx, bstride=(Lx + 2 * pad) * Cin), hip.operand(tmpw, k * Cin),
                         Cout, k * Cin, B * Lx, _dt(dtype), flags=hip.A_KSTRIDED | hip.B_KSTRIDED | hip.OUT_F32)
                _gather3(tmpw, grad_buffer(w), (Cout, Cin, k), (k * Cin, 1, Cin), accumulate=True)   # grad[co, ci, j] += tmpw[co, j, ci]
                _grad_done(w)
            if i > 0 or ctx.needs_input_grad[0]:
                def build(w=w, Cin=Cin, Cout=Cout, k=k):   # out[ci, jj, co] = w[co, ci, k - 1 - jj]
                    return _gather3(w.detach(), torch.empty(Cin, k * Cout, dtype=dtype, device=dev), (Cin, k, Cout), (k, -1, Cin * k),
                                    off=k - 1)
                Wd = weight_cache.get(("convw_d", dtype, id(w)), [w], build)
                din = torch.empty(B * Lx, Cin, dtype=torch.float32, device=dev)
                hip.gemm(hip.operand(dxp, Cout, rpb=Lx, bstride=(Lx + 2 * pad) * Cout), hip.operand(Wd, k * Cout), hip.operand(din, Cin),
                         B * Lx, Cin, k * Cout, _dt(dtype), flags=hip.OUT_F32,
                         R=hip.operand(d_after.view(B * Lx, -1), Cin) if i == 0 else None)
                dy = din
                if i == 0:
                    d_before = din
        gx = None
        if ctx.needs_input_grad[0]:
            gx = (d_before if dtype == torch.float32 else to_compute(d_before)).view(B, Lx, -1)
        return (gx, None, None) + (None,) * len(params)


def postnet(before, blocks, training, dropout_p):
    """blocks: the nn.Sequential blocks of espnet's Postnet (block[0] = Conv1d without bias, block[1] = BatchNorm1d)."""
    params, bufs, cfg = [], [], []
    for blk in blocks:
        conv, bn = blk[0], blk[1]
        params += [conv.weight, bn.weight, bn.bias]
        bufs.append((bn.running_mean, bn.running_var, bn.num_batches_tracked))
        cfg.append((float(bn.eps), bn.momentum))
    return PostnetFunction.apply(before, (bool(training), float(dropout_p), cfg), bufs, *params)
