"""CPU, world_size 2 over gloo: the flat-gradient data-parallel layer (speecht5_amd/ddp.py, product code) on a plain-torch
stand-in model with the same layer-boundary hooks as the SpeechT5 mirrors.

Checked (reference semantics: fairseq legacy_ddp + --find-unused-parameters + --update-freq, SpeechT5/README.md:86-88,103,
tasks/speecht5.py:538,556): every rank ends with the MEAN over all W*U micro-batches of each micro-batch's gradient,
bit-identical across ranks; parameters no local micro-batch touched take part (zeros); with two micro-batches per update
(mixed "modalities") a bucket is reduced once, after its last contribution; ranks that skip different layers (LayerDrop)
still issue their collectives in the same order; tied weights are reduced by finish() only."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from speecht5_amd import functional as Fn


class Layer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc = nn.Linear(d, d)

    def forward(self, x, skip=False):
        x = Fn.layer_boundary(x, self)
        if skip:
            return x
        return torch.tanh(self.fc(x))


class Toy(nn.Module):
    def __init__(self, d=8):
        super().__init__()
        self.emb = nn.Linear(d, d, bias=False)   # "tied embedding": used at the input AND by the head
        self.inp = nn.Linear(d, d)
        self.layers = nn.ModuleList([Layer(d) for _ in range(3)])
        self.unused = nn.Linear(d, d)       # never touched by any rank
        self.rank1_only = nn.Linear(d, 1)   # touched by one "modality" only
        self.head = nn.Linear(d, 1)
        self.tied_head = nn.Linear(d, d, bias=False)
        self.tied_head.weight = self.emb.weight

    def forward(self, x, use_extra, skip=()):
        x = self.inp(self.emb(x))
        for i, l in enumerate(self.layers):
            x = l(x, skip=i in skip)
        x = Fn.layer_boundary(x, self, "out")
        y = self.head(x) + self.tied_head(x).sum(-1, keepdim=True)
        if use_extra:
            y = y + self.rank1_only(x)
        return y


def _groups(model):
    from speecht5_amd.ddp import BucketGroup
    return [BucketGroup([model.head, model.rank1_only, model.tied_head], triggers=[(model, "out")])] + \
           [BucketGroup([l]) for l in reversed(list(model.layers))]


def _micro(X, rank, u, U):
    i = (rank * U + u) * 4
    return X[i:i + 4]


def _worker(rank, world, port, ret, U, skips, local=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speecht5_amd.ddp import FlatGradDataParallel
    torch.manual_seed(0)
    model = Toy()
    ddp = FlatGradDataParallel(model, bucket_groups=_groups(model))
    launched = []
    orig = dist.all_reduce

    def spy(t, *a, **k):   # record the issue order of the collectives by (offset, length) inside the flat buffer
        launched.append(((t.data_ptr() - ddp.flat.data_ptr()) // 4, t.numel()))
        return orig(t, *a, **k)
    dist.all_reduce = spy
    torch.manual_seed(1)
    X = torch.randn(4 * world * U, 8)
    ddp.zero_grad()
    during = []

    def fwd_bwd(u):
        # modality mix: micro-batch (rank + u) odd uses the extra head
        model(_micro(X, rank, u, U), use_extra=((rank + u) % 2 == 1), skip=skips[rank][u]).pow(2).mean().backward()
        during.append(len(launched))
    if local == "phased":
        # the several-rank replayed form with overlapped exchange (speecht5_amd/update.py): every micro-batch local; the LAST
        # one's forward under cut_points, its backward in phases, a contiguous bucket range handed to the group after each
        with ddp.local_phase():
            for u in range(U - 1):
                fwd_bwd(u)
            with ddp.cut_points([0, 2]) as cuts:
                loss = model(_micro(X, rank, U - 1, U), use_extra=((rank + U - 1) % 2 == 1), skip=skips[rank][U - 1]).pow(2).mean()
            assert [c[0] for c in cuts] == [2, 0], cuts          # forward order: the deeper cut first
            for fn, upto in ddp.backward_phases(loss, cuts):
                fn()
                ddp.flush_deferred()
                ddp.reduce_bucket_range(upto)
                during.append(len(launched))
        ddp.wait_reductions()
        ddp.flat.mul_(1.0 / world)
    elif local:
        # the replayed-step form (bench.py, N > 1): every micro-batch local, the second one into the twin gradient buffer,
        # then ONE all-reduce over the whole flat buffer
        with ddp.local_phase():
            for u in range(U):
                with ddp._grad_slot(u % 2):
                    fwd_bwd(u)
            ddp._pair_pending = U > 1
        ddp.all_reduce_gradients()
    else:
        ddp.accumulate(list(range(U)), fwd_bwd)
        ddp.finish()
    dist.all_reduce = orig
    # model.zero_grad() drops the views: the wrapper must notice and re-install them
    model.zero_grad(set_to_none=True)
    ddp.check_grad_views()
    views_ok = all(p.grad is not None and p.grad.data_ptr() == ddp.flat.data_ptr() + o * 4 for p, o in zip(ddp.params, ddp.offsets))
    ret[rank] = dict(flat=ddp.flat.clone(), launched=launched, during=during, nb=len(ddp.buckets), buckets=list(ddp.buckets),
                     names={n: o for (n, p), o in zip([(n, p) for n, p in model.named_parameters()], [None] * 99)},
                     emb_off=ddp.offsets[[id(p) for p in ddp.params].index(id(model.emb.weight))],
                     grads={n: p.grad.clone() for n, p in model.named_parameters()}, views_ok=views_ok)
    dist.destroy_process_group()


def _reference(world, U, skips):
    torch.manual_seed(0)
    model = Toy()
    torch.manual_seed(1)
    X = torch.randn(4 * world * U, 8)
    tot = None
    for r in range(world):
        for u in range(U):
            model.zero_grad()
            model(_micro(X, r, u, U), use_extra=((r + u) % 2 == 1), skip=skips[r][u]).pow(2).mean().backward()
            g = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
            tot = g if tot is None else {n: tot[n] + g[n] for n in g}
    return {n: v / world for n, v in tot.items()}   # mean over ranks of the per-rank SUM over micro-batches


def _run(U, skips, local=False):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() * 7 + U * 13 + len(str(skips)) + 5 * bool(local) + 11 * (local == "phased")) % 2000
    mp.spawn(_worker, args=(2, port, ret, U, skips, local), nprocs=2, join=True)
    return ret, _reference(2, U, skips)


def _check(ret, ref, one_message=False):
    for r in range(2):
        for n, v in ref.items():
            assert torch.allclose(ret[r]["grads"][n], v, atol=1e-6), (r, n, (ret[r]["grads"][n] - v).abs().max())
        assert float(ret[r]["grads"]["unused.weight"].abs().sum()) == 0.0
        assert ret[r]["views_ok"]
    assert torch.equal(ret[0]["flat"], ret[1]["flat"])
    # same collectives in the same order on both ranks, each bucket exactly once, in index order
    assert ret[0]["launched"] == ret[1]["launched"]
    if one_message:
        assert ret[0]["launched"] == [(0, ret[0]["flat"].numel())]
    else:
        assert ret[0]["launched"] == [(s, e - s) for s, e in ret[0]["buckets"]]


NOSKIP = [[(), ()], [(), ()]]


def test_two_rank_mean_equals_single_process():
    ret, ref = _run(1, NOSKIP)
    _check(ret, ref)
    # the head trigger and the layer triggers fired during backward: every bucket but the final one was launched before finish()
    assert ret[0]["nb"] == 5 and ret[0]["during"][-1] == 4
    # the tied weight lives in the final bucket (reduced by finish() only)
    s, e = ret[0]["buckets"][-1]
    assert s <= ret[0]["emb_off"] < e


def test_two_micro_batches_per_update_mixed_modalities():
    """--update-freq 2 (bench.py's step: speech then text micro-batch): nothing is reduced during the first backward,
    buckets go out during the second, the result is the mean over ranks of the accumulated gradients."""
    ret, ref = _run(2, NOSKIP)
    _check(ret, ref)
    for r in range(2):
        assert ret[r]["during"][0] == 0, "a bucket was all-reduced before its last micro-batch"
        assert ret[r]["during"][1] == 4


def test_ranks_skipping_different_layers_keep_collective_order():
    """LayerDrop draws differ per rank: rank 0 skips layer 1 in its last micro-batch, rank 1 skips layer 2.  A skipped layer
    still reports its bucket; collectives stay in bucket order on both ranks and the mean is right."""
    skips = [[(), (1,)], [(0,), (2,)]]
    ret, ref = _run(2, skips)
    _check(ret, ref)


def test_local_phase_then_one_all_reduce():
    """local_phase() + all_reduce_gradients(): the form a replayed (HIP graph) update takes on several ranks -- no bucket
    triggers during either backward, micro-batch 1 into the second gradient buffer, one all-reduce of everything; ranks that
    skip different layers included.  Same mean as the bucketed path."""
    skips = [[(), (1,)], [(0,), (2,)]]
    ret, ref = _run(2, skips, local=True)
    _check(ret, ref, one_message=True)
    for r in range(2):
        assert ret[r]["during"] == [0, 0], "a collective was issued inside local_phase()"


def test_phased_backward_hands_bucket_ranges_to_the_group_between_phases():
    """The overlapped exchange of a replayed several-rank update: the last micro-batch's autograd graph is cut at two bucket
    boundaries (ddp.cut_points), its backward runs in three phases, and after each phase ONE all-reduce over the contiguous
    range of buckets that just became complete is issued (heads | layers 2-1 | layer 0 + everything else) -- three collectives in
    the same order on both ranks (LayerDrop-divergent ranks included), same mean as the bucketed path."""
    skips = [[(), (1,)], [(0,), (2,)]]
    ret, ref = _run(2, skips, local="phased")
    for r in range(2):
        for n, v in ref.items():
            assert torch.allclose(ret[r]["grads"][n], v, atol=1e-6), (r, n, (ret[r]["grads"][n] - v).abs().max())
        b = ret[r]["buckets"]
        assert ret[r]["launched"] == [(b[0][0], b[0][1] - b[0][0]), (b[1][0], b[2][1] - b[1][0]), (b[3][0], b[4][1] - b[3][0])], ret[r]["launched"]
        assert ret[r]["during"] == [0, 1, 2, 3], ret[r]["during"]      # nothing during the first micro-batch, one range per phase
    assert torch.equal(ret[0]["flat"], ret[1]["flat"])


class _SharedKeyLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc = nn.Linear(d, d)

    def forward(self, x, keys):
        x = Fn.layer_boundary(x, self)
        return torch.tanh(self.fc(x) + (x.to(torch.bfloat16) @ keys).float())


class _SharedKeyStack(nn.Module):
    """Six layers that all read ONE bf16 tensor derived from a parameter (the encoder's relative-position keys)."""

    def __init__(self, d=16):
        super().__init__()
        self.table = nn.Parameter(torch.randn(d, d) * 0.3)
        self.layers = nn.ModuleList([_SharedKeyLayer(d) for _ in range(6)])
        self.head = nn.Linear(d, 1)

    def forward(self, x):
        keys = self.table.to(torch.bfloat16) * 1.0         # (the producer: one node between the parameter and its readers)
        for l in self.layers:
            x = Fn.layer_boundary(x, l)
            x = l(x, Fn.layer_boundary(keys, self, "shared"))
        return self.head(Fn.layer_boundary(x, self, "out")).sum()


def test_tensor_shared_by_a_cut_stack_keeps_its_summation_order():
    """A phased backward (ddp.cut_points / backward_phases) cuts the stack twice; the gradient of the tensor all layers read
    is a bf16 sum of six terms.  Its phased value must be THE SAME BITS as the uncut backward's: each region's leaf is seeded
    with the fold the later regions left, so the sequence of additions is unchanged (ddp._boundary, tag "shared")."""
    from speecht5_amd.ddp import BucketGroup, FlatGradDataParallel
    torch.manual_seed(3)
    model = _SharedKeyStack()
    ddp = FlatGradDataParallel(model, bucket_groups=[BucketGroup([model.head], triggers=[(model, "out")])] +
                               [BucketGroup([l]) for l in reversed(list(model.layers))])
    try:
        x = torch.randn(64, 16)
        ddp.zero_grad()
        model(x).backward()
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        mb = ddp.module_bucket
        ddp.zero_grad()
        with ddp.local_phase():
            with ddp.cut_points([mb[(id(model.layers[4]), None)], mb[(id(model.layers[2]), None)]]) as cuts:
                loss = model(x)
            assert sum(1 for c in cuts if c[0] != "shared") == 2
            regions = [rg for c in cuts if c[0] == "shared" for rg, _ in c[2]]
            assert regions == [0, 1, 2], regions
            phases = ddp.backward_phases(loss, cuts)
            assert len(phases) == 3
            for k, (fn, upto) in enumerate(phases):
                fn()
                if k < 2:
                    assert model.table.grad is None or not model.table.grad.any(), "the producer ran before its region"
        for n, p in model.named_parameters():
            assert torch.equal(p.grad, ref[n]), (n, float((p.grad - ref[n]).abs().max()))
        assert ref["table"].abs().max() > 0
    finally:
        ddp.close()


def test_shared_tensor_when_no_cut_materialises_or_the_cut_is_the_first_layer():
    """Edge cases of the shared-tensor bookkeeping: (a) the requested cut sits where nothing requires a gradient yet (the stack's
    input): no cut is made, one region, and the producer's backward still runs (once) inside the only phase; (b) a cut directly behind
    the first layer: the producer's region holds a single consumer.  Both must give the uncut backward's bits."""
    from speecht5_amd.ddp import BucketGroup, FlatGradDataParallel
    torch.manual_seed(4)
    model = _SharedKeyStack()
    ddp = FlatGradDataParallel(model, bucket_groups=[BucketGroup([model.head], triggers=[(model, "out")])] +
                               [BucketGroup([l]) for l in reversed(list(model.layers))])
    try:
        x = torch.randn(32, 16)
        ddp.zero_grad()
        model(x).backward()
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        mb = ddp.module_bucket
        for cut_layers, n_real in (([0], 0), ([1], 1), ([1, 5], 2)):
            ddp.zero_grad()
            with ddp.local_phase():
                with ddp.cut_points([mb[(id(model.layers[i]), None)] for i in cut_layers]) as cuts:
                    loss = model(x)
                assert sum(1 for c in cuts if c[0] != "shared") == n_real, (cut_layers, cuts)
                phases = ddp.backward_phases(loss, cuts)
                assert len(phases) == n_real + 1
                for fn, _ in phases:
                    fn()
            for n, p in model.named_parameters():
                assert torch.equal(p.grad, ref[n]), (cut_layers, n, float((p.grad - ref[n]).abs().max()))
    finally:
        ddp.close()


class _TappedStack(nn.Module):
    """A front end whose output feeds a layer stack AND, directly, a penalty term of the loss (the feature penalty on the convolution
    stack's output, speech_encoder_prenet.py:172-176): a path from the loss to the front end that crosses no layer boundary."""

    def __init__(self, d=16):
        super().__init__()
        self.front = nn.Linear(d, d)
        self.layers = nn.ModuleList([Layer(d) for _ in range(4)])
        self.head = nn.Linear(d, 1)
        self.front_runs = 0

    def forward(self, x):
        f = torch.tanh(self.front(x))
        f.register_hook(lambda g: setattr(self, "front_runs", self.front_runs + 1))
        pen = Fn.layer_boundary(f, self, "bypass").pow(2).mean()
        x = f
        for l in self.layers:
            x = l(x)
        return self.head(Fn.layer_boundary(x, self, "out")).sum() + 10.0 * pen


def test_loss_term_tapping_an_early_tensor_survives_a_cut_backward():
    """Round 6 (the speech micro-batch inside a phased backward): without the "bypass" boundary the first phase's loss.backward() runs
    the front end's backward with the penalty's gradient alone and the last phase runs it AGAIN ("backward through the graph a second
    time").  With it the tap's gradient is held back and joins the main gradient as a root of the front end's own phase: one backward
    through the front end, the uncut backward's bits -- with two cuts, one cut, and when no cut materialises (tap in the loss's region)."""
    from speecht5_amd.ddp import BucketGroup, FlatGradDataParallel
    torch.manual_seed(5)
    model = _TappedStack()
    ddp = FlatGradDataParallel(model, bucket_groups=[BucketGroup([model.head], triggers=[(model, "out")])] +
                               [BucketGroup([l]) for l in reversed(list(model.layers))])
    try:
        x = torch.randn(32, 16)
        ddp.zero_grad()
        model(x).backward()
        ref = {n: p.grad.clone() for n, p in model.named_parameters()}
        assert model.front_runs == 1
        mb = ddp.module_bucket
        for cut_keys, n_real in (([(id(model), "out"), (id(model.layers[2]), None)], 2), ([(id(model.layers[1]), None)], 1), ([], 0)):
            model.front_runs = 0
            ddp.zero_grad()
            with ddp.local_phase():
                with ddp.cut_points([mb[k] for k in cut_keys] or [10 ** 6]) as cuts:
                    loss = model(x)
                assert sum(1 for c in cuts if c[0] not in ("shared", "bypass")) == n_real and sum(1 for c in cuts if c[0] == "bypass") == 1
                phases = ddp.backward_phases(loss, cuts)
                assert len(phases) == n_real + 1
                for k, (fn, _) in enumerate(phases):
                    fn()
                    assert model.front_runs == (1 if k == n_real else 0), (cut_keys, k, model.front_runs)
            for n, p in model.named_parameters():
                assert torch.equal(p.grad, ref[n]), (n_real, n, float((p.grad - ref[n]).abs().max()))
    finally:
        ddp.close()
