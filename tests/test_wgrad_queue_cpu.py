"""Host logic of the weight-gradient groups (functional._wgrad_queue / flush_wgrads; the launches go to st5_gemm_tn_group) and of
the tied-parameter rule of FlatGradDataParallel, without a GPU: the library call and the stream handle are stand-ins.

What must hold: a group never exceeds one round of the chip (512 tiles of 128^2) or eight problems, is launched as
soon as it reaches 400 tiles, keeps FIFO order, never holds the same output (or bias-gradient column) twice, belongs to ONE stream
and keeps its operands alive until it is launched; a flush point reached with another stream's problems still queued is an error;
tied parameters are marked so that their weight gradients bypass queue and deferred reduction."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

from speecht5_amd import functional as Fn
from speecht5_amd import hip


class _Asum:
    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


def _fixture(monkeypatch, tile_mode):
    launched = []
    cur = {"s": 11}
    monkeypatch.setattr(hip, "stream", lambda: cur["s"])
    monkeypatch.setattr(hip, "gemm_tn_group", lambda problems, dt: launched.append((cur["s"], [p[2].ptr for p in problems], dt)))
    Fn._S.__dict__.pop("wq", None)
    hip.check(hip.lib().st5_gemm_set_tn_group_tile(tile_mode), "st5_gemm_set_tn_group_tile")     # (host state of the library: no GPU needed)
    return SimpleNamespace(launched=launched, cur=cur)


@pytest.fixture
def queue(monkeypatch):
    """128 x 128 accounting (st5_gemm_set_tn_group_tile(1); also what shapes that are no multiples of 256 get)."""
    yield _fixture(monkeypatch, 1)
    Fn._S.__dict__.pop("wq", None)
    hip.lib().st5_gemm_set_tn_group_tile(0)


@pytest.fixture
def queue256(monkeypatch):
    """The default: problems whose M, N are multiples of 256 are counted in 256 x 256 tiles (phased grouped kernel, round 6)."""
    yield _fixture(monkeypatch, 0)
    Fn._S.__dict__.pop("wq", None)


def _q(c_ptr, M, N, keep=None, asum=None, dt=None):
    op = SimpleNamespace(ptr=c_ptr)
    Fn._wgrad_queue(SimpleNamespace(ptr=1), SimpleNamespace(ptr=2), op, M, N, 4096, hip.BF16 if dt is None else dt, 0, asum, keep or ())


def test_an_encoder_layer_is_one_group_of_four(queue):
    # backward order of a post-LN encoder layer: fc2 (768 x 3072 = 144 tiles), fc1 (144), out-proj (36), QKV (108) = 432 tiles
    for c, (M, N) in enumerate([(768, 3072), (3072, 768), (768, 768), (2304, 768)]):
        assert not queue.launched
        _q(100 + c, M, N)
    assert queue.launched == [(11, [100, 101, 102, 103], hip.BF16)]
    Fn.flush_wgrads()
    assert len(queue.launched) == 1                       # nothing left behind


def test_a_group_never_exceeds_a_round_or_eight_problems_and_keeps_order(queue):
    _q(1, 768, 3072)           # 144
    _q(2, 768, 3072)           # 288
    _q(3, 2304, 3072)          # + 432 > 512: the first two go out, this one starts a new group and (432 >= 400) goes out at once
    assert queue.launched == [(11, [1, 2], hip.BF16), (11, [3], hip.BF16)]
    queue.launched.clear()
    for c in range(10):        # ten 36-tile problems: eight, then two at the flush point
        _q(10 + c, 768, 768)
    assert queue.launched == [(11, list(range(10, 18)), hip.BF16)]
    Fn.flush_wgrads()
    assert queue.launched[-1] == (11, [18, 19], hip.BF16)


def test_two_encoder_layers_are_one_round_of_the_phased_kernel(queue256):
    """Round 6: a Base encoder layer is 36 + 36 + 9 + 27 = 108 tiles of 256 x 256; a round is 256, launched from 200 on: two layers."""
    layer = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    for c, (M, N) in enumerate(layer + layer):
        assert not queue256.launched
        _q(100 + c, M, N)
    assert queue256.launched == [(11, list(range(100, 108)), hip.BF16)]
    queue256.launched.clear()
    # Large: 64 + 64 + 16 + 48 = 192 per layer; the next layer's fc2 would make 256 > ... no: 192 + 64 = 256 <= 256 and >= 200: five problems
    large = [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]
    for c, (M, N) in enumerate(large + large[:1]):
        _q(200 + c, M, N)
    assert queue256.launched == [(11, [200, 201, 202, 203, 204], hip.BF16)]


def test_block_tile_classes_are_never_mixed_in_one_launch(queue256):
    _q(1, 768, 768)            # phased class (multiples of 256)
    _q(2, 83, 768)             # the vocabulary projection: 128 x 128 class -> the phased problem goes out first
    assert queue256.launched == [(11, [1], hip.BF16)]
    _q(3, 768, 768)            # and back
    assert queue256.launched[-1] == (11, [2], hip.BF16)
    Fn.flush_wgrads()
    assert queue256.launched[-1] == (11, [3], hip.BF16)


def test_the_same_output_twice_is_never_in_one_launch(queue):
    _q(7, 768, 768)
    _q(8, 768, 768, asum=_Asum(500))
    _q(7, 768, 768)            # tied weights / a module applied twice: the first contribution is launched before the second queues
    assert queue.launched == [(11, [7, 8], hip.BF16)]
    _q(9, 768, 768, asum=_Asum(600))
    assert len(queue.launched) == 1       # (7 again, 9: different outputs, different bias columns)
    _q(10, 768, 768, asum=_Asum(600))     # the same bias-gradient column as 9
    assert queue.launched[-1] == (11, [7, 9], hip.BF16)
    Fn.flush_wgrads()
    assert queue.launched[-1] == (11, [10], hip.BF16)


def test_operands_are_held_until_the_launch(queue):
    t = torch.zeros(4)
    import sys
    before = sys.getrefcount(t)
    _q(1, 768, 768, keep=(t,))
    assert sys.getrefcount(t) > before
    Fn.flush_wgrads()
    assert sys.getrefcount(t) == before


def test_queues_are_per_stream_and_a_foreign_leftover_is_an_error(queue):
    _q(1, 768, 768)
    queue.cur["s"] = 22        # the second micro-batch's stream
    _q(2, 768, 768)
    with pytest.raises(AssertionError):
        Fn.flush_wgrads()      # stream 22's flush point with stream 11's problem still queued
    queue.cur["s"] = 11
    Fn.flush_wgrads()
    assert sorted(queue.launched) == [(11, [1], hip.BF16), (22, [2], hip.BF16)]


def test_tied_parameters_are_marked_for_program_order_reduction():
    from speecht5_amd.ddp import FlatGradDataParallel

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = nn.Embedding(16, 8)
            self.proj = nn.Linear(8, 16, bias=False)
            self.proj.weight = self.embed.weight
            self.fc = nn.Linear(8, 8)

    m = M()
    ddp = FlatGradDataParallel(m, process_group=None) if "process_group" in FlatGradDataParallel.__init__.__code__.co_varnames else FlatGradDataParallel(m)
    try:
        assert getattr(m.embed.weight, "_st5_multi_writer", False)
        assert not getattr(m.fc.weight, "_st5_multi_writer", False) and not getattr(m.fc.bias, "_st5_multi_writer", False)
    finally:
        ddp.close()
