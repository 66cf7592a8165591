"""Host half of the replayed step (speecht5_amd/functional.HostStaging, no GPU): value-less draws from the CPU random streams
(the encoder's per-layer LayerDrop draw, speecht5_amd/modules/encoder.py, reference encoder.py:104) must be consumed before every
replay exactly as an eagerly enqueued step consumes them -- otherwise the span masks drawn from the same numpy stream later
differ from the eager path's (found in round 2: replay and eager diverged from the third replay on)."""
import numpy as np

from speecht5_amd import functional as Fn


def _step(st, log):
    """What a training step does with the host random stream: 3 LayerDrop draws, one 'mask' draw whose value matters (the
    product stages it as a device input; here it is logged), 2 more LayerDrop draws."""
    for _ in range(3):
        st.draw(np.random.random)
    st.draw(lambda: log.append(float(np.random.random())) or 0.0)
    for _ in range(2):
        st.draw(np.random.random)


def test_replay_repeats_value_less_draws_in_recording_order():
    np.random.seed(3)
    eager, st = [], Fn.HostStaging()
    for _ in range(6):                       # eager: 6 steps
        _step(st, eager)
    assert not st.entries
    np.random.seed(3)
    got, st = [], Fn.HostStaging()
    for _ in range(2):                       # two recorded steps ...
        st.begin_step("record")
        _step(st, got)
    assert len(st.entries) == 6 and all(e[0] is None for e in st.entries)
    st.produce(0)                            # ... the host half of the step that gets captured (graph.StepGraph._pre_replay) ...
    st.begin_step("capture")                 # ... and the capture pass itself: draws nothing, hands back the recorded values
    pos = np.random.get_state()[2]
    n = len(got)
    _step(st, got)
    assert np.random.get_state()[2] == pos and len(got) == n
    st.mode = None
    for _ in range(3):                       # three more replays
        st.produce(0)
    assert got == eager


def test_draw_outside_a_recorded_step_is_a_plain_call():
    st = Fn.HostStaging()
    np.random.seed(0)
    a = st.draw(np.random.random)
    np.random.seed(0)
    assert a == np.random.random() and not st.entries


def test_alignment_weights_context_sets_and_restores_the_decoder_flag():
    """criterions.alignment_weights: a criterion's statement whether it reads the decoder's alignment weights lives only for
    its model(...) call (also when that call raises); models without a decoder are left alone."""
    from types import SimpleNamespace
    from speecht5_amd.criterions import alignment_weights
    model = SimpleNamespace(decoder=SimpleNamespace(materialise_alignment=True))
    with alignment_weights(model, False):
        assert model.decoder.materialise_alignment is False
        with alignment_weights(model, True):
            assert model.decoder.materialise_alignment is True
        assert model.decoder.materialise_alignment is False
    assert model.decoder.materialise_alignment is True
    try:
        with alignment_weights(model, False):
            raise RuntimeError("forward failed")
    except RuntimeError:
        pass
    assert model.decoder.materialise_alignment is True
    with alignment_weights(SimpleNamespace(), False):   # (e.g. an encoder-only wrapper)
        pass
