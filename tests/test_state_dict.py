"""CPU: the module tree exposes exactly the reference's parameter/buffer names and shapes
 -- released checkpoints must load -- and the C-ABI library exports every
symbol that include/speecht5_hip.h declares."""
import ctypes
import os
import re
from argparse import Namespace

import torch

from tests.util import Task, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_matches_reference_names_and_shapes():
    from speecht5_amd.speecht5 import T5TransformerModel
    m, _ = load_golden(None)
    model = T5TransformerModel.build_model(Namespace(**m["args"]), Task())
    own = model.state_dict()
    ref = m["state_dict"]
    assert set(own.keys()) == set(ref.keys()), (sorted(set(ref) - set(own))[:5], sorted(set(own) - set(ref))[:5])
    for k, v in ref.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    model.load_state_dict(dict(ref))
    for k, v in ref.items():
        assert torch.equal(model.state_dict()[k], v), k


def test_base_architecture_has_reference_parameter_count():
    from speecht5_amd.speecht5 import T5TransformerModel, t5_transformer_base
    args = Namespace(label_rates=50, sample_rate=16000, speech_odim=80, bert_init=True, use_codebook=True,
                     share_input_output_embed=True)
    t5_transformer_base(args)

    class BigTask(Task):
        def __init__(self):
            super().__init__(vocab=77, n_units=500)
    with torch.device("meta"):
        model = T5TransformerModel.build_model(args, BigTask()) if False else None
    model = T5TransformerModel.build_model(args, BigTask())
    n = sum(p.numel() for p in model.parameters())
    assert n == 154392031 - 0 or abs(n - 154392031) < 10, n  # reference Base: 154.392031 M (SURVEY.md probe)
    assert args.encoder_layers == 12 and args.decoder_layers == 6


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "speecht5_hip.h")).read()
    declared = set(re.findall(r"\b(st5_[a-z0-9_]+)\s*\(", hdr))
    from speecht5_amd import hip
    lib = ctypes.CDLL(os.path.join(ROOT, "speecht5_amd", "libspeecht5_hip.so"))
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/speecht5_hip.h but not exported"
    assert declared == set(hip.exported_symbols()), declared ^ set(hip.exported_symbols())
    assert b"gfx950" in hip.lib().st5_version()


def test_product_path_fails_loudly_without_the_library(monkeypatch):
    from speecht5_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "_LIB_PATH", "/nonexistent/libspeecht5_hip.so")
    import pytest
    with pytest.raises(hip.HipLibraryMissing):
        hip.lib()
