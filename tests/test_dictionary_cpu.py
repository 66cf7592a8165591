"""speecht5_amd.task._Dictionary against fairseq.data.Dictionary's documented behaviour for what the reference's data plane calls on
`task.dicts[...]`: LabelEncoder = `dictionary.encode_line(label, append_eos=False, add_if_not_exist=False)`
(tasks/speecht5.py:24-36, restated below because the module imports fairseq), the `#fairseq:overwrite` row rule of
add_from_file (a duplicate row is APPENDED and becomes the symbol's index, so later symbols keep fairseq's ids), and the
--iid-noise-target sentinels (tasks/speecht5.py:289-294)."""
from argparse import Namespace

import pytest
import torch

from speecht5_amd.task import SpeechT5Task, _Dictionary


class LabelEncoder:           # tasks/speecht5.py:24-36
    def __init__(self, dictionary):
        self.dictionary = dictionary

    def __call__(self, label):
        return self.dictionary.encode_line(label, append_eos=False, add_if_not_exist=False)


def test_encode_line_is_what_label_encoder_needs(tmp_path):
    p = tmp_path / "dict.txt"
    p.write_text("a 10\nb 7\nc 3\n")
    d = _Dictionary.load(str(p))
    assert len(d) == 7 and d.nspecial == 4 and d.count[4:] == [10, 7, 3]
    ids = LabelEncoder(d)("b  a zz c\n")
    assert ids.dtype == torch.int32 and ids.tolist() == [5, 4, d.unk(), 6]
    assert len(d) == 7                                   # add_if_not_exist=False: nothing was added
    ids = d.encode_line("c q")                          # fairseq defaults: add unknown words, append </s>
    assert ids.tolist() == [6, 7, d.eos()] and d[7] == "q" and d.index("q") == 7
    assert d.encode_line("a b", append_eos=False, reverse_order=True).tolist() == [5, 4]


def test_overwrite_rows_keep_fairseq_indices(tmp_path):
    p = tmp_path / "dict.txt"
    p.write_text("a 10\nb 7\na 5 #fairseq:overwrite\nc 3\n")
    d = _Dictionary.load(str(p))
    # fairseq: rows 4 (a), 5 (b), 6 (a again, now THE index of "a"), 7 (c)
    assert len(d) == 8 and d.index("a") == 6 and d.index("b") == 5 and d.index("c") == 7
    p.write_text("a 10\na 5\n")
    with pytest.raises(RuntimeError):
        _Dictionary.load(str(p))


def test_iid_noise_target_sentinels():
    text = _Dictionary(["x", "y"])
    task = SpeechT5Task(Namespace(t5_task="pretrain", iid_noise_target=True), {"text": text})
    assert task.uni_mask_idxs is not None and task.uni_mask_idxs.shape == (600,)
    assert text[int(task.uni_mask_idxs[0])] == "<mask>0" and text[int(task.uni_mask_idxs[599])] == "<mask>599"
    assert task.mask_idx == text.index("<mask>") and task.blank_symbol_idx == text.index("<ctc_blank>")
    assert int(task.uni_mask_idxs[0]) == task.blank_symbol_idx + 1
    assert SpeechT5Task(Namespace(t5_task="pretrain"), {"text": _Dictionary(["x"])}).uni_mask_idxs is None
