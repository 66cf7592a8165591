"""Plugging the GPU collaters (speecht5_amd.collate) into the reference's datasets (drop-in side of it).

The reference's data plane -- manifests, audio decoding, librosa features, tokenisation, batching by size -- stays its own CPU code
(task.load_dataset runs the plug-in's `load_dataset`).  What moves is the step that builds the batch tensors: `GpuCollated(dataset,
device)` wraps one of the four dataset classes of /root/reference/SpeechT5/speecht5/data/ (speech_dataset.SpeechPretrainDataset,
text_to_speech_dataset.TextToSpeechDataset, speech_to_text_dataset.SpeechToTextDataset, text_dataset.TextPretrainDataset) -- or anything
with their attributes -- so that `__getitem__` is the dataset's own (its tensors moved to the device once, per item) and `collater` is
the ragged-gather collater of speecht5_amd.collate, configured from the dataset's attributes.  Every other attribute (sizes,
ordered_indices, num_tokens, set_epoch ...) is the wrapped dataset's.  `wrap_datasets` does it for a MultitaskDataset's members in place
(data/multitask_dataset.py:31-92: its collater dispatches to the member a batch came from).

The batches are bit-identical to the wrapped collaters' (tests/test_collate_gpu.py, tests/test_collate2_gpu.py); the numpy / torch
generator streams the reference's collaters and noise draw from are consumed identically."""
import torch

from . import collate as C


def _to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, list):
        return [_to_device(v, device) for v in x]
    return x


def _collater_for(ds, device):
    name = type(ds).__name__
    if name == "SpeechPretrainDataset" or all(hasattr(ds, a) for a in ("pad_audio", "random_crop", "label_rates", "pad_list")):
        return C.SpeechPretrainCollater(device, sample_rate=ds.sample_rate, label_rates=ds.label_rates, pad_list=ds.pad_list,
                                        max_sample_size=ds.max_sample_size, pad_audio=ds.pad_audio, random_crop=ds.random_crop,
                                        reduction_factor=ds.reduction_factor, single_target=getattr(ds, "single_target", False))
    if name == "TextToSpeechDataset" or (hasattr(ds, "src_dict") and hasattr(ds, "reduction_factor")):
        return C.TextToSpeechCollater(device, pad_idx=ds.src_dict.pad(), reduction_factor=ds.reduction_factor)
    if name == "SpeechToTextDataset" or hasattr(ds, "tgt_dict"):
        return C.SpeechToTextCollater(device, pad_idx=ds.tgt_dict.pad(), eos_idx=ds.tgt_dict.eos())
    if name == "TextPretrainDataset" or (hasattr(ds, "vocab") and hasattr(ds, "mask_idx")):
        return C.TextPretrainCollater(device, pad_idx=ds.vocab.pad())
    raise TypeError(f"no GPU collater for dataset class {name}")


class GpuCollated:
    def __init__(self, dataset, device):
        self.dataset, self.device = dataset, torch.device(device)
        self.gpu_collater = _collater_for(dataset, self.device)

    def __getitem__(self, index):
        item = self.dataset[index]
        return {k: _to_device(v, self.device) for k, v in item.items()}

    def __len__(self):
        return len(self.dataset)

    def collater(self, samples, **kwargs):
        if kwargs.get("pad_to_length") is not None:
            raise NotImplementedError("pad_to_length is not used by the SpeechT5 recipes")
        return self.gpu_collater(samples)

    def __getattr__(self, name):          # (sizes, ordered_indices, num_tokens, size, set_epoch, prefetch ...)
        return getattr(self.dataset, name)


def wrap_datasets(dataset, device):
    """GpuCollated around `dataset`, or around every member of a MultitaskDataset (returned unchanged otherwise: its own collater
    dispatches to the wrapped members)."""
    members = getattr(dataset, "datasets", None)
    if isinstance(members, list) and hasattr(dataset, "sample_ratios"):
        dataset.datasets = [m if isinstance(m, GpuCollated) else GpuCollated(m, device) for m in members]
        return dataset
    return GpuCollated(dataset, device)
