"""Batches for the training loop: DataLoader workers run ``HandDataset.get_sample`` (file reads, JPEG decode, RNG draws,
annotation arithmetic - no pixel work); the training process renders each batch's images on its GPU.

Replaces ``torch.utils.data.DataLoader(train_dat, batch_size, shuffle, num_workers, pin_memory, drop_last)`` of
``traineval.py:219-227``: iterate it the same way, the batch dict has the same keys and shapes, and
``batch[TransQueries.images]`` is already a device tensor (``HandNet.forward`` calls ``.cuda()`` on it, a no-op then).
"""
import torch

from .handataset import HandDataset


def _as_list(samples):
    return samples


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, drop_last=False, device="cuda",
                 channels_last=False, stage=None, **loader_kwargs):
        self.dataset = dataset
        self.stage = stage if stage is not None else dataset.image_stage(device=device, channels_last=channels_last)
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                                  drop_last=drop_last, collate_fn=_as_list, **loader_kwargs)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for samples in self.loader:
            yield HandDataset.collate(samples, self.stage)
