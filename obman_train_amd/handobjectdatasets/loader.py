"""Batches for the training loop: DataLoader workers run ``HandDataset.get_sample`` (file reads, JPEG decode, RNG draws,
annotation arithmetic - no pixel work); the training process renders each batch's images on its GPU.

Replaces ``torch.utils.data.DataLoader(train_dat, batch_size, shuffle, num_workers, pin_memory, drop_last)`` of
``traineval.py:219-227``: iterate it the same way, the batch dict has the same keys and shapes, and
``batch[TransQueries.images]`` is already a device tensor (``HandNet.forward`` calls ``.cuda()`` on it, a no-op then).

With ``prefetch`` (default) the upload of the next batch's frames and its three input-stream kernels are issued on a side
HIP stream while the caller is still training on the current batch; the consumer's stream waits on an event before it touches
the batch, so the PCIe transfer (25 MB of uint8 frames at bs 64) disappears behind the step.
"""
import torch

from ..queries import TransQueries
from .handataset import HandDataset


def _as_list(samples):
    return samples


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size=1, shuffle=False, num_workers=0, drop_last=False, device="cuda",
                 channels_last=False, stage=None, prefetch=True, **loader_kwargs):
        self.dataset = dataset
        inner = dataset
        while not hasattr(inner, "image_stage") and hasattr(inner, "dataset"):  # torch.utils.data.Subset (get_dataset's limit_size)
            inner = inner.dataset
        self.stage = stage if stage is not None else inner.image_stage(device=device, channels_last=channels_last)
        self.prefetch = bool(prefetch)
        self.loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                                  drop_last=drop_last, collate_fn=_as_list, **loader_kwargs)

    def __len__(self):
        return len(self.loader)

    def _side_stream(self):
        device = getattr(self.stage, "device", None)
        if not self.prefetch or device is None or device.type != "cuda":
            return None
        return torch.cuda.Stream(device=device)

    def __iter__(self):
        side = self._side_stream()
        if side is None:
            for samples in self.loader:
                yield HandDataset.collate(samples, self.stage)
            return

        def launch(samples):  # everything the stage enqueues (upload + kernels) goes to the side stream
            side.wait_stream(torch.cuda.current_stream(side.device))  # ... after whatever produced reusable memory
            with torch.cuda.stream(side):
                batch = HandDataset.collate(samples, self.stage)
                ready = torch.cuda.Event()
                ready.record(side)
            return batch, ready

        pending = None
        for samples in self.loader:
            nxt = launch(samples)
            if pending is not None:
                yield self._hand_over(*pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(*pending)

    @staticmethod
    def _hand_over(batch, ready):
        consumer = torch.cuda.current_stream()
        consumer.wait_event(ready)
        images = batch.get(TransQueries.images)
        if torch.is_tensor(images) and images.is_cuda:
            images.record_stream(consumer)  # allocated on the side stream, consumed (and later freed) on this one
        return batch
