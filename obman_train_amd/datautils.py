"""Round-robin over several loaders, tagging every batch with where it came from.

Mirror of ``ConcatDataloader`` (reference ``mano_train/datautils.py:5-39``; ``traineval.py:232,271`` wraps its train and
validation loaders in it): batches are drawn from the loaders in turn until the first one is exhausted, and each batch dict
gains ``"dataset"``, ``"root"`` (``"palm"`` for stereohands / zimsynth, else ``"wrist"`` - ``HandNet.forward`` reads it),
``"use_stereohands"`` and ``"split"``.  Works with ``torch.utils.data.DataLoader`` and with ``DeviceBatchLoader``.
"""
from torch.utils.data import Subset

PALM_ROOTED = ("stereohands", "zimsynth")


def _pose_dataset_of(loader):
    dataset = loader.dataset
    while isinstance(dataset, Subset):
        dataset = dataset.dataset
    return dataset.pose_dataset


class ConcatDataloader:
    def __init__(self, dataloaders):
        self.loaders = list(dataloaders)

    def __len__(self):
        return min(len(loader) for loader in self.loaders) * len(self.loaders)

    def __iter__(self):
        streams = [(iter(loader), _pose_dataset_of(loader)) for loader in self.loaders]
        while True:
            for batches, pose in streams:
                try:
                    batch = next(batches)
                except StopIteration:
                    return
                name = getattr(pose, "name", type(pose).__name__.lower())
                batch["dataset"] = name
                batch["root"] = "palm" if name in PALM_ROOTED else "wrist"
                batch["use_stereohands"] = name == "stereohands"
                batch["split"] = getattr(pose, "split", None)
                yield batch
