"""Running averages of the per-step losses.  Mirror of ``mano_train/evaluation/evalutils.py:1-28`` (same class and
method names, ``average_meters[name].avg``), plus ``add_loss_dict`` which takes a whole device-side loss dict and reads
it back with ONE device->host copy (the reference calls ``.item()`` once per key per step, ``epochpass3d.py:111-117``)."""
from obman_train_amd.trainer import read_losses


class AverageMeter:
    """Latest value, weighted running sum and count; ``avg`` is derived."""

    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n


class AverageMeters:
    def __init__(self):
        self.average_meters = {}

    def add_loss_value(self, loss_name, loss_val, n=1):
        self.average_meters.setdefault(loss_name, AverageMeter()).update(loss_val, n=n)

    def add_loss_dict(self, losses, n=1):
        values = read_losses(losses)
        for name, val in losses.items():  # numpy / python scalars (the reference's contact_auc) pass through
            if name not in values and val is not None and not hasattr(val, "is_cuda"):
                values[name] = float(val)
        for name, val in values.items():
            self.add_loss_value(name, val, n=n)
        return values

    def add_loss_window(self, window):
        """A list of per-step loss dicts (device tensors) -> every step enters the running means exactly as if it had been
        added on its own (a key counts only the steps where it was not None), with ONE device->host transfer for the whole
        window.  Returns the last step's values (what the reference prints)."""
        import torch

        slots, parts = [], []
        for step, losses in enumerate(window):
            for name, val in losses.items():
                if val is None:
                    continue
                if torch.is_tensor(val):
                    slots.append((step, name, len(parts)))
                    parts.append(val.detach().reshape(-1)[0].float())
                else:
                    slots.append((step, name, float(val)))
        flat = torch.stack(parts).cpu().tolist() if parts else []
        last = {}
        for step, name, ref in slots:
            val = flat[ref] if isinstance(ref, int) else ref
            self.add_loss_value(name, val)
            if step == len(window) - 1:
                last[name] = val
        return last
