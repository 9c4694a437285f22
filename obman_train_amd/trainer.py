"""One training step with the semantics of the reference's hot loop (``epochpass3d.py:71-121``):
forward -> ``optimizer.zero_grad()`` -> ``backward()`` -> ``optimizer.step()``.  The per-loss
``.item()`` of the reference (one device->host sync per key per step, ``:111-117``) is NOT done here:
loss values stay on the device; ``read_losses`` batches them into one transfer when the caller logs."""
import torch


def train_step(model, optimizer, sample, buckets=None):
    total, results, losses = model.forward(sample)
    if buckets is not None and buckets.enabled:
        buckets.zero_grad()  # gradients are views into the all-reduce buckets: zero them in place
    else:
        optimizer.zero_grad(set_to_none=True)
    total.backward()
    if buckets is not None:
        buckets.finish()
    optimizer.step()
    return total, results, losses


def read_losses(losses):
    """{name: python float} with a single device->host copy."""
    keys = [k for k, v in losses.items() if torch.is_tensor(v)]
    if not keys:
        return {}
    flat = torch.stack([losses[k].detach().reshape(-1)[0].float() for k in keys]).cpu().tolist()
    return dict(zip(keys, flat))


def make_optimizer(model, name="adam", lr=1e-4, momentum=0.9, weight_decay=0.0):
    """traineval.py:104-127 (defaults nets3dopts.py:249-273)."""
    params = [p for p in model.parameters() if p.requires_grad]
    if name == "adam":
        fused = all(p.is_cuda for p in params)  # one multi-tensor kernel instead of ~10 foreach launches per step
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=fused)
    if name == "rms":
        return torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay)
    if name == "sgd":
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    raise ValueError(name)
