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


def make_optimizer(model, name="adam", lr=1e-4, momentum=0.9, weight_decay=0.0, capturable=False):
    """traineval.py:112-127 (defaults nets3dopts.py:249-273).  ``capturable``: the step may be recorded into a hipGraph
    (``GraphedTrainStep``): Adam keeps its step counters on the device and never reads them on the host."""
    params = [p for p in model.parameters() if p.requires_grad]
    if name == "adam":
        if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            # round 6: the step as this package's multi-tensor kernel (csrc/stepops.hip; device-side step counters, so it records
            # into a hipGraph whatever `capturable` says); a bf16-autocast encoder also gets bf16 shadow filters the kernel keeps
            # current (no per-step weight-cast launches)
            from . import optim

            base = getattr(model, "base_net", None)
            if base is not None and getattr(base, "autocast_dtype", None) == torch.bfloat16:
                optim.attach_bf16_shadows(base)
            return optim.ObmanAdam(params, lr=lr, weight_decay=weight_decay)
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    if name == "rms":
        return torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay)
    if name == "sgd":
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    raise ValueError(name)


def _same_entry(a, b):
    """Equality of two non-tensor sample entries (lists of strings, numpy arrays, scalars, None)."""
    if a is b:
        return True
    if a is None or b is None:
        return False
    try:
        import numpy as np

        if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
            return np.array_equal(np.asarray(a), np.asarray(b))
    except ImportError:
        pass
    try:
        return bool(a == b)
    except (ValueError, RuntimeError):  # "truth value of an array is ambiguous"
        return False


def _watchdog_period_s():
    """Poll period of ProcessGroupNCCL's watchdog loop: ``kWatchdogThreadSleepMillis`` = 100 ms, compiled in (no environment switch
    changes it in torch 2.10; ``TORCH_NCCL_HEARTBEAT_TIMEOUT_SEC`` and friends govern the monitor thread, not this loop)."""
    return 0.1


def wait_for_watchdog(buckets, dev, timeout_s=5.0):
    """Before a capture that will hold RCCL work: wait until the process group's watchdog thread has DROPPED every eager collective.

    ProcessGroupNCCL's watchdog polls the end events of the outstanding EAGER collectives (``hipEventQuery``) until it has seen
    them complete, then removes them from its list.  Once the recorded step pulls RCCL's stream into the capture, HIP answers such a
    query - on an event last recorded on that stream, eagerly, by the warm-up steps - with ``hipErrorCapturedEvent``; the watchdog
    rethrows and the process aborts (measured in round 5: 1 of 8 runs of the 1-rank bench command, always inside the capture window).

    Round 6: a CONDITION instead of round 5's fixed ``sleep(0.5)``.  The watchdog publishes its progress in the process group's
    status record, ``pg_status[pg]["last_completed_collective"]`` of the flight recorder's dump
    (``torch._C._distributed_c10d._dump_nccl_trace``): the sequence number of the last collective it has seen complete AND dropped
    from its list.  When that has reached ``last_enqueued_collective`` for every group, no eager work is left for it to query.  The
    record exists when the flight recorder is on (``TORCH_NCCL_TRACE_BUFFER_SIZE`` > 0: ``dp.init_rccl`` sets 64 entries; measured:
    the condition turns true 60 ms after the synchronize, ``tools/archive/r06/watchdog_probe.py``).  Where it does not (another torch
    build, the recorder switched off by the caller) the fallback is round 5's measured one: five watchdog poll periods of sleep.
    Returns a dict saying which of the two ended the wait (``bench.py`` prints it)."""
    import pickle
    import time

    torch.cuda.synchronize(dev)  # every eager collective has finished on the device: one watchdog pass drops them all
    t0 = time.perf_counter()
    dump = getattr(torch._C._distributed_c10d, "_dump_nccl_trace", None)
    polls = 0
    while dump is not None and time.perf_counter() - t0 < timeout_s:
        try:
            status = pickle.loads(dump(includeCollectives=False, includeStackTraces=False, onlyActive=False)).get("pg_status") or {}
        except Exception:  # noqa: BLE001 - no recorder in this build
            status = {}
        if not status:
            break
        polls += 1
        if all(int(s.get("last_completed_collective", -1)) >= int(s.get("last_enqueued_collective", 0)) for s in status.values()):
            return {"condition": "watchdog reached last_enqueued_collective", "groups": len(status), "polls": polls,
                    "waited_s": round(time.perf_counter() - t0, 4)}
        time.sleep(_watchdog_period_s() / 4)
    time.sleep(5 * _watchdog_period_s())
    return {"condition": "slept 5 watchdog periods (no flight-recorder status)", "polls": polls, "waited_s": round(time.perf_counter() - t0, 4)}


def _lambda_signature(model):
    """(module path, attribute, value) of every python-number loss weight of the model's loss objects / branches: attributes named
    ``*lambda*`` or ``*_weight`` (``HandNet.contact_lambda``, ``ManoLoss.lambda_verts``, ``AtlasLoss.trans_weight``, ...)."""
    sig = []
    owners = [("", model)] + [(n, m) for n, m in model.named_modules() if n]
    for name, owner in list(owners):
        for attr in ("mano_loss", "atlas_loss"):
            sub = getattr(owner, attr, None)
            if sub is not None and not isinstance(sub, torch.nn.Module):
                owners.append((name + "." + attr, sub))
    for name, owner in owners:
        for attr, val in vars(owner).items():
            if ("lambda" in attr or attr.endswith("_weight")) and isinstance(val, (int, float)) and not isinstance(val, bool):
                sig.append((name, attr, float(val)))
    return tuple(sorted(sig))


class GraphedTrainStep:
    """One training step (forward -> zero_grad -> backward -> optimizer.step) recorded ONCE into a hipGraph and replayed.

    Why: the step is ~1 000 kernel launches.  With fp32 convolutions the GPU needs 11 ms for them and the host 5 ms to enqueue
    them, so the host is idle half of the time; with the bf16 encoder the GPU needs 5.9 ms and the same 5 ms of enqueue leave
    no slack - and eight ranks of one node share the host's cores.  A replay is ONE launch call (~0.1 ms of host time).

    What makes the step recordable: every kernel of ``csrc/`` is launched on the caller's stream through the C-ABI with no
    hidden allocation or synchronisation, and the launchers clear their scratch with fill KERNELS, never ``hipMemsetAsync`` - a
    captured memset becomes a hipGraph memset node, and replays of the configs[2] step (the contact path clears three scratch
    buffers per step) died with GPU memory faults in 9 of 12 runs until the graph held kernel nodes only
    (profiles/r04_graph_fault.md; ``tests/test_memory_safety_gpu.py`` replays that configuration in both bf16 flavours); the model's forward has no device->host read (masked means, mesh IoU and the hand-side
    split are device-side); BatchNorm counters are bumped on the device; Adam is created with ``capturable=True``
    (``make_optimizer``).  MIOpen's solution search runs in the eager warm-up steps before the capture.

    The batch SHAPE (and the non-tensor entries of the sample: sides, root) is fixed at capture; ``__call__`` copies a new
    batch into the static input buffers and replays.  Outputs are static tensors overwritten by every replay (clone what must
    survive).

    Data parallel (``buckets`` = an enabled ``dp.GradientBuckets`` on RCCL; round 5): the WHOLE step is one graph - the
    post-accumulate hooks run while the backward is being recorded, so the pack copies and the ``all_reduce`` calls (RCCL kernels
    on RCCL's own high-priority stream, joined by events) become graph nodes in exactly the order and overlap the eager step
    has; ``finish()``'s waits become the join before the recorded optimizer step.  Needs a fixed autograd graph on every rank: a
    parameter without a gradient in the eager warm-up step is refused before anything is recorded (its "some rank had one" flag
    would need the host).  A host-driven backend (gloo) cannot be recorded.  (A split form for such backends - forward +
    backward graph, exchange from Python, optimizer graph - was built and REMOVED: on ROCm 7.0 / torch 2.10 a recorded backward that
    ends its graph produced garbage gradients for the encoder's early layers in most two-process runs, with autograd-allocated
    and with persistent gradient buffers alike; tools/archive/r05/dp_split_dbg*.py are the reproducers.)
    Lambdas baked into the step (``ops.weighted_terms`` keeps the loss weights as device tensors; the ones a capture used are held by
    this object: ``term_weights``) are those of the capture: ``__call__`` compares the model's lambda attributes with the captured
    ones and raises when a schedule has changed them - re-capture, or write the new values into ``term_weights`` in
    place and call ``accept_lambdas()``.

    Side effect of construction: the ``warmup`` eager steps and the capture pass are REAL train steps on the capture batch
    (``warmup`` optimizer updates, BatchNorm running statistics, Adam step counters move; the capture pass itself only
    records).  A training loop that must not see them passes ``restore_state=True``: model and optimizer state are
    snapshotted before the warm-up and copied back (in place - the recorded kernels keep their addresses) after the capture.

    ROCm 7.0 caveat (measured, tools/graph_probe.py): ``hipGraphInstantiate`` segfaults at the end of the capture while the
    OUTPUTS of an earlier eager step (loss tensor, results dict - and through them their autograd nodes) are still referenced.
    Drop them before constructing this object (``bench.py`` does); the constructor collects garbage first."""

    def __init__(self, model, optimizer, sample, warmup=3, restore_state=False, buckets=None):
        import copy
        import gc

        gc.collect()
        dev = next(model.parameters()).device
        self.model, self.optimizer = model, optimizer
        self.buckets = buckets if (buckets is not None and buckets.enabled) else None
        if self.buckets is not None and self.buckets.backend != "nccl":
            raise ValueError("a data-parallel step can only be recorded when its collectives are stream work (RCCL); backend %r drives "
                             "them from the host - run that step eagerly" % self.buckets.backend)
        self.mode = "fused" if self.buckets is not None else "single"
        if restore_state and not isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)):
            # state the warm-up creates is put back by zeroing it: exact for Adam's moments and step counter, NOT for e.g. SGD's
            # momentum buffer (first step: buf = grad, not momentum * 0 + grad under dampening / nesterov)
            raise ValueError("GraphedTrainStep(restore_state=True) supports Adam / AdamW only (lazily created optimizer state is "
                             "restored by zeroing it)")
        snapshot = None
        if restore_state:
            snapshot = ({k: v.detach().clone() for k, v in model.state_dict().items()}, copy.deepcopy(optimizer.state_dict()))
        self.static = {k: (v.detach().to(dev).clone(memory_format=torch.preserve_format) if torch.is_tensor(v) else v)
                       for k, v in sample.items()}
        self._fixed = {k: v for k, v in sample.items() if not torch.is_tensor(v)}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # eager steps on the side stream: lazy initialisation, MIOpen find, allocator growth
            for _ in range(max(int(warmup), 1)):
                train_step(model, optimizer, self.static, self.buckets)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if self.buckets is not None and self.buckets.some_rank_lacked_a_gradient():
            # found by the eager warm-up step, BEFORE anything is being recorded (an exception inside a capture that holds RCCL work
            # leaves the stream in capture mode).  Decided from the REDUCED flag words, so every rank of the group raises here
            # together (a rank-local check lets one rank raise while its peers record and replay collectives that
            # never complete).
            raise RuntimeError("GraphedTrainStep: some rank of the group had planned parameter(s) without a gradient in the warm-up step.  A recorded "
                               "data-parallel step needs a fixed autograd graph on every rank (the 'some rank had a gradient' flags "
                               "of dp.GradientBuckets need a host round trip): pass such parameters in `exclude`, or run this step "
                               "eagerly (this rank lacked %d)" % self.buckets.last_missing)
        self.watchdog_wait = None
        if self.mode == "fused":
            self.watchdog_wait = wait_for_watchdog(self.buckets, dev)
        from . import ops

        owned_before = set(ops._CAPTURE_OWNED)
        self.graph = torch.cuda.CUDAGraph()
        if self.mode == "single":
            with torch.cuda.graph(self.graph):
                self.total, self.results, self.losses = train_step(model, optimizer, self.static)
        else:
            # thread_local: the process group's watchdog thread may touch the runtime while this thread records
            b = self.buckets
            b.capturing = True
            try:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.total, self.results, self.losses = train_step(model, optimizer, self.static, b)
            finally:
                b.capturing = False
        # the loss-weight tensors this capture took out of ops' cache: kept alive here; [(values, tensor)] (a key another, uncollected
        # capture already owned stays with ops._CAPTURE_OWNED)
        mine = [k for k in ops._CAPTURE_OWNED if k not in owned_before]
        self.term_weights = [(k[1], ops._CAPTURE_OWNED.pop(k)) for k in mine]
        self._lambdas = _lambda_signature(model)
        if snapshot is not None:
            with torch.no_grad():
                live = model.state_dict()
                for k, v in snapshot[0].items():
                    live[k].copy_(v)
                # in place as well: capturable Adam's exp_avg / exp_avg_sq / step tensors are addresses inside the graph
                saved = snapshot[1]["state"]
                order = [p for g in optimizer.param_groups for p in g["params"]]
                for idx, p in enumerate(order):
                    cur = optimizer.state.get(p, {})
                    old = saved.get(idx)
                    for name, t in cur.items():
                        if not torch.is_tensor(t):
                            continue
                        if old is not None and name in old:
                            t.copy_(old[name])
                        else:  # state created by the warm-up itself (first step of a fresh optimizer): back to its initial value
                            t.zero_()
            from . import optim  # bf16 shadow filters (optim.ObmanAdam): rewritten IN PLACE from the restored fp32 filters

            optim.refresh_bf16_shadows(model)

    def accept_lambdas(self):
        """The caller has updated ``term_weights`` in place for the model's current lambdas."""
        self._lambdas = _lambda_signature(self.model)

    def __call__(self, sample):
        if _lambda_signature(self.model) != self._lambdas:
            raise ValueError("GraphedTrainStep: the model's loss weights changed since the capture (%s).  The recorded step multiplies "
                             "by the captured values: capture a new graph, or write the new values into `term_weights` and call "
                             "accept_lambdas()" % sorted(set(_lambda_signature(self.model)) ^ set(self._lambdas))[:4])
        for k, v in sample.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
            elif not _same_entry(self._fixed.get(k), v):
                raise ValueError("GraphedTrainStep: the non-tensor entry %r differs from the captured batch (%r != %r).  A "
                                 "recorded step replays the kernels of ONE batch layout - the hand-side split (`sides`) and "
                                 "every other host-side entry decide which kernels run and on how many rows - so every batch "
                                 "must carry the captured values; capture one graph per layout" % (k, v, self._fixed.get(k)))
        self.graph.replay()
        return self.total, self.results, self.losses
