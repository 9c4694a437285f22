"""GPU: memory-safety evidence for the HIP path.

(1) Guard pages.  ``tools/efence`` routes EVERY device allocation of a child process (inputs, outputs, workspaces, autograd-saved
    tensors, MIOpen workspaces) through an allocator that right-aligns (or left-aligns) the tensor against an UNMAPPED page, traces
    every launcher of the C-ABI and synchronises after it.  A kernel that touches one 16-byte unit outside any buffer it was handed
    kills the child with a GPU memory fault (exit code != 0) after naming itself; a write into the mapped slack on the other side
    trips a canary.  The tool proves itself first: in-range accesses pass, one byte past the end / before the start dies.
(2) hipGraph replays.  The captured train step of configs[2] (both reduced-precision flavours) is replayed 30 times and must
    reproduce the eager steps: the configuration died with memory faults under replay until the stream memsets of the launchers
    became fill kernels (profiles/r04_graph_fault.md).
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EFENCE = os.path.join(REPO, "tools", "efence", "efence.py")


def _run(args, timeout=900):
    env = dict(os.environ, OBMAN_MANO_SYNTHETIC="1")
    p = subprocess.run([sys.executable, EFENCE] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)
    tail = "\n".join((p.stderr or "").strip().splitlines()[-12:])
    return p.returncode, p.stdout + p.stderr, tail


def test_guard_page_allocator_detects_one_byte_out_of_bounds():
    rc, out, tail = _run(["--selftest"])
    assert rc == 0, tail
    for mode in ("inside", "read_past", "write_past", "left_read_before", "write_before"):
        line = [l for l in out.splitlines() if "selftest " + mode in l]
        assert line and " ok " in line[0], (mode, line)


@pytest.mark.parametrize("name,args", [
    ("configs[1] fp32 bs64", ["--config", "c2", "--batch", "64", "--steps", "2", "--eval"]),
    ("configs[2] bf16 bs64", ["--config", "c3", "--batch", "64", "--encoder-dtype", "bf16", "--decoder-dtype", "bf16", "--steps", "2",
                              "--eval"]),
    ("configs[2] bf16 bs64, guard BEFORE the tensors", ["--config", "c3", "--batch", "64", "--encoder-dtype", "bf16",
                                                         "--decoder-dtype", "bf16", "--steps", "1", "--left"]),
    ("configs[2] fp32 bs16", ["--config", "c3", "--batch", "16", "--steps", "1"]),
    ("configs[2] block-tiled bf16 rows kernels + five-tile dW2, bs16", ["--config", "c3", "--batch", "16", "--decoder-dtype", "bf16", "--steps", "1",
                                                                        "--env", "OBMAN_DEC_ROWS2=0", "--env", "OBMAN_DEC_TN2W=0"]),
    ("configs[4] bf16 bs8", ["--config", "c5", "--batch", "8", "--encoder-dtype", "bf16", "--decoder-dtype", "bf16", "--steps", "1"]),
    ("configs[4] fp32 bs4, guard BEFORE", ["--config", "c5", "--batch", "4", "--steps", "1", "--left"]),
])
def test_train_step_touches_nothing_outside_its_buffers(name, args):
    rc, out, tail = _run(args)
    assert rc == 0, "%s: exit %d\n%s" % (name, rc, tail)
    clean = [l for l in out.splitlines() if l.startswith("[efence] clean")]
    assert clean and "'canary_hits': 0" in clean[0], tail
    assert out.count("[obman-launch]") > 20  # the HIP launchers really ran (and were attributed) in the child


def _c3_model(flavour):
    import warnings

    from obman_train_amd.networks.handnet import HandNet
    from obman_train_amd.synthetic import CONFIGS

    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    model = HandNet(**CONFIGS["c3"]).cuda().train()
    if flavour in ("all-bf16",):
        model.base_net.autocast_dtype = torch.bfloat16
    if flavour in ("decoder-bf16", "all-bf16"):
        model.atlas_branch.decoder.mfma_dtype = "bf16"
    return model


@pytest.mark.parametrize("flavour", ["decoder-bf16", "all-bf16"])
def test_graphed_train_step_configs2_replays_match_eager(flavour):
    from obman_train_amd.queries import TransQueries
    from obman_train_amd.synthetic import make_batch
    from obman_train_amd.trainer import GraphedTrainStep, make_optimizer, train_step
    from tests.conftest import record_measurement

    os.environ.setdefault("OBMAN_MANO_SYNTHETIC", "1")
    torch.backends.cudnn.benchmark = False
    dev = torch.device("cuda", 0)
    sample = make_batch(16, dev, seed=3, image_size=256)
    sample[TransQueries.images] = sample[TransQueries.images].contiguous(memory_format=torch.channels_last)
    n_cmp, n_replays = 4, 30

    def eager_run():
        model = _c3_model(flavour)
        opt = make_optimizer(model, "adam", lr=1e-4, capturable=True)
        out = []
        for _ in range(n_cmp):
            total, _, losses = train_step(model, opt, sample)
            out.append((float(total), {k: float(v) for k, v in losses.items() if torch.is_tensor(v)}))
        return out

    eager, again = eager_run(), eager_run()  # two eager runs: what "the same steps" means on this stack (run-to-run noise)
    noise = [abs(a[0] - b[0]) / abs(a[0]) for a, b in zip(eager, again)]

    model = _c3_model(flavour)
    opt = make_optimizer(model, "adam", lr=1e-4, capturable=True)
    step = GraphedTrainStep(model, opt, sample, warmup=2, restore_state=True)  # the warm-up steps must not count
    got = []
    for i in range(n_replays):
        total, _, losses = step(sample)  # queued without waiting, like a training loop that logs rarely
        if i < n_cmp:
            got.append((total.clone(), {k: v.clone() for k, v in losses.items() if torch.is_tensor(v)}))
    torch.cuda.synchronize()
    final = float(total)
    assert final == final and abs(final) < 1e9
    worst = 0.0
    for i in range(n_cmp):
        rel = abs(float(got[i][0]) - eager[i][0]) / abs(eager[i][0])
        worst = max(worst, rel)
        # Replay 0 is a pure forward on the restored initial weights: same kernels, same inputs, same loss.  From the second
        # step on the weights carry an Adam update, whose first steps are ~lr * sign(gradient): MIOpen's atomically accumulated
        # weight gradients differ in the last bits from run to run, near-zero gradient entries flip their sign, and two EAGER
        # runs already drift apart by ~1e-3 per step (measured: 9e-4 at step 2; tests/test_handnet_gpu.py holds whole
        # flavours to 1e-2 over 5 steps for the same reason)
        # ... so the yardstick is measured in the same process: 4 x the eager-vs-eager difference of that step, with a floor
        # (the all-bf16 flavour is not run-to-run deterministic even in its first forward: 2.6e-3 measured; its later steps
        # scatter between 2e-3 and 1.4e-2 - so the largest difference of the run is the scale, not the step's own)
        bound = 4.0 * max(max(noise) if flavour == "all-bf16" else noise[i], 1e-5 if i == 0 else 1e-3 * i)
        assert rel <= bound, (i, float(got[i][0]), eager[i][0], bound, noise)
        assert set(got[i][1]) == set(eager[i][1])
    assert final < eager[0][0]  # 30 Adam steps on one batch: the loss went down
    record_measurement("graph_replay_vs_eager[%s]" % flavour, {"worst_rel_total_first_%d_steps" % n_cmp: worst, "eager_vs_eager": noise, "replays": n_replays,
                                                                 "loss_first": eager[0][0], "loss_last": final})
